// gsvc_amd/csrc/raster_common.h — per-Gaussian preprocess of the orthographic rasterizer (device) and the
// layout of the three state blobs shared by forward and backward.
//
// Replaces the external CUDA extension GSVC imports as
// diff_gaussian_rasterization.cuda_ortho_gaussian_rasterizer (reference README.md:52; call sites
// reference ortho_gaussian_renderer/renderer.py:63-98, preprocess.py:58-104).  Spec: DESIGN.md "Raster spec".
#pragma once
#include "common.h"

namespace gsvc {

constexpr int TILE = 16;
constexpr float LOWPASS = 0.3f;   // default of gsvc_raster_settings.low_pass
constexpr float ALPHA_MIN = 1.0f / 255.0f;
constexpr float ALPHA_MAX = 0.99f;
constexpr float T_MIN = 0.0001f;

// One Gaussian as the blend / sort kernels read it: 64 B = one cache line (a 48-B record straddles two lines half of the
// time); the blend kernels load the first three 16-B words, the sort kernel the last two.
struct alignas(16) GeomRec {
    float u, v, A, B;            // pixel-space centre, conic xx, xy
    float C, opacity, r, g;      // conic yy, opacity, colour
    float b;                     // colour
    uint32_t bbox_x, bbox_y;     // int16 pairs lo|hi<<16: pixels outside can never reach alpha >= 1/255
    int32_t goff;                // first of this Gaussian's rows in the backward's per-instance partial-sum buffer: the
                                 // instance in tile j of its rectangle (row-major) owns row goff + j (single-view binning only)
    uint32_t rect_x, rect_y;     // tile rectangle x0 | x1<<16, y0 | y1<<16 (as BinRec; 0 when culled)
    float depth;                 // view-space z
    uint32_t pad;
};
static_assert(sizeof(GeomRec) == 64, "GeomRec must be 64 bytes");
constexpr int ROW_FLOATS = 16;   // one backward partial-sum row: 9 sums + padding = one 64-B line, written whole

// One Gaussian as the binning kernels read it: 32 B.
constexpr int BIN_SLOTS = 4;     // instance slots resolved by K1's LDS histogram; further tiles ("extras") are placed by K3
constexpr int BIN_WG = 1024;     // Gaussians per binning workgroup (K1 and K3 use the same partition)
constexpr int HEAVY_EXTRAS = 2 * BIN_WG;   // a workgroup with more extras than this places them through an LDS histogram in K3
struct alignas(16) BinRec {
    float depth;
    uint32_t rect_x, rect_y;     // x0 | x1<<16,  y0 | y1<<16 (tile units, upper exclusive); 0 when culled
    uint32_t pad;                // pair mode: x range of the opposite view's rectangle, mirrored into this view's tiles
    int32_t slot[BIN_SLOTS];     // position of the Gaussian's first tiles inside each tile's segment
};
static_assert(sizeof(BinRec) == 32, "BinRec must be 32 bytes");

struct RasterLayout {
    int gx, gy, tiles;
    // geom blob
    uint64_t off_geom, off_bin, geom_bytes;
    // binning blob
    uint64_t off_counters, off_tile_offsets, off_tile_count, off_tile_extra, off_big_list, off_wg_extras, off_keys, off_point_list,
        off_inst_bbox, off_gslot, binning_bytes;
    // image blob
    uint64_t off_final_T, off_n_contrib, image_bytes;
};

inline RasterLayout raster_layout(const gsvc_raster_settings &s, int64_t P, int64_t max_instances)
{
    RasterLayout L;
    L.gx = (s.image_width + TILE - 1) / TILE;
    L.gy = (s.image_height + TILE - 1) / TILE;
    L.tiles = L.gx * L.gy;
    const uint64_t p = (uint64_t)(P > 0 ? P : 1), m = (uint64_t)(max_instances > 0 ? max_instances : 1);
    L.off_geom = 0;
    L.off_bin = align_up(p * sizeof(GeomRec), 256);
    L.geom_bytes = L.off_bin + align_up(p * sizeof(BinRec), 256);
    uint64_t o = 0;
    L.off_counters = o;      o += 256;
    L.off_tile_offsets = o;  o += align_up((uint64_t)(L.tiles + 1) * 4, 256);
    L.off_tile_count = o;    o += align_up((uint64_t)L.tiles * 4, 256);
    L.off_tile_extra = o;    o += align_up((uint64_t)L.tiles * 4, 256);
    L.off_big_list = o;      o += align_up((uint64_t)L.tiles * 4, 256);
    L.off_wg_extras = o;     o += align_up(((p + BIN_WG - 1) / BIN_WG) * 4, 256);     // per K1 workgroup: instances beyond the slots
    L.off_keys = o;          o += align_up(m * 8, 256);
    L.off_point_list = o;    o += align_up(m * 4, 256);
    L.off_inst_bbox = o;     o += align_up(m * 8, 256);
    L.off_gslot = o;         o += align_up(m * 4, 256);      // per sorted instance: its row in the backward's partial-sum buffer
    L.binning_bytes = o;
    // final_T / n_contrib are stored tile-major: [tile][quadrant][lane] (whole 16x16 tiles, also at the image border)
    const uint64_t hw = (uint64_t)L.tiles * (TILE * TILE);
    L.off_final_T = 0;
    L.off_n_contrib = align_up(hw * 4, 256);
    L.image_bytes = L.off_n_contrib + align_up(hw * 4, 256);
    return L;
}

// settings as the kernels take them (by value, in SGPRs)
struct RasterParams {
    int H, W, gx, gy;
    float x_min, y_min, scale, threshold, scale_modifier;
    float bg0, bg1, bg2;
    float m[12];  // rows 0..2 of the view matrix
    uint32_t flags;     // GSVC_RASTER_* convention switches
    float low_pass;     // resolved value (default 0.3)
    float pix_off;      // 0.5 (pixel centres on integers) or 0
};

inline RasterParams make_params(const gsvc_raster_settings &s)
{
    RasterParams p;
    p.H = s.image_height; p.W = s.image_width;
    p.gx = (p.W + TILE - 1) / TILE; p.gy = (p.H + TILE - 1) / TILE;
    p.x_min = s.x_min; p.y_min = s.y_min; p.scale = s.scale; p.threshold = s.threshold;
    p.scale_modifier = s.scale_modifier;
    p.bg0 = s.bg[0]; p.bg1 = s.bg[1]; p.bg2 = s.bg[2];
    for (int i = 0; i < 12; i++) p.m[i] = s.viewmatrix[i];
    p.flags = s.flags;
    p.low_pass = (s.flags & GSVC_RASTER_NO_LOW_PASS) ? 0.0f : (s.low_pass != 0.0f ? s.low_pass : LOWPASS);
    p.pix_off = (s.flags & GSVC_RASTER_PIXEL_CORNER) ? 0.0f : 0.5f;
    return p;
}

#ifdef __HIPCC__

__device__ __forceinline__ uint32_t order_bits(float f)
{
    uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

__device__ __forceinline__ int tile_clamp(float t, int g)
{
    t = fmaxf(t, -1.0f);
    t = fminf(t, (float)g + 1.0f);
    int i = (int)t;
    i = i < 0 ? 0 : i;
    i = i > g ? g : i;
    return i;
}

struct PreOut {
    float xv;          // view-space x (the opposite view of a pair sees -xv)
    int radius_raw;    // 3-sigma radius before the screen-rectangle test (0 when culled by slab / covariance)
    float u, v, depth;
    float A, B, C;     // conic
    float a, b, c;     // 2-D covariance (low-pass included)
    int radius;
    int x0, y0, x1, y1;
};

// Can the Gaussian (centre (u,v), conic A,B,C, opacity o) reach alpha >= 1/255 at any pixel centre of the 8x8 quadrant whose
// first pixel is (x0, y0)?  alpha = o exp(-q/2) with q(d) = A dx^2 + 2 B dx dy + C dy^2, so the question is whether the
// minimum of the convex quadratic q over the square [x0, x0+7] x [y0, y0+7] is <= 2 ln(255 o): 0 if the centre is inside,
// else the smallest of the four edge minima (each a clamped 1-D parabola).  Tighter than the axis-aligned alpha bbox for
// rotated / elongated footprints; conservative (margins, NaN keeps), so the composited result is unchanged.
__device__ __forceinline__ bool ellipse_hits_quad(float u, float v, float A, float B, float C, float o, float x0, float y0)
{
    const float lx = x0 - u, hx = x0 + 7.0f - u, ly = y0 - v, hy = y0 + 7.0f - v;
    if (lx <= 0.f && hx >= 0.f && ly <= 0.f && hy >= 0.f) return true;
    const float tau2 = 2.0f * (__logf(255.0f * o) + 1e-3f);
    const float mbc = -B / C, mba = -B / A;
    float qmin;
    {
        const float dy = fminf(fmaxf(mbc * lx, ly), hy);
        qmin = A * lx * lx + 2.0f * B * lx * dy + C * dy * dy;
    }
    {
        const float dy = fminf(fmaxf(mbc * hx, ly), hy);
        qmin = fminf(qmin, A * hx * hx + 2.0f * B * hx * dy + C * dy * dy);
    }
    {
        const float dx = fminf(fmaxf(mba * ly, lx), hx);
        qmin = fminf(qmin, A * dx * dx + 2.0f * B * dx * ly + C * ly * ly);
    }
    {
        const float dx = fminf(fmaxf(mba * hy, lx), hx);
        qmin = fminf(qmin, A * dx * dx + 2.0f * B * dx * hy + C * hy * hy);
    }
    return !(qmin > tau2 * 1.0005f + 2e-3f);
}

// Steps 1-7 of the raster spec.  Every operation here decides an integer downstream (radius, tile
// rectangle, depth key), so it is plain IEEE binary32 in a fixed order with no FMA contraction — the CPU
// oracle repeats it bit for bit.  sqrt and division are the correctly rounded forms.
__device__ __forceinline__ int preprocess_gaussian(const RasterParams &st, float px, float py, float pz,
                                                   float s0, float s1, float s2, float qr, float qx, float qy,
                                                   float qz, PreOut &o)
{
#pragma clang fp contract(off)
    const float *M = st.m;
    float xv = M[0] * px + M[1] * py + M[2] * pz + M[3];
    float yv = M[4] * px + M[5] * py + M[6] * pz + M[7];
    float zv = M[8] * px + M[9] * py + M[10] * pz + M[11];
    o.radius = 0;
    o.radius_raw = 0;
    o.xv = xv;
    if (st.flags & GSVC_RASTER_SLAB_ONE_SIDED) {
        if (!(zv <= 0.0f && zv >= -st.threshold)) return 0;
    } else if (!(fabsf(zv) <= st.threshold)) return 0;

    float R00 = 1.f - 2.f * (qy * qy + qz * qz);
    float R01 = 2.f * (qx * qy - qr * qz);
    float R02 = 2.f * (qx * qz + qr * qy);
    float R10 = 2.f * (qx * qy + qr * qz);
    float R11 = 1.f - 2.f * (qx * qx + qz * qz);
    float R12 = 2.f * (qy * qz - qr * qx);
    float R20 = 2.f * (qx * qz - qr * qy);
    float R21 = 2.f * (qy * qz + qr * qx);
    float R22 = 1.f - 2.f * (qx * qx + qy * qy);
    float S0 = st.scale_modifier * s0, S1 = st.scale_modifier * s1, S2 = st.scale_modifier * s2;
    float L00 = R00 * S0, L01 = R01 * S1, L02 = R02 * S2;
    float L10 = R10 * S0, L11 = R11 * S1, L12 = R12 * S2;
    float L20 = R20 * S0, L21 = R21 * S1, L22 = R22 * S2;
    float c00 = L00 * L00 + L01 * L01 + L02 * L02;
    float c01 = L00 * L10 + L01 * L11 + L02 * L12;
    float c02 = L00 * L20 + L01 * L21 + L02 * L22;
    float c11 = L10 * L10 + L11 * L11 + L12 * L12;
    float c12 = L10 * L20 + L11 * L21 + L12 * L22;
    float c22 = L20 * L20 + L21 * L21 + L22 * L22;

    float T00 = st.scale * M[0], T01 = st.scale * M[1], T02 = st.scale * M[2];
    float T10 = st.scale * M[4], T11 = st.scale * M[5], T12 = st.scale * M[6];
    float U00 = c00 * T00 + c01 * T01 + c02 * T02;
    float U01 = c01 * T00 + c11 * T01 + c12 * T02;
    float U02 = c02 * T00 + c12 * T01 + c22 * T02;
    float U10 = c00 * T10 + c01 * T11 + c02 * T12;
    float U11 = c01 * T10 + c11 * T11 + c12 * T12;
    float U12 = c02 * T10 + c12 * T11 + c22 * T12;
    float ca = T00 * U00 + T01 * U01 + T02 * U02 + st.low_pass;
    float cb = T00 * U10 + T01 * U11 + T02 * U12;
    float cc = T10 * U10 + T11 * U11 + T12 * U12 + st.low_pass;
    float det = ca * cc - cb * cb;
    if (!(det != 0.0f)) return 0;
    float det_inv = (1.0f / det);
    o.a = ca; o.b = cb; o.c = cc;
    o.A = cc * det_inv;
    o.B = -cb * det_inv;
    o.C = ca * det_inv;
    float mid = 0.5f * (ca + cc);
    float disc = sqrtf(fmaxf(0.1f, mid * mid - det));
    float lam = fmaxf(mid + disc, mid - disc);
    float rad_f = ceilf(3.0f * sqrtf(lam));
    if (!(rad_f >= 0.0f)) return 0;
    if (rad_f > 1.0e9f) rad_f = 1.0e9f;
    int radius = (int)rad_f;

    float u = (xv - st.x_min) * st.scale - st.pix_off;
    float v = (yv - st.y_min) * st.scale - st.pix_off;
    float rf = (float)radius;
    int x0 = tile_clamp(((u - rf) / (float)TILE), st.gx);
    int x1 = tile_clamp(((u + rf + (float)(TILE - 1)) / (float)TILE), st.gx);
    int y0 = tile_clamp(((v - rf) / (float)TILE), st.gy);
    int y1 = tile_clamp(((v + rf + (float)(TILE - 1)) / (float)TILE), st.gy);
    o.u = u; o.v = v; o.depth = zv;
    o.radius_raw = radius;
    o.x0 = x0; o.y0 = y0; o.x1 = x1; o.y1 = y1;
    if ((x1 - x0) * (y1 - y0) <= 0) return 0;
    o.radius = radius;
    return radius;
}

#endif  // __HIPCC__

}  // namespace gsvc

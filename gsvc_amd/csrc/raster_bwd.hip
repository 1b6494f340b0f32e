// gsvc_amd/csrc/raster_bwd.hip — backward of the orthographic tile rasterizer, gfx950.
//
// Replaces the implicit autograd backward of the external CUDA extension (consumers: reference
// pipeline/train.py:462 `loss.backward()`, scene/gaussian_model.py:1311 `viewspace_points.grad`).
//
//   B1 blend_bwd     one wave per 16x16 tile (lane l owns pixel l of each 8x8 quadrant) walks the tile's sorted list
//                    back to front in chunks of 64 entries: bbox + exact ellipse test per quadrant, replay of the alpha
//                    compositing per pixel, per Gaussian nine sums (moments of h = G dL/dalpha and the colour
//                    gradients) reduced over the wave with permlane swaps + DPP (four entries per pass), written per chunk
//                    as whole 64-byte rows — one per (tile, Gaussian) instance, at the row index the sort kernel left in
//                    gslot — with plain stores: no atomics, no zero-fill pass, deterministic.
//   B2 gaussian_bwd  one lane per Gaussian: adds up the Gaussian's rows (contiguous, one per tile of its rectangle), then
//                    conic -> 2-D covariance -> 3-D covariance -> (scale, quaternion),
//                    screen-space -> world mean (constant orthographic Jacobian, no covariance->mean term).
#include "raster_common.h"

#include <cstdlib>

namespace gsvc {

__device__ __forceinline__ int sext16b(uint32_t v) { return (int)(int16_t)(v & 0xffffu); }

__device__ __forceinline__ bool bbox_hits_b(uint2 bb, int qx0, int qy0)
{
    return !(sext16b(bb.x) > qx0 + 7 || sext16b(bb.x >> 16) < qx0 || sext16b(bb.y) > qy0 + 7 || sext16b(bb.y >> 16) < qy0);
}

// Does the alpha bounding box touch the 16x16 tile at (tx0, ty0) at all?  The tile lists come from the 3-sigma rectangles of
// the spec, the alpha box is tighter (it knows the opacity): 16-28 % of the instances cannot reach alpha >= 1/255 anywhere in
// their tile.  Such an instance gets no row in the partial-sum buffer — B1 does not write it, B2 (same test, same integers)
// does not read it.
__device__ __forceinline__ bool bbox_hits_tile(uint32_t bx, uint32_t by, int tx0, int ty0)
{
    return !(sext16b(bx) > tx0 + 15 || sext16b(bx >> 16) < tx0 || sext16b(by) > ty0 + 15 || sext16b(by >> 16) < ty0);
}

// Pixel state of the back-to-front replay.  The colours only ever enter through their dot product with dL/dpixel, so the
// composited-behind colour is kept as ONE number: bd = (colour seen behind the current entry, at unit transmittance) . d
struct PixState {
    float T, tb, d0, d1, d2, bd;  // tb = final_T * (bg . dL/dpixel)
    int last;                     // n_contrib; 0 for a pixel outside the image (no list position is <= 0)
};

struct Sums9 {
    float h, x, y, xx, xy, yy, r, g, b;
};

// Nine sums of FOUR list entries in one pass: 74 cross-lane instructions per four entries (18.5 per entry; rounds 1-3 reduced
// two entries per pass in 61: 30.5 per entry).  Every halving stage pairs registers of DIFFERENT entries, so each stage halves the number of live
// registers as well as the number of lanes per value:
//   stage 32  (A, B) and (C, D): v_permlane32_swap + add per value leaves A's 32 half-sums in the lower and B's in the upper half
//   stage 16  (AB, CD): v_permlane16_swap + add -> nine registers whose 16-lane rows hold the partial sums of entry A | C | B | D
//   stage 8/4 row_ror:8 and row_half_mirror adds whose bank_mask writes only half of the destination's banks: two registers merge
//             into one per stage without a select (9 -> 5 -> 3 registers)
//   stage 2/1 quad_perm adds: afterwards every lane of the quad (row r, bank b) of q.xx holds the total of value {0,2,1,3}[b] of
//             row r's entry, of q.g value {4,6,5,7}[b], and every lane of row r of q.b the ninth.
__device__ __forceinline__ void r4_pair(Sums9 &a, Sums9 &b, const bool rows16)
{
    if (!rows16)
        asm volatile("s_nop 1\n"
                     "v_permlane32_swap_b32 %0, %9\n"
                     "v_permlane32_swap_b32 %1, %10\n"
                     "v_permlane32_swap_b32 %2, %11\n"
                     "v_permlane32_swap_b32 %3, %12\n"
                     "v_permlane32_swap_b32 %4, %13\n"
                     "v_permlane32_swap_b32 %5, %14\n"
                     "v_permlane32_swap_b32 %6, %15\n"
                     "v_permlane32_swap_b32 %7, %16\n"
                     "v_permlane32_swap_b32 %8, %17\n"
                     "s_nop 1\n"
                     "v_add_f32 %0, %0, %9\n"
                     "v_add_f32 %1, %1, %10\n"
                     "v_add_f32 %2, %2, %11\n"
                     "v_add_f32 %3, %3, %12\n"
                     "v_add_f32 %4, %4, %13\n"
                     "v_add_f32 %5, %5, %14\n"
                     "v_add_f32 %6, %6, %15\n"
                     "v_add_f32 %7, %7, %16\n"
                     "v_add_f32 %8, %8, %17\n"
                     : "+v"(a.h), "+v"(a.x), "+v"(a.y), "+v"(a.xx), "+v"(a.xy), "+v"(a.yy), "+v"(a.r), "+v"(a.g), "+v"(a.b), "+v"(b.h),
                       "+v"(b.x), "+v"(b.y), "+v"(b.xx), "+v"(b.xy), "+v"(b.yy), "+v"(b.r), "+v"(b.g), "+v"(b.b));
    else
        asm volatile("s_nop 1\n"
                     "v_permlane16_swap_b32 %0, %9\n"
                     "v_permlane16_swap_b32 %1, %10\n"
                     "v_permlane16_swap_b32 %2, %11\n"
                     "v_permlane16_swap_b32 %3, %12\n"
                     "v_permlane16_swap_b32 %4, %13\n"
                     "v_permlane16_swap_b32 %5, %14\n"
                     "v_permlane16_swap_b32 %6, %15\n"
                     "v_permlane16_swap_b32 %7, %16\n"
                     "v_permlane16_swap_b32 %8, %17\n"
                     "s_nop 1\n"
                     "v_add_f32 %0, %0, %9\n"
                     "v_add_f32 %1, %1, %10\n"
                     "v_add_f32 %2, %2, %11\n"
                     "v_add_f32 %3, %3, %12\n"
                     "v_add_f32 %4, %4, %13\n"
                     "v_add_f32 %5, %5, %14\n"
                     "v_add_f32 %6, %6, %15\n"
                     "v_add_f32 %7, %7, %16\n"
                     "v_add_f32 %8, %8, %17\n"
                     : "+v"(a.h), "+v"(a.x), "+v"(a.y), "+v"(a.xx), "+v"(a.xy), "+v"(a.yy), "+v"(a.r), "+v"(a.g), "+v"(a.b), "+v"(b.h),
                       "+v"(b.x), "+v"(b.y), "+v"(b.xx), "+v"(b.xy), "+v"(b.yy), "+v"(b.r), "+v"(b.g), "+v"(b.b));
}

// q0..q8 = h x y xx xy yy r g b.  A DPP read of a register needs two wait states after the VALU write: the order below keeps two
// other instructions between every write and its dependent read.
__device__ __forceinline__ void r4_tail(Sums9 &q)
{
    asm volatile("s_nop 1\n"
                 "v_add_f32_dpp %1, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xc\n"
                 "v_add_f32_dpp %3, %3, %3 row_ror:8 row_mask:0xf bank_mask:0xc\n"
                 "v_add_f32_dpp %5, %5, %5 row_ror:8 row_mask:0xf bank_mask:0xc\n"
                 "v_add_f32_dpp %7, %7, %7 row_ror:8 row_mask:0xf bank_mask:0xc\n"
                 "v_add_f32_dpp %8, %8, %8 row_ror:8 row_mask:0xf bank_mask:0xf\n"
                 "v_add_f32_dpp %1, %0, %0 row_ror:8 row_mask:0xf bank_mask:0x3\n"
                 "v_add_f32_dpp %3, %2, %2 row_ror:8 row_mask:0xf bank_mask:0x3\n"
                 "v_add_f32_dpp %5, %4, %4 row_ror:8 row_mask:0xf bank_mask:0x3\n"
                 "v_add_f32_dpp %7, %6, %6 row_ror:8 row_mask:0xf bank_mask:0x3\n"
                 "v_add_f32_dpp %8, %8, %8 row_half_mirror row_mask:0xf bank_mask:0xf\n"
                 "v_add_f32_dpp %3, %3, %3 row_half_mirror row_mask:0xf bank_mask:0xa\n"
                 "v_add_f32_dpp %7, %7, %7 row_half_mirror row_mask:0xf bank_mask:0xa\n"
                 "v_add_f32_dpp %3, %1, %1 row_half_mirror row_mask:0xf bank_mask:0x5\n"
                 "v_add_f32_dpp %7, %5, %5 row_half_mirror row_mask:0xf bank_mask:0x5\n"
                 "v_add_f32_dpp %8, %8, %8 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
                 "s_nop 0\n"
                 "v_add_f32_dpp %3, %3, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
                 "v_add_f32_dpp %7, %7, %7 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
                 "v_add_f32_dpp %8, %8, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                 "v_add_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                 "v_add_f32_dpp %7, %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                 : "+v"(q.h), "+v"(q.x), "+v"(q.y), "+v"(q.xx), "+v"(q.xy), "+v"(q.yy), "+v"(q.r), "+v"(q.g), "+v"(q.b));
}


// One list entry at one pixel, straight-line: a pixel the entry did not contribute to (behind the pixel's last
// contributor, alpha < 1/255, power > 0, outside the image) runs the same instructions with alpha = 0 and G = 0, which
// leaves T and bd unchanged (rcp(1) = 1, 0 * cd + 1 * bd = bd) and adds zeros.  No EXEC-mask branches: the compiler's
// branchy form spent a third of its instructions on zero-filling the nine sums on every path.
template <bool CLAMP_STOP, bool ZERO_BG>
__device__ __forceinline__ bool bwd_pixel(PixState &p, float dx, float dy, const float4 &a, const float4 &b, float cb,
                                          int contributor, Sums9 &s)
{
    float t1 = a.z * dx;                        // -power*log2(e) = A' dx^2 + C' dy^2 + B' dx dy, conic pre-scaled
    t1 = fmaf(a.w, dy, t1);
    float pw = (b.x * dy) * dy;
    pw = fmaf(dx, t1, pw);
    const float Graw = __builtin_amdgcn_exp2f(-pw);
    const float araw = fminf(ALPHA_MAX, b.y * Graw);
    const bool valid = (contributor <= p.last) & !(pw < 0.0f) & !(araw < ALPHA_MIN);
    const float alpha = valid ? araw : 0.0f;
    // CLAMP_STOP (GSVC_RASTER_CLAMP_STOPS_GRADIENT): where min(0.99, .) is active alpha does not depend on the Gaussian
    const float G = (CLAMP_STOP ? (valid & !(b.y * Graw > ALPHA_MAX)) : valid) ? Graw : 0.0f;
    const float cd = fmaf(cb, p.d2, fmaf(b.w, p.d1, b.z * p.d0));   // colour . dL/dpixel
    const float oma = 1.0f - alpha;
    const float inv = __builtin_amdgcn_rcpf(oma);
    p.T *= inv;
    const float w = alpha * p.T;
    s.r = fmaf(w, p.d0, s.r); s.g = fmaf(w, p.d1, s.g); s.b = fmaf(w, p.d2, s.b);
    // ZERO_BG (black background, the reference's default): the background's share T_final (bg . d) / (1 - alpha) is zero
    const float dLda = ZERO_BG ? (cd - p.bd) * p.T : fmaf(cd - p.bd, p.T, -(p.tb * inv));
    p.bd = fmaf(alpha, cd, oma * p.bd);
    // moments of h = G * dL/dalpha; the conic / opacity factors are applied once per Gaussian in B2
    const float h = G * dLda;
    const float hx = h * dx, hy = h * dy;
    s.h += h; s.x += hx; s.y += hy;
    s.xx = fmaf(hx, dx, s.xx); s.xy = fmaf(hx, dy, s.xy); s.yy = fmaf(hy, dy, s.yy);
    return valid;
}

// The same step with the entry's log2(alpha) as a POLYNOMIAL in the pixel's offset (x, y) from the tile centre (|x|, |y| <= 7.5):
//   t = n0 + n1 x + n2 y + n3 x^2 + n4 y^2 + n5 xy = log2(o) - [A'(U - x)^2 + C'(V - y)^2 + B'(U - x)(V - y)],  (U, V) = centre - tile centre
// (the forward's compositing loop does the same per quadrant: csrc/raster_fwd.hip blend_pass).  x, y and their products are
// per-lane CONSTANTS, so the power costs five multiply-adds (was: two subtractions for dx, dy + five), the opacity rides in n0
// (no o * G product) and the six moments are multiply-adds against the same constants: sums of h', h' x, h' y, h' x^2, h' xy,
// h' y^2 with h' = o G dL/dalpha — moments about the TILE centre, which the row's writer shifts to the Gaussian's centre and
// divides by o once per (tile, Gaussian) (k_blend_bwd_tile's flush).  33 vector instructions per (entry, quadrant), was 39.
// Only for conics that are positive definite with a margin (the forward's rule): "power > 0" cannot occur there.
template <bool CLAMP_STOP, bool ZERO_BG>
__device__ __forceinline__ bool bwd_pixel_poly(PixState &p, float x, float y, float x2, float y2, float xy, const float4 &a,
                                               const float4 &b, float cb, int contributor, Sums9 &s)
{
    float t = fmaf(b.y, xy, a.x);
    t = fmaf(a.y, x, t);
    t = fmaf(a.z, y, t);
    t = fmaf(a.w, x2, t);
    t = fmaf(b.x, y2, t);
    const float oG = __builtin_amdgcn_exp2f(t);            // opacity x Gaussian
    const float araw = fminf(ALPHA_MAX, oG);
    const bool valid = (contributor <= p.last) & !(araw < ALPHA_MIN);
    const float alpha = valid ? araw : 0.0f;
    const float Gh = (CLAMP_STOP ? (valid & !(oG > ALPHA_MAX)) : valid) ? oG : 0.0f;
    const float cd = fmaf(cb, p.d2, fmaf(b.w, p.d1, b.z * p.d0));   // colour . dL/dpixel
    const float oma = 1.0f - alpha;
    const float inv = __builtin_amdgcn_rcpf(oma);
    p.T *= inv;
    const float w = alpha * p.T;
    s.r = fmaf(w, p.d0, s.r); s.g = fmaf(w, p.d1, s.g); s.b = fmaf(w, p.d2, s.b);
    // ZERO_BG (black background, the reference's default): the background's share T_final (bg . d) / (1 - alpha) is zero
    const float dLda = ZERO_BG ? (cd - p.bd) * p.T : fmaf(cd - p.bd, p.T, -(p.tb * inv));
    p.bd = fmaf(alpha, cd, oma * p.bd);
    const float h = Gh * dLda;
    s.h += h;
    s.x = fmaf(h, x, s.x); s.y = fmaf(h, y, s.y);
    s.xx = fmaf(h, x2, s.xx); s.xy = fmaf(h, xy, s.xy); s.yy = fmaf(h, y2, s.yy);
    return valid;
}

// ONE wave per 16x16 tile; lane l owns pixel l of each of the four 8x8 quadrants and walks the quadrants a Gaussian can
// reach one after the other (wave-uniform 4-bit mask from the bbox + ellipse tests), adding its nine partial sums in
// registers: one cross-lane reduction per (tile, Gaussian), two list entries per reduction pass.  The nine totals of
// every list entry are written — with plain, whole-line stores, no atomics — to that instance's own 64-byte row of the
// partial-sum buffer (row index from the sort: gslot); entries nothing reached get a row of zeros.  B2 then adds up
// each Gaussian's rows (they are contiguous and in tile order: the sum order is fixed, the backward is deterministic).
#ifndef GSVC_BWD_WAVES
#define GSVC_BWD_WAVES 4      // 115-120 VGPRs (27 live sums of the four-entry reduction + 28 of pixel state + the 12 per-lane
                              // polynomial constants); at 5 waves (96) the spills cost more than the fifth wave gives (measured)
#endif
// DBG: the timing experiments / lane-efficiency probe selected by the run-time word `dbg` (GSVC_BWD_DEBUG); the production
// instantiation (DBG = false) carries none of their code or registers
template <bool CLAMP_STOP, bool DBG, bool ZERO_BG>
__global__ void __launch_bounds__(64, GSVC_BWD_WAVES) k_blend_bwd_tile(RasterParams st, const int32_t *__restrict__ tile_offsets,
                                                       const int32_t *__restrict__ point_list,
                                                       const uint2 *__restrict__ inst_bbox,
                                                       const int32_t *__restrict__ gslot,
                                                       const GeomRec *__restrict__ geom,
                                                       const float *__restrict__ final_T,
                                                       const int32_t *__restrict__ n_contrib,
                                                       const float *__restrict__ dL_dimage, float *__restrict__ rows,
                                                       const gsvc_raster_counters *__restrict__ counters, int dbg_word)
{
    const int dbg = DBG ? dbg_word : 0;
    __shared__ float4 s_f0[64];     // u v A' B'
    __shared__ float4 s_f1[64];     // C' opacity r g
    __shared__ float2 s_f2[64];     // b, tag = chunk entry | quadrant mask << 8 | list position << 12
    __shared__ float s_out[64][9];  // per chunk entry: sum h, h dx, h dy, h dx^2, h dx dy, h dy^2, w d0, w d1, w d2
    __shared__ float s_park[5][64]; // per lane, across the replay loop: what the flush needs of the lane's own entry (registers
                                    // are what bounds the waves per SIMD here; these six would be live through the whole loop)
    const int lane = threadIdx.x;
    const int tx0 = blockIdx.x * TILE, ty0 = blockIdx.y * TILE;
    const int tile = blockIdx.y * st.gx + blockIdx.x;
    const int HW = st.H * st.W;
    PixState ps[4];
    // the lane's pixel of quadrant q relative to the tile centre: x = (q & 1 ? xb : xa), y = (q >> 1 ? yb : ya), and the products
    const float xa = (float)(lane & 7) - 7.5f, xb = xa + 8.0f, ya = (float)(lane >> 3) - 7.5f, yb = ya + 8.0f;
    const float px_[4] = {xa, xb, xa, xb}, py_[4] = {ya, ya, yb, yb};
    const float px2_[4] = {xa * xa, xb * xb, xa * xa, xb * xb}, py2_[4] = {ya * ya, ya * ya, yb * yb, yb * yb};
    const float pxy_[4] = {xa * ya, xb * ya, xa * yb, xb * yb};
    const float tcx = (float)tx0 + 7.5f, tcy = (float)ty0 + 7.5f;
    int wl[4];
    // every load of the prologue is issued before anything is waited for (one memory round trip): the tile's list bounds,
    // the forward's per-pixel state (tile-major: coalesced) and dL/dpixel — the latter as one 16-byte load per lane and
    // channel (lane l: row l / 4 of the tile, pixels 4 (l % 4) .. + 3), handed to the quadrant layout through LDS
    float Tf[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        Tf[q] = final_T[(tile * 4 + q) * 64 + lane];
        ps[q].last = n_contrib[(tile * 4 + q) * 64 + lane];
    }
    const bool wide = (st.W & 3) == 0;
    if (wide) {
        const int py = ty0 + (lane >> 2), px = tx0 + 4 * (lane & 3);
        const bool in = py < st.H && px < st.W;
        float4 g[3];
#pragma unroll
        for (int c = 0; c < 3; c++)
            g[c] = in ? *reinterpret_cast<const float4 *>(dL_dimage + (size_t)c * HW + (size_t)py * st.W + px)
                      : make_float4(0.f, 0.f, 0.f, 0.f);
        float4 *t0 = s_f0, *t1 = s_f1, *t2 = reinterpret_cast<float4 *>(&s_out[0][0]);
        t0[lane] = g[0]; t1[lane] = g[1]; t2[lane] = g[2];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int o = (8 * (q >> 1) + (lane >> 3)) * TILE + 8 * (q & 1) + (lane & 7);
            ps[q].d0 = reinterpret_cast<const float *>(t0)[o];
            ps[q].d1 = reinterpret_cast<const float *>(t1)[o];
            ps[q].d2 = reinterpret_cast<const float *>(t2)[o];
        }
        __builtin_amdgcn_wave_barrier();
    } else {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int px = tx0 + 8 * (q & 1) + (lane & 7), py = ty0 + 8 * (q >> 1) + (lane >> 3);
            const bool inq = px < st.W && py < st.H;
            const int pix = py * st.W + px;
            ps[q].d0 = inq ? dL_dimage[pix] : 0.f;
            ps[q].d1 = inq ? dL_dimage[HW + pix] : 0.f;
            ps[q].d2 = inq ? dL_dimage[2 * HW + pix] : 0.f;
        }
    }
    const int beg = tile_offsets[tile], end = tile_offsets[tile + 1];
    if (counters->overflow || beg == end) return;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        PixState &p = ps[q];
        p.tb = ZERO_BG ? 0.f : Tf[q] * (st.bg0 * p.d0 + st.bg1 * p.d1 + st.bg2 * p.d2);
        p.T = Tf[q];
        p.bd = 0.f;
        int m = p.last;
#pragma unroll
        for (int k = 32; k >= 1; k >>= 1) m = max(m, __shfl_xor(m, k, 64));
        wl[q] = __builtin_amdgcn_readfirstlane(m);
    }
    const int tile_last = max(max(wl[0], wl[1]), max(wl[2], wl[3]));

    // entries behind the last contributor of every pixel of the tile: nothing to replay, their rows are zero
    for (int k = beg + tile_last + lane; k < end; k += 64) {
        const uint2 bbk = inst_bbox[k];
        if (!bbox_hits_tile(bbk.x, bbk.y, tx0, ty0)) continue;
        float4 *row = reinterpret_cast<float4 *>(rows + (size_t)gslot[k] * ROW_FLOATS);
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        row[0] = z; row[1] = z; row[2] = z; row[3] = z;
    }

    // chunk = list entries [c1-64, c1) from the back; lane e holds entry k = c1 - 1 - e.  The next chunk's list words are
    // requested while the current chunk is replayed.
    uint2 bb_n = make_uint2(0u, 0u);
    int id_n = -1, gs_n = 0;
    {
        const int k = beg + tile_last - 1 - lane;
        if (k >= beg) { bb_n = inst_bbox[k]; id_n = point_list[k]; gs_n = gslot[k]; }
    }
    if (dbg & 4) return;                                      // timing experiment only: prologue + zero rows
    int probe_replays = 0, probe_lanes = 0, probe_entries = 0;
    for (int c1 = beg + tile_last; c1 > beg; c1 -= 64) {
        const int k = c1 - 1 - lane;
        const uint2 bb = bb_n;
        const int id = id_n, gs = gs_n;
        int qm = 0;
        if (k >= beg) {
            const int rel = k - beg;
#pragma unroll
            for (int q = 0; q < 4; q++)
                if (rel < wl[q] && bbox_hits_b(bb, tx0 + 8 * (q & 1), ty0 + 8 * (q >> 1))) qm |= 1 << q;
        }
        // exact ellipse-vs-quadrant test on the bbox survivors (same rule as the forward: raster_common.h)
        float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0;
        float r2x = 0.f;
        const float4 *rec = reinterpret_cast<const float4 *>(geom + (id < 0 ? 0 : id));
        if (qm != 0 && !(dbg & 64)) {
            r0 = rec[0];
            r1 = rec[1];
            r2x = rec[2].x;
        }
        {
            const int kn = c1 - 64 - 1 - lane;      // behind the gathers: waiting for them leaves these in flight
            id_n = -1;
            if (kn >= beg) { bb_n = inst_bbox[kn]; id_n = point_list[kn]; gs_n = gslot[kn]; }
        }
        if (qm != 0 && !(dbg & 32)) {
#pragma unroll
            for (int q = 0; q < 4; q++)
                if ((qm & (1 << q)) && !ellipse_hits_quad(r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, (float)(tx0 + 8 * (q & 1)),
                                                          (float)(ty0 + 8 * (q >> 1))))
                    qm &= ~(1 << q);
        }
        const unsigned long long mask = __ballot(qm != 0);
        const int pos = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
        // conic positive definite with a margin and opacity > 0 (the forward's rule): the chunk takes the polynomial replay; a chunk
        // that holds any other entry (NaN / indefinite conics) the literal one
        const bool safe = r0.z > 0.f && r1.x > 0.f && r0.z * r1.x >= 1.002f * (r0.w * r0.w) && r1.y > 0.f;
        const bool generic = DBG || __ballot(qm != 0 && !safe) != 0ull;
        if (qm != 0) {
            const float Ap = (0.5f * 1.44269504088896340736f) * r0.z, Bp = 1.44269504088896340736f * r0.w,
                        Cp = (0.5f * 1.44269504088896340736f) * r1.x;
            if (generic) {
                s_f0[pos] = make_float4(r0.x, r0.y, Ap, Bp);
                s_f1[pos] = make_float4(Cp, r1.y, r1.z, r1.w);
            } else {
                const float U = r0.x - tcx, V = r0.y - tcy;
                const float n0 = __builtin_amdgcn_logf(r1.y) - (Ap * U * U + Cp * V * V + Bp * U * V);
                s_f0[pos] = make_float4(n0, 2.0f * Ap * U + Bp * V, 2.0f * Cp * V + Bp * U, -Ap);
                s_f1[pos] = make_float4(-Cp, -Bp, r1.z, r1.w);
            }
            s_f2[pos] = make_float2(r2x, __int_as_float(lane | (qm << 8) | ((k - beg + 1) << 12)));
        }
        const int cnt = (dbg & 8) ? 0 : __popcll(mask);      // timing experiment only: chunk overhead without entries
        {
            const bool fl = k >= beg && bbox_hits_tile(bb.x, bb.y, tx0, ty0) && !(dbg & 16);      // this lane writes a row
            s_park[0][lane] = __int_as_float(fl ? gs : -1);
            s_park[1][lane] = __int_as_float(qm != 0 ? pos : -1);
            s_park[2][lane] = r0.x - tcx;
            s_park[3][lane] = r0.y - tcy;
            s_park[4][lane] = qm != 0 ? 1.0f / r1.y : 0.f;
        }
        probe_entries += cnt;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        auto replay = [&](int j, Sums9 &s, int &e) {
            const float4 a = s_f0[j];
            const float4 b = s_f1[j];
            const float2 c = s_f2[j];
            const int tag = __builtin_amdgcn_readfirstlane(__float_as_int(c.y));
            const int contributor = tag >> 12, quads = (tag >> 8) & 0xf;
            e = tag & 0xff;
            s.h = s.x = s.y = s.xx = s.xy = s.yy = s.r = s.g = s.b = 0.f;
            const float dxb = a.x - (tcx + xa), dyb = a.y - (tcy + ya);      // exact: small integers + halves
            if (dbg & 1) { s.h = dxb; s.x = dyb; return; }     // timing experiment only (GSVC_BWD_DEBUG): no replay
#pragma unroll
            for (int q = 0; q < 4; q++)
                if (quads & (1 << q)) {
                    const bool v = bwd_pixel<CLAMP_STOP, ZERO_BG>(ps[q], dxb - (float)(8 * (q & 1)), dyb - (float)(8 * (q >> 1)), a, b, c.x, contributor, s);
                    if (dbg & 128) {      // lane-efficiency probe (tools/bwd_lane_efficiency.py): replays and the lanes they were for
                        probe_replays++;
                        probe_lanes += __popcll(__ballot(v));
                    }
                }
        };
        auto replay_poly = [&](int j, Sums9 &s, int &e) {
            const float4 a = s_f0[j];
            const float4 b = s_f1[j];
            const float2 c = s_f2[j];
            const int tag = __builtin_amdgcn_readfirstlane(__float_as_int(c.y));
            const int contributor = tag >> 12, quads = (tag >> 8) & 0xf;
            e = tag & 0xff;
            s.h = s.x = s.y = s.xx = s.xy = s.yy = s.r = s.g = s.b = 0.f;
#pragma unroll
            for (int q = 0; q < 4; q++)
                if (quads & (1 << q))
                    bwd_pixel_poly<CLAMP_STOP, ZERO_BG>(ps[q], px_[q], py_[q], px2_[q], py2_[q], pxy_[q], a, b, c.x, contributor, s);
        };
        // four entries per reduction pass (r4_pair / r4_tail above); rows of the reduced registers = entries j, j + 2, j + 1, j + 3.
        // Lane (row r, bank b, t): t = 0 stores value {0,2,1,3}[b], t = 1 value {4,6,5,7}[b], (b, t) = (0, 2) the ninth, into the
        // sums of chunk position j + {0,2,1,3}[r]; a group's missing entries (cnt not a multiple of 4) contribute zeros.
        {
            const int r = lane >> 4, bk = (lane >> 2) & 3, t = lane & 3;
            const int perm_r = ((r & 1) << 1) | (r >> 1), perm_b = ((bk & 1) << 1) | (bk >> 1);
            const bool wr = t < 2 || (t == 2 && bk == 0);
            const int off = perm_r * 9 + (t == 0 ? perm_b : (t == 1 ? 4 + perm_b : 8));
            float *so = &s_out[0][0];
            auto pass = [&](auto &&replay) {
            for (int j = 0; j < cnt; j += 4) {
                Sums9 sa, sb, sc, sd;
                int e_;
                replay(j, sa, e_);
                if (j + 1 < cnt) replay(j + 1, sb, e_); else sb.h = sb.x = sb.y = sb.xx = sb.xy = sb.yy = sb.r = sb.g = sb.b = 0.f;
                if (!(dbg & 2)) r4_pair(sa, sb, false);      // (dbg & 2: timing experiment only, no reduction)
                if (j + 2 < cnt) replay(j + 2, sc, e_); else sc.h = sc.x = sc.y = sc.xx = sc.xy = sc.yy = sc.r = sc.g = sc.b = 0.f;
                if (j + 3 < cnt) replay(j + 3, sd, e_); else sd.h = sd.x = sd.y = sd.xx = sd.xy = sd.yy = sd.r = sd.g = sd.b = 0.f;
                if (!(dbg & 2)) {
                    r4_pair(sc, sd, false);
                    r4_pair(sa, sc, true);
                    r4_tail(sa);
                } else {
                    sa.h += sb.h + sc.h + sd.h;
                }
                const float v = t == 0 ? sa.xx : (t == 1 ? sa.g : sa.b);
                if (wr) so[j * 9 + off] = v;
            }
            };
            if (generic) pass(replay); else pass(replay_poly);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // flush: every entry of the chunk whose alpha box touches the tile writes its whole row (replayed entries their sums,
        // the others zeros)
        const int gs_f = __float_as_int(s_park[0][lane]), pos_f = __float_as_int(s_park[1][lane]);
        if (gs_f >= 0) {
            float v[9];
#pragma unroll
            for (int c = 0; c < 9; c++) v[c] = pos_f >= 0 ? s_out[pos_f][c] : 0.f;      // sums by chunk position
            if (!generic) {
                // the polynomial replay's moments are of o h about the tile centre: shift to the Gaussian's centre, take o out
                const float U = s_park[2][lane], V = s_park[3][lane], io = s_park[4][lane];
                const float m0 = v[0], mx = v[1], my = v[2], mxx = v[3], mxy = v[4], myy = v[5];
                v[0] = m0 * io;
                v[1] = fmaf(U, m0, -mx) * io;
                v[2] = fmaf(V, m0, -my) * io;
                v[3] = fmaf(U, fmaf(U, m0, -2.0f * mx), mxx) * io;
                v[4] = (fmaf(U, fmaf(V, m0, -my), mxy) - V * mx) * io;
                v[5] = fmaf(V, fmaf(V, m0, -2.0f * my), myy) * io;
            }
            float4 *row = reinterpret_cast<float4 *>(rows + (size_t)gs_f * ROW_FLOATS);
            row[0] = make_float4(v[0], v[1], v[2], v[3]);
            row[1] = make_float4(v[4], v[5], v[6], v[7]);
            row[2] = make_float4(v[8], 0.f, 0.f, 0.f);
            row[3] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        __builtin_amdgcn_wave_barrier();
    }
    if ((dbg & 128) && lane == 0) {
        // bytes 64 .. 87 of the 256-byte counters block (unused by gsvc_raster_counters): replays, valid lanes, replayed entries
        unsigned long long *pr = reinterpret_cast<unsigned long long *>(const_cast<gsvc_raster_counters *>(counters)) + 8;
        atomicAdd(pr + 0, (unsigned long long)probe_replays);
        atomicAdd(pr + 1, (unsigned long long)probe_lanes);
        atomicAdd(pr + 2, (unsigned long long)probe_entries);
    }
}

#ifndef GSVC_GBWD_COOP_MIN_ROWS
#define GSVC_GBWD_COOP_MIN_ROWS 64
#endif
constexpr int GBWD_COOP_MIN_ROWS = GSVC_GBWD_COOP_MIN_ROWS;      // rectangles from this many tiles are added up by the whole wave

__global__ void __launch_bounds__(256) k_gaussian_bwd(RasterParams st, int P, const float *__restrict__ means3D,
                                                      const float *__restrict__ scales,
                                                      const float *__restrict__ rotations,
                                                      const float *__restrict__ opacities,
                                                      const int32_t *__restrict__ radii, const GeomRec *__restrict__ geom,
                                                      const float *__restrict__ rows,
                                                      const gsvc_raster_counters *__restrict__ counters,
                                                      float *__restrict__ dL_dmeans3D, float *__restrict__ dL_dmeans2D,
                                                      float *__restrict__ dL_dcolors, float *__restrict__ dL_dopacities,
                                                      float *__restrict__ dL_dscales, float *__restrict__ dL_drotations)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    float g3[3] = {0, 0, 0}, g2[3] = {0, 0, 0}, gc[3] = {0, 0, 0}, gs[3] = {0, 0, 0}, gq[4] = {0, 0, 0, 0}, go = 0.f;
    const bool overflow = counters->overflow != 0;
    const bool act = i < P && !overflow && radii[i] > 0;
    // this Gaussian's rows of the partial-sum buffer: one per tile of its rectangle, contiguous, added in a FIXED order (tile order by
    // the Gaussian's own lane; for a large rectangle by the whole wave: lane l adds rows l, l + 64, ... in tile order, then a fixed
    // butterfly adds the lanes) — the backward stays bit-repeatable.  Late in a fit a Gaussian covers ~57 tiles on average and
    // hundreds at the tail (profiles/r05 late_stage_profile): a lane walking 300 rows alone, four loads per dependent round trip, kept
    // its 63 neighbours waiting (measured: 312 us per launch at 65 k active Gaussians and 3.7 M instances, 4.6 ms of lane-serial
    // latency for ~140 MB of rows).
    int n_rows = 0, tw = 1, rx0 = 0, ry0 = 0, goff = 0;
    uint32_t abx = 0u, aby = 0u;
    if (act) {
        const float4 w2 = reinterpret_cast<const float4 *>(geom + i)[2];
        const float4 w3 = reinterpret_cast<const float4 *>(geom + i)[3];
        const uint32_t rx = __float_as_uint(w3.x), ry = __float_as_uint(w3.y);
        rx0 = (int)(rx & 0xffff); ry0 = (int)(ry & 0xffff);
        tw = (int)(rx >> 16) - rx0;
        n_rows = tw * ((int)(ry >> 16) - ry0);
        goff = __float_as_int(w2.w);
        abx = __float_as_uint(w2.y); aby = __float_as_uint(w2.z);      // alpha bounding box
    }
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
    float a2 = 0.f;
    const bool big = act && n_rows >= GBWD_COOP_MIN_ROWS;
    if (act && !big) {
        const float4 *row = reinterpret_cast<const float4 *>(rows + (size_t)goff * ROW_FLOATS);
        int jx = 0, tpx = rx0 * TILE, tpy = ry0 * TILE;
        // four rows per round: their loads are issued together, then added in tile order (a load per iteration made the loop one
        // dependent memory round trip per tile of the rectangle)
        for (int j = 0; j < n_rows; j += 4, row += 4 * (ROW_FLOATS / 4)) {
            float4 x0[4], x1[4];
            float x2[4];
            bool hit[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                hit[u] = j + u < n_rows && bbox_hits_tile(abx, aby, tpx, tpy);     // rows exist only where the box touches the tile (B1)
                if (++jx == tw) { jx = 0; tpx = rx0 * TILE; tpy += TILE; } else tpx += TILE;
                const float4 *r4 = row + u * (ROW_FLOATS / 4);
                x0[u] = hit[u] ? r4[0] : make_float4(0.f, 0.f, 0.f, 0.f);
                x1[u] = hit[u] ? r4[1] : make_float4(0.f, 0.f, 0.f, 0.f);
                x2[u] = hit[u] ? r4[2].x : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (!hit[u]) continue;          // (adding a skipped row's zeros could turn a -0 sum into +0: keep the old sums bit for bit)
                a0.x += x0[u].x; a0.y += x0[u].y; a0.z += x0[u].z; a0.w += x0[u].w;
                a1.x += x1[u].x; a1.y += x1[u].y; a1.z += x1[u].z; a1.w += x1[u].w;
                a2 += x2[u];
            }
        }
    }
    // large rectangles: the wave adds one Gaussian's rows together (every lane of the wave takes part, also those past P)
    for (unsigned long long todo = __ballot(big); todo != 0ull; todo &= todo - 1ull) {
        const int src = __builtin_ctzll(todo);
        const int n = __shfl(n_rows, src, 64), w = __shfl(tw, src, 64), x0t = __shfl(rx0, src, 64), y0t = __shfl(ry0, src, 64);
        const int gof = __shfl(goff, src, 64);
        const uint32_t bx = (uint32_t)__shfl((int)abx, src, 64), by = (uint32_t)__shfl((int)aby, src, 64);
        const float4 *row0 = reinterpret_cast<const float4 *>(rows + (size_t)gof * ROW_FLOATS);
        float v[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int j = lane; j < n; j += 128) {          // two rows in flight per lane
            float4 p0[2], p1[2];
            float p2[2];
            bool hit[2];
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int jj = j + 64 * u;
                const int ty = jj / w, tx = jj - ty * w;
                hit[u] = jj < n && bbox_hits_tile(bx, by, (x0t + tx) * TILE, (y0t + ty) * TILE);
                const float4 *r4 = row0 + (size_t)jj * (ROW_FLOATS / 4);
                p0[u] = hit[u] ? r4[0] : make_float4(0.f, 0.f, 0.f, 0.f);
                p1[u] = hit[u] ? r4[1] : make_float4(0.f, 0.f, 0.f, 0.f);
                p2[u] = hit[u] ? r4[2].x : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 2; u++) {
                if (!hit[u]) continue;
                v[0] += p0[u].x; v[1] += p0[u].y; v[2] += p0[u].z; v[3] += p0[u].w;
                v[4] += p1[u].x; v[5] += p1[u].y; v[6] += p1[u].z; v[7] += p1[u].w;
                v[8] += p2[u];
            }
        }
#pragma unroll
        for (int c = 0; c < 9; c++)
#pragma unroll
            for (int k = 32; k >= 1; k >>= 1) v[c] += __shfl_xor(v[c], k, 64);
        if (lane == src) {
            a0 = make_float4(v[0], v[1], v[2], v[3]);
            a1 = make_float4(v[4], v[5], v[6], v[7]);
            a2 = v[8];
        }
    }
    if (i >= P) return;
    if (act) {
        float du, dv, dA, dB, dC;
        PreOut o;
        preprocess_gaussian(st, means3D[3 * i], means3D[3 * i + 1], means3D[3 * i + 2], scales[3 * i], scales[3 * i + 1],
                            scales[3 * i + 2], rotations[4 * i], rotations[4 * i + 1], rotations[4 * i + 2],
                            rotations[4 * i + 3], o);
        // accumulator holds moments of h = G dL/dalpha: (sum h, h dx, h dy, h dx^2, h dx dy, h dy^2, colour grads)
        const float op = opacities[i];
        const float sh = a0.x, sx = a0.y, sy = a0.z, sxx = a0.w, sxy = a1.x, syy = a1.y;
        go = sh;
        du = -op * (o.A * sx + o.B * sy);
        dv = -op * (o.C * sy + o.B * sx);
        dA = -0.5f * op * sxx; dB = -op * sxy; dC = -0.5f * op * syy;
        gc[0] = a1.z; gc[1] = a1.w; gc[2] = a2;
        const bool pixel_units = st.flags & GSVC_RASTER_MEANS2D_PIXEL_UNITS;
        g2[0] = pixel_units ? du : du * 0.5f * (float)st.W;
        g2[1] = pixel_units ? dv : dv * 0.5f * (float)st.H;
        const float *M = st.m;
        for (int j = 0; j < 3; j++) g3[j] = st.scale * (M[j] * du + M[4 + j] * dv);

        const float s0 = scales[3 * i], s1 = scales[3 * i + 1], s2 = scales[3 * i + 2];
        const float4 q = reinterpret_cast<const float4 *>(rotations)[i];
        const float ca = o.a, cb = o.b, cc = o.c;
        const float det = ca * cc - cb * cb;
        const float inv2 = 1.0f / (det * det + 1e-7f);
        const float dLa = inv2 * (-cc * cc * dA + cb * cc * dB + (det - ca * cc) * dC);
        const float dLc = inv2 * (-ca * ca * dC + ca * cb * dB + (det - ca * cc) * dA);
        const float dLb = inv2 * (2.0f * cb * cc * dA - (det + 2.0f * cb * cb) * dB + 2.0f * ca * cb * dC);
        const float T0[3] = {st.scale * M[0], st.scale * M[1], st.scale * M[2]};
        const float T1[3] = {st.scale * M[4], st.scale * M[5], st.scale * M[6]};
        float G3[3][3];
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++)
                G3[r][c] = T0[r] * dLa * T0[c] + 0.5f * dLb * (T0[r] * T1[c] + T1[r] * T0[c]) + T1[r] * dLc * T1[c];
        const float qr = q.x, qx = q.y, qy = q.z, qz = q.w;
        const float R[3][3] = {{1.f - 2.f * (qy * qy + qz * qz), 2.f * (qx * qy - qr * qz), 2.f * (qx * qz + qr * qy)},
                               {2.f * (qx * qy + qr * qz), 1.f - 2.f * (qx * qx + qz * qz), 2.f * (qy * qz - qr * qx)},
                               {2.f * (qx * qz - qr * qy), 2.f * (qy * qz + qr * qx), 1.f - 2.f * (qx * qx + qy * qy)}};
        const float S[3] = {st.scale_modifier * s0, st.scale_modifier * s1, st.scale_modifier * s2};
        float dR[3][3];
        for (int k = 0; k < 3; k++) {
            float sv = 0.f;
            for (int r = 0; r < 3; r++) {
                float v = 0.f;
                for (int c = 0; c < 3; c++) v += G3[r][c] * R[c][k] * S[k];
                v *= 2.0f;            // dL/dL[r][k], L = R diag(S)
                sv += v * R[r][k];
                dR[r][k] = v * S[k];
            }
            gs[k] = sv * st.scale_modifier;
        }
        gq[0] = 2.f * (-qz * dR[0][1] + qy * dR[0][2] + qz * dR[1][0] - qx * dR[1][2] - qy * dR[2][0] + qx * dR[2][1]);
        gq[1] = 2.f * (qy * dR[0][1] + qz * dR[0][2] + qy * dR[1][0] - 2.f * qx * dR[1][1] - qr * dR[1][2] + qz * dR[2][0] +
                       qr * dR[2][1] - 2.f * qx * dR[2][2]);
        gq[2] = 2.f * (-2.f * qy * dR[0][0] + qx * dR[0][1] + qr * dR[0][2] + qx * dR[1][0] + qz * dR[1][2] - qr * dR[2][0] +
                       qz * dR[2][1] - 2.f * qy * dR[2][2]);
        gq[3] = 2.f * (-2.f * qz * dR[0][0] - qr * dR[0][1] + qx * dR[0][2] + qr * dR[1][0] - 2.f * qz * dR[1][1] +
                       qy * dR[1][2] + qx * dR[2][0] + qy * dR[2][1]);
    }
    for (int j = 0; j < 3; j++) {
        dL_dmeans3D[3 * i + j] = g3[j];
        dL_dmeans2D[3 * i + j] = g2[j];
        dL_dcolors[3 * i + j] = gc[j];
        dL_dscales[3 * i + j] = gs[j];
    }
    dL_dopacities[i] = go;
    reinterpret_cast<float4 *>(dL_drotations)[i] = make_float4(gq[0], gq[1], gq[2], gq[3]);
}

}  // namespace gsvc

using namespace gsvc;

extern "C" int64_t gsvc_raster_backward_scratch_bytes(int64_t P, int64_t max_instances)
{
    (void)P;
    return (max_instances > 0 ? max_instances : 1) * (int64_t)(ROW_FLOATS * sizeof(float));
}

extern "C" int gsvc_raster_backward(const gsvc_raster_settings *settings, int64_t P, int64_t max_instances,
                                    const float *means3D, const float *colors, const float *opacities,
                                    const float *scales, const float *rotations, const int32_t *radii, const void *geom,
                                    const void *binning, const void *image_state, const float *dL_dimage,
                                    float *dL_dmeans3D, float *dL_dmeans2D, float *dL_dcolors, float *dL_dopacities,
                                    float *dL_dscales, float *dL_drotations, void *scratch, void *stream)
{
    (void)colors;
    GSVC_REQUIRE(settings != nullptr, "raster_backward: settings is NULL");
    GSVC_REQUIRE(P >= 0 && P < (int64_t)1 << 31 && max_instances >= 0, "raster_backward: bad sizes");
    if (P == 0) return GSVC_OK;
    GSVC_REQUIRE(means3D && opacities && scales && rotations && radii && geom && binning && image_state && dL_dimage && scratch,
                 "raster_backward: NULL input pointer");
    GSVC_REQUIRE(dL_dmeans3D && dL_dmeans2D && dL_dcolors && dL_dopacities && dL_dscales && dL_drotations,
                 "raster_backward: NULL output pointer");
    const RasterParams p = make_params(*settings);
    const RasterLayout L = raster_layout(*settings, P, max_instances);
    hipStream_t s = (hipStream_t)stream;
    const char *bin = (const char *)binning;
    auto *counters = (const gsvc_raster_counters *)(bin + L.off_counters);
    auto *tile_offsets = (const int32_t *)(bin + L.off_tile_offsets);
    auto *point_list = (const int32_t *)(bin + L.off_point_list);
    auto *inst_bbox = (const uint2 *)(bin + L.off_inst_bbox);
    auto *final_T = (const float *)((const char *)image_state + L.off_final_T);
    auto *n_contrib = (const int32_t *)((const char *)image_state + L.off_n_contrib);
    auto *gslot = (const int32_t *)(bin + L.off_gslot);
    static const int dbg_env = getenv("GSVC_BWD_DEBUG") ? atoi(getenv("GSVC_BWD_DEBUG")) : 0;   // kernel-timing experiments
    const int dbg = dbg_env | (bwd_probe_enabled() ? 128 : 0);     // gsvc_profile_enable(2): the lane-efficiency probe, at run time
    {
        ProfScope _prof("k_blend_bwd", s);
#define GSVC_BWD_LAUNCH(CS, DB, ZB) hipLaunchKernelGGL((k_blend_bwd_tile<CS, DB, ZB>), dim3(L.gx, L.gy), dim3(64), 0, s, p, tile_offsets, point_list, \
            inst_bbox, gslot, (const GeomRec *)geom, final_T, n_contrib, dL_dimage, (float *)scratch, counters, dbg)
        const bool cs = p.flags & GSVC_RASTER_CLAMP_STOPS_GRADIENT;
        const bool zb = p.bg0 == 0.f && p.bg1 == 0.f && p.bg2 == 0.f;
        if (dbg) { if (cs) GSVC_BWD_LAUNCH(true, true, false); else GSVC_BWD_LAUNCH(false, true, false); }
        else if (zb) { if (cs) GSVC_BWD_LAUNCH(true, false, true); else GSVC_BWD_LAUNCH(false, false, true); }
        else { if (cs) GSVC_BWD_LAUNCH(true, false, false); else GSVC_BWD_LAUNCH(false, false, false); }
#undef GSVC_BWD_LAUNCH
    }
    {
        ProfScope _prof("k_gaussian_bwd", s);
        hipLaunchKernelGGL(k_gaussian_bwd, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, s, p, (int)P, means3D,
                           scales, rotations, opacities, radii, (const GeomRec *)geom, (const float *)scratch, counters,
                           dL_dmeans3D, dL_dmeans2D, dL_dcolors, dL_dopacities, dL_dscales, dL_drotations);
    }
    return check_launch("raster_backward");
}

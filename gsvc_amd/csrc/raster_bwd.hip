// gsvc_amd/csrc/raster_bwd.hip — backward of the orthographic tile rasterizer, gfx950.
//
// Replaces the implicit autograd backward of the external CUDA extension (consumers: reference
// pipeline/train.py:462 `loss.backward()`, scene/gaussian_model.py:1311 `viewspace_points.grad`).
//
//   B1 blend_bwd     one wave per 16x16 tile (lane l owns pixel l of each 8x8 quadrant) walks the tile's sorted list
//                    back to front in chunks of 64 entries: bbox + exact ellipse test per quadrant, replay of the alpha
//                    compositing per pixel, per Gaussian nine sums (moments of h = G dL/dalpha and the colour
//                    gradients) reduced over the wave with permlane swaps + DPP, flushed per chunk with 36-byte
//                    atomic segments into the per-Gaussian accumulator.
//   B2 gaussian_bwd  one lane per Gaussian: conic -> 2-D covariance -> 3-D covariance -> (scale, quaternion),
//                    screen-space -> world mean (constant orthographic Jacobian, no covariance->mean term).
#include "raster_common.h"

namespace gsvc {

constexpr int ACC_STRIDE = 16;  // floats per Gaussian in the accumulator (9 used, 64-B rows)

// Nine per-Gaussian sums over the wave in 24 cross-lane instructions (a plain DPP butterfly needs 54): values 0..7 are
// reduced "transposed" — every halving step also halves the number of live registers (v_permlane32_swap / v_permlane16_swap move half of one register into the
// idle half of its partner, one add then reduces two values at once) — so that afterwards a0 holds, in lane group
// g = lane / 8, the wave total of value g; the ninth value is reduced the classic way into lane 63.
__device__ __forceinline__ void wave_sum9_spread(float &a0, float &a1, float &a2, float &a3, float &a4, float &a5,
                                                 float &a6, float &a7, float &a8)
{
    const unsigned long long odd_half_rows = 0xFF00FF00FF00FF00ull;
    asm volatile("s_nop 1\n"
                 "v_permlane32_swap_b32 %0, %4\n"
                 "v_permlane32_swap_b32 %1, %5\n"
                 "v_permlane32_swap_b32 %2, %6\n"
                 "v_permlane32_swap_b32 %3, %7\n"
                 "v_add_f32_dpp %8, %8, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                 "s_nop 1\n"
                 "v_add_f32 %0, %0, %4\n"
                 "v_add_f32 %1, %1, %5\n"
                 "v_add_f32 %2, %2, %6\n"
                 "v_add_f32 %3, %3, %7\n"
                 "v_add_f32_dpp %8, %8, %8 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
                 "s_nop 1\n"
                 "v_permlane16_swap_b32 %0, %2\n"
                 "v_permlane16_swap_b32 %1, %3\n"
                 "v_add_f32_dpp %8, %8, %8 row_shr:4 row_mask:0xf bank_mask:0xf\n"
                 "s_nop 1\n"
                 "v_add_f32 %0, %0, %2\n"
                 "v_add_f32 %1, %1, %3\n"
                 "v_add_f32_dpp %8, %8, %8 row_shr:8 row_mask:0xf bank_mask:0xf\n"
                 "s_nop 1\n"
                 "v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n"
                 "v_add_f32_dpp %1, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf\n"
                 "v_add_f32_dpp %8, %8, %8 row_bcast:15 row_mask:0xa bank_mask:0xf\n"
                 "s_nop 1\n"
                 "v_cndmask_b32_e64 %0, %0, %1, %9\n"
                 "v_add_f32_dpp %8, %8, %8 row_bcast:31 row_mask:0xc bank_mask:0xf\n"
                 "s_nop 1\n"
                 "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                 "s_nop 1\n"
                 "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
                 "s_nop 1\n"
                 "v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(a8)
                 : "s"(odd_half_rows));
}

__device__ __forceinline__ int sext16b(uint32_t v) { return (int)(int16_t)(v & 0xffffu); }

__device__ __forceinline__ bool bbox_hits_b(uint2 bb, int qx0, int qy0)
{
    return !(sext16b(bb.x) > qx0 + 7 || sext16b(bb.x >> 16) < qx0 || sext16b(bb.y) > qy0 + 7 || sext16b(bb.y >> 16) < qy0);
}

// ONE wave per 16x16 tile; lane l owns pixel l of each of the four 8x8 quadrants and
// walks the quadrants a Gaussian can reach one after the other (wave-uniform 4-bit mask from the bbox test), adding
// its nine partial sums in registers.  One DPP reduction per (tile, Gaussian) instead of one per (quadrant,
// Gaussian), no cross-wave combine, no barriers; a chunk's 64 x 9 sums are flushed with nine 64-lane atomic
// wave-instructions (36 contiguous bytes per list entry).
struct PixState {
    float T, tb, d0, d1, d2, behind0, behind1, behind2;  // tb = final_T * (bg . dL/dpixel)
    int last;
};

__device__ __forceinline__ void bwd_pixel(PixState &p, bool inq, float dx, float dy, const float4 &a, const float4 &b,
                                          float cb, int contributor, float &s_h, float &s_x, float &s_y, float &s_xx,
                                          float &s_xy, float &s_yy, float &s_r, float &s_g, float &s_b)
{
    const float pw = a.z * dx * dx + b.x * dy * dy + a.w * dx * dy;   // -power*log2(e), conic pre-scaled
    const float G = __builtin_amdgcn_exp2f(-pw);
    const float alpha = fminf(ALPHA_MAX, b.y * G);
    const bool valid = inq && (contributor <= p.last) && !(pw < 0.0f) && !(alpha < ALPHA_MIN);
    if (valid) {
        const float oma = 1.0f - alpha;
        const float inv = __builtin_amdgcn_rcpf(oma);
        p.T *= inv;
        const float w = alpha * p.T;
        // `behind` = colour composited behind this Gaussian (everything already visited)
        float dLda = (b.z - p.behind0) * p.d0 + (b.w - p.behind1) * p.d1 + (cb - p.behind2) * p.d2;
        s_r += w * p.d0; s_g += w * p.d1; s_b += w * p.d2;
        dLda = dLda * p.T - p.tb * inv;
        p.behind0 = alpha * b.z + oma * p.behind0;
        p.behind1 = alpha * b.w + oma * p.behind1;
        p.behind2 = alpha * cb + oma * p.behind2;
        // moments of h = G * dL/dalpha; the conic / opacity factors are applied once per Gaussian in B2
        const float h = G * dLda;
        const float hx = h * dx, hy = h * dy;
        s_h += h; s_x += hx; s_y += hy; s_xx += hx * dx; s_xy += hx * dy; s_yy += hy * dy;
    }
}

__global__ void __launch_bounds__(64, 4) k_blend_bwd_tile(RasterParams st, const int32_t *__restrict__ tile_offsets,
                                                       const int32_t *__restrict__ point_list,
                                                       const uint2 *__restrict__ inst_bbox,
                                                       const GeomRec *__restrict__ geom,
                                                       const float *__restrict__ final_T,
                                                       const int32_t *__restrict__ n_contrib,
                                                       const float *__restrict__ dL_dimage, float *__restrict__ acc,
                                                       const gsvc_raster_counters *__restrict__ counters)
{
    __shared__ float4 s_f0[64];     // u v A' B'
    __shared__ float4 s_f1[64];     // C' opacity r g
    __shared__ float2 s_f2[64];     // b, tag = chunk entry | quadrant mask << 8 | list position << 12
    __shared__ float s_out[64][9];  // per chunk entry: sum h, h dx, h dy, h dx^2, h dx dy, h dy^2, w d0, w d1, w d2
    __shared__ int s_id[64];
    if (counters->overflow) return;
    const int lane = threadIdx.x;
    const int tx0 = blockIdx.x * TILE, ty0 = blockIdx.y * TILE;
    const int tile = blockIdx.y * st.gx + blockIdx.x;
    const int beg = tile_offsets[tile];
    const int HW = st.H * st.W;
    PixState ps[4];
    const float fx0 = (float)(tx0 + (lane & 7)), fy0 = (float)(ty0 + (lane >> 3));
    bool inq[4];
    int wl[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int px = tx0 + 8 * (q & 1) + (lane & 7), py = ty0 + 8 * (q >> 1) + (lane >> 3);
        inq[q] = px < st.W && py < st.H;
        const int pix = py * st.W + px;
        PixState &p = ps[q];
        const float Tf = inq[q] ? final_T[pix] : 0.f;
        p.last = inq[q] ? n_contrib[pix] : 0;
        p.d0 = inq[q] ? dL_dimage[pix] : 0.f;
        p.d1 = inq[q] ? dL_dimage[HW + pix] : 0.f;
        p.d2 = inq[q] ? dL_dimage[2 * HW + pix] : 0.f;
        p.tb = Tf * (st.bg0 * p.d0 + st.bg1 * p.d1 + st.bg2 * p.d2);
        p.T = Tf;
        p.behind0 = p.behind1 = p.behind2 = 0.f;
        int m = p.last;
#pragma unroll
        for (int k = 32; k >= 1; k >>= 1) m = max(m, __shfl_xor(m, k, 64));
        wl[q] = __builtin_amdgcn_readfirstlane(m);
    }
    const int tile_last = max(max(wl[0], wl[1]), max(wl[2], wl[3]));

    for (int c1 = beg + tile_last; c1 > beg; c1 -= 64) {
        // chunk = list entries [c1-64, c1) from the back; chunk-local index e = c1 - 1 - k
        const int k = c1 - 1 - lane;
        int qm = 0, id = -1;
        if (k >= beg) {
            const uint2 bb = inst_bbox[k];
            id = point_list[k];
            const int rel = k - beg;
#pragma unroll
            for (int q = 0; q < 4; q++)
                if (rel < wl[q] && bbox_hits_b(bb, tx0 + 8 * (q & 1), ty0 + 8 * (q >> 1))) qm |= 1 << q;
        }
        s_id[lane] = id;
        if (__ballot(qm != 0) == 0ull) continue;
        // exact ellipse-vs-quadrant test on the bbox survivors (same rule as the forward: raster_common.h)
        float4 r0, r1;
        const float4 *rec = reinterpret_cast<const float4 *>(geom + (id < 0 ? 0 : id));
        if (qm != 0) {
            r0 = rec[0];
            r1 = rec[1];
#pragma unroll
            for (int q = 0; q < 4; q++)
                if ((qm & (1 << q)) && !ellipse_hits_quad(r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, (float)(tx0 + 8 * (q & 1)),
                                                          (float)(ty0 + 8 * (q >> 1))))
                    qm &= ~(1 << q);
        }
        const unsigned long long mask = __ballot(qm != 0);
        if (mask == 0ull) continue;
        if (qm != 0) {
            const int pos = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
            s_f0[pos] = make_float4(r0.x, r0.y, (0.5f * 1.44269504088896340736f) * r0.z, 1.44269504088896340736f * r0.w);
            s_f1[pos] = make_float4((0.5f * 1.44269504088896340736f) * r1.x, r1.y, r1.z, r1.w);
            s_f2[pos] = make_float2(rec[2].x, __int_as_float(lane | (qm << 8) | ((k - beg + 1) << 12)));
        }
        const int cnt = __popcll(mask);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        unsigned long long touched = 0ull;
        for (int j = 0; j < cnt; j++) {
            const float4 a = s_f0[j];
            const float4 b = s_f1[j];
            const float2 c = s_f2[j];
            const int tag = __builtin_amdgcn_readfirstlane(__float_as_int(c.y));
            const int contributor = tag >> 12, quads = (tag >> 8) & 0xf, e = tag & 0xff;
            float s_h = 0.f, s_x = 0.f, s_y = 0.f, s_xx = 0.f, s_xy = 0.f, s_yy = 0.f, s_r = 0.f, s_g = 0.f, s_b = 0.f;
            const float dxb = a.x - fx0, dyb = a.y - fy0;
#pragma unroll
            for (int q = 0; q < 4; q++)
                if (quads & (1 << q))
                    bwd_pixel(ps[q], inq[q], dxb - (float)(8 * (q & 1)), dyb - (float)(8 * (q >> 1)), a, b, c.x, contributor,
                              s_h, s_x, s_y, s_xx, s_xy, s_yy, s_r, s_g, s_b);
            wave_sum9_spread(s_h, s_x, s_y, s_xx, s_xy, s_yy, s_r, s_g, s_b);
            touched |= 1ull << e;
            // lane 8g holds the total of value g (g < 8) in s_h's register, lane 63 the total of the ninth
            if ((lane & 7) == 0) s_out[e][lane >> 3] = s_h;
            if (lane == 63) s_out[e][8] = s_b;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // flush: lane -> (entry, component); 9 consecutive lanes = one 36-byte atomic segment
#pragma unroll
        for (int r = 0; r < 9; r++) {
            const int v = r * 64 + lane;
            const int e = v / 9, comp = v - e * 9;
            if ((touched >> e) & 1ull) atomicAdd(acc + (size_t)s_id[e] * ACC_STRIDE + comp, s_out[e][comp]);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

__global__ void __launch_bounds__(256) k_gaussian_bwd(RasterParams st, int P, const float *__restrict__ means3D,
                                                      const float *__restrict__ scales,
                                                      const float *__restrict__ rotations,
                                                      const float *__restrict__ opacities,
                                                      const int32_t *__restrict__ radii, const float *__restrict__ acc,
                                                      float *__restrict__ dL_dmeans3D, float *__restrict__ dL_dmeans2D,
                                                      float *__restrict__ dL_dcolors, float *__restrict__ dL_dopacities,
                                                      float *__restrict__ dL_dscales, float *__restrict__ dL_drotations)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    float g3[3] = {0, 0, 0}, g2[3] = {0, 0, 0}, gc[3] = {0, 0, 0}, gs[3] = {0, 0, 0}, gq[4] = {0, 0, 0, 0}, go = 0.f;
    if (radii[i] > 0) {
        const float4 a0 = reinterpret_cast<const float4 *>(acc + (size_t)i * ACC_STRIDE)[0];
        const float4 a1 = reinterpret_cast<const float4 *>(acc + (size_t)i * ACC_STRIDE)[1];
        const float a2 = acc[(size_t)i * ACC_STRIDE + 8];
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
        g2[0] = du * 0.5f * (float)st.W;
        g2[1] = dv * 0.5f * (float)st.H;
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
    if (hipMemsetAsync(scratch, 0, (size_t)P * ACC_STRIDE * sizeof(float), s) != hipSuccess) {
        set_error("raster_backward: hipMemsetAsync failed");
        return GSVC_E_LAUNCH;
    }
    {
        ProfScope _prof("k_blend_bwd", s);
        hipLaunchKernelGGL(k_blend_bwd_tile, dim3(L.gx, L.gy), dim3(64), 0, s, p, tile_offsets, point_list, inst_bbox,
                           (const GeomRec *)geom, final_T, n_contrib, dL_dimage, (float *)scratch, counters);
    }
    {
        ProfScope _prof("k_gaussian_bwd", s);
        hipLaunchKernelGGL(k_gaussian_bwd, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, s, p, (int)P, means3D,
                           scales, rotations, opacities, radii, (const float *)scratch, dL_dmeans3D, dL_dmeans2D,
                           dL_dcolors, dL_dopacities, dL_dscales, dL_drotations);
    }
    return check_launch("raster_backward");
}

// gsvc_amd/csrc/raster_bwd.hip — backward of the orthographic tile rasterizer, gfx950.
//
// Replaces the implicit autograd backward of the external CUDA extension (consumers: reference
// pipeline/train.py:462 `loss.backward()`, scene/gaussian_model.py:1311 `viewspace_points.grad`).
//
//   B1 blend_bwd     one 256-lane workgroup per 16x16 tile walks the tile's sorted list back to front
//                    (LDS-staged batches), replays alpha compositing per pixel and produces, per Gaussian,
//                    nine partial sums (d/du, d/dv, d/dA, d/dB, d/dC, d/dopacity, d/drgb).  Each wave
//                    reduces its 64 pixels in registers and issues ONE atomic wave-instruction (9 lanes,
//                    36 contiguous bytes) into the per-Gaussian accumulator; waves none of whose pixels
//                    touched the Gaussian skip both.
//   B2 gaussian_bwd  one lane per Gaussian: conic -> 2-D covariance -> 3-D covariance -> (scale, quaternion),
//                    screen-space -> world mean (constant orthographic Jacobian, no covariance->mean term).
#include "raster_common.h"

namespace gsvc {

constexpr int ACC_STRIDE = 16;  // floats per Gaussian in the accumulator (9 used, 64-B rows)

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

__global__ void __launch_bounds__(256) k_blend_bwd(RasterParams st, const int32_t *__restrict__ tile_offsets,
                                                   const int32_t *__restrict__ point_list,
                                                   const GeomRec *__restrict__ geom,
                                                   const float *__restrict__ final_T,
                                                   const int32_t *__restrict__ n_contrib,
                                                   const float *__restrict__ dL_dimage, float *__restrict__ acc,
                                                   const gsvc_raster_counters *__restrict__ counters)
{
    __shared__ int sid[256];
    __shared__ float4 s0[256];  // u v A B
    __shared__ float4 s1[256];  // C opacity r g
    __shared__ float s2[256];   // b
    if (counters->overflow) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int lx = tid & 15, ly = tid >> 4;
    const int tile = blockIdx.y * st.gx + blockIdx.x;
    const int px = blockIdx.x * TILE + lx, py = blockIdx.y * TILE + ly;
    const bool inside = px < st.W && py < st.H;
    const int beg = tile_offsets[tile], end = tile_offsets[tile + 1];
    const int n = end - beg;
    const float fx = (float)px, fy = (float)py;
    const int HW = st.H * st.W, pix = py * st.W + px;

    const float Tf = inside ? final_T[pix] : 0.f;
    const int last = inside ? n_contrib[pix] : 0;
    const float d0 = inside ? dL_dimage[pix] : 0.f;
    const float d1 = inside ? dL_dimage[HW + pix] : 0.f;
    const float d2 = inside ? dL_dimage[2 * HW + pix] : 0.f;
    const float bg_dot = st.bg0 * d0 + st.bg1 * d1 + st.bg2 * d2;
    float T = Tf;
    float behind0 = 0.f, behind1 = 0.f, behind2 = 0.f, last_alpha = 0.f, lc0 = 0.f, lc1 = 0.f, lc2 = 0.f;

    // the whole workgroup can stop once it is past every pixel's last contributor
    int wg_last = last;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) wg_last = max(wg_last, __shfl_xor(wg_last, m, 64));
    __shared__ int s_last[4];
    if (lane == 0) s_last[tid >> 6] = wg_last;
    __syncthreads();
    wg_last = max(max(s_last[0], s_last[1]), max(s_last[2], s_last[3]));
    const int skip = n - wg_last;  // list tail nobody reached

    for (int base = skip; base < n; base += 256) {
        __syncthreads();
        const int k = end - 1 - (base + tid);
        if (k >= beg) {
            const int id = point_list[k];
            const float4 *src = reinterpret_cast<const float4 *>(geom + id);
            sid[tid] = id;
            s0[tid] = src[0];
            s1[tid] = src[1];
            s2[tid] = src[2].x;
        }
        __syncthreads();
        const int m = min(256, n - base);
        for (int j = 0; j < m; j++) {
            const int contributor = n - base - j;  // 1-based position in the tile list
            const float4 a = s0[j];
            const float4 b = s1[j];
            const float cb = s2[j];
            const float dx = a.x - fx, dy = a.y - fy;
            const float power = -0.5f * (a.z * dx * dx + b.x * dy * dy) - a.w * dx * dy;
            const float G = __expf(power);
            const float alpha = fminf(ALPHA_MAX, b.y * G);
            const bool valid = (contributor <= last) && (power <= 0.0f) && (alpha >= ALPHA_MIN);
            if (__ballot(valid) == 0ull) continue;  // wave-uniform
            float v_du = 0.f, v_dv = 0.f, v_dA = 0.f, v_dB = 0.f, v_dC = 0.f, v_do = 0.f, v_r = 0.f, v_g = 0.f, v_b = 0.f;
            if (valid) {
                T = T / (1.0f - alpha);
                const float w = alpha * T;
                behind0 = last_alpha * lc0 + (1.0f - last_alpha) * behind0;
                behind1 = last_alpha * lc1 + (1.0f - last_alpha) * behind1;
                behind2 = last_alpha * lc2 + (1.0f - last_alpha) * behind2;
                lc0 = b.z; lc1 = b.w; lc2 = cb;
                float dL_dalpha = (b.z - behind0) * d0 + (b.w - behind1) * d1 + (cb - behind2) * d2;
                v_r = w * d0; v_g = w * d1; v_b = w * d2;
                dL_dalpha *= T;
                last_alpha = alpha;
                dL_dalpha += (-Tf / (1.0f - alpha)) * bg_dot;
                const float dL_dG = b.y * dL_dalpha;
                const float gdx = G * dx, gdy = G * dy;
                v_du = dL_dG * (-gdx * a.z - gdy * a.w);
                v_dv = dL_dG * (-gdy * b.x - gdx * a.w);
                v_dA = -0.5f * gdx * dx * dL_dG;
                v_dB = -gdx * dy * dL_dG;
                v_dC = -0.5f * gdy * dy * dL_dG;
                v_do = G * dL_dalpha;
            }
            v_du = wave_sum(v_du); v_dv = wave_sum(v_dv); v_dA = wave_sum(v_dA); v_dB = wave_sum(v_dB);
            v_dC = wave_sum(v_dC); v_do = wave_sum(v_do); v_r = wave_sum(v_r); v_g = wave_sum(v_g); v_b = wave_sum(v_b);
            float out = v_du;
            out = lane == 1 ? v_dv : out;
            out = lane == 2 ? v_dA : out;
            out = lane == 3 ? v_dB : out;
            out = lane == 4 ? v_dC : out;
            out = lane == 5 ? v_do : out;
            out = lane == 6 ? v_r : out;
            out = lane == 7 ? v_g : out;
            out = lane == 8 ? v_b : out;
            if (lane < 9) atomicAdd(acc + (size_t)sid[j] * ACC_STRIDE + lane, out);
        }
    }
}

__global__ void __launch_bounds__(256) k_gaussian_bwd(RasterParams st, int P, const float *__restrict__ means3D,
                                                      const float *__restrict__ scales,
                                                      const float *__restrict__ rotations,
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
        const float du = a0.x, dv = a0.y, dA = a0.z, dB = a0.w, dC = a1.x;
        go = a1.y; gc[0] = a1.z; gc[1] = a1.w; gc[2] = a2;
        g2[0] = du * 0.5f * (float)st.W;
        g2[1] = dv * 0.5f * (float)st.H;
        const float *M = st.m;
        for (int j = 0; j < 3; j++) g3[j] = st.scale * (M[j] * du + M[4 + j] * dv);

        const float px = means3D[3 * i], py = means3D[3 * i + 1], pz = means3D[3 * i + 2];
        const float s0 = scales[3 * i], s1 = scales[3 * i + 1], s2 = scales[3 * i + 2];
        const float4 q = reinterpret_cast<const float4 *>(rotations)[i];
        PreOut o;
        preprocess_gaussian(st, px, py, pz, s0, s1, s2, q.x, q.y, q.z, q.w, o);
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
    (void)colors; (void)opacities;
    GSVC_REQUIRE(settings != nullptr, "raster_backward: settings is NULL");
    GSVC_REQUIRE(P >= 0 && P < (int64_t)1 << 31 && max_instances >= 0, "raster_backward: bad sizes");
    if (P == 0) return GSVC_OK;
    GSVC_REQUIRE(means3D && scales && rotations && radii && geom && binning && image_state && dL_dimage && scratch,
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
    auto *final_T = (const float *)((const char *)image_state + L.off_final_T);
    auto *n_contrib = (const int32_t *)((const char *)image_state + L.off_n_contrib);
    if (hipMemsetAsync(scratch, 0, (size_t)P * ACC_STRIDE * sizeof(float), s) != hipSuccess) {
        set_error("raster_backward: hipMemsetAsync failed");
        return GSVC_E_LAUNCH;
    }
    { ProfScope _prof("k_blend_bwd", s); hipLaunchKernelGGL(k_blend_bwd, dim3(L.gx, L.gy), dim3(256), 0, s, p, tile_offsets, point_list,
                       (const GeomRec *)geom, final_T, n_contrib, dL_dimage, (float *)scratch, counters); }
    { ProfScope _prof("k_gaussian_bwd", s); hipLaunchKernelGGL(k_gaussian_bwd, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, s, p, (int)P, means3D, scales,
                       rotations, radii, (const float *)scratch, dL_dmeans3D, dL_dmeans2D, dL_dcolors, dL_dopacities,
                       dL_dscales, dL_drotations); }
    return check_launch("raster_backward");
}

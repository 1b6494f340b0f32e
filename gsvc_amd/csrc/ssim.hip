// gsvc_amd/csrc/ssim.hip — fused SSIM + L1 image loss (forward and backward), gfx950.
//
// Replaces the distortion part of the fitting step: reference utils/loss_utils.py:20-72 (l1_loss_func and
// ssim_func/_ssim: five depthwise 11x11 Gaussian-window conv2d calls with zero padding plus ~15 elementwise
// kernels per image pair, and their autograd).  One kernel each way:
//   forward   per 32x32 tile: img1/img2 with a 5-pixel halo staged in LDS, separable 11-tap blur of the five
//             moments (x, y, x^2, y^2, xy) through LDS, the SSIM map, its sum, |x-y| summed, and the three
//             per-pixel partials d(map)/d(mu1), d(map)/d(E[x^2]), d(map)/d(E[xy]) for the backward
//   backward  per 32x32 tile: the three partial maps (times dL/dmap) blurred again (the window is symmetric,
//             so the transposed convolution is the same convolution) and combined with x, y; + the L1 sign
// Both passes are REGISTER-BLOCKED (round 3): a thread produces 4 adjacent outputs of a row (horizontal pass: 14 + 2 staged
// values per image through four 16-byte LDS reads) or of a column (vertical pass: 14 LDS reads per moment), so an output
// costs 7 + 17.5 LDS reads instead of 22 + 55, and the 32x32 tile reads 1.7x its pixels from HBM instead of 2.6x (16x16).
#include "common.h"

#include <cmath>

namespace gsvc {

constexpr int SS_TILE = 32;
constexpr int SS_R = 5;                     // window radius (11 taps)
constexpr int SS_HALO = SS_TILE + 2 * SS_R;  // 42
constexpr int SS_LD = 44;                   // row stride of the staged images: 16-byte aligned groups of 4 columns
constexpr int SS_HLD = SS_TILE + 1;         // row stride of the horizontally blurred moments
constexpr float SS_C1 = 0.01f * 0.01f;
constexpr float SS_C2 = 0.03f * 0.03f;
constexpr int SS_SLOTS = 1024;             // rows of the partial-sum workspace

struct SsimWindow {
    float w[11];
};

// The eleven window taps as VECTOR registers.  Passed by value they live in SGPRs, and a multiply-add that reads an SGPR issues at
// 4.5 cycles per wave where the register-only form issues at 2.7 (four waves per SIMD; DESIGN section 4b, tools/micro/valu_issue.hip)
// — and these kernels are nothing but such multiply-adds (434 of the forward's 717 vector arithmetic instructions).  The compiler
// keeps a uniform value in an SGPR whenever it can see that it is uniform: the move is hidden from it.
struct WinRegs {
    float w[11];
};
__device__ __forceinline__ WinRegs window_in_vgprs(const SsimWindow &win)
{
    WinRegs r;
#pragma unroll
    for (int k = 0; k < 11; k++) asm volatile("v_mov_b32 %0, %1" : "=v"(r.w[k]) : "s"(win.w[k]));
    return r;
}

__device__ __forceinline__ float block_sum_256(float v, float *smem)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) smem[wave] = v;
    __syncthreads();
    return smem[0] + smem[1] + smem[2] + smem[3];
}

// img1b != NULL: the first image is the two-view frame 0.5 * (img1 + flip_W(img1b)) (reference pipeline/train.py:368-375), formed
// on load (and stored to avg_out where the caller wants it) instead of by three elementwise launches each way.
__global__ void __launch_bounds__(256) k_ssim_fwd(SsimWindow win_s, const float *__restrict__ img1, const float *__restrict__ img1b,
                                                  float *__restrict__ avg_out,
                                                  const float *__restrict__ img2, int H, int W,
                                                  float *__restrict__ partials /* [SS_SLOTS][2]: ssim, l1 */,
                                                  float *__restrict__ dm_dmu1, float *__restrict__ dm_de11,
                                                  float *__restrict__ dm_de12)
{
    __shared__ __attribute__((aligned(16))) float sx[SS_HALO][SS_LD];
    __shared__ __attribute__((aligned(16))) float sy[SS_HALO][SS_LD];
    __shared__ float hb[5][SS_HALO][SS_HLD];  // horizontally blurred moments
    __shared__ float red[4];
    const WinRegs win = window_in_vgprs(win_s);
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * SS_TILE, y0 = blockIdx.y * SS_TILE;
    const size_t plane = (size_t)blockIdx.z * H * W;
    // every load of the tile is issued before the first is waited for (7 per thread and image: with one load per loop
    // iteration the block spent its time in 7 dependent memory round trips)
    constexpr int SS_LOADS = (SS_HALO * SS_HALO + 255) / 256;
    float xa[SS_LOADS], xb[SS_LOADS], ya[SS_LOADS];
#pragma unroll
    for (int it = 0; it < SS_LOADS; it++) {
        const int i = tid + 256 * it;
        const int r = i / SS_HALO, c = i - r * SS_HALO;
        const int gy = y0 + r - SS_R, gx = x0 + c - SS_R;
        const bool in = i < SS_HALO * SS_HALO && gy >= 0 && gy < H && gx >= 0 && gx < W;
        xa[it] = in ? img1[plane + (size_t)gy * W + gx] : 0.f;
        xb[it] = (img1b && in) ? img1b[plane + (size_t)gy * W + (W - 1 - gx)] : 0.f;
        ya[it] = in ? img2[plane + (size_t)gy * W + gx] : 0.f;
    }
#pragma unroll
    for (int it = 0; it < SS_LOADS; it++) {
        const int i = tid + 256 * it;
        const int r = i / SS_HALO, c = i - r * SS_HALO;
        if (i < SS_HALO * SS_HALO) {
            sx[r][c] = img1b ? (xa[it] + xb[it]) / 2.0f : xa[it];
            sy[r][c] = ya[it];
        }
    }
    if (tid < 2 * SS_HALO) {      // the two pad columns the 16-byte reads touch (never used in arithmetic): keep them finite
        sx[tid >> 1][SS_HALO + (tid & 1)] = 0.f;
        sy[tid >> 1][SS_HALO + (tid & 1)] = 0.f;
    }
    __syncthreads();
    // horizontal pass: group g = (row r, columns 4 cg .. 4 cg + 3)
    for (int g = tid; g < SS_HALO * (SS_TILE / 4); g += 256) {
        const int r = g >> 3, c0 = 4 * (g & 7);
        float xv[16], yv[16];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const float4 a = *reinterpret_cast<const float4 *>(&sx[r][c0 + 4 * q]);
            const float4 b = *reinterpret_cast<const float4 *>(&sy[r][c0 + 4 * q]);
            xv[4 * q] = a.x; xv[4 * q + 1] = a.y; xv[4 * q + 2] = a.z; xv[4 * q + 3] = a.w;
            yv[4 * q] = b.x; yv[4 * q + 1] = b.y; yv[4 * q + 2] = b.z; yv[4 * q + 3] = b.w;
        }
        // the products once per staged element (the reference blurs img1 * img1 etc.: products first), 5 FMAs per tap
        float xx[14], yy[14], xy[14];
#pragma unroll
        for (int k = 0; k < 14; k++) { xx[k] = xv[k] * xv[k]; yy[k] = yv[k] * yv[k]; xy[k] = xv[k] * yv[k]; }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f;
#pragma unroll
            for (int k = 0; k < 11; k++) {
                const float w = win.w[k];
                a0 = fmaf(w, xv[j + k], a0); a1 = fmaf(w, yv[j + k], a1); a2 = fmaf(w, xx[j + k], a2);
                a3 = fmaf(w, yy[j + k], a3); a4 = fmaf(w, xy[j + k], a4);
            }
            hb[0][r][c0 + j] = a0; hb[1][r][c0 + j] = a1; hb[2][r][c0 + j] = a2; hb[3][r][c0 + j] = a3; hb[4][r][c0 + j] = a4;
        }
    }
    __syncthreads();
    // vertical pass: thread = (column lx, rows 4 gq .. 4 gq + 3)
    const int lx = tid & 31, gq = tid >> 5;
    float mom[5][4];
#pragma unroll
    for (int m = 0; m < 5; m++) {
        float v[14];
#pragma unroll
        for (int k = 0; k < 14; k++) v[k] = hb[m][4 * gq + k][lx];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            float a = 0.f;
#pragma unroll
            for (int k = 0; k < 11; k++) a += win.w[k] * v[j + k];
            mom[m][j] = a;
        }
    }
    const int gx = x0 + lx;
    float ssim_sum = 0.f, l1_sum = 0.f;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int ly = 4 * gq + j, gy = y0 + ly;
        if (gx < W && gy < H) {
            const float mu1 = mom[0][j], mu2 = mom[1][j], e11 = mom[2][j], e22 = mom[3][j], e12 = mom[4][j];
            const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
            const float s1 = e11 - mu1_sq, s2 = e22 - mu2_sq, s12 = e12 - mu12;
            const float num1 = 2.f * mu12 + SS_C1, num2 = 2.f * s12 + SS_C2;
            const float A = mu1_sq + mu2_sq + SS_C1, B = s1 + s2 + SS_C2;
            const float inv = 1.0f / (A * B);
            const float ssim = num1 * num2 * inv;
            const float xc = sx[ly + SS_R][lx + SS_R], yc = sy[ly + SS_R][lx + SS_R];
            ssim_sum += ssim;
            l1_sum += fabsf(xc - yc);
            const size_t o = plane + (size_t)gy * W + gx;
            if (avg_out) avg_out[o] = xc;
            if (dm_dmu1) {
                // map as a function of (mu1, E[x^2], E[xy]) with sigma1^2 = E[x^2] - mu1^2, sigma12 = E[xy] - mu1 mu2
                dm_dmu1[o] = 2.f * mu2 * (num2 - num1) * inv - 2.f * mu1 * ssim * (1.0f / A - 1.0f / B);
                dm_de11[o] = -ssim / B;
                dm_de12[o] = 2.f * num1 * inv;
            }
        }
    }
    const float ssum = block_sum_256(ssim_sum, red);
    const float lsum = block_sum_256(l1_sum, red);
    if (tid == 0) {
        // spread the per-block sums over SS_SLOTS rows: atomics on one address serialise at the memory side
        const unsigned slot = (blockIdx.x + blockIdx.y * gridDim.x + blockIdx.z * gridDim.x * gridDim.y) % SS_SLOTS;
        atomicAdd(&partials[2 * slot + 0], ssum);
        atomicAdd(&partials[2 * slot + 1], lsum);
    }
}

__global__ void __launch_bounds__(256) k_ssim_finalize(const float *__restrict__ partials, float *__restrict__ sums)
{
    __shared__ float red[4];
    float a = 0.f, b = 0.f;
    for (int i = threadIdx.x; i < SS_SLOTS; i += 256) { a += partials[2 * i]; b += partials[2 * i + 1]; }
    a = block_sum_256(a, red);
    b = block_sum_256(b, red);
    if (threadIdx.x == 0) { sums[0] = a; sums[1] = b; }
}

// dL/dimg1 = conv(g*dm_dmu1) + 2 x conv(g*dm_de11) + y conv(g*dm_de12) + g_l1 * sign(x - y),
// g = grads[0] / (C*H*W) (mean of the map), g_l1 = grads[1] / (C*H*W)
__global__ void __launch_bounds__(256) k_ssim_bwd(SsimWindow win_s, const float *__restrict__ img1, const float *__restrict__ img1b,
                                                  float *__restrict__ dL_dimg1b,
                                                  const float *__restrict__ img2, int H, int W, float inv_count,
                                                  const float *__restrict__ grads, const float *__restrict__ dm_dmu1,
                                                  const float *__restrict__ dm_de11, const float *__restrict__ dm_de12,
                                                  float *__restrict__ dL_dimg1)
{
    __shared__ __attribute__((aligned(16))) float sm[3][SS_HALO][SS_LD];
    __shared__ float hb[3][SS_HALO][SS_HLD];
    const WinRegs win = window_in_vgprs(win_s);
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * SS_TILE, y0 = blockIdx.y * SS_TILE;
    const size_t plane = (size_t)blockIdx.z * H * W;
    constexpr int SS_LOADS = (SS_HALO * SS_HALO + 255) / 256;
    float ma[3][SS_LOADS];
#pragma unroll
    for (int it = 0; it < SS_LOADS; it++) {          // all loads first (see k_ssim_fwd)
        const int i = tid + 256 * it;
        const int r = i / SS_HALO, c = i - r * SS_HALO;
        const int gy = y0 + r - SS_R, gx = x0 + c - SS_R;
        const bool in = i < SS_HALO * SS_HALO && gy >= 0 && gy < H && gx >= 0 && gx < W;
        const size_t o = plane + (size_t)gy * W + gx;
        ma[0][it] = in ? dm_dmu1[o] : 0.f;
        ma[1][it] = in ? dm_de11[o] : 0.f;
        ma[2][it] = in ? dm_de12[o] : 0.f;
    }
#pragma unroll
    for (int it = 0; it < SS_LOADS; it++) {
        const int i = tid + 256 * it;
        const int r = i / SS_HALO, c = i - r * SS_HALO;
        if (i < SS_HALO * SS_HALO) {
            sm[0][r][c] = ma[0][it]; sm[1][r][c] = ma[1][it]; sm[2][r][c] = ma[2][it];
        }
    }
    if (tid < 2 * SS_HALO) {
#pragma unroll
        for (int m = 0; m < 3; m++) sm[m][tid >> 1][SS_HALO + (tid & 1)] = 0.f;
    }
    __syncthreads();
    for (int g = tid; g < SS_HALO * (SS_TILE / 4); g += 256) {
        const int r = g >> 3, c0 = 4 * (g & 7);
#pragma unroll
        for (int m = 0; m < 3; m++) {
            float v[16];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const float4 a = *reinterpret_cast<const float4 *>(&sm[m][r][c0 + 4 * q]);
                v[4 * q] = a.x; v[4 * q + 1] = a.y; v[4 * q + 2] = a.z; v[4 * q + 3] = a.w;
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                float a = 0.f;
#pragma unroll
                for (int k = 0; k < 11; k++) a += win.w[k] * v[j + k];
                hb[m][r][c0 + j] = a;
            }
        }
    }
    __syncthreads();
    const int lx = tid & 31, gq = tid >> 5;
    float bl[3][4];
#pragma unroll
    for (int m = 0; m < 3; m++) {
        float v[14];
#pragma unroll
        for (int k = 0; k < 14; k++) v[k] = hb[m][4 * gq + k][lx];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            float a = 0.f;
#pragma unroll
            for (int k = 0; k < 11; k++) a += win.w[k] * v[j + k];
            bl[m][j] = a;
        }
    }
    const int gx = x0 + lx;
    const float g = grads[0] * inv_count, gl = grads[1] * inv_count;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int gy = y0 + 4 * gq + j;
        if (gx < W && gy < H) {
            const size_t o = plane + (size_t)gy * W + gx;
            const size_t ob = plane + (size_t)gy * W + (W - 1 - gx);
            float x = img1[o];
            if (img1b) x = (x + img1b[ob]) / 2.0f;
            const float y = img2[o];
            const float d = x - y;
            const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
            const float v = g * (bl[0][j] + 2.f * x * bl[1][j] + y * bl[2][j]) + gl * sgn;
            if (img1b) {
                dL_dimg1[o] = 0.5f * v;
                dL_dimg1b[ob] = 0.5f * v;
            } else {
                dL_dimg1[o] = v;
            }
        }
    }
}

static SsimWindow make_window()
{
    // same construction as the reference: exp(-(i-5)^2 / (2*1.5^2)) in double, normalised, stored as float
    SsimWindow w;
    double g[11], s = 0.0;
    for (int i = 0; i < 11; i++) {
        g[i] = std::exp(-(double)((i - 5) * (i - 5)) / (2.0 * 1.5 * 1.5));
        s += g[i];
    }
    // reference: torch.Tensor([...]) rounds each tap to float first, then divides by the float sum
    float gf[11], sf = 0.f;
    for (int i = 0; i < 11; i++) gf[i] = (float)g[i];
    for (int i = 0; i < 11; i++) sf += gf[i];
    for (int i = 0; i < 11; i++) w.w[i] = gf[i] / sf;
    (void)s;
    return w;
}

}  // namespace gsvc

using namespace gsvc;

static int ssim_forward_impl(const float *img1, const float *img1b, float *avg_out, const float *img2, int32_t C, int32_t H, int32_t W,
                             float *sums, float *workspace, float *dm_dmu1, float *dm_de11, float *dm_de12, void *stream);

extern "C" int gsvc_ssim_l1_forward(const float *img1, const float *img2, int32_t C, int32_t H, int32_t W, float *sums,
                                    float *workspace, float *dm_dmu1, float *dm_de11, float *dm_de12, void *stream)
{
    return ssim_forward_impl(img1, nullptr, nullptr, img2, C, H, W, sums, workspace, dm_dmu1, dm_de11, dm_de12, stream);
}

extern "C" int gsvc_ssim_l1_pair_forward(const float *img_f, const float *img_b, const float *img2, int32_t C, int32_t H, int32_t W,
                                         float *sums, float *workspace, float *dm_dmu1, float *dm_de11, float *dm_de12,
                                         float *avg_out, void *stream)
{
    GSVC_REQUIRE(img_b, "ssim_l1_pair_forward: NULL pointer");
    return ssim_forward_impl(img_f, img_b, avg_out, img2, C, H, W, sums, workspace, dm_dmu1, dm_de11, dm_de12, stream);
}

static int ssim_forward_impl(const float *img1, const float *img1b, float *avg_out, const float *img2, int32_t C, int32_t H, int32_t W,
                             float *sums, float *workspace, float *dm_dmu1, float *dm_de11, float *dm_de12, void *stream)
{
    GSVC_REQUIRE(img1 && img2 && sums && workspace, "ssim_l1_forward: NULL pointer");
    GSVC_REQUIRE(C > 0 && H > 0 && W > 0, "ssim_l1_forward: bad shape");
    GSVC_REQUIRE((dm_dmu1 && dm_de11 && dm_de12) || (!dm_dmu1 && !dm_de11 && !dm_de12),
                 "ssim_l1_forward: pass all three partial maps or none");
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(workspace, 0, 2 * SS_SLOTS * sizeof(float), s) != hipSuccess) {
        set_error("ssim_l1_forward: hipMemsetAsync failed");
        return GSVC_E_LAUNCH;
    }
    static const SsimWindow win = make_window();
    {
        ProfScope _prof("k_ssim_fwd", s);
        hipLaunchKernelGGL(k_ssim_fwd, dim3((W + SS_TILE - 1) / SS_TILE, (H + SS_TILE - 1) / SS_TILE, C), dim3(256), 0, s,
                           win, img1, img1b, avg_out, img2, H, W, workspace, dm_dmu1, dm_de11, dm_de12);
    }
    {
        ProfScope _prof("k_ssim_finalize", s);
        hipLaunchKernelGGL(k_ssim_finalize, dim3(1), dim3(256), 0, s, workspace, sums);
    }
    return check_launch("ssim_l1_forward");
}

static int ssim_backward_impl(const float *img1, const float *img1b, float *dL_dimg1b, const float *img2, int32_t C, int32_t H, int32_t W,
                              const float *grads, const float *dm_dmu1, const float *dm_de11, const float *dm_de12, float *dL_dimg1,
                              void *stream);

extern "C" int gsvc_ssim_l1_backward(const float *img1, const float *img2, int32_t C, int32_t H, int32_t W,
                                     const float *grads, const float *dm_dmu1, const float *dm_de11,
                                     const float *dm_de12, float *dL_dimg1, void *stream)
{
    return ssim_backward_impl(img1, nullptr, nullptr, img2, C, H, W, grads, dm_dmu1, dm_de11, dm_de12, dL_dimg1, stream);
}

extern "C" int gsvc_ssim_l1_pair_backward(const float *img_f, const float *img_b, const float *img2, int32_t C, int32_t H, int32_t W,
                                          const float *grads, const float *dm_dmu1, const float *dm_de11, const float *dm_de12,
                                          float *dL_dimg_f, float *dL_dimg_b, void *stream)
{
    GSVC_REQUIRE(img_b && dL_dimg_b, "ssim_l1_pair_backward: NULL pointer");
    return ssim_backward_impl(img_f, img_b, dL_dimg_b, img2, C, H, W, grads, dm_dmu1, dm_de11, dm_de12, dL_dimg_f, stream);
}

static int ssim_backward_impl(const float *img1, const float *img1b, float *dL_dimg1b, const float *img2, int32_t C, int32_t H, int32_t W,
                              const float *grads, const float *dm_dmu1, const float *dm_de11, const float *dm_de12, float *dL_dimg1,
                              void *stream)
{
    GSVC_REQUIRE(img1 && img2 && grads && dm_dmu1 && dm_de11 && dm_de12 && dL_dimg1, "ssim_l1_backward: NULL pointer");
    GSVC_REQUIRE(C > 0 && H > 0 && W > 0, "ssim_l1_backward: bad shape");
    hipStream_t s = (hipStream_t)stream;
    static const SsimWindow win = make_window();
    {
        ProfScope _prof("k_ssim_bwd", s);
        hipLaunchKernelGGL(k_ssim_bwd, dim3((W + SS_TILE - 1) / SS_TILE, (H + SS_TILE - 1) / SS_TILE, C), dim3(256), 0, s,
                           win, img1, img1b, dL_dimg1b, img2, H, W, 1.0f / ((float)C * (float)H * (float)W), grads, dm_dmu1, dm_de11,
                           dm_de12, dL_dimg1);
    }
    return check_launch("ssim_l1_backward");
}

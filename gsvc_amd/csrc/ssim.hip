// gsvc_amd/csrc/ssim.hip — fused SSIM + L1 image loss (forward and backward), gfx950.
//
// Replaces the distortion part of the fitting step: reference utils/loss_utils.py:20-72 (l1_loss_func and
// ssim_func/_ssim: five depthwise 11x11 Gaussian-window conv2d calls with zero padding plus ~15 elementwise
// kernels per image pair, and their autograd).  One kernel each way:
//   forward   per 16x16 tile: img1/img2 with a 5-pixel halo staged in LDS, separable 11-tap blur of the five
//             moments (x, y, x^2, y^2, xy) through LDS, the SSIM map, its sum, |x-y| summed, and the three
//             per-pixel partials d(map)/d(mu1), d(map)/d(E[x^2]), d(map)/d(E[xy]) for the backward
//   backward  per 16x16 tile: the three partial maps (times dL/dmap) blurred again (the window is symmetric,
//             so the transposed convolution is the same convolution) and combined with x, y; + the L1 sign
// HBM-bound: forward reads 2 images, writes 3 maps; backward reads 2 images + 3 maps, writes 1 gradient.
#include "common.h"

#include <cmath>

namespace gsvc {

constexpr int SS_TILE = 16;
constexpr int SS_R = 5;                     // window radius (11 taps)
constexpr int SS_HALO = SS_TILE + 2 * SS_R;  // 26
constexpr float SS_C1 = 0.01f * 0.01f;
constexpr float SS_C2 = 0.03f * 0.03f;
constexpr int SS_SLOTS = 1024;             // rows of the partial-sum workspace

struct SsimWindow {
    float w[11];
};

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
__global__ void __launch_bounds__(256) k_ssim_fwd(SsimWindow win, const float *__restrict__ img1, const float *__restrict__ img1b,
                                                  float *__restrict__ avg_out,
                                                  const float *__restrict__ img2, int H, int W,
                                                  float *__restrict__ partials /* [SS_SLOTS][2]: ssim, l1 */,
                                                  float *__restrict__ dm_dmu1, float *__restrict__ dm_de11,
                                                  float *__restrict__ dm_de12)
{
    __shared__ float sx[SS_HALO][SS_HALO + 1];
    __shared__ float sy[SS_HALO][SS_HALO + 1];
    __shared__ float hb[5][SS_HALO][SS_TILE + 1];  // horizontally blurred moments
    __shared__ float red[4];
    const int tid = threadIdx.x, lx = tid & 15, ly = tid >> 4;
    const int x0 = blockIdx.x * SS_TILE, y0 = blockIdx.y * SS_TILE;
    const size_t plane = (size_t)blockIdx.z * H * W;
    for (int i = tid; i < SS_HALO * SS_HALO; i += 256) {
        const int r = i / SS_HALO, c = i - r * SS_HALO;
        const int gy = y0 + r - SS_R, gx = x0 + c - SS_R;
        const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
        float xv = in ? img1[plane + (size_t)gy * W + gx] : 0.f;
        if (img1b && in) xv = (xv + img1b[plane + (size_t)gy * W + (W - 1 - gx)]) / 2.0f;
        sx[r][c] = xv;
        sy[r][c] = in ? img2[plane + (size_t)gy * W + gx] : 0.f;
    }
    __syncthreads();
    if (avg_out) {
        const int gxo = x0 + lx, gyo = y0 + ly;
        if (gxo < W && gyo < H) avg_out[plane + (size_t)gyo * W + gxo] = sx[ly + SS_R][lx + SS_R];
    }
    for (int i = tid; i < SS_HALO * SS_TILE; i += 256) {
        const int r = i / SS_TILE, c = i - r * SS_TILE;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f;
#pragma unroll
        for (int k = 0; k < 11; k++) {
            const float x = sx[r][c + k], y = sy[r][c + k], w = win.w[k];
            a0 += w * x; a1 += w * y; a2 += w * x * x; a3 += w * y * y; a4 += w * x * y;
        }
        hb[0][r][c] = a0; hb[1][r][c] = a1; hb[2][r][c] = a2; hb[3][r][c] = a3; hb[4][r][c] = a4;
    }
    __syncthreads();
    float mu1 = 0.f, mu2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
    for (int k = 0; k < 11; k++) {
        const float w = win.w[k];
        mu1 += w * hb[0][ly + k][lx]; mu2 += w * hb[1][ly + k][lx]; e11 += w * hb[2][ly + k][lx];
        e22 += w * hb[3][ly + k][lx]; e12 += w * hb[4][ly + k][lx];
    }
    const int gx = x0 + lx, gy = y0 + ly;
    const bool in = gx < W && gy < H;
    float ssim = 0.f, l1 = 0.f;
    if (in) {
        const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
        const float s1 = e11 - mu1_sq, s2 = e22 - mu2_sq, s12 = e12 - mu12;
        const float num1 = 2.f * mu12 + SS_C1, num2 = 2.f * s12 + SS_C2;
        const float A = mu1_sq + mu2_sq + SS_C1, B = s1 + s2 + SS_C2;
        const float inv = 1.0f / (A * B);
        ssim = num1 * num2 * inv;
        l1 = fabsf(sx[ly + SS_R][lx + SS_R] - sy[ly + SS_R][lx + SS_R]);
        if (dm_dmu1) {
            const size_t o = plane + (size_t)gy * W + gx;
            // map as a function of (mu1, E[x^2], E[xy]) with sigma1^2 = E[x^2] - mu1^2, sigma12 = E[xy] - mu1 mu2
            dm_dmu1[o] = 2.f * mu2 * (num2 - num1) * inv - 2.f * mu1 * ssim * (1.0f / A - 1.0f / B);
            dm_de11[o] = -ssim / B;
            dm_de12[o] = 2.f * num1 * inv;
        }
    }
    const float ssum = block_sum_256(ssim, red);
    const float lsum = block_sum_256(l1, red);
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
__global__ void __launch_bounds__(256) k_ssim_bwd(SsimWindow win, const float *__restrict__ img1, const float *__restrict__ img1b,
                                                  float *__restrict__ dL_dimg1b,
                                                  const float *__restrict__ img2, int H, int W, float inv_count,
                                                  const float *__restrict__ grads, const float *__restrict__ dm_dmu1,
                                                  const float *__restrict__ dm_de11, const float *__restrict__ dm_de12,
                                                  float *__restrict__ dL_dimg1)
{
    __shared__ float sm[3][SS_HALO][SS_HALO + 1];
    __shared__ float hb[3][SS_HALO][SS_TILE + 1];
    const int tid = threadIdx.x, lx = tid & 15, ly = tid >> 4;
    const int x0 = blockIdx.x * SS_TILE, y0 = blockIdx.y * SS_TILE;
    const size_t plane = (size_t)blockIdx.z * H * W;
    for (int i = tid; i < SS_HALO * SS_HALO; i += 256) {
        const int r = i / SS_HALO, c = i - r * SS_HALO;
        const int gy = y0 + r - SS_R, gx = x0 + c - SS_R;
        const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
        const size_t o = plane + (size_t)gy * W + gx;
        sm[0][r][c] = in ? dm_dmu1[o] : 0.f;
        sm[1][r][c] = in ? dm_de11[o] : 0.f;
        sm[2][r][c] = in ? dm_de12[o] : 0.f;
    }
    __syncthreads();
    for (int i = tid; i < SS_HALO * SS_TILE; i += 256) {
        const int r = i / SS_TILE, c = i - r * SS_TILE;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
        for (int k = 0; k < 11; k++) {
            const float w = win.w[k];
            a0 += w * sm[0][r][c + k]; a1 += w * sm[1][r][c + k]; a2 += w * sm[2][r][c + k];
        }
        hb[0][r][c] = a0; hb[1][r][c] = a1; hb[2][r][c] = a2;
    }
    __syncthreads();
    float b0 = 0.f, b1 = 0.f, b2 = 0.f;
#pragma unroll
    for (int k = 0; k < 11; k++) {
        const float w = win.w[k];
        b0 += w * hb[0][ly + k][lx]; b1 += w * hb[1][ly + k][lx]; b2 += w * hb[2][ly + k][lx];
    }
    const int gx = x0 + lx, gy = y0 + ly;
    if (gx < W && gy < H) {
        const size_t o = plane + (size_t)gy * W + gx;
        const size_t ob = plane + (size_t)gy * W + (W - 1 - gx);
        float x = img1[o];
        if (img1b) x = (x + img1b[ob]) / 2.0f;
        const float y = img2[o];
        const float g = grads[0] * inv_count, gl = grads[1] * inv_count;
        const float d = x - y;
        const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
        const float v = g * (b0 + 2.f * x * b1 + y * b2) + gl * sgn;
        if (img1b) {
            dL_dimg1[o] = 0.5f * v;
            dL_dimg1b[ob] = 0.5f * v;
        } else {
            dL_dimg1[o] = v;
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

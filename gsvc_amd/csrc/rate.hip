// gsvc_amd/csrc/rate.hip — fused entropy-rate estimator (forward and analytic backward), gfx950.
//
// Replaces the chain  clamp -> Normal.cdf x2 -> sub -> Low_bound -> -log2 -> (mask) -> sum  of reference
// utils/entropy_models.py:32-68 (EntropyGaussian.forward, non-quantized branch) and the Low_bound autograd
// function (:159-175) whose backward round-trips through NumPy on the host in the reference.  Net Low_bound
// rule reproduced: the gradient passes only where the un-clamped likelihood is >= 2^-16.
//
// HBM-bound elementwise work: x, mean, scale are read once with 16-byte loads where the row length allows,
// the per-row Q is a broadcast, bits are optionally written, and the (weighted) sum is reduced per wave and
// added with one float atomic per workgroup.
#include "common.h"

namespace gsvc {

constexpr float LOW_BOUND = 1.52587890625e-05f;  // 2^-16
constexpr float INV_SQRT2 = 0.70710678118654752440f;
constexpr float INV_SQRT_2PI = 0.39894228040143267794f;
constexpr float INV_LN2 = 1.44269504088896340736f;

__device__ __forceinline__ float normal_cdf(float v, float mu, float inv_sigma)
{
    // torch.distributions.Normal.cdf: 0.5 * (1 + erf((v - mu) * (1/sigma) / sqrt(2)))
    return 0.5f * (1.0f + erff((v - mu) * inv_sigma * INV_SQRT2));
}

__device__ __forceinline__ float block_sum_256(float v, float *smem)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) smem[wave] = v;
    __syncthreads();
    return smem[0] + smem[1] + smem[2] + smem[3];
}

__global__ void __launch_bounds__(256) k_rate_fwd(const float *__restrict__ x, const float *__restrict__ mean,
                                                  const float *__restrict__ scale, const float *__restrict__ Q,
                                                  float Q_scalar, const float *__restrict__ weight,
                                                  const float *__restrict__ x_lo, const float *__restrict__ x_hi,
                                                  int bounds_per_row, long long total, int c, float *__restrict__ bits,
                                                  float *__restrict__ bits_sum)
{
    __shared__ float smem[4];
    float lo = (x_lo && !bounds_per_row) ? *x_lo : -INFINITY, hi = (x_hi && !bounds_per_row) ? *x_hi : INFINITY;
    float acc = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long row = i / c;
        const float q = Q ? Q[row] : Q_scalar;
        if (bounds_per_row) { lo = x_lo[row]; hi = x_hi[row]; }
        const float xv = fminf(fmaxf(x[i], lo), hi);
        const float mu = mean[i], inv_sigma = 1.0f / scale[i];
        const float lower = normal_cdf(xv - 0.5f * q, mu, inv_sigma);
        const float upper = normal_cdf(xv + 0.5f * q, mu, inv_sigma);
        const float lik = fmaxf(upper - lower, LOW_BOUND);
        const float b = -log2f(lik);
        if (bits) bits[i] = b;
        acc += weight ? b * weight[i] : b;
    }
    if (bits_sum) {
        const float s = block_sum_256(acc, smem);
        if (threadIdx.x == 0) atomicAdd(bits_sum, s);
    }
}

// one workgroup per group of rows so the per-row dQ reduces on chip: each wave owns whole rows
__global__ void __launch_bounds__(256) k_rate_bwd(const float *__restrict__ x, const float *__restrict__ mean,
                                                  const float *__restrict__ scale, const float *__restrict__ Q,
                                                  float Q_scalar, const float *__restrict__ weight,
                                                  const float *__restrict__ x_lo, const float *__restrict__ x_hi,
                                                  int bounds_per_row, long long n, int c,
                                                  const float *__restrict__ gscale_dev,
                                                  float *__restrict__ dx, float *__restrict__ dmean,
                                                  float *__restrict__ dscale, float *__restrict__ dQ,
                                                  float *__restrict__ dweight)
{
    float lo = (x_lo && !bounds_per_row) ? *x_lo : -INFINITY, hi = (x_hi && !bounds_per_row) ? *x_hi : INFINITY;
    const float gs = gscale_dev ? *gscale_dev : 1.0f;
    const int lane = threadIdx.x & 63;
    const long long wave_global = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long n_waves = ((long long)gridDim.x * blockDim.x) >> 6;
    for (long long row = wave_global; row < n; row += n_waves) {
        const float q = Q ? Q[row] : Q_scalar;
        if (bounds_per_row) { lo = x_lo[row]; hi = x_hi[row]; }
        float dq_acc = 0.f;
        for (int col = lane; col < c; col += 64) {
            const long long i = row * c + col;
            const float xr = x[i];
            const float xv = fminf(fmaxf(xr, lo), hi);
            const float mu = mean[i], sigma = scale[i], inv_sigma = 1.0f / sigma;
            const float zl = (xv - 0.5f * q - mu) * inv_sigma, zu = (xv + 0.5f * q - mu) * inv_sigma;
            const float lower = 0.5f * (1.0f + erff(zl * INV_SQRT2));
            const float upper = 0.5f * (1.0f + erff(zu * INV_SQRT2));
            const float lik_raw = upper - lower;
            const float lik = fmaxf(lik_raw, LOW_BOUND);
            const float w = weight ? weight[i] : 1.0f;
            // d(-log2 lik)/d lik, zero below the bound (Low_bound net rule)
            const float dl = (lik_raw >= LOW_BOUND) ? (-INV_LN2 / lik) * w * gs : 0.0f;
            const float pl = __expf(-0.5f * zl * zl) * INV_SQRT_2PI * inv_sigma;
            const float pu = __expf(-0.5f * zu * zu) * INV_SQRT_2PI * inv_sigma;
            const float dlik_dx = pu - pl;
            if (dx) dx[i] = (xr >= lo && xr <= hi) ? dl * dlik_dx : 0.0f;
            if (dmean) dmean[i] = -dl * dlik_dx;
            if (dscale) dscale[i] = -dl * (zu * pu - zl * pl);
            if (dweight) dweight[i] = -log2f(lik) * gs;
            dq_acc += dl * 0.5f * (pu + pl);
        }
        if (dQ) {
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) dq_acc += __shfl_xor(dq_acc, m, 64);
            if (lane == 0) dQ[row] += dq_acc;
        }
    }
}

}  // namespace gsvc

using namespace gsvc;

extern "C" int gsvc_rate_forward(const float *x, const float *mean, const float *scale, const float *Q, float Q_scalar,
                                 const float *weight, const float *x_lo, const float *x_hi, int32_t bounds_per_row,
                                 int64_t n, int64_t c, float *bits, float *bits_sum, void *stream)
{
    GSVC_REQUIRE(n >= 0 && c >= 0 && c < (1 << 30), "rate_forward: bad shape");
    if (n == 0 || c == 0) return GSVC_OK;
    GSVC_REQUIRE(x && mean && scale, "rate_forward: NULL input");
    const long long total = (long long)n * c;
    long long blocks = (total + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    { ProfScope _prof("k_rate_fwd", (hipStream_t)stream); hipLaunchKernelGGL(k_rate_fwd, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, mean, scale, Q, Q_scalar,
                       weight, x_lo, x_hi, (int)bounds_per_row, total, (int)c, bits, bits_sum); }
    return check_launch("rate_forward");
}

extern "C" int gsvc_rate_backward(const float *x, const float *mean, const float *scale, const float *Q, float Q_scalar,
                                  const float *weight, const float *x_lo, const float *x_hi, int32_t bounds_per_row,
                                  int64_t n, int64_t c, const float *gscale_dev, float *dx, float *dmean, float *dscale, float *dQ,
                                  float *dweight, void *stream)
{
    GSVC_REQUIRE(n >= 0 && c >= 0 && c < (1 << 30), "rate_backward: bad shape");
    if (n == 0 || c == 0) return GSVC_OK;
    GSVC_REQUIRE(x && mean && scale, "rate_backward: NULL input");
    long long blocks = (n + 3) / 4;  // 4 waves (rows) per workgroup
    if (blocks > 4096) blocks = 4096;
    { ProfScope _prof("k_rate_bwd", (hipStream_t)stream); hipLaunchKernelGGL(k_rate_bwd, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, mean, scale, Q, Q_scalar,
                       weight, x_lo, x_hi, (int)bounds_per_row, (long long)n, (int)c, gscale_dev, dx, dmean, dscale, dQ, dweight); }
    return check_launch("rate_backward");
}

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
            const float pl = expf(-0.5f * zl * zl) * INV_SQRT_2PI * inv_sigma;
            const float pu = expf(-0.5f * zu * zu) * INV_SQRT_2PI * inv_sigma;
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

// ---------------------------------------------------------------------------------------------------------
// Sampled rate of the R renders of a fitting step (reference ortho_gaussian_renderer/guassian.py:73-132, per render): the 5 %
// sample's rows are gathered, priced and summed per render and attribute group in one launch; the backward scatters straight
// into the dense gradients of the gathered-from tensors.  Was ~45 PyTorch launches each way around three k_rate launches.
//   group g in {feature, scaling, offsets}: x_g [rows, C_g], step Q_g [rows], model mean_g / scale_g [ctx rows, C_g];
//   selected row i: s = sel[i] (row of x / Q / mask), e = sel_ctx[i] (row of mean / scale), render r = segment of s;
//   clamp: x_mean_g -+ 15000 * (mean of Q_g over the selected rows of render r); offsets weighted by mask[s, col / 3].
constexpr int RS_MAX_R = 16;

struct RateSampleArgs {
    const float *x[3], *mean[3], *scale[3], *Q[3];
    const float *mask;              // [rows, K] (offset masks), weight of group 2
    const long long *sel, *sel_ctx;
    const float *x_mean;            // device, 3 floats: mean of the whole parameter tensor behind each group
    int C[3];
    long long bound[RS_MAX_R + 1];  // row offsets of the renders
    long long n_sel;
    int R;
};

__device__ __forceinline__ int rs_render_of(const RateSampleArgs &a, long long row)
{
    int r = 0;
    while (r + 1 < a.R && row >= a.bound[r + 1]) r++;
    return r;
}

// stats[r][0..2] = sum of Q_g over the selected rows of render r, stats[r][3] = their count.  Two stages in a fixed order
// (the same bits every run): RS_STAT_BLOCKS workgroups each leave the sums of a strided share of the rows, one workgroup
// adds those.  (One workgroup over all rows took 50-64 us: ~100 dependent gather round trips.)
constexpr int RS_STAT_BLOCKS = 64;

__global__ void __launch_bounds__(256) k_rate_sample_stats(RateSampleArgs a, float *__restrict__ spart)
{
    __shared__ float sm[4][RS_MAX_R][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float acc[RS_MAX_R][4];
#pragma unroll
    for (int r = 0; r < RS_MAX_R; r++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[r][j] = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < a.n_sel; i += 256LL * RS_STAT_BLOCKS) {
        const long long srow = a.sel[i];
        const int r = rs_render_of(a, srow);
        const float q0 = a.Q[0][srow], q1 = a.Q[1][srow], q2 = a.Q[2][srow];
#pragma unroll
        for (int rr = 0; rr < RS_MAX_R; rr++)
            if (rr == r) { acc[rr][0] += q0; acc[rr][1] += q1; acc[rr][2] += q2; acc[rr][3] += 1.f; }
    }
    for (int r = 0; r < a.R; r++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            float v = acc[r][j];
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
            if (lane == 0) sm[wave][r][j] = v;
        }
    __syncthreads();
    if (threadIdx.x < a.R * 4) {
        const int r = threadIdx.x >> 2, j = threadIdx.x & 3;
        spart[blockIdx.x * 64 + r * 4 + j] = (sm[0][r][j] + sm[1][r][j]) + (sm[2][r][j] + sm[3][r][j]);
    }
}

__global__ void __launch_bounds__(64) k_rate_sample_stats_sum(const float *__restrict__ spart, int R, float *__restrict__ stats)
{
    if ((int)threadIdx.x >= R * 4) return;
    float v[RS_STAT_BLOCKS];
#pragma unroll
    for (int b = 0; b < RS_STAT_BLOCKS; b++) v[b] = spart[b * 64 + threadIdx.x];      // all in flight together
    float acc = 0.f;
#pragma unroll
    for (int b = 0; b < RS_STAT_BLOCKS; b++) acc += v[b];
    stats[threadIdx.x] = acc;
}

// MODE 0: forward — per-block partial sums of weight * bits per (group, render) -> part[block][3][R]
// MODE 1: backward — gS[r][g] = dL/d(sum of render r, group g); dx / dQ rows stored (selected rows are distinct), dmean /
//         dscale / dmask added atomically (several selected rows may share a context row) into zero-filled dense tensors
template <int MODE>
__global__ void __launch_bounds__(256) k_rate_sample(RateSampleArgs a, const float *__restrict__ stats, float *__restrict__ part,
                                                     const float *__restrict__ gS, float *const dx0, float *const dx1, float *const dx2,
                                                     float *const dm0, float *const dm1, float *const dm2, float *const ds0,
                                                     float *const ds1, float *const ds2, float *const dq0, float *const dq1,
                                                     float *const dq2, float *__restrict__ dmask, int K)
{
    __shared__ float sacc[4][RS_MAX_R];          // one row of per-render sums per WAVE (added in wave order at the end: fixed order)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = blockIdx.y;
    if (MODE == 0) {
        if (threadIdx.x < 4 * RS_MAX_R) (&sacc[0][0])[threadIdx.x] = 0.f;
        __syncthreads();
    }
    const int C = a.C[g];
    const float *__restrict__ X = a.x[g], *__restrict__ MU = a.mean[g], *__restrict__ SG = a.scale[g], *__restrict__ QQ = a.Q[g];
    float *const DX = g == 0 ? dx0 : (g == 1 ? dx1 : dx2), *const DM = g == 0 ? dm0 : (g == 1 ? dm1 : dm2);
    float *const DS = g == 0 ? ds0 : (g == 1 ? ds1 : ds2), *const DQ = g == 0 ? dq0 : (g == 1 ? dq1 : dq2);
    const long long wave_global = ((long long)blockIdx.x * 256 + threadIdx.x) >> 6;
    const long long n_waves = ((long long)gridDim.x * 256) >> 6;
    for (long long i = wave_global; i < a.n_sel; i += n_waves) {
        const long long srow = a.sel[i], erow = a.sel_ctx ? a.sel_ctx[i] : srow;
        const int r = rs_render_of(a, srow);
        const float q = QQ[srow];
        const float qm = stats[r * 4 + g] / fmaxf(stats[r * 4 + 3], 1.f);
        const float xm = a.x_mean[g];
        const float lo = xm - 15000.f * qm, hi = xm + 15000.f * qm;
        const float gs = MODE == 1 ? gS[r * 3 + g] : 0.f;
        float row_acc = 0.f, dq_acc = 0.f;
        for (int col = lane; col < C; col += 64) {
            const float xr = X[srow * C + col];
            const float xv = fminf(fmaxf(xr, lo), hi);
            const float mu = MU[erow * C + col], sigma = SG[erow * C + col], inv_sigma = 1.0f / sigma;
            const float zl = (xv - 0.5f * q - mu) * inv_sigma, zu = (xv + 0.5f * q - mu) * inv_sigma;
            const float lower = 0.5f * (1.0f + erff(zl * INV_SQRT2));
            const float upper = 0.5f * (1.0f + erff(zu * INV_SQRT2));
            const float lik_raw = upper - lower;
            const float lik = fmaxf(lik_raw, LOW_BOUND);
            const float w = (g == 2 && a.mask) ? a.mask[srow * K + col / 3] : 1.0f;
            const float b = -log2f(lik);
            if (MODE == 0) {
                row_acc += b * w;
            } else {
                const float dl = (lik_raw >= LOW_BOUND) ? (-INV_LN2 / lik) * w * gs : 0.0f;
                const float pl = expf(-0.5f * zl * zl) * INV_SQRT_2PI * inv_sigma;
                const float pu = expf(-0.5f * zu * zu) * INV_SQRT_2PI * inv_sigma;
                const float dlik_dx = pu - pl;
                if (DX) DX[srow * C + col] = (xr >= lo && xr <= hi) ? dl * dlik_dx : 0.0f;
                if (DM) atomicAdd(DM + erow * C + col, -dl * dlik_dx);
                if (DS) atomicAdd(DS + erow * C + col, -dl * (zu * pu - zl * pl));
                if (g == 2 && dmask && a.mask) atomicAdd(dmask + srow * K + col / 3, b * gs);
                dq_acc += dl * 0.5f * (pu + pl);
            }
        }
        if (MODE == 0) {
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) row_acc += __shfl_xor(row_acc, m, 64);
            if (lane == 0) sacc[wave][r] += row_acc;          // the wave's own slot, rows in the wave's loop order
        } else if (DQ) {
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) dq_acc += __shfl_xor(dq_acc, m, 64);
            if (lane == 0) DQ[srow] = dq_acc;
        }
    }
    if (MODE == 0) {
        __syncthreads();
        if (threadIdx.x < a.R)
            part[((size_t)blockIdx.x * 3 + g) * RS_MAX_R + threadIdx.x] =
                (sacc[0][threadIdx.x] + sacc[1][threadIdx.x]) + (sacc[2][threadIdx.x] + sacc[3][threadIdx.x]);
    }
}

// S[r][g] = sum over the blocks' partial sums: one workgroup per (g, r), each thread a strided share of the blocks, then a
// fixed tree -> the same bits every run (the serial loop this replaces took 236 us for 1024 blocks: 1024 dependent loads)
__global__ void __launch_bounds__(256) k_rate_sample_finalize(const float *__restrict__ part, int blocks, int R, float *__restrict__ S)
{
    __shared__ float sm[256];
    const int g = blockIdx.x / R, r = blockIdx.x - g * R;
    float acc = 0.f;
    for (int b = threadIdx.x; b < blocks; b += 256) acc += part[((size_t)b * 3 + g) * RS_MAX_R + r];
    sm[threadIdx.x] = acc;
    __syncthreads();
#pragma unroll
    for (int w = 128; w >= 1; w >>= 1) {
        if ((int)threadIdx.x < w) sm[threadIdx.x] += sm[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) S[r * 3 + g] = sm[0];
}

// Means of the three whole parameter tensors behind the rate's clamp bounds (reference utils/entropy_models.py: x_mean =
// pc._anchor_feat.mean() etc., taken over ALL anchors each time the rate is evaluated): one pass over the 86 floats per anchor
// instead of three reductions, an exp pass and their temporaries (0.18 ms of PyTorch launches per step).  Fixed order:
// PM_BLOCKS block sums per tensor, then one workgroup adds them.
constexpr int PM_BLOCKS = 2048;      // 8 workgroups per CU: enough 16-byte loads in flight to stream at the HBM rate

// sum of f(x) over a contiguous tensor by the workgroups [0, nblk) it was given: 16-byte loads, FOUR of them in flight per lane,
// four independent running sums (one dependent add chain per lane over 4-byte loads from 256 workgroups read 84 MB in 141 us =
// 0.6 TB/s; this form 29 us = 2.9 TB/s, the same as with every workgroup walking the three tensors in turn); the order is fixed
// by (nblk, n) alone
template <bool EXP>
__device__ __forceinline__ float pm_stream_sum(const float *__restrict__ x, long long n, int blk, int nblk)
{
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    const long long stride = (long long)nblk * 256, first = (long long)blk * 256 + threadIdx.x;
    auto f = [](float v) { return EXP ? expf(v) : v; };
    if ((reinterpret_cast<uintptr_t>(x) & 15) == 0) {
        const float4 *x4 = reinterpret_cast<const float4 *>(x);
        const long long n4 = n >> 2;
        long long i = first;
        for (; i + 3 * stride < n4; i += 4 * stride) {
            const float4 u = x4[i], v = x4[i + stride], w = x4[i + 2 * stride], z = x4[i + 3 * stride];
            s0 += (f(u.x) + f(v.x)) + (f(w.x) + f(z.x)); s1 += (f(u.y) + f(v.y)) + (f(w.y) + f(z.y));
            s2 += (f(u.z) + f(v.z)) + (f(w.z) + f(z.z)); s3 += (f(u.w) + f(v.w)) + (f(w.w) + f(z.w));
        }
        for (; i < n4; i += stride) {
            const float4 u = x4[i];
            s0 += f(u.x); s1 += f(u.y); s2 += f(u.z); s3 += f(u.w);
        }
        for (long long j = 4 * n4 + first; j < n; j += stride) s0 += f(x[j]);
    } else {
        for (long long j = first; j < n; j += stride) s0 += f(x[j]);
    }
    return (s0 + s1) + (s2 + s3);
}

// workgroups [0, fb) sum tensor a, [fb, fc) tensor b, [fc, gridDim) tensor c (dealt by size on the host); part[block] = its sum
__global__ void __launch_bounds__(256) k_param_means_part(const float *__restrict__ a, long long na, const float *__restrict__ b,
                                                          long long nb, int b_exp, const float *__restrict__ c, long long nc,
                                                          int fb, int fc, float *__restrict__ part)
{
    __shared__ float red[4];
    const int blk = blockIdx.x;
    float v;
    if (blk < fb) v = pm_stream_sum<false>(a, na, blk, fb);
    else if (blk < fc) v = b_exp ? pm_stream_sum<true>(b, nb, blk - fb, fc - fb) : pm_stream_sum<false>(b, nb, blk - fb, fc - fb);
    else v = pm_stream_sum<false>(c, nc, blk - fc, (int)gridDim.x - fc);
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) part[blk] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ void __launch_bounds__(256) k_param_means_sum(const float *__restrict__ part, int blocks, int fb, int fc, long long na,
                                                         long long nb, long long nc, float *__restrict__ out)
{
    __shared__ float sm[3][256];
    const int lo[3] = {0, fb, fc}, hi[3] = {fb, fc, blocks};
#pragma unroll
    for (int k = 0; k < 3; k++) {
        float v = 0.f;
        for (int bl = lo[k] + (int)threadIdx.x; bl < hi[k]; bl += 256) v += part[bl];
        sm[k][threadIdx.x] = v;
    }
    __syncthreads();
    for (int w = 128; w >= 1; w >>= 1) {
        if ((int)threadIdx.x < w) {
#pragma unroll
            for (int k = 0; k < 3; k++) sm[k][threadIdx.x] += sm[k][threadIdx.x + w];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out[0] = sm[0][0] / (float)max(na, 1LL);
        out[1] = sm[1][0] / (float)max(nb, 1LL);
        out[2] = sm[2][0] / (float)max(nc, 1LL);
    }
}

// Normalisation of the sampled rate (reference guassian.py:110-132 per render): bits per coded parameter of the three groups
// and of all of them, scaled by the render's keep rate = its rows with a live offset / its rows.  The PyTorch form was ~20
// few-element launches forward (mask sums, a searchsorted for the sample sizes, divisions) and as many backward.
constexpr int RN_BLOCKS = 256;
struct RateNormBounds { long long b[RS_MAX_R + 1]; };

// part[block][r] = rows of render r in the block's share of the rows whose K offset masks sum to more than zero
__global__ void __launch_bounds__(256) k_rate_live_part(const float *__restrict__ offset_masks, long long rows, int K, RateNormBounds rb, int R,
                                                        int *__restrict__ part)
{
    __shared__ int cnt[RS_MAX_R];
    if (threadIdx.x < RS_MAX_R) cnt[threadIdx.x] = 0;
    __syncthreads();
    const long long per = (rows + gridDim.x - 1) / gridDim.x, r0 = per * blockIdx.x;
    const long long r1 = r0 + per < rows ? r0 + per : rows;
    for (long long row = r0 + threadIdx.x; row < r1; row += 256) {
        float s = 0.f;
        for (int k = 0; k < K; k++) s += offset_masks[row * K + k];
        if (s > 0.f) {
            int r = 0;
            while (r + 1 < R && row >= rb.b[r + 1]) r++;
            atomicAdd(&cnt[r], 1);
        }
    }
    __syncthreads();
    if (threadIdx.x < RS_MAX_R) part[blockIdx.x * RS_MAX_R + threadIdx.x] = cnt[threadIdx.x];
}

__device__ __forceinline__ long long rn_lower_bound(const long long *__restrict__ a, long long n, long long v)
{
    long long lo = 0, hi = n;
    while (lo < hi) {
        const long long mid = (lo + hi) >> 1;
        if (a[mid] < v) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// out[r] = {all, features, scalings, offsets} bits per parameter; coef[r] = d out[r][i] / d (the S it divides); sum = sum_r out[r][0]
__global__ void __launch_bounds__(256) k_rate_normalise(const float *__restrict__ S, const int *__restrict__ part, int blocks,
                                                        const long long *__restrict__ sel, long long n_sel, RateNormBounds rb, int R, float d0,
                                                        float d1, float d2, float *__restrict__ out, float *__restrict__ coef, float *__restrict__ sum)
{
    __shared__ int live_s[RS_MAX_R];
    __shared__ long long edge[RS_MAX_R + 1];
    __shared__ float tot[RS_MAX_R];
    const int t = threadIdx.x;
    if (t < RS_MAX_R) live_s[t] = 0;
    __syncthreads();
    // the blocks' live-row counts: thread t takes blocks t, t + 256, ... (integer sums: any order)
    for (int r = 0; r < R; r++) {
        int v = 0;
        for (int b = t; b < blocks; b += 256) v += part[b * RS_MAX_R + r];
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
        if ((t & 63) == 0 && v) atomicAdd(&live_s[r], v);
    }
    if (t <= R) edge[t] = rn_lower_bound(sel, n_sel, rb.b[t]);      // the R + 1 searches side by side
    __syncthreads();
    const int r = t;
    if (r < R) {
        const long long rows = rb.b[r + 1] - rb.b[r];
        const float kr = (float)live_s[r] / (float)(rows > 1 ? rows : 1);
        const float ns = (float)(edge[r + 1] - edge[r]);
        const float N0 = ns * d0, N1 = ns * d1, N2 = ns * d2, NA = (N0 + N1) + N2;
        const float s0 = S[r * 3], s1 = S[r * 3 + 1], s2 = S[r * 3 + 2];
        out[r * 4] = ((s0 + s1) + s2) / NA * kr;
        out[r * 4 + 1] = s0 / N0 * kr;
        out[r * 4 + 2] = s1 / N1 * kr;
        out[r * 4 + 3] = s2 / N2 * kr;
        coef[r * 4] = kr / NA; coef[r * 4 + 1] = kr / N0; coef[r * 4 + 2] = kr / N1; coef[r * 4 + 3] = kr / N2;
        tot[r] = out[r * 4];
    }
    __syncthreads();
    if (t == 0) {
        float a = 0.f;
        for (int q = 0; q < R; q++) a += tot[q];
        sum[0] = a;
    }
}

__global__ void __launch_bounds__(64) k_rate_normalise_bwd(const float *__restrict__ coef, const float *__restrict__ g_out,
                                                           const float *__restrict__ g_sum, int R, float *__restrict__ gS)
{
    const int i = threadIdx.x;
    if (i >= 3 * R) return;
    const int r = i / 3, g = i - 3 * r;
    const float ga = (g_out ? g_out[r * 4] : 0.f) + (g_sum ? g_sum[0] : 0.f);
    gS[i] = ga * coef[r * 4] + (g_out ? g_out[r * 4 + 1 + g] * coef[r * 4 + 1 + g] : 0.f);
}

// Densification statistics of the un-compacted renders of a step (reference scene/gaussian_model.py:1281-1314 through the
// nested masks visible anchor -> opacity > 0 -> radius > 0): per anchor the positive opacities of its K Gaussians and one visit,
// per (anchor, slot) the screen-space gradient norm and one count where the Gaussian was rasterised.  One launch instead of
// eleven (clamp, sums, a norm, four index_adds and their temporaries).
__global__ void __launch_bounds__(256) k_training_statis(const long long *__restrict__ vis, const float *__restrict__ opacity,
                                                         const uint8_t *__restrict__ seen, const float *__restrict__ grad, int grad_stride,
                                                         long long rows, int K, float *__restrict__ opacity_accum,
                                                         float *__restrict__ anchor_denom, float *__restrict__ grad_accum,
                                                         float *__restrict__ denom)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * K) return;
    const long long row = i / K;
    const int k = (int)(i - row * K);
    const long long a = vis[row];
    if (seen[i]) {
        const float gx = grad[i * grad_stride], gy = grad[i * grad_stride + 1];
        atomicAdd(grad_accum + a * K + k, sqrtf(gx * gx + gy * gy));
        atomicAdd(denom + a * K + k, 1.f);
    }
    if (k == 0) {
        float s = 0.f;
        for (int q = 0; q < K; q++) s += fmaxf(opacity[row * K + q], 0.f);
        atomicAdd(opacity_accum + a, s);
        atomicAdd(anchor_denom + a, 1.f);
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

static int rate_sample_args(gsvc::RateSampleArgs &a, const gsvc_rate_sample *d, const char *what)
{
    GSVC_REQUIRE(d && d->renders >= 1 && d->renders <= gsvc::RS_MAX_R && d->n_sel >= 0 && d->row_bounds, "%s: bad arguments", what);
    for (int g = 0; g < 3; g++) {
        GSVC_REQUIRE(d->x[g] && d->mean[g] && d->scale[g] && d->Q[g] && d->C[g] > 0, "%s: NULL input (group %d)", what, g);
        a.x[g] = d->x[g]; a.mean[g] = d->mean[g]; a.scale[g] = d->scale[g]; a.Q[g] = d->Q[g];
        a.C[g] = d->C[g];
    }
    GSVC_REQUIRE(!d->mask || (d->K > 0 && d->C[2] == 3 * d->K), "%s: the offsets group must have 3 * K columns", what);
    GSVC_REQUIRE(d->x_mean, "%s: x_mean is NULL", what);
    a.x_mean = d->x_mean;
    a.mask = d->mask; a.sel = (const long long *)d->sel; a.sel_ctx = (const long long *)d->sel_ctx;
    for (int r = 0; r <= d->renders; r++) a.bound[r] = d->row_bounds[r];
    a.n_sel = d->n_sel; a.R = d->renders;
    return GSVC_OK;
}

extern "C" int64_t gsvc_rate_sample_scratch_floats(int64_t n_sel)
{
    long long blocks = (n_sel + 3) / 4;
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    return 64 + blocks * 3 * gsvc::RS_MAX_R + 64 * gsvc::RS_STAT_BLOCKS;
}

extern "C" int gsvc_rate_sample_forward(const gsvc_rate_sample *d, float *scratch, float *S, void *stream)
{
    gsvc::RateSampleArgs a;
    if (int rc = rate_sample_args(a, d, "rate_sample_forward")) return rc;
    GSVC_REQUIRE(scratch && S && (d->n_sel == 0 || d->sel), "rate_sample_forward: NULL pointer");
    hipStream_t s = (hipStream_t)stream;
    long long blocks = (d->n_sel + 3) / 4;
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    gsvc::ProfScope _prof("k_rate_sample", s);
    float *spart = scratch + 64 + blocks * 3 * gsvc::RS_MAX_R;
    hipLaunchKernelGGL(gsvc::k_rate_sample_stats, dim3(gsvc::RS_STAT_BLOCKS), dim3(256), 0, s, a, spart);
    hipLaunchKernelGGL(gsvc::k_rate_sample_stats_sum, dim3(1), dim3(64), 0, s, spart, d->renders, scratch);
    hipLaunchKernelGGL(gsvc::k_rate_sample<0>, dim3((unsigned)blocks, 3), dim3(256), 0, s, a, scratch, scratch + 64, nullptr, nullptr,
                       nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, d->K);
    hipLaunchKernelGGL(gsvc::k_rate_sample_finalize, dim3(3 * d->renders), dim3(256), 0, s, scratch + 64, (int)blocks, d->renders, S);
    return gsvc::check_launch("rate_sample_forward");
}

extern "C" int gsvc_rate_sample_backward(const gsvc_rate_sample *d, const float *scratch, const float *gS, float *const *dx,
                                         float *const *dmean, float *const *dscale, float *const *dQ, float *dmask, void *stream)
{
    gsvc::RateSampleArgs a;
    if (int rc = rate_sample_args(a, d, "rate_sample_backward")) return rc;
    GSVC_REQUIRE(scratch && gS && dx && dmean && dscale && dQ, "rate_sample_backward: NULL pointer");
    if (d->n_sel == 0) return GSVC_OK;
    hipStream_t s = (hipStream_t)stream;
    long long blocks = (d->n_sel + 3) / 4;
    if (blocks > 1024) blocks = 1024;
    gsvc::ProfScope _prof("k_rate_sample_bwd", s);
    hipLaunchKernelGGL(gsvc::k_rate_sample<1>, dim3((unsigned)blocks, 3), dim3(256), 0, s, a, scratch, nullptr, gS, dx[0], dx[1], dx[2],
                       dmean[0], dmean[1], dmean[2], dscale[0], dscale[1], dscale[2], dQ[0], dQ[1], dQ[2], dmask, d->K);
    return gsvc::check_launch("rate_sample_backward");
}

extern "C" int64_t gsvc_param_means_scratch_floats(void) { return 3 * gsvc::PM_BLOCKS; }

extern "C" int gsvc_param_means(const float *feat, int64_t n_feat, const float *scaling, int64_t n_scaling, int32_t scaling_exp,
                                const float *offset, int64_t n_offset, float *scratch, float *out3, void *stream)
{
    GSVC_REQUIRE(n_feat >= 0 && n_scaling >= 0 && n_offset >= 0, "param_means: bad sizes");
    GSVC_REQUIRE(scratch && out3 && (n_feat == 0 || feat) && (n_scaling == 0 || scaling) && (n_offset == 0 || offset),
                 "param_means: NULL pointer");
    hipStream_t s = (hipStream_t)stream;
    gsvc::ProfScope _prof("k_param_means", s);
    // workgroups dealt to the three tensors by size (a tensor with elements gets at least one)
    const double tot = (double)n_feat + (double)n_scaling + (double)n_offset;
    int wa = 0, wb = 0, wc = 0;
    if (tot > 0) {
        wa = n_feat ? (int)(gsvc::PM_BLOCKS * ((double)n_feat / tot)) : 0;
        wb = n_scaling ? (int)(gsvc::PM_BLOCKS * ((double)n_scaling / tot)) : 0;
        if (n_feat && wa < 1) wa = 1;
        if (n_scaling && wb < 1) wb = 1;
        wc = gsvc::PM_BLOCKS - wa - wb;
        if (wc < 0) { wb += wc; wc = 0; }
        if (n_offset == 0) { wa += wc; wc = 0; if (n_feat == 0) { wb += wa; wa = 0; } }
        else if (wc < 1) { wc = 1; if (wa > wb) wa--; else wb--; }
    }
    const int fb = wa, fc = wa + wb;
    hipLaunchKernelGGL(gsvc::k_param_means_part, dim3(gsvc::PM_BLOCKS), dim3(256), 0, s, feat, (long long)n_feat, scaling,
                       (long long)n_scaling, (int)scaling_exp, offset, (long long)n_offset, fb, fc, scratch);
    hipLaunchKernelGGL(gsvc::k_param_means_sum, dim3(1), dim3(256), 0, s, scratch, gsvc::PM_BLOCKS, fb, fc, (long long)n_feat,
                       (long long)n_scaling, (long long)n_offset, out3);
    return gsvc::check_launch("param_means");
}

static bool rn_bounds(const int64_t *row_bounds, int32_t R, gsvc::RateNormBounds &rb)
{
    if (!row_bounds || R < 1 || R > gsvc::RS_MAX_R) return false;
    for (int r = 0; r <= gsvc::RS_MAX_R; r++) rb.b[r] = row_bounds[r <= R ? r : R];
    for (int r = 0; r < R; r++)
        if (rb.b[r + 1] < rb.b[r]) return false;
    return true;
}

extern "C" int64_t gsvc_rate_normalise_scratch_bytes(void) { return (int64_t)gsvc::RN_BLOCKS * gsvc::RS_MAX_R * sizeof(int); }

extern "C" int gsvc_rate_normalise_forward(const float *S, const float *offset_masks, int32_t K, const int64_t *sel, int64_t n_sel,
                                           const int64_t *row_bounds_host, int32_t R, const float *dims3_host, void *scratch,
                                           float *out, float *coef, float *sum, void *stream)
{
    gsvc::RateNormBounds rb;
    GSVC_REQUIRE(rn_bounds(row_bounds_host, R, rb) && K > 0 && n_sel >= 0 && dims3_host, "rate_normalise_forward: 1..16 renders, K > 0");
    GSVC_REQUIRE(S && scratch && out && coef && sum && (n_sel == 0 || sel) && (rb.b[R] == 0 || offset_masks), "rate_normalise_forward: NULL pointer");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(gsvc::k_rate_live_part, dim3(gsvc::RN_BLOCKS), dim3(256), 0, s, offset_masks, (long long)rb.b[R], (int)K, rb, (int)R,
                       (int *)scratch);
    hipLaunchKernelGGL(gsvc::k_rate_normalise, dim3(1), dim3(256), 0, s, S, (const int *)scratch, gsvc::RN_BLOCKS, (const long long *)sel,
                       (long long)n_sel, rb, (int)R, dims3_host[0], dims3_host[1], dims3_host[2], out, coef, sum);
    return gsvc::check_launch("rate_normalise_forward");
}

extern "C" int gsvc_rate_normalise_backward(const float *coef, const float *g_out, const float *g_sum, int32_t R, float *gS, void *stream)
{
    GSVC_REQUIRE(coef && gS && R >= 1 && R <= gsvc::RS_MAX_R, "rate_normalise_backward: bad arguments");
    hipLaunchKernelGGL(gsvc::k_rate_normalise_bwd, dim3(1), dim3(64), 0, (hipStream_t)stream, coef, g_out, g_sum, (int)R, gS);
    return gsvc::check_launch("rate_normalise_backward");
}

extern "C" int gsvc_training_statis(const int64_t *vis, const float *opacity, const uint8_t *seen, const float *grad, int32_t grad_stride,
                                    int64_t rows, int32_t K, float *opacity_accum, float *anchor_denom, float *grad_accum, float *denom,
                                    void *stream)
{
    GSVC_REQUIRE(rows >= 0 && K > 0 && grad_stride >= 2, "training_statis: bad shape");
    if (rows == 0) return GSVC_OK;
    GSVC_REQUIRE(vis && opacity && seen && grad && opacity_accum && anchor_denom && grad_accum && denom, "training_statis: NULL pointer");
    hipStream_t s = (hipStream_t)stream;
    gsvc::ProfScope _prof("k_training_statis", s);
    hipLaunchKernelGGL(gsvc::k_training_statis, dim3((unsigned)((rows * K + 255) / 256)), dim3(256), 0, s, (const long long *)vis, opacity, seen,
                       grad, (int)grad_stride, (long long)rows, (int)K, opacity_accum, anchor_denom, grad_accum, denom);
    return gsvc::check_launch("training_statis");
}

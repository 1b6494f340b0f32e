// gsvc_amd/csrc/quant.hip — training-time noise quantisation of the per-anchor attributes, fused, gfx950.
//
// Reference utils/encodings.py:395-409 (UniformQuantizer) as the batched step applies it to R renders at once: for every
// row m of render r (rows [off[r], off[r+1])), with a per-row step Q_m (or one scalar Q),
//     centre_r = mean(x over render r) / mean(Q over render r)          (no gradient)
//     y = clamp(x / Q_m, centre_r - 15000, centre_r + 15000) * Q_m + noise * Q_m,   noise ~ U(-1/2, 1/2) drawn by the caller
// In PyTorch that is ~25 small launches per tensor and direction (segment means through index_add, broadcasts, clamp with
// tensor bounds); here: one reduction (+ a one-block finalize), one elementwise forward, one backward.
#include "common.h"

namespace gsvc {

constexpr int Q_MAX_RENDERS = 8;
constexpr float Q_CLAMP_STEPS = 15000.0f;

struct QSeg {
    long long off[Q_MAX_RENDERS + 1];   // first row of render r
    int R;
};

__device__ __forceinline__ int q_render_of(const QSeg &seg, long long row)
{
    int r = 0;
    while (r + 1 < seg.R && row >= seg.off[r + 1]) r++;
    return r;
}

__device__ __forceinline__ float q_wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__device__ __forceinline__ float q_block_sum(float v, float *red)
{
    v = q_wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// part[(r*nbx + bx)*2 + {0,1}] = block sums of x (all columns) and of q over 64-row slabs of render r
__global__ void __launch_bounds__(256) k_quant_sums(const float *__restrict__ x, const float *__restrict__ q, QSeg seg, int C,
                                                    float *__restrict__ part)
{
    __shared__ float red[4];
    const int r = blockIdx.y;
    const long long row0 = seg.off[r] + (long long)blockIdx.x * 64, row1 = min(seg.off[r + 1], row0 + 64);
    float sx = 0.f, sq = 0.f;
    if (row0 < row1) {
        const long long e0 = row0 * C, e1 = row1 * C;
        for (long long e = e0 + threadIdx.x; e < e1; e += 256) sx += x[e];
        if (q && row0 + threadIdx.x < row1) sq = q[row0 + threadIdx.x];
    }
    sx = q_block_sum(sx, red);
    sq = q_block_sum(sq, red);
    if (threadIdx.x == 0) {
        float *d = part + ((size_t)r * gridDim.x + blockIdx.x) * 2;
        d[0] = sx; d[1] = sq;
    }
}

// centre[r] = (sum x / (rows_r C)) / (sum q / rows_r)   or   / q_scalar
__global__ void __launch_bounds__(256) k_quant_centre(const float *__restrict__ part, int nbx, QSeg seg, int C, int has_q,
                                                      float q_scalar, float *__restrict__ centre)
{
    __shared__ float red[4];
    for (int r = 0; r < seg.R; r++) {
        float sx = 0.f, sq = 0.f;
        for (int b = threadIdx.x; b < nbx; b += 256) {
            sx += part[((size_t)r * nbx + b) * 2];
            sq += part[((size_t)r * nbx + b) * 2 + 1];
        }
        sx = q_block_sum(sx, red);
        sq = q_block_sum(sq, red);
        if (threadIdx.x == 0) {
            const float rows = (float)(seg.off[r + 1] - seg.off[r]);
            const float xm = sx / fmaxf(rows * (float)C, 1.f);
            const float qm = has_q ? sq / fmaxf(rows, 1.f) : q_scalar;
            centre[r] = xm / qm;
        }
    }
}

__global__ void __launch_bounds__(256) k_quant_fwd(const float *__restrict__ x, const float *__restrict__ q, float q_scalar,
                                                   const float *__restrict__ noise, const float *__restrict__ centre, QSeg seg,
                                                   int C, float *__restrict__ y)
{
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long n = seg.off[seg.R] * C;
    if (e >= n) return;
    const long long row = e / C;
    const float c = centre[q_render_of(seg, row)];
    const float Q = q ? q[row] : q_scalar;
    const float t = fminf(fmaxf(x[e] / Q, c - Q_CLAMP_STEPS), c + Q_CLAMP_STEPS);
    y[e] = t * Q + noise[e] * Q;
}

// dx = g where the clamp is inactive; dq[row] = sum_c g * (clamped value if the clamp is active, else 0, + noise).
// A block takes rpb = 256 / C whole rows (one lane per element), the row sums meet in LDS.
__global__ void __launch_bounds__(256) k_quant_bwd(const float *__restrict__ g, const float *__restrict__ x,
                                                   const float *__restrict__ q, float q_scalar, const float *__restrict__ noise,
                                                   const float *__restrict__ centre, QSeg seg, int C, int rpb,
                                                   float *__restrict__ dx, float *__restrict__ dq)
{
    __shared__ float part[256];
    const int t = threadIdx.x, lr = t / C;
    const long long rows = seg.off[seg.R];
    const long long row = (long long)blockIdx.x * rpb + lr;
    const bool live = lr < rpb && row < rows;
    float contrib = 0.f;
    if (live) {
        const long long e = row * C + (t - lr * C);
        const float c = centre[q_render_of(seg, row)];
        const float Q = q ? q[row] : q_scalar;
        const float v = x[e] / Q, lo = c - Q_CLAMP_STEPS, hi = c + Q_CLAMP_STEPS;
        const bool inside = v >= lo && v <= hi;
        const float ge = g[e];
        dx[e] = inside ? ge : 0.f;
        contrib = ge * ((inside ? 0.f : fminf(fmaxf(v, lo), hi)) + noise[e]);
    }
    part[t] = contrib;
    __syncthreads();
    if (dq && t < rpb) {
        const long long r2 = (long long)blockIdx.x * rpb + t;
        if (r2 < rows) {
            float a = 0.f;
            for (int c = 0; c < C; c++) a += part[t * C + c];
            dq[r2] = a;
        }
    }
}

static bool q_fill(const int64_t *off_host, int R, QSeg &seg)
{
    if (R < 1 || R > Q_MAX_RENDERS) return false;
    seg.R = R;
    for (int r = 0; r <= R; r++) seg.off[r] = off_host[r];
    for (int r = R + 1; r <= Q_MAX_RENDERS; r++) seg.off[r] = off_host[R];
    return true;
}

static int q_nbx(const QSeg &seg)
{
    long long longest = 1;
    for (int r = 0; r < seg.R; r++) longest = seg.off[r + 1] - seg.off[r] > longest ? seg.off[r + 1] - seg.off[r] : longest;
    return (int)((longest + 63) / 64);
}

// STE_binary forward of a hash table (reference utils/encodings.py:375-392: sign with 0 -> +1) together with the count of its
// +1 entries, which the hash-bit term of the loss needs (pipeline/train.py:456): one pass instead of compare / where / sum.
__global__ void __launch_bounds__(256) k_ste_binary_count(const float *__restrict__ x, long long n, float *__restrict__ y,
                                                          float *__restrict__ count)
{
    __shared__ float red[4];
    float c = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const bool pos = x[i] >= 0.f;
        y[i] = pos ? 1.0f : -1.0f;
        c += pos ? 1.f : 0.f;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) c += __shfl_xor(c, m, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(count, (red[0] + red[1]) + (red[2] + red[3]));      // integer-valued sums below 2^24: exact
}

// The same for every hash table of the model in ONE launch (four tables per fitting step), the hash-bit term itself, and the
// straight-through backward with the bit term's gradient folded in — the per-table PyTorch form was ~40 one-table launches a step.
constexpr int STE_MAX_TABLES = 8;
constexpr int STE_CHUNK = 4096;          // elements per block
struct SteTables {
    const float *x[STE_MAX_TABLES];
    const float *g[STE_MAX_TABLES];
    float *y[STE_MAX_TABLES];
    long long n[STE_MAX_TABLES];
    int first_block[STE_MAX_TABLES + 1];
    int tables;
};

__device__ __forceinline__ int ste_table_of(const SteTables &t, int b)
{
    int k = 0;
#pragma unroll
    for (int q = 1; q < STE_MAX_TABLES; q++)
        if (q < t.tables && b >= t.first_block[q]) k = q;
    return k;
}

__global__ void __launch_bounds__(256) k_ste_binary_count_many(SteTables t, float *__restrict__ counts)
{
    __shared__ float red[4];
    const int k = ste_table_of(t, (int)blockIdx.x);
    const float *__restrict__ x = t.x[0];
    float *__restrict__ y = t.y[0];
    long long n = t.n[0];
    int fb = t.first_block[0];
#pragma unroll
    for (int q = 1; q < STE_MAX_TABLES; q++)
        if (k == q) { x = t.x[q]; y = t.y[q]; n = t.n[q]; fb = t.first_block[q]; }
    const long long i0 = (long long)((int)blockIdx.x - fb) * STE_CHUNK;
    float c = 0.f;
#pragma unroll 4
    for (int e = threadIdx.x; e < STE_CHUNK; e += 256) {
        const long long i = i0 + e;
        if (i < n) {
            const bool pos = x[i] >= 0.f;
            y[i] = pos ? 1.0f : -1.0f;
            c += pos ? 1.f : 0.f;
        }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) c += __shfl_xor(c, m, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(counts + k, (red[0] + red[1]) + (red[2] + red[3]));      // integer-valued sums below 2^24: exact
}

// out[0] = bits = n1 (-log2 p) + n0 (-log2 (1 - p)) + 32, p = n1 / total clamped to [1e-6, 1 - 1e-6] (reference
// utils/encodings.py:34-51, float32 as there); out[1] = d bits / d n1 = log2((1 - p) / p) (the terms through p cancel where p
// is not clamped and vanish where it is)
__global__ void k_table_bits(const float *__restrict__ counts, int tables, float total, float *__restrict__ out)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float ones = 0.f;
    for (int k = 0; k < tables; k++) ones += counts[k];
    const float p = fminf(fmaxf(ones / total, 1e-6f), 1.f - 1e-6f);
    out[0] = ones * (-log2f(p)) + (total - ones) * (-log2f(1.f - p)) + 32.f;
    out[1] = log2f((1.f - p) / p);
}

// grad x = (|x| <= 1) ? grad y + 0.5 * grad count : 0   (count = (sum of y + n) / 2: every entry's share of it is 1/2)
__global__ void __launch_bounds__(256) k_ste_binary_bwd_many(SteTables t, const float *__restrict__ count_grads, int count_stride)
{
    const int k = ste_table_of(t, (int)blockIdx.x);
    const float *__restrict__ x = t.x[0], *__restrict__ g = t.g[0];
    float *__restrict__ y = t.y[0];
    long long n = t.n[0];
    int fb = t.first_block[0];
#pragma unroll
    for (int q = 1; q < STE_MAX_TABLES; q++)
        if (k == q) { x = t.x[q]; g = t.g[q]; y = t.y[q]; n = t.n[q]; fb = t.first_block[q]; }
    const float add = count_grads ? 0.5f * count_grads[k * count_stride] : 0.f;
    const long long i0 = (long long)((int)blockIdx.x - fb) * STE_CHUNK;
#pragma unroll 4
    for (int e = threadIdx.x; e < STE_CHUNK; e += 256) {
        const long long i = i0 + e;
        if (i < n) y[i] = fabsf(x[i]) <= 1.f ? (g ? g[i] : 0.f) + add : 0.f;
    }
}

// STE_multistep of a render's rows (reference utils/encodings.py:395-420 as guassian.py:205-207 calls it: the centre's numerator is
// the mean of the WHOLE parameter, handed in; the bounds are truncated to integers): bounds[r] = trunc(x_mean / mean_r(q) -+ 15000)
__global__ void __launch_bounds__(256) k_ste_bounds(const float *__restrict__ part, int nbx, QSeg seg, int has_q, float q_scalar,
                                                    const float *__restrict__ x_mean, float *__restrict__ bounds)
{
    __shared__ float red[4];
    for (int r = 0; r < seg.R; r++) {
        float sq = 0.f;
        for (int b = threadIdx.x; b < nbx; b += 256) sq += part[((size_t)r * nbx + b) * 2 + 1];
        sq = q_block_sum(sq, red);
        if (threadIdx.x == 0) {
            const float rows = (float)(seg.off[r + 1] - seg.off[r]);
            const float qm = has_q ? sq / fmaxf(rows, 1.f) : q_scalar;
            const float c = x_mean[0] / qm;
            bounds[2 * r] = truncf(c - Q_CLAMP_STEPS);
            bounds[2 * r + 1] = truncf(c + Q_CLAMP_STEPS);
        }
    }
}

// y = x1 + (round(x1 / Q) Q - x1), x1 = clamp(x / Q, lo_r, hi_r) Q — the expression of the tensor form, operation by operation
__global__ void __launch_bounds__(256) k_ste_fwd(const float *__restrict__ x, const float *__restrict__ q, float q_scalar,
                                                 const float *__restrict__ bounds, QSeg seg, int C, float *__restrict__ y)
{
#pragma clang fp contract(off)
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long n = seg.off[seg.R] * C;
    if (e >= n) return;
    const long long row = e / C;
    const int r = q_render_of(seg, row);
    const float Q = q ? q[row] : q_scalar;
    const float x1 = fminf(fmaxf(x[e] / Q, bounds[2 * r]), bounds[2 * r + 1]) * Q;
    const float rq = rintf(x1 / Q) * Q;
    y[e] = x1 + (rq - x1);
}

}  // namespace gsvc

using namespace gsvc;

extern "C" int gsvc_ste_quant_forward(const float *x, const float *q_rows, float q_scalar, const float *x_mean,
                                      const int64_t *row_offsets_host, int32_t R, int32_t C, float *scratch, float *bounds, float *y,
                                      void *stream)
{
    QSeg seg;
    GSVC_REQUIRE(row_offsets_host && q_fill(row_offsets_host, R, seg) && C > 0, "ste_quant_forward: 1..8 renders, C > 0");
    GSVC_REQUIRE(scratch && bounds && x_mean, "ste_quant_forward: NULL pointer");
    hipStream_t s = (hipStream_t)stream;
    const long long rows = seg.off[R];
    GSVC_REQUIRE(rows == 0 || (x && y), "ste_quant_forward: NULL pointer");
    const int nbx = q_nbx(seg);
    ProfScope _p("k_ste_quant", s);
    if (q_rows) hipLaunchKernelGGL(k_quant_sums, dim3(nbx, R), dim3(256), 0, s, x, q_rows, seg, C, scratch);
    hipLaunchKernelGGL(k_ste_bounds, dim3(1), dim3(256), 0, s, scratch, nbx, seg, q_rows ? 1 : 0, q_scalar, x_mean, bounds);
    if (rows)
        hipLaunchKernelGGL(k_ste_fwd, dim3((unsigned)((rows * C + 255) / 256)), dim3(256), 0, s, x, q_rows, q_scalar, bounds, seg, C, y);
    return check_launch("ste_quant_forward");
}

extern "C" int64_t gsvc_noise_quant_scratch_floats(const int64_t *row_offsets_host, int32_t R)
{
    QSeg seg;
    if (!row_offsets_host || !q_fill(row_offsets_host, R, seg)) return -1;
    return (int64_t)2 * R * q_nbx(seg);
}

extern "C" int gsvc_noise_quant_forward(const float *x, const float *q_rows, float q_scalar, const float *noise,
                                        const int64_t *row_offsets_host, int32_t R, int32_t C, float *scratch, float *centre,
                                        float *y, void *stream)
{
    QSeg seg;
    GSVC_REQUIRE(row_offsets_host && q_fill(row_offsets_host, R, seg) && C > 0, "noise_quant_forward: 1..8 renders, C > 0");
    GSVC_REQUIRE(scratch && centre, "noise_quant_forward: NULL pointer");
    hipStream_t s = (hipStream_t)stream;
    const long long rows = seg.off[R];
    GSVC_REQUIRE(rows == 0 || (x && noise && y), "noise_quant_forward: NULL pointer");
    const int nbx = q_nbx(seg);
    {
        ProfScope _p("k_quant_sums", s);
        hipLaunchKernelGGL(k_quant_sums, dim3(nbx, R), dim3(256), 0, s, x, q_rows, seg, C, scratch);
    }
    hipLaunchKernelGGL(k_quant_centre, dim3(1), dim3(256), 0, s, scratch, nbx, seg, C, q_rows ? 1 : 0, q_scalar, centre);
    if (rows) {
        ProfScope _p("k_quant_fwd", s);
        hipLaunchKernelGGL(k_quant_fwd, dim3((unsigned)((rows * C + 255) / 256)), dim3(256), 0, s, x, q_rows, q_scalar, noise, centre,
                           seg, C, y);
    }
    return check_launch("noise_quant_forward");
}

extern "C" int gsvc_noise_quant_backward(const float *grad_y, const float *x, const float *q_rows, float q_scalar,
                                         const float *noise, const float *centre, const int64_t *row_offsets_host, int32_t R,
                                         int32_t C, float *grad_x, float *grad_q_rows, void *stream)
{
    QSeg seg;
    GSVC_REQUIRE(row_offsets_host && q_fill(row_offsets_host, R, seg) && C > 0 && C <= 256, "noise_quant_backward: 1..8 renders, 0 < C <= 256");
    const long long rows = seg.off[R];
    if (rows == 0) return GSVC_OK;
    GSVC_REQUIRE(grad_y && x && noise && centre && grad_x, "noise_quant_backward: NULL pointer");
    hipStream_t s = (hipStream_t)stream;
    const int rpb = 256 / C;
    ProfScope _p("k_quant_bwd", s);
    hipLaunchKernelGGL(k_quant_bwd, dim3((unsigned)((rows + rpb - 1) / rpb)), dim3(256), 0, s, grad_y, x, q_rows, q_scalar, noise, centre,
                       seg, C, rpb, grad_x, grad_q_rows);
    return check_launch("noise_quant_backward");
}

extern "C" int gsvc_ste_binary_count(const float *x, int64_t n, float *y, float *count, void *stream)
{
    GSVC_REQUIRE(n >= 0 && n < (1ll << 24), "ste_binary_count: the count must stay exact in float32 (n < 2^24)");
    GSVC_REQUIRE(count, "ste_binary_count: NULL pointer");
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(count, 0, sizeof(float), s) != hipSuccess) {
        gsvc::set_error("ste_binary_count: hipMemsetAsync failed");
        return GSVC_E_LAUNCH;
    }
    if (n == 0) return GSVC_OK;
    GSVC_REQUIRE(x && y, "ste_binary_count: NULL pointer");
    long long blocks = (n + 1023) / 1024;
    if (blocks > 256) blocks = 256;          // one atomic per block on the same word: keep them few
    gsvc::ProfScope _prof("k_ste_binary", s);
    hipLaunchKernelGGL(gsvc::k_ste_binary_count, dim3((unsigned)blocks), dim3(256), 0, s, x, (long long)n, y, count);
    return gsvc::check_launch("ste_binary_count");
}

static bool ste_fill(const float *const *x, const float *const *g, float *const *y, const int64_t *n, int32_t tables, SteTables &t, int &blocks)
{
    if (!x || !y || !n || tables < 1 || tables > STE_MAX_TABLES) return false;
    blocks = 0;
    for (int k = 0; k < STE_MAX_TABLES; k++) {
        const int q = k < tables ? k : 0;
        if (n[q] < 0 || n[q] >= (1ll << 24) || (n[q] > 0 && (!x[q] || !y[q]))) return false;
        t.x[k] = x[q]; t.g[k] = g ? g[q] : nullptr; t.y[k] = y[q]; t.n[k] = n[q];
        t.first_block[k] = blocks;
        if (k < tables) blocks += (int)((n[q] + STE_CHUNK - 1) / STE_CHUNK);
    }
    t.first_block[STE_MAX_TABLES] = blocks;
    t.tables = tables;
    return true;
}

extern "C" int gsvc_ste_binary_count_many(const float *const *x, float *const *y, const int64_t *n, int32_t tables, float *counts,
                                          void *stream)
{
    SteTables t;
    int blocks = 0;
    GSVC_REQUIRE(ste_fill(x, nullptr, y, n, tables, t, blocks) && counts, "ste_binary_count_many: 1..8 tables of fewer than 2^24 entries");
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(counts, 0, sizeof(float) * tables, s) != hipSuccess) {
        gsvc::set_error("ste_binary_count_many: hipMemsetAsync failed");
        return GSVC_E_LAUNCH;
    }
    if (blocks == 0) return GSVC_OK;
    gsvc::ProfScope _prof("k_ste_binary", s);
    hipLaunchKernelGGL(gsvc::k_ste_binary_count_many, dim3((unsigned)blocks), dim3(256), 0, s, t, counts);
    return gsvc::check_launch("ste_binary_count_many");
}

extern "C" int gsvc_ste_binary_backward_many(const float *const *x, const float *const *grad_y, const int64_t *n, int32_t tables,
                                             const float *count_grads, int32_t count_grad_stride, float *const *grad_x, void *stream)
{
    SteTables t;
    int blocks = 0;
    GSVC_REQUIRE(ste_fill(x, grad_y, grad_x, n, tables, t, blocks), "ste_binary_backward_many: 1..8 tables of fewer than 2^24 entries");
    if (blocks == 0) return GSVC_OK;
    hipStream_t s = (hipStream_t)stream;
    gsvc::ProfScope _prof("k_ste_binary_bwd", s);
    hipLaunchKernelGGL(gsvc::k_ste_binary_bwd_many, dim3((unsigned)blocks), dim3(256), 0, s, t, count_grads, (int)count_grad_stride);
    return gsvc::check_launch("ste_binary_backward_many");
}

extern "C" int gsvc_table_bits(const float *counts, int32_t tables, int64_t total, float *out, void *stream)
{
    GSVC_REQUIRE(counts && out && tables >= 1 && total > 0, "table_bits: bad arguments");
    hipLaunchKernelGGL(gsvc::k_table_bits, dim3(1), dim3(64), 0, (hipStream_t)stream, counts, tables, (float)total, out);
    return gsvc::check_launch("table_bits");
}

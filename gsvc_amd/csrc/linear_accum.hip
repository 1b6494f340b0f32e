// gsvc_amd/csrc/linear_accum.hip — Y[M,N] = sum_p X_p[M,K_p] W_p in ONE launch (gfx950, fp32 MFMA).
//
// The six sub-networks of the three EntropyParamsNets read the same hash-grid feature matrix (reference
// scene/gaussian_model.py:198-232, 1569-1597), so the backward pass adds six products G_p W_p into its gradient.  As six launches
// of the layer kernel (the last five with the accumulate epilogue) every product paid the launch, the weight staging and a
// wave's serial first block alone, and the [rows, 192] sum crossed memory eleven times: 232 us per fitting step for 52 k rows.
// Here a wave keeps the accumulators of its (at most two) 16-row blocks in registers across the products; the workgroup
// re-stages the weight image between them and the sum is stored once.
#include "linear_ws.h"

namespace gsvc {

constexpr int ACC_MAX_JOBS = 8, ACC_THREADS = 512, ACC_NT = LIN_NT_MAX;

struct AccumJobs {
    int n;
    const float *X[ACC_MAX_JOBS];
    const float *W[ACC_MAX_JOBS];
    int K[ACC_MAX_JOBS];
};

// 64 k-rows of W [K][N] (N % 4 == 0, 16-byte aligned) as 6 float4 per thread: piece i = row k_lo + i % 64, columns 4 (i / 64) .. + 3.
// The lanes of a wave take 64 consecutive k of one column group: their LDS writes land on 64 consecutive words (with consecutive
// column groups per lane the addresses were 4 ld = 32 (mod 64) words apart — every lane on the same bank).
constexpr int ACC_PIECES = 64 * (ACC_NT * 16 / 4) / ACC_THREADS;
__device__ __forceinline__ void acc_stage_load(const float *__restrict__ W, int tid, int k_lo, int K, int N, float4 (&v)[ACC_PIECES])
{
#pragma unroll
    for (int u = 0; u < ACC_PIECES; u++) {
        const int i = tid + u * ACC_THREADS, k = k_lo + (i & 63), c = (i >> 6) * 4;
        v[u] = (k < K && c < N) ? *reinterpret_cast<const float4 *>(W + (size_t)k * N + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}
__device__ __forceinline__ void acc_stage_write(float *__restrict__ lds, int tid, int ld, int k_lo, const float4 (&v)[ACC_PIECES])
{
#pragma unroll
    for (int u = 0; u < ACC_PIECES; u++) {
        const int i = tid + u * ACC_THREADS, k = k_lo + (i & 63), c = (i >> 6) * 4;
        lds[c * ld + k] = v[u].x; lds[(c + 1) * ld + k] = v[u].y; lds[(c + 2) * ld + k] = v[u].z; lds[(c + 3) * ld + k] = v[u].w;
    }
}

// one product for the wave's two blocks: the workgroup's weight image replaced, then block 0's and block 1's MFMAs
__device__ __forceinline__ float4 acc_load_a(bool v4, __amdgpu_buffer_rsrc_t r, int off, int k0, int K)
{
    return v4 ? ws_load_a<4, true>(r, off, k0, K) : ws_load_a<2, true>(r, off, k0, K);      // uniform: 16- or 8-byte aligned rows
}

__device__ __forceinline__ void accum_product(const float *__restrict__ X, const float *__restrict__ W, int K, int N, long long rb0,
                                              long long rb1, long long RB, long long M, float *lds, int tid, int fr, int kq,
                                              v4f (&acc)[2][ACC_NT])
{
    constexpr int KGM = ACC_NT;
    const int ld = ws_ld(K), KG = (K + 15) >> 4, k16 = KG * 16;
    const bool v4 = (K & 3) == 0 && (reinterpret_cast<uintptr_t>(X) & 15) == 0;
    __syncthreads();      // every wave is done with the previous product's image
    if ((reinterpret_cast<uintptr_t>(W) & 15) == 0 && (N & 3) == 0) {
        // W [K][N] -> the zero-padded image [192][ld] in pieces of 64 k-rows: 6 float4 per thread and piece, the next piece's
        // loads in flight while this one is written (all of a 192 x 160 matrix at once is 72 registers per thread: with the 96
        // accumulators live that spilled)
        float4 v0[ACC_PIECES], v1[ACC_PIECES];
        acc_stage_load(W, tid, 0, K, N, v0);
        if (k16 > 64) acc_stage_load(W, tid, 64, K, N, v1);
        acc_stage_write(lds, tid, ld, 0, v0);
        if (k16 > 128) acc_stage_load(W, tid, 128, K, N, v0);
        if (k16 > 64) acc_stage_write(lds, tid, ld, 64, v1);
        if (k16 > 128) acc_stage_write(lds, tid, ld, 128, v0);
    } else {
        for (int i = tid; i < k16 * ACC_NT * 16; i += ACC_THREADS) {
            const int k = i / (ACC_NT * 16), n = i - k * (ACC_NT * 16);
            lds[n * ld + k] = (n < N && k < K) ? W[(size_t)k * N + n] : 0.f;
        }
    }
    // block 0's fragments are requested behind the staging (its pieces and they together would not fit the register budget next
    // to the 96 accumulators; their flight overlaps the wait at the barrier); block 1's take a group's registers as soon as block
    // 0's MFMAs have consumed it (k_linear_ws's register-neutral prefetch)
    float4 a[KGM];
    const __amdgpu_buffer_rsrc_t r0 = ws_block_rsrc(X, rb0, RB, M, K), r1 = ws_block_rsrc(X, rb1, RB, M, K);
    const int voff = fr * K * 4 + 16 * kq;
#pragma unroll
    for (int g = 0; g < KGM; g++) a[g] = acc_load_a(v4, r0, voff + 64 * g, 16 * g + 4 * kq, K);
    __syncthreads();
    const float *wb = lds + fr * ld + 4 * kq;
#pragma unroll
    for (int b = 0; b < 2; b++) {
#pragma unroll
        for (int g = 0; g < KGM; g++) {
            if (g < KG) {
                const float4 x = a[g];
#pragma unroll
                for (int t0 = 0; t0 < ACC_NT; t0 += 4) {
                    float4 w[4];
#pragma unroll
                    for (int tt = 0; tt < 4; tt++) w[tt] = *reinterpret_cast<const float4 *>(wb + (t0 + tt) * 16 * ld + 16 * g);
#pragma unroll
                    for (int tt = 0; tt < 4; tt++) acc[b][t0 + tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[tt].x, x.x, acc[b][t0 + tt], 0, 0, 0);
#pragma unroll
                    for (int tt = 0; tt < 4; tt++) acc[b][t0 + tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[tt].y, x.y, acc[b][t0 + tt], 0, 0, 0);
#pragma unroll
                    for (int tt = 0; tt < 4; tt++) acc[b][t0 + tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[tt].z, x.z, acc[b][t0 + tt], 0, 0, 0);
#pragma unroll
                    for (int tt = 0; tt < 4; tt++) acc[b][t0 + tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[tt].w, x.w, acc[b][t0 + tt], 0, 0, 0);
                }
                if (b == 0) a[g] = acc_load_a(v4, r1, voff + 64 * g, 16 * g + 4 * kq, K);
            }
            __builtin_amdgcn_sched_barrier(0);      // keep the groups in order (as k_linear_ws: hoisted B reads blow the register budget)
        }
    }
}

// Persistent grid of <= 256 workgroups x 8 waves; wave w of workgroup b owns the blocks b + w * grid and that + 8 * grid (the host
// guarantees ceil(M / 16) <= 16 * grid).  A wave without rows multiplies zeros (empty descriptors) and stores nothing.
__global__ void __launch_bounds__(ACC_THREADS) k_linear_accum_many(AccumJobs jobs, float *__restrict__ Y, long long M, int N, int sv)
{
    extern __shared__ float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, kq = lane >> 4;
    const long long RB = (M + 15) >> 4, stride = (long long)gridDim.x * (ACC_THREADS / 64);
    const long long rb0 = (long long)wave * gridDim.x + blockIdx.x, rb1 = rb0 + stride;
    v4f acc[2][ACC_NT];
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
        for (int t = 0; t < ACC_NT; t++) acc[b][t] = (v4f){0.f, 0.f, 0.f, 0.f};
    for (int p = 0; p < jobs.n; p++) {
        // by-value arrays indexed by the (uniform) loop counter: picked with selects, not through scratch
        const float *X = jobs.X[0], *W = jobs.W[0];
        int K = jobs.K[0];
#pragma unroll
        for (int q = 1; q < ACC_MAX_JOBS; q++)
            if (p == q) { X = jobs.X[q]; W = jobs.W[q]; K = jobs.K[q]; }
        // (rows of 8-byte alignment at least: the host checks; the k-groups past K cost a uniform branch each)
        accum_product(X, W, K, N, rb0, rb1, RB, M, lds, tid, fr, kq, acc);
    }
    const int c00 = 4 * kq, yoff = (fr * N + c00) * 4;
#pragma unroll
    for (int b = 0; b < 2; b++) {
        const __amdgpu_buffer_rsrc_t ry = ws_block_rsrc(Y, b ? rb1 : rb0, RB, M, N);
#pragma unroll
        for (int t = 0; t < ACC_NT; t++) {
            float v[4] = {acc[b][t][0], acc[b][t][1], acc[b][t][2], acc[b][t][3]};
            ws_store4(ry, yoff + 64 * t, c00 + 16 * t, N, sv, v);
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------------
// The forward counterpart: Y_p[M,N_p] = X[M,K] W_p^T + b_p (+ the GELU pair) for several layers that read the SAME input — the six
// first layers of the entropy networks.  A wave reads the fragments of its (at most two) blocks once and keeps them across
// the products; the workgroup re-stages the weight image per product.  W_p is [N_p][K] (output-major, as nn.Linear stores it).
constexpr int SH_MAX_JOBS = 8, SH_NT = 10;      // N_p <= 160: with K = 192 the image is 160 x 200 words = 128 KB

struct SharedInputJobs {
    int n;
    const float *W[SH_MAX_JOBS], *bias[SH_MAX_JOBS];
    float *Y[SH_MAX_JOBS], *Y2[SH_MAX_JOBS];      // Y2 != NULL: Y = pre-activation, Y2 = GELU(Y)
    int N[SH_MAX_JOBS];
};

// 32 n-rows of W [N][K] (K % 4 == 0, 16-byte aligned rows) as float4 pieces along k: piece i = row n_lo + i / kp, k = 4 (i % kp)
constexpr int SH_ROWS = 32, SH_PIECES = SH_ROWS * (ACC_NT * 16 / 4) / ACC_THREADS;      // 3 per thread at K = 192
__device__ __forceinline__ void sh_stage_load(const float *__restrict__ W, int tid, int n_lo, int N, int K, int kp, float4 (&v)[SH_PIECES])
{
#pragma unroll
    for (int u = 0; u < SH_PIECES; u++) {
        const int i = tid + u * ACC_THREADS, r = i / kp, k = (i - r * kp) * 4, n = n_lo + r;
        v[u] = (r < SH_ROWS && n < N && k < K) ? *reinterpret_cast<const float4 *>(W + (size_t)n * K + k) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}
__device__ __forceinline__ void sh_stage_write(float *__restrict__ lds, int tid, int ld, int n_lo, int kp, const float4 (&v)[SH_PIECES])
{
#pragma unroll
    for (int u = 0; u < SH_PIECES; u++) {
        const int i = tid + u * ACC_THREADS, r = i / kp, k = (i - r * kp) * 4;
        if (r < SH_ROWS) *reinterpret_cast<float4 *>(lds + (n_lo + r) * ld + k) = v[u];
    }
}

__global__ void __launch_bounds__(ACC_THREADS) k_linear_shared_input(const float *__restrict__ X, SharedInputJobs jobs, long long M, int K,
                                                                     int img_floats)
{
    extern __shared__ float lds[];
    float *s_bias = lds + img_floats;      // behind the weight image (static LDS would come off the 160 KB the attribute asks for)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, kq = lane >> 4;
    const long long RB = (M + 15) >> 4, stride = (long long)gridDim.x * (ACC_THREADS / 64);
    const long long rb0 = (long long)wave * gridDim.x + blockIdx.x, rb1 = rb0 + stride;
    const int ld = ws_ld(K), KG = (K + 15) >> 4, kp = (KG * 16) >> 2;
    // the two blocks' fragments: read once, used by every product
    float4 a[2][ACC_NT];
    {
        const __amdgpu_buffer_rsrc_t r0 = ws_block_rsrc(X, rb0, RB, M, K), r1 = ws_block_rsrc(X, rb1, RB, M, K);
        const int voff = fr * K * 4 + 16 * kq;
#pragma unroll
        for (int g = 0; g < ACC_NT; g++) {
            a[0][g] = ws_load_a<4, true>(r0, voff + 64 * g, 16 * g + 4 * kq, K);
            a[1][g] = ws_load_a<4, true>(r1, voff + 64 * g, 16 * g + 4 * kq, K);
        }
    }
    const float *wb = lds + fr * ld + 4 * kq;
    for (int p = 0; p < jobs.n; p++) {
        const float *W = jobs.W[0], *bias = jobs.bias[0];
        float *Y = jobs.Y[0], *Y2 = jobs.Y2[0];
        int N = jobs.N[0];
#pragma unroll
        for (int q = 1; q < SH_MAX_JOBS; q++)
            if (p == q) { W = jobs.W[q]; bias = jobs.bias[q]; Y = jobs.Y[q]; Y2 = jobs.Y2[q]; N = jobs.N[q]; }
        const int NT = (N + 15) >> 4;
        __syncthreads();      // every wave is done with the previous product's image and bias
        {
            // the image [16 NT][ld] in pieces of 32 n-rows, the next piece's loads in flight while this one is written
            float4 v0[SH_PIECES], v1[SH_PIECES];
            sh_stage_load(W, tid, 0, N, K, kp, v0);
            for (int n_lo = 0; n_lo < NT * 16; n_lo += 2 * SH_ROWS) {
                if (n_lo + SH_ROWS < NT * 16) sh_stage_load(W, tid, n_lo + SH_ROWS, N, K, kp, v1);
                sh_stage_write(lds, tid, ld, n_lo, kp, v0);
                if (n_lo + 2 * SH_ROWS < NT * 16) sh_stage_load(W, tid, n_lo + 2 * SH_ROWS, N, K, kp, v0);
                if (n_lo + SH_ROWS < NT * 16) sh_stage_write(lds, tid, ld, n_lo + SH_ROWS, kp, v1);
            }
            if (tid < SH_NT * 16) s_bias[tid] = (bias && tid < N) ? bias[tid] : 0.f;
        }
        __syncthreads();
        const int sv = (N % 4 == 0 && ((reinterpret_cast<uintptr_t>(Y) | reinterpret_cast<uintptr_t>(Y2)) & 15) == 0) ? 4
                       : ((N % 2 == 0 && ((reinterpret_cast<uintptr_t>(Y) | reinterpret_cast<uintptr_t>(Y2)) & 7) == 0) ? 2 : 1);
        const int c00 = 4 * kq, yoff = (fr * N + c00) * 4;
#pragma unroll
        for (int b = 0; b < 2; b++) {
            v4f acc[SH_NT];
#pragma unroll
            for (int t = 0; t < SH_NT; t++) acc[t] = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int g = 0; g < ACC_NT; g++) {
                if (g < KG) {
                    const float4 x = a[b][g];
#pragma unroll
                    for (int t0 = 0; t0 < SH_NT; t0 += 2) {
                        if (t0 < NT) {      // tiles in pairs (NT is uniform): N = 50 takes 4 of the 10
                            const float4 w0 = *reinterpret_cast<const float4 *>(wb + t0 * 16 * ld + 16 * g);
                            const float4 w1 = *reinterpret_cast<const float4 *>(wb + (t0 + 1) * 16 * ld + 16 * g);
                            acc[t0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.x, x.x, acc[t0], 0, 0, 0);
                            acc[t0 + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.x, x.x, acc[t0 + 1], 0, 0, 0);
                            acc[t0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.y, x.y, acc[t0], 0, 0, 0);
                            acc[t0 + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.y, x.y, acc[t0 + 1], 0, 0, 0);
                            acc[t0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.z, x.z, acc[t0], 0, 0, 0);
                            acc[t0 + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.z, x.z, acc[t0 + 1], 0, 0, 0);
                            acc[t0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.w, x.w, acc[t0], 0, 0, 0);
                            acc[t0 + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.w, x.w, acc[t0 + 1], 0, 0, 0);
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            const long long rb = b ? rb1 : rb0;
            const __amdgpu_buffer_rsrc_t ry = ws_block_rsrc(Y, rb, RB, M, N), r2 = ws_block_rsrc(Y2, Y2 ? rb : RB, RB, M, N);
#pragma unroll
            for (int t = 0; t < SH_NT; t++) {
                if (t < NT) {
                    const float4 bv = *reinterpret_cast<const float4 *>(s_bias + c00 + 16 * t);
                    float v[4] = {acc[t][0] + bv.x, acc[t][1] + bv.y, acc[t][2] + bv.z, acc[t][3] + bv.w};
                    ws_store4(ry, yoff + 64 * t, c00 + 16 * t, N, sv, v);
                    float v2[4] = {gelu_f(v[0]), gelu_f(v[1]), gelu_f(v[2]), gelu_f(v[3])};
                    ws_store4(r2, yoff + 64 * t, c00 + 16 * t, N, sv, v2);      // dropped through the empty descriptor when Y2 is NULL
                }
            }
        }
    }
}

}  // namespace gsvc

using namespace gsvc;

extern "C" int gsvc_linear_accumulate_many(const gsvc_accum_job *jobs, int32_t n_jobs, float *Y, int64_t M, int32_t N, void *stream)
{
    GSVC_REQUIRE(jobs && n_jobs >= 1 && M >= 0 && N > 0, "linear_accumulate_many: bad arguments");
    if (n_jobs > ACC_MAX_JOBS || N > LIN_NT_MAX * 16 || M > (int64_t)16 * 256 * 16) {
        set_error("linear_accumulate_many: at most %d products, N <= %d, M <= 65536", ACC_MAX_JOBS, LIN_NT_MAX * 16);
        return GSVC_E_UNSUPPORTED;
    }
    if (M == 0) return GSVC_OK;
    GSVC_REQUIRE(Y, "linear_accumulate_many: NULL pointer");
    AccumJobs t;
    t.n = n_jobs;
    int kmax = 1;
    for (int i = 0; i < ACC_MAX_JOBS; i++) {
        const gsvc_accum_job &q = jobs[i < n_jobs ? i : 0];
        GSVC_REQUIRE(q.X && q.W && q.K > 0, "linear_accumulate_many: bad job %d", i);
        if (q.K > LIN_NT_MAX * 16 || (q.K & 1) || (reinterpret_cast<uintptr_t>(q.X) & 7)) {
            set_error("linear_accumulate_many: K=%d: at most %d, even, rows 8-byte aligned", q.K, LIN_NT_MAX * 16);
            return GSVC_E_UNSUPPORTED;
        }
        t.X[i] = q.X; t.W[i] = q.W; t.K[i] = q.K;
        if (q.K > kmax) kmax = q.K;
    }
    const size_t lds = (size_t)ACC_NT * 16 * ws_ld(kmax) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_linear_accum_many), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    const long long RB = (M + 15) / 16;
    long long grid = (RB + 7) / 8;      // one block per wave while that is <= 256 workgroups, then two
    if (grid > 256) grid = 256;
    const int sv = (N % 4 == 0 && (reinterpret_cast<uintptr_t>(Y) & 15) == 0) ? 4 : ((N % 2 == 0 && (reinterpret_cast<uintptr_t>(Y) & 7) == 0) ? 2 : 1);
    hipStream_t s = (hipStream_t)stream;
    ProfScope _prof("k_linear_accum", s);
    hipLaunchKernelGGL(k_linear_accum_many, dim3((unsigned)grid), dim3(ACC_THREADS), lds, s, t, Y, (long long)M, (int)N, sv);
    return check_launch("linear_accumulate_many");
}

extern "C" int gsvc_linear_forward_shared_input(const float *X, int64_t M, int32_t K, const gsvc_shared_input_job *jobs, int32_t n_jobs,
                                                void *stream)
{
    GSVC_REQUIRE(jobs && n_jobs >= 1 && M >= 0 && K > 0, "linear_forward_shared_input: bad arguments");
    if (n_jobs > SH_MAX_JOBS || K > LIN_NT_MAX * 16 || (K & 3) || (reinterpret_cast<uintptr_t>(X) & 15) || M > (int64_t)16 * 256 * 16) {
        set_error("linear_forward_shared_input: at most %d products, K <= %d and a multiple of 4, X 16-byte aligned, M <= 65536", SH_MAX_JOBS,
                  LIN_NT_MAX * 16);
        return GSVC_E_UNSUPPORTED;
    }
    if (M == 0) return GSVC_OK;
    GSVC_REQUIRE(X, "linear_forward_shared_input: NULL pointer");
    SharedInputJobs t;
    t.n = n_jobs;
    int nmax = 1;
    for (int i = 0; i < SH_MAX_JOBS; i++) {
        const gsvc_shared_input_job &q = jobs[i < n_jobs ? i : 0];
        GSVC_REQUIRE(q.W && q.Y && q.N > 0, "linear_forward_shared_input: bad job %d", i);
        if (q.N > SH_NT * 16 || (reinterpret_cast<uintptr_t>(q.W) & 15)) {
            set_error("linear_forward_shared_input: N=%d: at most %d, W 16-byte aligned", q.N, SH_NT * 16);
            return GSVC_E_UNSUPPORTED;
        }
        t.W[i] = q.W; t.bias[i] = q.bias; t.Y[i] = q.Y; t.Y2[i] = q.Y2; t.N[i] = q.N;
        if (q.N > nmax) nmax = q.N;
    }
    const int img_floats = ((nmax + 31) / 32 * 32) * ws_ld(K);      // whole tile PAIRS (the product loop's unit)
    const size_t lds = ((size_t)img_floats + SH_NT * 16) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_linear_shared_input), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    const long long RB = (M + 15) / 16;
    long long grid = (RB + 7) / 8;
    if (grid > 256) grid = 256;
    hipStream_t s = (hipStream_t)stream;
    ProfScope _prof("k_linear_shared_input", s);
    hipLaunchKernelGGL(k_linear_shared_input, dim3((unsigned)grid), dim3(ACC_THREADS), lds, s, X, t, (long long)M, (int)K, img_floats);
    return check_launch("linear_forward_shared_input");
}

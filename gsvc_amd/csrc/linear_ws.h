// gsvc_amd/csrc/linear_ws.h — the weight-stationary MFMA linear kernel (template) shared by the translation units that
// instantiate it: linear.hip (plain epilogue), linear_epi_{a,b,c,d}.hip (epilogue programs, by column-tile count),
// linear_wgrad.hip (uses the fragment-load helpers).  Split so that the ~650 kernel instantiations compile in parallel.
#pragma once
#include "common.h"

namespace gsvc {

typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int LIN_NT_MAX = 12;      // K, N <= 192
constexpr int GSVC_LIN_EPI_REGS = 36;     // 2 x 16 auxiliary operands + descriptors held through an EPI epilogue

// ---------------------------------------------------------------------------------------------------------
// Weight-stationary: the whole zero-padded weight matrix sits in LDS for the life
// of a persistent workgroup (<= 150 KiB of the 160 KiB), and X never touches LDS: each wave owns 16-row blocks and
// reads them straight into MFMA A fragments, one 16-byte load per lane per 16 k's, the next row block in flight
// while the current one is multiplied.  The four k's of an MFMA step may be ANY four k's as long as A and B agree,
// so lane (r, kq) takes X[r][16g + 4kq .. +3] as one float4 and the matching B fragment is one ds_read_b128 of
// W[n][16g + 4kq .. +3]; the i-th component of both feeds the i-th MFMA of the group.  The LDS row stride is
// 8 (mod 64) dwords, which makes those b128 reads conflict-free for the hardware's 16-lane groups.
// Per 16 rows x 16 k's x 16 columns: one 4-cycle LDS read per four 32-cycle MFMAs -> the kernel is bound by MFMA
// issue (K, N ~ 192) or by the HBM stream of X and Y (K, N <= 100).
__host__ __device__ inline int ws_ld(int K) { return ((K + 63) / 64) * 64 + 8; }

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
constexpr int BUF_OOB = 0x7fffffff;     // a byte offset past any descriptor's range: loads return 0, stores are dropped

// Fragment load through a buffer descriptor that covers exactly the valid rows of one 16-row block: rows past M
// and (by the explicit offset select) k's past K come back as zeros from the hardware range check — no branches,
// no clamps, so the compiler counts outstanding loads exactly and the loads can stay in flight across the MFMAs.
template <int VEC, bool CHECK>
__device__ __forceinline__ float4 ws_load_a(__amdgpu_buffer_rsrc_t rs, int off, int k0, int K)
{
    // off = byte offset of X[row][k0] inside the block; CHECK = this k-group may reach past K (only the last two can)
    float4 v;
    if (VEC == 4) {
        const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rs, (!CHECK || k0 < K) ? off : BUF_OOB, 0, 0);
        v = make_float4(__uint_as_float(t.x), __uint_as_float(t.y), __uint_as_float(t.z), __uint_as_float(t.w));
    } else if (VEC == 2) {
        const u32x2 lo = __builtin_amdgcn_raw_buffer_load_b64(rs, (!CHECK || k0 < K) ? off : BUF_OOB, 0, 0);
        const u32x2 hi = __builtin_amdgcn_raw_buffer_load_b64(rs, (!CHECK || k0 + 2 < K) ? off + 8 : BUF_OOB, 0, 0);
        v = make_float4(__uint_as_float(lo.x), __uint_as_float(lo.y), __uint_as_float(hi.x), __uint_as_float(hi.y));
    } else {
        const unsigned x0 = __builtin_amdgcn_raw_buffer_load_b32(rs, (!CHECK || k0 < K) ? off : BUF_OOB, 0, 0);
        const unsigned x1 = __builtin_amdgcn_raw_buffer_load_b32(rs, (!CHECK || k0 + 1 < K) ? off + 4 : BUF_OOB, 0, 0);
        const unsigned x2 = __builtin_amdgcn_raw_buffer_load_b32(rs, (!CHECK || k0 + 2 < K) ? off + 8 : BUF_OOB, 0, 0);
        const unsigned x3 = __builtin_amdgcn_raw_buffer_load_b32(rs, (!CHECK || k0 + 3 < K) ? off + 12 : BUF_OOB, 0, 0);
        v = make_float4(__uint_as_float(x0), __uint_as_float(x1), __uint_as_float(x2), __uint_as_float(x3));
    }
    return v;
}

// All KGM fragments of one row block.  Groups 0..KGM-3 are always whole (K > 16 (KGM-2)), so their offsets are
// one VGPR plus an instruction immediate.
template <int VEC, int KGM>
__device__ __forceinline__ float4 ws_load_group(__amdgpu_buffer_rsrc_t rs, int voff, int kq, int K, int g)
{
    if (g < KGM - 2) return ws_load_a<VEC, false>(rs, voff + 64 * g, 16 * g + 4 * kq, K);
    return ws_load_a<VEC, true>(rs, voff + 64 * g, 16 * g + 4 * kq, K);
}

// Descriptor of the 16-row block `rb` of a row-major [M][ld] float matrix (zero bytes when rb is past the end).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t ws_block_rsrc(const float *base, long long rb, long long RB, long long M, int ld)
{
    const long long rows = rb < RB ? min((long long)16, M - rb * 16) : 0;
    const float *p = base + (rb < RB ? rb : 0) * 16 * ld;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(reinterpret_cast<uintptr_t>(p) & 0xffffffffu));
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(reinterpret_cast<uintptr_t>(p) >> 32));
    const int bytes = __builtin_amdgcn_readfirstlane((int)(rows * ld * 4));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(((uintptr_t)hi << 32) | lo), 0, bytes, 0x00020000);
}

// Epilogue program of the EPI variants (gsvc_linear_forward_ex): what happens to v = acc + bias before / beside the store.
// Every auxiliary matrix has Y's shape [M][N], so one offset serves them all.  The activations of the MLPs ride here:
// forward GELU leaves the pre-activation (Y) and the activated value (Y2), the dX products of a backward pass multiply by
// the derivative of the previous layer's activation, FiLM's product and its product rule are formed where gamma / g appear.
struct LinEpi {
    int mode;
    const float *aux1, *aux2;
    float *y2, *y3;
};

__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_grad_f(float x)
{
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    const float pdf = expf(-0.5f * x * x) * 0.39894228040143267794f;
    return cdf + x * pdf;
}

// Epilogue IO.  The MFMAs are issued with the weight fragment as the A operand, so a lane's four accumulator values of a
// column tile are FOUR CONSECUTIVE COLUMNS of one row: lane (fr, kq) holds Y[row fr][16 t + 4 kq .. + 3] — one 16-byte
// access per lane and tile (a wave instruction covers 16 rows x 64 bytes; the row-major D layout needed four 4-byte
// accesses of 4 rows x 64 bytes each).  sv = 4 / 2 / 1: widest access N and the pointers allow; a column group is inside
// [0, N) as a whole (sv 4), in halves (sv 2) or per element (sv 1); rows past M fall outside the block descriptor.
__device__ __forceinline__ void ws_store4(__amdgpu_buffer_rsrc_t r, int off, int c0, int N, int sv, const float (&v)[4])
{
    if (sv == 4) {
        const u32x4 t = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
        __builtin_amdgcn_raw_buffer_store_b128(t, r, c0 < N ? off : BUF_OOB, 0, 0);
    } else if (sv == 2) {
        const u32x2 lo = {__float_as_uint(v[0]), __float_as_uint(v[1])}, hi = {__float_as_uint(v[2]), __float_as_uint(v[3])};
        __builtin_amdgcn_raw_buffer_store_b64(lo, r, c0 < N ? off : BUF_OOB, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b64(hi, r, c0 + 2 < N ? off + 8 : BUF_OOB, 0, 0);
    } else {
#pragma unroll
        for (int i = 0; i < 4; i++)
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[i]), r, c0 + i < N ? off + 4 * i : BUF_OOB, 0, 0);
    }
}

__device__ __forceinline__ void ws_load4(__amdgpu_buffer_rsrc_t r, int off, int c0, int N, int sv, float (&v)[4])
{
    if (sv == 4) {
        const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(r, c0 < N ? off : BUF_OOB, 0, 0);
        v[0] = __uint_as_float(t.x); v[1] = __uint_as_float(t.y); v[2] = __uint_as_float(t.z); v[3] = __uint_as_float(t.w);
    } else if (sv == 2) {
        const u32x2 lo = __builtin_amdgcn_raw_buffer_load_b64(r, c0 < N ? off : BUF_OOB, 0, 0);
        const u32x2 hi = __builtin_amdgcn_raw_buffer_load_b64(r, c0 + 2 < N ? off + 8 : BUF_OOB, 0, 0);
        v[0] = __uint_as_float(lo.x); v[1] = __uint_as_float(lo.y); v[2] = __uint_as_float(hi.x); v[3] = __uint_as_float(hi.y);
    } else {
#pragma unroll
        for (int i = 0; i < 4; i++)
            v[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, c0 + i < N ? off + 4 * i : BUF_OOB, 0, 0));
    }
}

// One row block's epilogue for a fixed program: the auxiliary operands of four column tiles are requested together
// (one round trip per four tiles while the accumulators stay put), then computed and stored.
template <int NT, int MODE>
__device__ __forceinline__ void ws_epilogue(const v4f (&acc)[NT], const float *sbias, __amdgpu_buffer_rsrc_t ry, int yoff, int c00,
                                            int sv, long long rb, long long RB, long long M, int N, const LinEpi &ep)
{
    constexpr bool FILMS = MODE == GSVC_LIN_FILM || MODE == GSVC_LIN_FILM_GRAD;
    constexpr bool HAS1 = MODE >= GSVC_LIN_MUL_GELU_GRAD, HAS2 = FILMS;
    constexpr bool OUT2 = MODE == GSVC_LIN_GELU_DUAL || FILMS, OUT3 = MODE == GSVC_LIN_FILM_GRAD;
    const __amdgpu_buffer_rsrc_t r1 = ws_block_rsrc(HAS1 ? ep.aux1 : nullptr, HAS1 ? rb : RB, RB, M, N);
    const __amdgpu_buffer_rsrc_t r2 = ws_block_rsrc(HAS2 ? ep.aux2 : nullptr, HAS2 ? rb : RB, RB, M, N);
    const __amdgpu_buffer_rsrc_t q2 = ws_block_rsrc(OUT2 ? ep.y2 : nullptr, OUT2 ? rb : RB, RB, M, N);
    const __amdgpu_buffer_rsrc_t q3 = ws_block_rsrc(OUT3 ? ep.y3 : nullptr, OUT3 ? rb : RB, RB, M, N);
#pragma unroll
    for (int t0 = 0; t0 < NT; t0 += 4) {
        float x1[4][4], x2[4][4];
#pragma unroll
        for (int tt = 0; tt < 4; tt++) {
            const int t = t0 + tt;
            if (t >= NT) continue;
            if (HAS1) ws_load4(r1, yoff + 64 * t, c00 + 16 * t, N, sv, x1[tt]);
            if (HAS2) ws_load4(r2, yoff + 64 * t, c00 + 16 * t, N, sv, x2[tt]);
        }
#pragma unroll
        for (int tt = 0; tt < 4; tt++) {
            const int t = t0 + tt;
            if (t >= NT) continue;
            const float4 bv = *reinterpret_cast<const float4 *>(sbias + c00 + 16 * t);
            const float bb[4] = {bv.x, bv.y, bv.z, bv.w};
            float v[4], v2[4], v3[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                v[i] = acc[t][i] + bb[i];
                v2[i] = v3[i] = 0.f;
                if (MODE == GSVC_LIN_GELU_DUAL) v2[i] = gelu_f(v[i]);
                if (MODE == GSVC_LIN_TANH) v[i] = tanhf(v[i]);
                if (MODE == GSVC_LIN_SIGMOID) v[i] = 1.0f / (1.0f + expf(-v[i]));
                if (MODE == GSVC_LIN_MUL_GELU_GRAD) v[i] *= gelu_grad_f(x1[tt][i]);
                if (MODE == GSVC_LIN_MUL_RELU_MASK) v[i] = x1[tt][i] > 0.f ? v[i] : 0.f;
                if (MODE == GSVC_LIN_FILM) v2[i] = fmaf(v[i], x1[tt][i], x2[tt][i]);
                if (MODE == GSVC_LIN_FILM_GRAD) { v2[i] = v[i] * x1[tt][i]; v3[i] = v[i] * x2[tt][i]; }
                if (MODE == GSVC_LIN_ADD) v[i] += x1[tt][i];
            }
            ws_store4(ry, yoff + 64 * t, c00 + 16 * t, N, sv, v);
            if (OUT2) ws_store4(q2, yoff + 64 * t, c00 + 16 * t, N, sv, v2);
            if (OUT3) ws_store4(q3, yoff + 64 * t, c00 + 16 * t, N, sv, v3);
        }
    }
}

// W ([N][K], or [K][N] when w_in_out) -> the zero-padded LDS image [NT * 16][ld], VW floats per piece along W's contiguous
// dimension; every piece of the thread is in flight before the first LDS write.
template <int NT, int KGM, int THREADS, int VW>
__device__ __forceinline__ void ws_stage_weights(const float *__restrict__ W, float *__restrict__ lds, int tid, int ld, int k16, int K, int N,
                                                 int w_in_out)
{
    constexpr int PIECES = (NT * 16 * KGM * 16 / VW + THREADS - 1) / THREADS;      // per thread (k16 <= 16 KGM)
    const int inner = (w_in_out ? NT * 16 : k16) / VW;                              // pieces along the contiguous dimension (padded)
    const int total = inner * (w_in_out ? k16 : NT * 16);
    float v[PIECES][VW];
#pragma unroll
    for (int u = 0; u < PIECES; u++) {
        const int i = tid + u * THREADS;
        const int o = i / inner, c = (i - o * inner) * VW;          // outer index, first inner index
        const bool ok = i < total && (w_in_out ? (o < K && c < N) : (o < N && c < K));
        const float *src = W + (size_t)o * (w_in_out ? N : K) + c;
        if (VW == 4) {
            const float4 t = ok ? *reinterpret_cast<const float4 *>(src) : make_float4(0.f, 0.f, 0.f, 0.f);
            v[u][0] = t.x; v[u][1] = t.y; v[u][2] = t.z; v[u][3] = t.w;
        } else {
            const float2 t = ok ? *reinterpret_cast<const float2 *>(src) : make_float2(0.f, 0.f);
            v[u][0] = t.x; v[u][1] = t.y;
        }
    }
#pragma unroll
    for (int u = 0; u < PIECES; u++) {
        const int i = tid + u * THREADS;
        if (i >= total) continue;
        const int o = i / inner, c = (i - o * inner) * VW;
#pragma unroll
        for (int e = 0; e < VW; e++) {
            if (w_in_out) lds[(c + e) * ld + o] = v[u][e];          // piece = W[k = o][n = c ..]: column k of VW LDS rows
            else lds[o * ld + c + e] = v[u][e];                     // piece = W[n = o][k = c ..]
        }
    }
}

// NT = ceil(N/16) column tiles, KGM = compile-time bound on ceil(K/16), VEC = widest aligned load of an X row.
// THREADS = 1024 (4 waves per SIMD, 128 VGPRs) for the small shapes, 512 (2 waves per SIMD, 256 VGPRs) for the
// large ones.  Software pipeline without extra registers: as soon as the MFMAs of k-group g have consumed a[g],
// a[g] is reloaded with the NEXT row block's fragment, so the X stream overlaps the rest of the multiply and the
// store of the current block.
// w_in_out != 0: W is given as [K][N] (input-major), i.e. Y = X W — the dX = G W product of the backward pass
// without a transposed copy of W.
template <int NT, int KGM, int VEC, int THREADS, bool EPI>
__global__ void __launch_bounds__(THREADS) k_linear_ws(const float *__restrict__ X, const float *__restrict__ W,
                                                       const float *__restrict__ bias, float *__restrict__ Y,
                                                       long long M, int K, int N, int w_in_out, int relu, int sv, LinEpi ep)
{
    extern __shared__ float lds[];
    constexpr int WAVES = THREADS / 64;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, kq = lane >> 4;
    const int ld = ws_ld(K), KG = (K + 15) >> 4, k16 = KG * 16;
    const long long RB = (M + 15) >> 4, stride = (long long)gridDim.x * WAVES;
    // 16-row blocks are dealt wave-major: block rb goes to wave rb / gridDim of workgroup rb % gridDim.  M = 200 k rows are 3.05
    // blocks per wave with 4096 waves: dealt workgroup-major the 212 blocks of the fourth round all landed on the first 13
    // workgroups, whose SIMDs then had 16 blocks against 12 everywhere else — the kernel lasted a third longer than its average
    // CU needed (tools/micro/gemm_overlap.hip: 53 -> 43 us for the 100 x 100 layer's loop)
    long long rb = (long long)wave * gridDim.x + blockIdx.x;

    // first row block's fragments fly while the weights are staged
    float4 a[KGM];
    const int voff = fr * K * 4 + 16 * kq;
    {
        const __amdgpu_buffer_rsrc_t rx = ws_block_rsrc(X, rb, RB, M, K);
#pragma unroll
        for (int g = 0; g < KGM; g++) a[g] = ws_load_group<VEC, KGM>(rx, voff, kq, K, g);
    }
    // weights -> zero-padded LDS.  When W's rows allow 16- or 8-byte loads every thread requests ALL its pieces before it writes
    // the first (one memory round trip for the whole matrix; the scalar form below makes two to three, ~2 us each, per launch:
    // 192 x 150 went 131 -> 107 us, 100 x 100 55 -> 52)
    const int w_inner = w_in_out ? N : K;
    const bool w_v4 = (reinterpret_cast<uintptr_t>(W) & 15) == 0 && (w_inner & 3) == 0;
    const bool w_v2 = (reinterpret_cast<uintptr_t>(W) & 7) == 0 && (w_inner & 1) == 0;
    if (w_v4) ws_stage_weights<NT, KGM, THREADS, 4>(W, lds, tid, ld, k16, K, N, w_in_out);
    else if (w_v2) ws_stage_weights<NT, KGM, THREADS, 2>(W, lds, tid, ld, k16, K, N, w_in_out);
    else {
        const int total = k16 * NT * 16;
        for (int base = tid; base < total; base += THREADS * 8) {
            float v[8];
            int dst[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int i = base + u * THREADS;
                int n, k;
                if (w_in_out) { k = i / (NT * 16); n = i - k * (NT * 16); }
                else { n = i / k16; k = i - n * k16; }
                const bool ok = i < total && n < N && k < K;
                const size_t src = w_in_out ? (size_t)k * N + n : (size_t)n * K + k;
                v[u] = ok ? W[src] : 0.f;
                dst[u] = i < total ? n * ld + k : -1;
            }
#pragma unroll
            for (int u = 0; u < 8; u++)
                if (dst[u] >= 0) lds[dst[u]] = v[u];
        }
    }
    float *sbias = lds + NT * 16 * ld;     // bias (zero-padded) behind the weights: read back with ds_read, not VGPR-resident
    if (tid < NT * 16) sbias[tid] = (bias && tid < N) ? bias[tid] : 0.f;
    __syncthreads();
    // everything loaded so far has landed before the loop starts: inside it the only outstanding memory operations
    // are the loop's own, which lets the compiler wait for exactly the fragment it needs (vmcnt(n), not vmcnt(0))
#pragma unroll
    for (int g = 0; g < KGM; g++) asm volatile("" : "+v"(a[g].x), "+v"(a[g].y), "+v"(a[g].z), "+v"(a[g].w));

    const float *wb = lds + fr * ld + 4 * kq;
    const int c00 = 4 * kq, yoff = (fr * N + c00) * 4;      // this lane's four consecutive columns of row fr (per tile: + 16 t)
    for (; rb < RB; rb += stride) {
        const __amdgpu_buffer_rsrc_t rx = ws_block_rsrc(X, rb + stride, RB, M, K);   // empty past the end
        v4f acc[NT];
#pragma unroll
        for (int t = 0; t < NT; t++) acc[t] = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < KGM; g++) {
            if (g < KGM - 1 || g < KG) {      // KG is KGM or KGM-1: only the last group is a run-time decision
                const float4 ag = a[g];
#pragma unroll
                for (int t0 = 0; t0 < NT; t0 += 4) {
                    float4 b[4];
#pragma unroll
                    for (int tt = 0; tt < 4; tt++)
                        if (t0 + tt < NT) b[tt] = *reinterpret_cast<const float4 *>(wb + (t0 + tt) * 16 * ld + 16 * g);
#pragma unroll
                    for (int tt = 0; tt < 4; tt++)
                        if (t0 + tt < NT) acc[t0 + tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[tt].x, ag.x, acc[t0 + tt], 0, 0, 0);
#pragma unroll
                    for (int tt = 0; tt < 4; tt++)
                        if (t0 + tt < NT) acc[t0 + tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[tt].y, ag.y, acc[t0 + tt], 0, 0, 0);
#pragma unroll
                    for (int tt = 0; tt < 4; tt++)
                        if (t0 + tt < NT) acc[t0 + tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[tt].z, ag.z, acc[t0 + tt], 0, 0, 0);
#pragma unroll
                    for (int tt = 0; tt < 4; tt++)
                        if (t0 + tt < NT) acc[t0 + tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[tt].w, ag.w, acc[t0 + tt], 0, 0, 0);
                }
                a[g] = ws_load_group<VEC, KGM>(rx, voff, kq, K, g);
            }
            __builtin_amdgcn_sched_barrier(0);      // keep the groups in order: hoisting every B read blows the VGPR budget
        }
        // accumulators (weights as the A operand): lane (fr, kq) holds Y[row fr][16 t + 4 kq .. + 3]; rows past M fall outside
        // the descriptor and are dropped by the range check, padded columns are sent out of range explicitly
        const __amdgpu_buffer_rsrc_t ry = ws_block_rsrc(Y, rb, RB, M, N);
        if constexpr (!EPI) {
#pragma unroll
            for (int t = 0; t < NT; t++) {
                const float4 bv = *reinterpret_cast<const float4 *>(sbias + c00 + 16 * t);
                float v[4] = {acc[t][0] + bv.x, acc[t][1] + bv.y, acc[t][2] + bv.z, acc[t][3] + bv.w};
                if (relu) {
#pragma unroll
                    for (int i = 0; i < 4; i++) v[i] = fmaxf(v[i], 0.f);
                }
                ws_store4(ry, yoff + 64 * t, c00 + 16 * t, N, sv, v);
            }
        } else {
            switch (ep.mode) {
                case GSVC_LIN_GELU_DUAL: ws_epilogue<NT, GSVC_LIN_GELU_DUAL>(acc, sbias, ry, yoff, c00, sv, rb, RB, M, N, ep); break;
                case GSVC_LIN_TANH: ws_epilogue<NT, GSVC_LIN_TANH>(acc, sbias, ry, yoff, c00, sv, rb, RB, M, N, ep); break;
                case GSVC_LIN_SIGMOID: ws_epilogue<NT, GSVC_LIN_SIGMOID>(acc, sbias, ry, yoff, c00, sv, rb, RB, M, N, ep); break;
                case GSVC_LIN_MUL_GELU_GRAD: ws_epilogue<NT, GSVC_LIN_MUL_GELU_GRAD>(acc, sbias, ry, yoff, c00, sv, rb, RB, M, N, ep); break;
                case GSVC_LIN_MUL_RELU_MASK: ws_epilogue<NT, GSVC_LIN_MUL_RELU_MASK>(acc, sbias, ry, yoff, c00, sv, rb, RB, M, N, ep); break;
                case GSVC_LIN_FILM: ws_epilogue<NT, GSVC_LIN_FILM>(acc, sbias, ry, yoff, c00, sv, rb, RB, M, N, ep); break;
                case GSVC_LIN_FILM_GRAD: ws_epilogue<NT, GSVC_LIN_FILM_GRAD>(acc, sbias, ry, yoff, c00, sv, rb, RB, M, N, ep); break;
                default: ws_epilogue<NT, GSVC_LIN_ADD>(acc, sbias, ry, yoff, c00, sv, rb, RB, M, N, ep); break;
            }
        }
    }
}

template <int NT, int KGM, int VEC, int THREADS, bool EPI>
static void launch_ws4(const float *X, const float *W, const float *b, float *Y, long long M, int K, int N, int w_in_out,
                       int relu, hipStream_t s, const LinEpi &ep)
{
    const size_t lds = (size_t)NT * 16 * (ws_ld(K) + 1) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_linear_ws<NT, KGM, VEC, THREADS, EPI>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    constexpr int WAVES = THREADS / 64;
    const long long RB = (M + 15) / 16, want = (RB + WAVES - 1) / WAVES;
    const unsigned grid = (unsigned)(want < 256 ? want : 256);      // persistent: one workgroup per CU
    ProfScope _prof("k_linear_ws", s);
    // widest epilogue access: every [M][N] matrix it touches must keep 16- / 8-byte alignment from row to row
    const uintptr_t al = reinterpret_cast<uintptr_t>(Y) | reinterpret_cast<uintptr_t>(ep.aux1) | reinterpret_cast<uintptr_t>(ep.aux2) |
                         reinterpret_cast<uintptr_t>(ep.y2) | reinterpret_cast<uintptr_t>(ep.y3);
    const int sv = (N % 4 == 0 && (al & 15) == 0) ? 4 : ((N % 2 == 0 && (al & 7) == 0) ? 2 : 1);
    hipLaunchKernelGGL((k_linear_ws<NT, KGM, VEC, THREADS, EPI>), dim3(grid), dim3(THREADS), lds, s, X, W, b, Y, M, K, N,
                       w_in_out, relu, sv, ep);
}

template <int NT, int KGM, int VEC, bool EPI>
static void launch_ws3(const float *X, const float *W, const float *b, float *Y, long long M, int K, int N, int w_in_out,
                       int relu, hipStream_t s, const LinEpi &ep)
{
    // 1024 threads leave 128 VGPRs per lane: accumulators + A fragments + addressing must fit without spilling
    // (the EPI variants hold 16 auxiliary operands and four more descriptors through the epilogue)
    constexpr int budget = (VEC == 4 ? 60 : (VEC == 2 ? 52 : 40)) - (NT > 8 ? 8 : 0) - (EPI ? GSVC_LIN_EPI_REGS : 0);
    if constexpr (NT * 4 + KGM * 4 <= budget) launch_ws4<NT, KGM, VEC, 1024, EPI>(X, W, b, Y, M, K, N, w_in_out, relu, s, ep);
    else launch_ws4<NT, KGM, VEC, 512, EPI>(X, W, b, Y, M, K, N, w_in_out, relu, s, ep);
}

template <int NT, int KGM, bool EPI>
static void launch_ws2(const float *X, const float *W, const float *b, float *Y, long long M, int K, int N, int w_in_out,
                       int relu, hipStream_t s, const LinEpi &ep)
{
    const bool a16 = (K % 4 == 0) && ((reinterpret_cast<uintptr_t>(X) & 15) == 0);
    const bool a8 = (K % 2 == 0) && ((reinterpret_cast<uintptr_t>(X) & 7) == 0);
    if (a16) launch_ws3<NT, KGM, 4, EPI>(X, W, b, Y, M, K, N, w_in_out, relu, s, ep);
    else if (a8) launch_ws3<NT, KGM, 2, EPI>(X, W, b, Y, M, K, N, w_in_out, relu, s, ep);
    else launch_ws3<NT, KGM, 1, EPI>(X, W, b, Y, M, K, N, w_in_out, relu, s, ep);
}

template <int NT, bool EPI>
static void launch_ws(const float *X, const float *W, const float *b, float *Y, long long M, int K, int N, int w_in_out,
                      int relu, hipStream_t s, const LinEpi &ep)
{
    const int kg = (K + 15) / 16;
    if (kg <= 2) launch_ws2<NT, 2, EPI>(X, W, b, Y, M, K, N, w_in_out, relu, s, ep);
    else if (kg <= 4) launch_ws2<NT, 4, EPI>(X, W, b, Y, M, K, N, w_in_out, relu, s, ep);
    else if (kg <= 6) launch_ws2<NT, 6, EPI>(X, W, b, Y, M, K, N, w_in_out, relu, s, ep);
    else if (kg <= 8) launch_ws2<NT, 8, EPI>(X, W, b, Y, M, K, N, w_in_out, relu, s, ep);
    else if (kg <= 10) launch_ws2<NT, 10, EPI>(X, W, b, Y, M, K, N, w_in_out, relu, s, ep);
    else launch_ws2<NT, 12, EPI>(X, W, b, Y, M, K, N, w_in_out, relu, s, ep);
}

template <bool EPI, int NT_LO = 1, int NT_HI = 12>
static void launch_ws_n(const float *X, const float *W, const float *b, float *Y, long long M, int K, int N, int w_in_out,
                        int relu, hipStream_t s, const LinEpi &ep)
{
    switch ((N + 15) / 16) {
        case 1: if constexpr (NT_LO <= 1 && 1 <= NT_HI) launch_ws<1, EPI>(X, W, b, Y, M, K, N, w_in_out, relu, s, ep); break;
        case 2: if constexpr (NT_LO <= 2 && 2 <= NT_HI) launch_ws<2, EPI>(X, W, b, Y, M, K, N, w_in_out, relu, s, ep); break;
        case 3: if constexpr (NT_LO <= 3 && 3 <= NT_HI) launch_ws<3, EPI>(X, W, b, Y, M, K, N, w_in_out, relu, s, ep); break;
        case 4: if constexpr (NT_LO <= 4 && 4 <= NT_HI) launch_ws<4, EPI>(X, W, b, Y, M, K, N, w_in_out, relu, s, ep); break;
        case 5: if constexpr (NT_LO <= 5 && 5 <= NT_HI) launch_ws<5, EPI>(X, W, b, Y, M, K, N, w_in_out, relu, s, ep); break;
        case 6: if constexpr (NT_LO <= 6 && 6 <= NT_HI) launch_ws<6, EPI>(X, W, b, Y, M, K, N, w_in_out, relu, s, ep); break;
        case 7: if constexpr (NT_LO <= 7 && 7 <= NT_HI) launch_ws<7, EPI>(X, W, b, Y, M, K, N, w_in_out, relu, s, ep); break;
        case 8: if constexpr (NT_LO <= 8 && 8 <= NT_HI) launch_ws<8, EPI>(X, W, b, Y, M, K, N, w_in_out, relu, s, ep); break;
        case 9: if constexpr (NT_LO <= 9 && 9 <= NT_HI) launch_ws<9, EPI>(X, W, b, Y, M, K, N, w_in_out, relu, s, ep); break;
        case 10: if constexpr (NT_LO <= 10 && 10 <= NT_HI) launch_ws<10, EPI>(X, W, b, Y, M, K, N, w_in_out, relu, s, ep); break;
        case 11: if constexpr (NT_LO <= 11 && 11 <= NT_HI) launch_ws<11, EPI>(X, W, b, Y, M, K, N, w_in_out, relu, s, ep); break;
        default: if constexpr (NT_LO <= 12 && 12 <= NT_HI) launch_ws<12, EPI>(X, W, b, Y, M, K, N, w_in_out, relu, s, ep); break;
    }
}


}  // namespace gsvc

// gsvc_amd/csrc/linear.hip — tall-skinny fp32 linear layer Y[M,N] = X[M,K] W[N,K]^T + b on MFMA, gfx950.
//
// The generator / deformation / entropy-parameter MLPs of GSVC (reference scene/gaussian_model.py:150-232,
// 411-501) are chains of nn.Linear with K, N <= 192 applied to ~50k-200k anchor rows per step.  Those GEMMs are
// tall and skinny (arithmetic intensity 12-48 FLOP/B): between the HBM and the fp32-MFMA roofs, and the library
// GEMM reached only 7-30 TF on them.  v_mfma_f32_16x16x4_f32: exact fp32 products, fp32 accumulate — same
// numerics class as the reference's fp32 GEMM, different summation order.
#include "common.h"

namespace gsvc {

typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int LIN_NT_MAX = 12;      // K, N <= 192

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

// NT = ceil(N/16) column tiles, KGM = compile-time bound on ceil(K/16), VEC = widest aligned load of an X row.
// THREADS = 1024 (4 waves per SIMD, 128 VGPRs) for the small shapes, 512 (2 waves per SIMD, 256 VGPRs) for the
// large ones.  Software pipeline without extra registers: as soon as the MFMAs of k-group g have consumed a[g],
// a[g] is reloaded with the NEXT row block's fragment, so the X stream overlaps the rest of the multiply and the
// store of the current block.
// w_in_out != 0: W is given as [K][N] (input-major), i.e. Y = X W — the dX = G W product of the backward pass
// without a transposed copy of W.
template <int NT, int KGM, int VEC, int THREADS>
__global__ void __launch_bounds__(THREADS) k_linear_ws(const float *__restrict__ X, const float *__restrict__ W,
                                                       const float *__restrict__ bias, float *__restrict__ Y,
                                                       long long M, int K, int N, int w_in_out, int relu)
{
    extern __shared__ float lds[];
    constexpr int WAVES = THREADS / 64;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, kq = lane >> 4;
    const int ld = ws_ld(K), KG = (K + 15) >> 4, k16 = KG * 16;
    const long long RB = (M + 15) >> 4, stride = (long long)gridDim.x * WAVES;
    long long rb = (long long)blockIdx.x * WAVES + wave;

    // first row block's fragments fly while the weights are staged
    float4 a[KGM];
    const int voff = fr * K * 4 + 16 * kq;
    {
        const __amdgpu_buffer_rsrc_t rx = ws_block_rsrc(X, rb, RB, M, K);
#pragma unroll
        for (int g = 0; g < KGM; g++) a[g] = ws_load_group<VEC, KGM>(rx, voff, kq, K, g);
    }
    {
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
    int yoff[4];
#pragma unroll
    for (int i = 0; i < 4; i++) yoff[i] = ((4 * kq + i) * N + fr) * 4;
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
                        if (t0 + tt < NT) acc[t0 + tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ag.x, b[tt].x, acc[t0 + tt], 0, 0, 0);
#pragma unroll
                    for (int tt = 0; tt < 4; tt++)
                        if (t0 + tt < NT) acc[t0 + tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ag.y, b[tt].y, acc[t0 + tt], 0, 0, 0);
#pragma unroll
                    for (int tt = 0; tt < 4; tt++)
                        if (t0 + tt < NT) acc[t0 + tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ag.z, b[tt].z, acc[t0 + tt], 0, 0, 0);
#pragma unroll
                    for (int tt = 0; tt < 4; tt++)
                        if (t0 + tt < NT) acc[t0 + tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ag.w, b[tt].w, acc[t0 + tt], 0, 0, 0);
                }
                a[g] = ws_load_group<VEC, KGM>(rx, voff, kq, K, g);
            }
            __builtin_amdgcn_sched_barrier(0);      // keep the groups in order: hoisting every B read blows the VGPR budget
        }
        // D fragment: lane holds rows 4*(lane/16)+i (i=0..3) of column lane%16; rows past M fall outside the
        // descriptor and are dropped by the range check, padded columns are sent out of range explicitly
        const __amdgpu_buffer_rsrc_t ry = ws_block_rsrc(Y, rb, RB, M, N);
#pragma unroll
        for (int t = 0; t < NT; t++) {
            const bool in = t < NT - 1 || t * 16 + fr < N;
            const float bvt = sbias[t * 16 + fr];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                float v = acc[t][i] + bvt;
                if (relu) v = fmaxf(v, 0.f);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), ry, in ? yoff[i] + 64 * t : BUF_OOB, 0, 0);
            }
        }
    }
}

template <int NT, int KGM, int VEC, int THREADS>
static void launch_ws4(const float *X, const float *W, const float *b, float *Y, long long M, int K, int N, int w_in_out,
                       int relu, hipStream_t s)
{
    const size_t lds = (size_t)NT * 16 * (ws_ld(K) + 1) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_linear_ws<NT, KGM, VEC, THREADS>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    constexpr int WAVES = THREADS / 64;
    const long long RB = (M + 15) / 16, want = (RB + WAVES - 1) / WAVES;
    const unsigned grid = (unsigned)(want < 256 ? want : 256);      // persistent: one workgroup per CU
    ProfScope _prof("k_linear_ws", s);
    hipLaunchKernelGGL((k_linear_ws<NT, KGM, VEC, THREADS>), dim3(grid), dim3(THREADS), lds, s, X, W, b, Y, M, K, N,
                       w_in_out, relu);
}

template <int NT, int KGM, int VEC>
static void launch_ws3(const float *X, const float *W, const float *b, float *Y, long long M, int K, int N, int w_in_out,
                       int relu, hipStream_t s)
{
    // 1024 threads leave 128 VGPRs per lane: accumulators + A fragments + addressing must fit without spilling
    constexpr int budget = (VEC == 4 ? 60 : (VEC == 2 ? 52 : 40)) - (NT > 8 ? 8 : 0);
    if constexpr (NT * 4 + KGM * 4 <= budget) launch_ws4<NT, KGM, VEC, 1024>(X, W, b, Y, M, K, N, w_in_out, relu, s);
    else launch_ws4<NT, KGM, VEC, 512>(X, W, b, Y, M, K, N, w_in_out, relu, s);
}

template <int NT, int KGM>
static void launch_ws2(const float *X, const float *W, const float *b, float *Y, long long M, int K, int N, int w_in_out,
                       int relu, hipStream_t s)
{
    const bool a16 = (K % 4 == 0) && ((reinterpret_cast<uintptr_t>(X) & 15) == 0);
    const bool a8 = (K % 2 == 0) && ((reinterpret_cast<uintptr_t>(X) & 7) == 0);
    if (a16) launch_ws3<NT, KGM, 4>(X, W, b, Y, M, K, N, w_in_out, relu, s);
    else if (a8) launch_ws3<NT, KGM, 2>(X, W, b, Y, M, K, N, w_in_out, relu, s);
    else launch_ws3<NT, KGM, 1>(X, W, b, Y, M, K, N, w_in_out, relu, s);
}

template <int NT>
static void launch_ws(const float *X, const float *W, const float *b, float *Y, long long M, int K, int N, int w_in_out,
                      int relu, hipStream_t s)
{
    const int kg = (K + 15) / 16;
    if (kg <= 2) launch_ws2<NT, 2>(X, W, b, Y, M, K, N, w_in_out, relu, s);
    else if (kg <= 4) launch_ws2<NT, 4>(X, W, b, Y, M, K, N, w_in_out, relu, s);
    else if (kg <= 6) launch_ws2<NT, 6>(X, W, b, Y, M, K, N, w_in_out, relu, s);
    else if (kg <= 8) launch_ws2<NT, 8>(X, W, b, Y, M, K, N, w_in_out, relu, s);
    else if (kg <= 10) launch_ws2<NT, 10>(X, W, b, Y, M, K, N, w_in_out, relu, s);
    else launch_ws2<NT, 12>(X, W, b, Y, M, K, N, w_in_out, relu, s);
}

// ---------------------------------------------------------------------------------------------------------
// Weight gradient dW[N,K] = G[M,N]^T X[M,K] (+ db[N] = column sums of G): a reduction over the ~180k anchor rows
// with a tiny output.  A library GEMM tiles the OUTPUT (a dozen workgroups on 256 CUs); here the rows are split:
// one persistent workgroup per CU, one wave per (64 x 64 output block, row split).  Both operands stream from HBM
// exactly once, straight into MFMA fragments, no LDS staging:
//   * an MFMA step reduces 4 rows; lane (j, mq) reads row m0+mq.  A = G^T, so lane j supplies output row n and
//     its natural load is VEC consecutive columns n..n+VEC-1 of one G row: component i of that vector is the A
//     operand of a *strided* tile (rows {16 VEC p + VEC j + i}), so one 16-byte load feeds 4 tiles; same for X/B.
//     A wave-instruction reads 4 rows x 256 contiguous bytes.
//   * 16 MFMAs per 8 operand registers; the operand registers of step s are reloaded with the next 16-row chunk
//     right after its MFMAs (register-neutral prefetch, as in k_linear_ws).
//   * the waves of a block's row splits are summed in LDS, then the workgroup adds its 64 x 64 blocks to dW with
//     contiguous 256-byte atomic segments (256 workgroups x N x K floats in total).
constexpr int WG_MAX_WAVES = 12;

template <int VEC>
__device__ __forceinline__ void wg_load(__amdgpu_buffer_rsrc_t rs, int off, bool valid, float (&dst)[4], int p)
{
    // piece p of a 64-column block: VEC floats at byte offset off (+ the piece's immediate), zeros when !valid
    const int o = valid ? off : BUF_OOB;
    if (VEC == 4) {
        const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rs, o, 0, 0);
        dst[0] = __uint_as_float(t.x); dst[1] = __uint_as_float(t.y); dst[2] = __uint_as_float(t.z); dst[3] = __uint_as_float(t.w);
    } else if (VEC == 2) {
        const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(rs, o, 0, 0);
        dst[2 * p] = __uint_as_float(t.x); dst[2 * p + 1] = __uint_as_float(t.y);
    } else {
        dst[p] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, o, 0, 0));
    }
}

// operands of one 4-row step for a 64-column block starting at column c0 of a [rows][ld] matrix
template <int VEC>
__device__ __forceinline__ void wg_load_step(__amdgpu_buffer_rsrc_t rs, int row_off, int c0, int j, int ld, float (&dst)[4])
{
#pragma unroll
    for (int p = 0; p < 4 / VEC; p++) {
        const int col = c0 + 16 * VEC * p + VEC * j;
        wg_load<VEC>(rs, row_off + col * 4, col < ld, dst, p);
    }
}

template <int VG, int VX>
__global__ void __launch_bounds__(64 * WG_MAX_WAVES) k_linear_wgrad(const float *__restrict__ G, const float *__restrict__ X,
                                                                   float *__restrict__ part, int want_db,
                                                                   long long M, int N, int K, int BN, int BK, int RS)
{
    extern __shared__ float sm[];      // [BN*BK][64][64]
    __shared__ float sdb[LIN_NT_MAX * 16];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pairs = BN * BK;
    const int pair = wave % pairs, rs = wave / pairs;
    const int bn = pair / BK, bk = pair - bn * BK;
    const int j = lane & 15, mq = lane >> 4;
    const long long RB = (M + 15) >> 4;
    const long long workers = (long long)gridDim.x * RS;
    long long rb = (long long)blockIdx.x * RS + rs;
    if (tid < LIN_NT_MAX * 16) sdb[tid] = 0.f;
    __syncthreads();

    v4f acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) acc[a][b] = (v4f){0.f, 0.f, 0.f, 0.f};
    float gsum[4] = {0.f, 0.f, 0.f, 0.f};
    float fa[4][4], fb[4][4];          // [step][tile]
    {
        const __amdgpu_buffer_rsrc_t rg = ws_block_rsrc(G, rb, RB, M, N), rx = ws_block_rsrc(X, rb, RB, M, K);
#pragma unroll
        for (int s = 0; s < 4; s++) {
            wg_load_step<VG>(rg, (4 * s + mq) * N * 4, 64 * bn, j, N, fa[s]);
            wg_load_step<VX>(rx, (4 * s + mq) * K * 4, 64 * bk, j, K, fb[s]);
        }
    }
    // the first chunk has landed before the loop: inside it only the loop's own loads are outstanding, so the wait
    // in front of step s is vmcnt(6) (the three later steps' reloads stay in flight), not vmcnt(0)
#pragma unroll
    for (int s = 0; s < 4; s++)
#pragma unroll
        for (int t = 0; t < 4; t++) asm volatile("" : "+v"(fa[s][t]), "+v"(fb[s][t]));
    for (; rb < RB; rb += workers) {
        const __amdgpu_buffer_rsrc_t rg = ws_block_rsrc(G, rb + workers, RB, M, N);      // empty past the end
        const __amdgpu_buffer_rsrc_t rx = ws_block_rsrc(X, rb + workers, RB, M, K);
#pragma unroll
        for (int s = 0; s < 4; s++) {
#pragma unroll
            for (int tn = 0; tn < 4; tn++) {
                gsum[tn] += fa[s][tn];
#pragma unroll
                for (int tk = 0; tk < 4; tk++)
                    acc[tn][tk] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[s][tn], fb[s][tk], acc[tn][tk], 0, 0, 0);
            }
            wg_load_step<VG>(rg, (4 * s + mq) * N * 4, 64 * bn, j, N, fa[s]);
            wg_load_step<VX>(rx, (4 * s + mq) * K * 4, 64 * bk, j, K, fb[s]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // bias gradient: column sums of G (waves of the first k-block; the row splits meet in LDS)
    if (want_db && bk == 0) {
#pragma unroll
        for (int tn = 0; tn < 4; tn++) {
            float v = gsum[tn];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            const int n = 64 * bn + 16 * VG * (tn / VG) + VG * j + (tn % VG);
            if (mq == 0 && n < N) atomicAdd(sdb + n, v);
        }
    }
    // sum the row splits of each 64 x 64 block in LDS (one wave per block and round), then write this workgroup's
    // partial dW (and db) to its slot of `part`; k_linear_wgrad_reduce adds the slots (no global atomics: 256
    // workgroups adding into the same 40 KB serialise on a handful of memory channels, measured +80 us)
    float *blk = sm + pair * 4096;
    for (int round = 0; round < RS; round++) {
        if (rs == round) {
#pragma unroll
            for (int tn = 0; tn < 4; tn++)
#pragma unroll
                for (int tk = 0; tk < 4; tk++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int nl = 16 * VG * (tn / VG) + VG * (4 * mq + r) + (tn % VG);
                        const int kl = 16 * VX * (tk / VX) + VX * j + (tk % VX);
                        float *d = blk + nl * 64 + kl;
                        *d = (round == 0) ? acc[tn][tk][r] : *d + acc[tn][tk][r];
                    }
        }
        __syncthreads();
    }
    float *out = part + (size_t)blockIdx.x * ((size_t)N * K + (want_db ? N : 0));
    for (int i = tid; i < pairs * 4096; i += blockDim.x) {
        const int pr = i >> 12, nl = (i >> 6) & 63, kl = i & 63;
        const int n = 64 * (pr / BK) + nl, k = 64 * (pr % BK) + kl;
        if (n < N && k < K) out[(size_t)n * K + k] = sm[i];
    }
    if (want_db)
        for (int n = tid; n < N; n += blockDim.x) out[(size_t)N * K + n] = sdb[n];
}

// dst[i] = sum over the workgroup slots of part[slot][i]: 64 outputs per workgroup, the slots dealt to its 4 waves
__global__ void __launch_bounds__(256) k_linear_wgrad_reduce(const float *__restrict__ part, int slots, int n, float *__restrict__ dW,
                                                            int nk, float *__restrict__ db)
{
    __shared__ float red[4][64];
    const int li = threadIdx.x & 63, sg = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + li;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (i < n) {
        int s = sg;
        for (; s + 12 < slots; s += 16) {
            a0 += part[(size_t)s * n + i];
            a1 += part[(size_t)(s + 4) * n + i];
            a2 += part[(size_t)(s + 8) * n + i];
            a3 += part[(size_t)(s + 12) * n + i];
        }
        for (; s < slots; s += 4) a0 += part[(size_t)s * n + i];
    }
    red[sg][li] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (sg == 0 && i < n) {
        const float v = (red[0][li] + red[1][li]) + (red[2][li] + red[3][li]);
        if (i < nk) dW[i] = v;
        else db[i - nk] = v;
    }
}

template <int VG, int VX>
static void launch_wgrad2(const float *G, const float *X, float *dW, float *db, float *part, int slots, long long M, int N,
                          int K, hipStream_t s)
{
    const int BN = (N + 63) / 64, BK = (K + 63) / 64, pairs = BN * BK;
    const int RS = pairs >= WG_MAX_WAVES ? 1 : WG_MAX_WAVES / pairs;
    const size_t lds = (size_t)pairs * 4096 * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_linear_wgrad<VG, VX>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        attr_set = true;
    }
    const long long RB = (M + 15) / 16, want = (RB + RS - 1) / RS;
    const int grid = (int)(want < slots ? want : slots);
    {
        ProfScope _prof("k_linear_wgrad", s);
        hipLaunchKernelGGL((k_linear_wgrad<VG, VX>), dim3(grid), dim3(64 * pairs * RS), lds, s, G, X, part, db ? 1 : 0, M, N, K,
                           BN, BK, RS);
    }
    const int n = N * K + (db ? N : 0);
    ProfScope _prof("k_linear_wgrad_reduce", s);
    hipLaunchKernelGGL(k_linear_wgrad_reduce, dim3((n + 63) / 64), dim3(256), 0, s, part, grid, n, dW, N * K, db);
}

static int vec_of(const float *p, int ld)
{
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    if (ld % 4 == 0 && (a & 15) == 0) return 4;
    if (ld % 2 == 0 && (a & 7) == 0) return 2;
    return 1;
}

}  // namespace gsvc

using namespace gsvc;

extern "C" int gsvc_linear_forward(const float *X, const float *W, const float *bias, float *Y, int64_t M, int32_t K,
                                      int32_t N, int32_t w_in_out, int32_t relu, void *stream)
{
    GSVC_REQUIRE(M >= 0 && K > 0 && N > 0, "linear_forward: bad shape");
    if (N > LIN_NT_MAX * 16 || K > LIN_NT_MAX * 16) {
        set_error("linear_forward: N=%d / K=%d exceed %d", N, K, LIN_NT_MAX * 16);
        return GSVC_E_UNSUPPORTED;
    }
    if (M == 0) return GSVC_OK;
    GSVC_REQUIRE(X && W && Y, "linear_forward: NULL pointer");
    hipStream_t s = (hipStream_t)stream;
    switch ((N + 15) / 16) {
        case 1: launch_ws<1>(X, W, bias, Y, M, K, N, w_in_out, relu, s); break;
        case 2: launch_ws<2>(X, W, bias, Y, M, K, N, w_in_out, relu, s); break;
        case 3: launch_ws<3>(X, W, bias, Y, M, K, N, w_in_out, relu, s); break;
        case 4: launch_ws<4>(X, W, bias, Y, M, K, N, w_in_out, relu, s); break;
        case 5: launch_ws<5>(X, W, bias, Y, M, K, N, w_in_out, relu, s); break;
        case 6: launch_ws<6>(X, W, bias, Y, M, K, N, w_in_out, relu, s); break;
        case 7: launch_ws<7>(X, W, bias, Y, M, K, N, w_in_out, relu, s); break;
        case 8: launch_ws<8>(X, W, bias, Y, M, K, N, w_in_out, relu, s); break;
        case 9: launch_ws<9>(X, W, bias, Y, M, K, N, w_in_out, relu, s); break;
        case 10: launch_ws<10>(X, W, bias, Y, M, K, N, w_in_out, relu, s); break;
        case 11: launch_ws<11>(X, W, bias, Y, M, K, N, w_in_out, relu, s); break;
        default: launch_ws<12>(X, W, bias, Y, M, K, N, w_in_out, relu, s); break;
    }
    return check_launch("linear_forward");
}

extern "C" int64_t gsvc_linear_wgrad_workspace(int32_t N, int32_t K)
{
    return (int64_t)256 * ((int64_t)N * K + N);      // one slot per workgroup (one per CU), floats
}

extern "C" int gsvc_linear_wgrad(const float *G, const float *X, float *dW, float *db, int64_t M, int32_t N, int32_t K,
                                 float *workspace, int64_t workspace_floats, void *stream)
{
    GSVC_REQUIRE(M >= 0 && K > 0 && N > 0, "linear_wgrad: bad shape");
    if (N > LIN_NT_MAX * 16 || K > LIN_NT_MAX * 16) {
        set_error("linear_wgrad: N=%d / K=%d exceed %d", N, K, LIN_NT_MAX * 16);
        return GSVC_E_UNSUPPORTED;
    }
    GSVC_REQUIRE(G && X && dW && workspace, "linear_wgrad: NULL pointer");
    const int64_t per_slot = (int64_t)N * K + (db ? N : 0);
    int64_t slots = workspace_floats / per_slot;
    GSVC_REQUIRE(slots >= 1, "linear_wgrad: workspace smaller than one slot (N*K + N floats)");
    if (slots > 256) slots = 256;
    hipStream_t s = (hipStream_t)stream;
    if (M == 0) {
        (void)hipMemsetAsync(dW, 0, sizeof(float) * (size_t)N * K, s);
        if (db) (void)hipMemsetAsync(db, 0, sizeof(float) * (size_t)N, s);
        return GSVC_OK;
    }
    const int vg = vec_of(G, N), vx = vec_of(X, K);
#define WG_CASE(a, b) if (vg == a && vx == b) launch_wgrad2<a, b>(G, X, dW, db, workspace, (int)slots, M, N, K, s)
    WG_CASE(4, 4); WG_CASE(4, 2); WG_CASE(4, 1); WG_CASE(2, 4); WG_CASE(2, 2); WG_CASE(2, 1); WG_CASE(1, 4); WG_CASE(1, 2); WG_CASE(1, 1);
#undef WG_CASE
    return check_launch("linear_wgrad");
}

// gsvc_amd/csrc/linear_wgrad.hip — weight / bias gradients of the MLP layers (dW = G^T X over ~200k rows) on MFMA, gfx950.
#include "linear_ws.h"

namespace gsvc {

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
    __shared__ float sdb[WG_MAX_WAVES][LIN_NT_MAX * 16];      // per row split: summed in a fixed order below (no float atomics)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pairs = BN * BK;
    const int pair = wave % pairs, rs = wave / pairs;
    const int bn = pair / BK, bk = pair - bn * BK;
    const int j = lane & 15, mq = lane >> 4;
    const long long RB = (M + 15) >> 4;
    const long long workers = (long long)gridDim.x * RS;
    long long rb = (long long)rs * gridDim.x + blockIdx.x;          // chunks dealt row-split-major (see k_linear_ws)
    for (int i = tid; i < WG_MAX_WAVES * LIN_NT_MAX * 16; i += blockDim.x) (&sdb[0][0])[i] = 0.f;
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
    // lockstep over the chunks (see wg_main_t): the waves that read the same rows ask for them together
    const long long n_it = (RB - (long long)blockIdx.x + workers - 1) / workers;
    for (long long it = 0; it < n_it; it++, rb += workers) {
        __builtin_amdgcn_s_barrier();
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
            if (mq == 0 && n < N) sdb[rs][n] = v;      // one wave per (row split, column)
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
        for (int n = tid; n < N; n += blockDim.x) {
            float v = sdb[0][n];
            for (int r = 1; r < RS; r++) v += sdb[r][n];
            out[(size_t)N * K + n] = v;
        }
}

// ---------------------------------------------------------------------------------------------------------
// The same product with 16-COLUMN granularity (round 2).  k_linear_wgrad above gives every wave a 64 x 64 output block, so
// N = K = 66 is computed as 128 x 128 (27 % useful MFMA work), 100 x 100 as 128 x 128 (61 %).  Here the ceil(N / 16) column
// tiles of G are dealt to ceil(tiles / 4) blocks as evenly as possible (5 tiles -> 3 + 2, 7 -> 4 + 3), same for X: a wave
// owns a (16 TN) x (16 TK) block with TN, TK in 1..4 and runs the loop instantiated for exactly that pair — chosen by ONE
// wave-uniform switch outside the loop (the attempt with per-tile branches around loads and MFMAs inside the loop lost
// the compiler's exact vmcnt accounting and was slower than the padding it saved).  A lane loads TN (TK) consecutive columns
// of its row with one 4 / 8 / 12 / 16-byte buffer load; component i is the operand of strided tile i (columns
// {TN j + i}).  Columns past N inside a row's last vector are the next row's values (past the block: zeros, the range check
// is per dword): they only reach output rows / columns >= N / K, which are never stored.
template <int T>
__device__ __forceinline__ void wg_load_t(__amdgpu_buffer_rsrc_t rs, int off, float (&dst)[4])
{
    if (T == 4) {
        const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
        dst[0] = __uint_as_float(t.x); dst[1] = __uint_as_float(t.y); dst[2] = __uint_as_float(t.z); dst[3] = __uint_as_float(t.w);
    } else if (T == 3) {
        typedef unsigned int u32x3 __attribute__((ext_vector_type(3)));
        const u32x3 t = __builtin_amdgcn_raw_buffer_load_b96(rs, off, 0, 0);
        dst[0] = __uint_as_float(t.x); dst[1] = __uint_as_float(t.y); dst[2] = __uint_as_float(t.z);
    } else if (T == 2) {
        const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(rs, off, 0, 0);
        dst[0] = __uint_as_float(t.x); dst[1] = __uint_as_float(t.y);
    } else {
        dst[0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, 0));
    }
}

template <int TN, int TK>
__device__ __forceinline__ void wg_main_t(const float *__restrict__ G, const float *__restrict__ X, long long M, int N, int K, int n0,
                                          int k0, long long rb, long long workers, int j, int mq, v4f (&acc)[4][4], float (&gsum)[4], long long n_it)
{
    const long long RB = (M + 15) >> 4;
    const int ca = n0 + TN * j, cb = k0 + TK * j;          // this lane's first column of G / X
#ifdef GSVC_WG_NO_DUP      // timing experiment: every operand block loaded by ONE wave of the workgroup (results wrong)
    const bool va = ca < N && k0 == 0, vb = cb < K && n0 == 0;
#else
    const bool va = ca < N, vb = cb < K;
#endif
    float fa[4][4], fb[4][4];          // [step][tile]
    {
        const __amdgpu_buffer_rsrc_t rg = ws_block_rsrc(G, rb, RB, M, N), rx = ws_block_rsrc(X, rb, RB, M, K);
#pragma unroll
        for (int s = 0; s < 4; s++) {
            wg_load_t<TN>(rg, va ? ((4 * s + mq) * N + ca) * 4 : BUF_OOB, fa[s]);
            wg_load_t<TK>(rx, vb ? ((4 * s + mq) * K + cb) * 4 : BUF_OOB, fb[s]);
        }
    }
#pragma unroll
    for (int s = 0; s < 4; s++) {
#pragma unroll
        for (int t = 0; t < TN; t++) asm volatile("" : "+v"(fa[s][t]));
#pragma unroll
        for (int t = 0; t < TK; t++) asm volatile("" : "+v"(fb[s][t]));
    }
    // The waves of a workgroup walk their chunks in LOCKSTEP (a barrier per 16-row chunk, the same trip count for all: a wave
    // past its last chunk multiplies the zeros of an empty descriptor).  The four waves of a row split read the same rows — the
    // same G block by two of them, the same X block by two, and the blocks of a row share cache lines — and left to
    // themselves they drift apart (16 vs 9 MFMAs per step), so that every line came from L2 two to four times: with the waves
    // asking together the launch's operand stream went from 2.7 to 3.6 TB/s (k_linear_wgrad_many 161 -> 123 us in the fitting step).
    for (long long it = 0; it < n_it; it++, rb += workers) {
        __builtin_amdgcn_s_barrier();
        const __amdgpu_buffer_rsrc_t rg = ws_block_rsrc(G, rb + workers, RB, M, N);      // empty past the end
        const __amdgpu_buffer_rsrc_t rx = ws_block_rsrc(X, rb + workers, RB, M, K);
#pragma unroll
        for (int s = 0; s < 4; s++) {
#pragma unroll
            for (int tn = 0; tn < TN; tn++) {
                gsum[tn] += fa[s][tn];
#pragma unroll
                for (int tk = 0; tk < TK; tk++) {
#ifdef GSVC_WG_NO_MFMA      // timing experiment (tools/scratch): the operand stream alone
                    acc[tn][tk][0] += fa[s][tn] + fb[s][tk];
#else
                    acc[tn][tk] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[s][tn], fb[s][tk], acc[tn][tk], 0, 0, 0);
#endif
                }
            }
#ifndef GSVC_WG_NO_LOAD     // timing experiment: the MFMAs alone (operands of the first chunk reused)
            wg_load_t<TN>(rg, va ? ((4 * s + mq) * N + ca) * 4 : BUF_OOB, fa[s]);
            wg_load_t<TK>(rx, vb ? ((4 * s + mq) * K + cb) * 4 : BUF_OOB, fb[s]);
#endif
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// tiles [first, first + count) of block b when `tiles` tiles are dealt to `blocks` blocks as evenly as possible
__device__ __forceinline__ void wg_block_tiles(int tiles, int blocks, int b, int &first, int &count)
{
    const int base = tiles / blocks, rem = tiles - base * blocks;
    first = b * base + min(b, rem);
    count = base + (b < rem ? 1 : 0);
}

// floats of one output block's LDS image in the 16-column kernel: 64 columns x 16 rows per tile of its widest n-block
__host__ __device__ inline int wgt_block_floats(int N, int BN) { return 64 * 16 * ((((N + 15) >> 4) + BN - 1) / BN); }

// wg of nwg: this workgroup's index among the workgroups that work on this product (a launch may carry several products:
// k_linear_wgrad_many below)
__device__ __forceinline__ void wgrad_t_body(const float *__restrict__ G, const float *__restrict__ X, float *__restrict__ part, int want_db,
                                             long long M, int N, int K, int BN, int BK, int RS, int wg, int nwg)
{
    extern __shared__ float sm[];      // [BN*BK][64][64]
    __shared__ float sdb[WG_MAX_WAVES][LIN_NT_MAX * 16];      // per row split: summed in a fixed order below (no float atomics)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pairs = BN * BK;
    // row split rs handles the pairs rotated by rs: the blocks differ in size (4 x 4 .. 1 x 1 tiles), and with pair = wave %
    // pairs every wave of one SIMD (wave % 4) would own the same block
    int rs = wave / pairs, pair = (wave + rs) % pairs;
    if (pairs == 4 && RS == 3) {
        // the common case (2 x 2 blocks, e.g. 4 x 4 / 4 x 3 / 3 x 4 / 3 x 3 tiles = 16 / 12 / 12 / 9 MFMAs per step): the twelve
        // (block, row split) pairs are dealt so that the three waves of every SIMD carry 37 / 37 / 37 / 36 of them (the rotation
        // above gives 40 / 33 / 37 / 37 and the kernel lasts as long as its busiest SIMD)
        pair = (int)((0xBF9940u >> (2 * wave)) & 3u);      // waves 0..11 -> 0 0 0 1  1 2 1 2  3 3 3 2
        rs = (int)((0xA450A4u >> (2 * wave)) & 3u);         // waves 0..11 -> 0 1 2 2  0 0 1 1  0 1 2 2
    }
    const int bn = pair / BK, bk = pair - bn * BK;
    const int j = lane & 15, mq = lane >> 4;
    int tn0, TN, tk0, TK;
    wg_block_tiles((N + 15) >> 4, BN, bn, tn0, TN);
    wg_block_tiles((K + 15) >> 4, BK, bk, tk0, TK);
    const int n0 = 16 * tn0, k0 = 16 * tk0;
    const long long workers = (long long)nwg * RS;
    const long long rb = (long long)rs * nwg + wg;      // chunks dealt row-split-major (see k_linear_ws)
    for (int i = tid; i < WG_MAX_WAVES * LIN_NT_MAX * 16; i += blockDim.x) (&sdb[0][0])[i] = 0.f;
    __syncthreads();

    v4f acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) acc[a][b] = (v4f){0.f, 0.f, 0.f, 0.f};
    float gsum[4] = {0.f, 0.f, 0.f, 0.f};
    const long long n_it = (((M + 15) >> 4) - wg + workers - 1) / workers;      // chunks of the workgroup's first row split = trip count of all its waves
#define WG_BODY(a, b) case (a) * 4 + (b): wg_main_t<a, b>(G, X, M, N, K, n0, k0, rb, workers, j, mq, acc, gsum, n_it); break
    switch (TN * 4 + TK) {
        WG_BODY(1, 1); WG_BODY(1, 2); WG_BODY(1, 3); WG_BODY(1, 4);
        WG_BODY(2, 1); WG_BODY(2, 2); WG_BODY(2, 3); WG_BODY(2, 4);
        WG_BODY(3, 1); WG_BODY(3, 2); WG_BODY(3, 3); WG_BODY(3, 4);
        WG_BODY(4, 1); WG_BODY(4, 2); WG_BODY(4, 3); WG_BODY(4, 4);
        default: break;
    }
#undef WG_BODY
    // bias gradient: column sums of G (waves of the first k-block; the row splits meet in LDS)
    if (want_db && bk == 0) {
#pragma unroll
        for (int tn = 0; tn < 4; tn++) {
            float v = gsum[tn];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            const int n = n0 + TN * j + tn;
            if (tn < TN && mq == 0 && n < N) sdb[rs][n] = v;      // one wave per (row split, column)
        }
    }
    // sum the row splits of each block in LDS (one wave per block and round), then this workgroup's partial dW (and db) goes
    // to its slot of `part`
    // a block's LDS image: 16 x (largest tile count of an n-block) rows of 64 floats — 12 blocks of 3 x 4 tiles (N = 150, K = 192)
    // are 144 KB where twelve 64 x 64 images would not fit
    const int PB = wgt_block_floats(N, BN);
    float *blk = sm + pair * PB;
    for (int round = 0; round < RS; round++) {
        if (rs == round) {
#pragma unroll
            for (int tn = 0; tn < 4; tn++)
#pragma unroll
                for (int tk = 0; tk < 4; tk++)
                    if (tn < TN && tk < TK) {
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const int nl = TN * (4 * mq + r) + tn, kl = TK * j + tk;
                            float *d = blk + nl * 64 + kl;
                            *d = (round == 0) ? acc[tn][tk][r] : *d + acc[tn][tk][r];
                        }
                    }
        }
        __syncthreads();
    }
    float *out = part + (size_t)wg * ((size_t)N * K + (want_db ? N : 0));
    // the blocks' tile ranges once per workgroup (two integer divisions per ELEMENT of the 64 x 64 blocks were ~1000 instructions
    // per thread of this epilogue)
    __shared__ int s_blk[WG_MAX_WAVES][4];      // first output row, rows, first output column, columns of block pr
    if (tid < pairs) {
        int f, c, g, e;
        wg_block_tiles((N + 15) >> 4, BN, tid / BK, f, c);
        wg_block_tiles((K + 15) >> 4, BK, tid % BK, g, e);
        s_blk[tid][0] = 16 * f; s_blk[tid][1] = 16 * c; s_blk[tid][2] = 16 * g; s_blk[tid][3] = 16 * e;
    }
    __syncthreads();
    for (int i = tid; i < pairs * PB; i += blockDim.x) {
        const int pr = i / PB, rem = i - pr * PB, nl = rem >> 6, kl = rem & 63;
        const int n = s_blk[pr][0] + nl, k = s_blk[pr][2] + kl;
        if (nl < s_blk[pr][1] && kl < s_blk[pr][3] && n < N && k < K) out[(size_t)n * K + k] = sm[i];
    }
    if (want_db)
        for (int n = tid; n < N; n += blockDim.x) {
            float v = sdb[0][n];
            for (int r = 1; r < RS; r++) v += sdb[r][n];
            out[(size_t)N * K + n] = v;
        }
}

__global__ void __launch_bounds__(64 * WG_MAX_WAVES) k_linear_wgrad_t(const float *__restrict__ G, const float *__restrict__ X,
                                                                     float *__restrict__ part, int want_db, long long M, int N, int K,
                                                                     int BN, int BK, int RS)
{
    wgrad_t_body(G, X, part, want_db, M, N, K, BN, BK, RS, (int)blockIdx.x, (int)gridDim.x);
}

// Several products in ONE launch (the weight gradients of a whole network: 6-7 layers over the same rows): the workgroups are
// dealt to the products in proportion to their MFMA work, each product's workgroups fill its own slots.  One prologue, one
// row-split meeting and one partial last round per network instead of one per layer (a launch of this kernel costs ~10-15 us
// beyond its MFMAs: at 40 launches per fitting step that was a quarter of the weight-gradient time).  Every product of a batch
// runs 12 waves (output blocks x row splits = 12).
constexpr int WG_MANY = 12;
struct WgManyJobs {
    int n;
    const float *G[WG_MANY], *X[WG_MANY];
    float *part[WG_MANY];
    long long M[WG_MANY];
    int want_db[WG_MANY], N[WG_MANY], K[WG_MANY], BN[WG_MANY], BK[WG_MANY], RS[WG_MANY], first[WG_MANY + 1];
};
template <typename T>
__device__ __forceinline__ T wg_pick(const T (&a)[WG_MANY], int i)
{
    T v = a[0];
#pragma unroll
    for (int q = 1; q < WG_MANY; q++) v = i == q ? a[q] : v;
    return v;
}

__global__ void __launch_bounds__(64 * WG_MAX_WAVES) k_linear_wgrad_many(WgManyJobs t)
{
    int j = 0;
#pragma unroll
    for (int q = 1; q < WG_MANY; q++)
        if (q < t.n && (int)blockIdx.x >= t.first[q]) j = q;
    int f0 = t.first[0];
#pragma unroll
    for (int q = 1; q < WG_MANY; q++) f0 = j == q ? t.first[q] : f0;
    int f1 = t.first[WG_MANY];
#pragma unroll
    for (int q = WG_MANY - 1; q >= 1; q--) f1 = j == q - 1 ? t.first[q] : f1;
    wgrad_t_body(wg_pick(t.G, j), wg_pick(t.X, j), wg_pick(t.part, j), wg_pick(t.want_db, j), wg_pick(t.M, j), wg_pick(t.N, j),
                 wg_pick(t.K, j), wg_pick(t.BN, j), wg_pick(t.BK, j), wg_pick(t.RS, j), (int)blockIdx.x - f0, f1 - f0);
}

// dst[i] = sum over the workgroup slots of part[slot][i]: 64 outputs per workgroup, the slots dealt to its 16 waves (256 slots:
// four rounds of four independent loads per thread — with 4 waves it was sixteen rounds, ~10 us of load latency per launch)
constexpr int WGR_WAVES = 16;

__device__ __forceinline__ float wg_reduce_slots(const float *__restrict__ part, int slots, int n, int i, int li, int sg, float (*red)[64])
{
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (i < n) {
        int s = sg;
        for (; s + 3 * WGR_WAVES < slots; s += 4 * WGR_WAVES) {
            a0 += part[(size_t)s * n + i];
            a1 += part[(size_t)(s + WGR_WAVES) * n + i];
            a2 += part[(size_t)(s + 2 * WGR_WAVES) * n + i];
            a3 += part[(size_t)(s + 3 * WGR_WAVES) * n + i];
        }
        for (; s < slots; s += WGR_WAVES) a0 += part[(size_t)s * n + i];
    }
    red[sg][li] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    float v = 0.f;
    if (sg == 0) {
#pragma unroll
        for (int w = 0; w < WGR_WAVES; w++) v += red[w][li];      // fixed order
    }
    return v;
}

__global__ void __launch_bounds__(64 * WGR_WAVES) k_linear_wgrad_reduce(const float *__restrict__ part, int slots, int n, float *__restrict__ dW,
                                                                       int nk, float *__restrict__ db)
{
    __shared__ float red[WGR_WAVES][64];
    const int li = threadIdx.x & 63, sg = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + li;
    const float v = wg_reduce_slots(part, slots, n, i, li, sg, red);
    if (sg == 0 && i < n) {
        if (i < nk) dW[i] = v;
        else db[i - nk] = v;
    }
}

// The same sum for up to 8 weight gradients in one launch (the layers of one network's backward pass): a block finds its
// job through the by-value table of first blocks.
constexpr int WGR_JOBS = 16;
struct WgReduceJobs {
    const float *part[WGR_JOBS];
    float *dW[WGR_JOBS], *db[WGR_JOBS];
    int slots[WGR_JOBS], n[WGR_JOBS], nk[WGR_JOBS], first_block[WGR_JOBS + 1];
};

__global__ void __launch_bounds__(64 * WGR_WAVES) k_linear_wgrad_reduce_many(WgReduceJobs t, int jobs)
{
    __shared__ float red[WGR_WAVES][64];
    int job = 0;
    while (job + 1 < jobs && (int)blockIdx.x >= t.first_block[job + 1]) job++;
    const int li = threadIdx.x & 63, sg = threadIdx.x >> 6;
    const int i = ((int)blockIdx.x - t.first_block[job]) * 64 + li;
    const float v = wg_reduce_slots(t.part[job], t.slots[job], t.n[job], i, li, sg, red);
    if (sg == 0 && i < t.n[job]) {
        if (i < t.nk[job]) t.dW[job][i] = v;
        else t.db[job][i - t.nk[job]] = v;
    }
}

template <int VG, int VX>
static int launch_wgrad2(const float *G, const float *X, float *dW, float *db, bool want_db, float *part, int slots, long long M,
                         int N, int K, hipStream_t s)
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
        hipLaunchKernelGGL((k_linear_wgrad<VG, VX>), dim3(grid), dim3(64 * pairs * RS), lds, s, G, X, part, want_db ? 1 : 0, M, N, K,
                           BN, BK, RS);
    }
    if (!dW) return grid;      // partial sums only: the caller adds the slots later (gsvc_linear_wgrad_reduce_many)
    const int n = N * K + (want_db ? N : 0);
    ProfScope _prof("k_linear_wgrad_reduce", s);
    hipLaunchKernelGGL(k_linear_wgrad_reduce, dim3((n + 63) / 64), dim3(64 * WGR_WAVES), 0, s, part, grid, n, dW, N * K, db);
    return grid;
}

static int launch_wgrad_t(const float *G, const float *X, float *dW, float *db, bool want_db, float *part, int slots, long long M,
                          int N, int K, int max_tn, int max_tk, hipStream_t s)
{
    // at most max_tn / max_tk tiles per block: a lane's vector load is as wide as its rows' alignment allows (a 16-byte load
    // of rows that are only 8-byte aligned is split by the memory pipeline: K = 50 ran 20 % slower than with two 8-byte loads)
    const int BN = ((N + 15) / 16 + max_tn - 1) / max_tn, BK = ((K + 15) / 16 + max_tk - 1) / max_tk, pairs = BN * BK;
    const int RS = pairs >= WG_MAX_WAVES ? 1 : WG_MAX_WAVES / pairs;
    const size_t lds = (size_t)pairs * wgt_block_floats(N, BN) * sizeof(float);
    if (lds > 150 * 1024) {      // never dispatch a workgroup the CU cannot hold (the caller checks; this is the last line)
        set_error("linear_wgrad: %d output blocks do not fit one workgroup's LDS", pairs);
        return -1;
    }
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_linear_wgrad_t), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        attr_set = true;
    }
    const long long RB = (M + 15) / 16, want = (RB + RS - 1) / RS;
    const int grid = (int)(want < slots ? want : slots);
    {
        ProfScope _prof("k_linear_wgrad", s);
        hipLaunchKernelGGL(k_linear_wgrad_t, dim3(grid), dim3(64 * pairs * RS), lds, s, G, X, part, want_db ? 1 : 0, M, N, K, BN, BK, RS);
    }
    if (!dW) return grid;      // partial sums only: the caller adds the slots later (gsvc_linear_wgrad_reduce_many)
    const int n = N * K + (want_db ? N : 0);
    ProfScope _prof("k_linear_wgrad_reduce", s);
    hipLaunchKernelGGL(k_linear_wgrad_reduce, dim3((n + 63) / 64), dim3(64 * WGR_WAVES), 0, s, part, grid, n, dW, N * K, db);
    return grid;
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

extern "C" int64_t gsvc_linear_wgrad_workspace(int32_t N, int32_t K)
{
    return (int64_t)256 * ((int64_t)N * K + N);      // one slot per workgroup (one per CU), floats
}

static int wgrad_launch(const float *G, const float *X, float *dW, float *db, bool want_db, int64_t M, int32_t N, int32_t K,
                        float *workspace, int64_t workspace_floats, hipStream_t s, const char *what, int *slots_used)
{
    GSVC_REQUIRE(M > 0 && K > 0 && N > 0, "%s: bad shape", what);
    if (N > LIN_NT_MAX * 16 || K > LIN_NT_MAX * 16) {
        set_error("%s: N=%d / K=%d exceed %d", what, N, K, LIN_NT_MAX * 16);
        return GSVC_E_UNSUPPORTED;
    }
    GSVC_REQUIRE(G && X && workspace, "%s: NULL pointer", what);
    const int64_t per_slot = (int64_t)N * K + (want_db ? N : 0);
    int64_t slots = workspace_floats / per_slot;
    GSVC_REQUIRE(slots >= 1, "%s: workspace smaller than one slot (N*K + N floats)", what);
    if (slots > 256) slots = 256;
    const int vg = vec_of(G, N), vx = vec_of(X, K);
    int used = 0;
    static const bool old_kernel = getenv("GSVC_WGRAD_BLOCK64") != nullptr;      // A/B: the 64 x 64-block kernel
    // tiles per block: 4 (16-byte loads) only for 16-byte aligned rows — a 16-byte load of rows aligned to 8 is split by the
    // memory pipeline (K = 50 ran 20 % slower than with two 8-byte loads); 12- and 8-byte loads are fine at 4-byte alignment
    const int mt_g = vg == 4 ? 4 : 3, mt_x = vx == 4 ? 4 : 3;
    const int bn_t = ((N + 15) / 16 + mt_g - 1) / mt_g, bk_t = ((K + 15) / 16 + mt_x - 1) / mt_x;
    if (!old_kernel && bn_t * bk_t <= WG_MAX_WAVES && (size_t)bn_t * bk_t * wgt_block_floats(N, bn_t) * sizeof(float) <= 150 * 1024) {      // the blocks' LDS images: the limit of one workgroup
        used = launch_wgrad_t(G, X, dW, db, want_db, workspace, (int)slots, M, N, K, mt_g, mt_x, s);
        if (used < 0) return GSVC_E_UNSUPPORTED;
        if (slots_used) *slots_used = used;
        return check_launch(what);
    }
#define WG_CASE(a, b) if (vg == a && vx == b) used = launch_wgrad2<a, b>(G, X, dW, db, want_db, workspace, (int)slots, M, N, K, s)
    WG_CASE(4, 4); WG_CASE(4, 2); WG_CASE(4, 1); WG_CASE(2, 4); WG_CASE(2, 2); WG_CASE(2, 1); WG_CASE(1, 4); WG_CASE(1, 2); WG_CASE(1, 1);
#undef WG_CASE
    if (slots_used) *slots_used = used;
    return check_launch(what);
}

extern "C" int gsvc_linear_wgrad(const float *G, const float *X, float *dW, float *db, int64_t M, int32_t N, int32_t K,
                                 float *workspace, int64_t workspace_floats, void *stream)
{
    GSVC_REQUIRE(M >= 0 && K > 0 && N > 0, "linear_wgrad: bad shape");
    GSVC_REQUIRE(dW, "linear_wgrad: NULL pointer");
    hipStream_t s = (hipStream_t)stream;
    if (M == 0) {
        (void)hipMemsetAsync(dW, 0, sizeof(float) * (size_t)N * K, s);
        if (db) (void)hipMemsetAsync(db, 0, sizeof(float) * (size_t)N, s);
        return GSVC_OK;
    }
    return wgrad_launch(G, X, dW, db, db != nullptr, M, N, K, workspace, workspace_floats, s, "linear_wgrad", nullptr);
}

extern "C" int gsvc_linear_wgrad_partial(const float *G, const float *X, int32_t want_db, int64_t M, int32_t N, int32_t K,
                                         float *workspace, int64_t workspace_floats, int32_t *slots_used, void *stream)
{
    GSVC_REQUIRE(slots_used, "linear_wgrad_partial: NULL pointer");
    return wgrad_launch(G, X, nullptr, nullptr, want_db != 0, M, N, K, workspace, workspace_floats, (hipStream_t)stream,
                        "linear_wgrad_partial", slots_used);
}

// Partial sums of several products: those whose blocking gives 12 waves go out together, at most WG_MANY per launch
// (k_linear_wgrad_many); any other shape takes the single-product path.
extern "C" int gsvc_linear_wgrad_partial_many(gsvc_wgrad_partial_job *jobs, int32_t n_jobs, void *stream)
{
    GSVC_REQUIRE(jobs && n_jobs >= 0, "linear_wgrad_partial_many: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    static const bool single = getenv("GSVC_WGRAD_NO_BATCH") != nullptr;      // A/B timing
    struct Plan { int job, BN, BK, RS, pairs; double work; };
    Plan batch[WG_MANY];
    int nb = 0;
    auto flush = [&]() -> int {
        if (nb == 0) return GSVC_OK;
        WgManyJobs t;
        t.n = nb;
        double total = 0;
        for (int i = 0; i < nb; i++) total += batch[i].work;
        // workgroups per product in proportion to its MFMA work, at least one, at most its slots / its 16-row chunks
        int wgs[WG_MANY], sum = 0;
        size_t max_lds = 0;
        for (int i = 0; i < nb; i++) {
            const gsvc_wgrad_partial_job &q = jobs[batch[i].job];
            const long long per_slot = (long long)q.N * q.K + (q.want_db ? q.N : 0);
            long long cap = q.workspace_floats / per_slot;
            const long long chunks = ((q.M + 15) / 16 + batch[i].RS - 1) / batch[i].RS;
            if (cap > chunks) cap = chunks;
            if (cap > 256) cap = 256;
            int w = (int)(256.0 * batch[i].work / total + 0.5);
            if (w < 1) w = 1;
            if (w > cap) w = (int)cap;
            wgs[i] = w;
            sum += w;
            const size_t need = (size_t)batch[i].pairs * wgt_block_floats(q.N, batch[i].BN) * sizeof(float);
            if (need > max_lds) max_lds = need;
        }
        while (sum > 256) {      // rounding: take from the largest
            int big = 0;
            for (int i = 1; i < nb; i++)
                if (wgs[i] > wgs[big]) big = i;
            wgs[big]--;
            sum--;
        }
        int at = 0;
        for (int i = 0; i < WG_MANY; i++) {
            const int k = i < nb ? i : 0;
            const gsvc_wgrad_partial_job &q = jobs[batch[k].job];
            t.G[i] = q.G; t.X[i] = q.X; t.part[i] = q.workspace; t.M[i] = q.M; t.want_db[i] = q.want_db ? 1 : 0; t.N[i] = q.N; t.K[i] = q.K;
            t.BN[i] = batch[k].BN; t.BK[i] = batch[k].BK; t.RS[i] = batch[k].RS;
            t.first[i] = at;
            if (i < nb) {
                jobs[batch[i].job].slots_used = wgs[i];
                at += wgs[i];
            }
        }
        t.first[WG_MANY] = at;
        for (int i = nb; i < WG_MANY; i++) t.first[i] = at;
        static bool attr_set = false;
        if (!attr_set) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_linear_wgrad_many), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
            attr_set = true;
        }
        static const bool trace = getenv("GSVC_WGRAD_TRACE") != nullptr;      // one line per launch: the products and their workgroups
        if (trace) {
            fprintf(stderr, "wgrad_many:");
            for (int i = 0; i < nb; i++)
                fprintf(stderr, " [M=%lld N=%d K=%d wgs=%d]", (long long)t.M[i], t.N[i], t.K[i], wgs[i]);
            fprintf(stderr, "\n");
        }
        ProfScope _prof("k_linear_wgrad", s);
        hipLaunchKernelGGL(k_linear_wgrad_many, dim3(at), dim3(64 * WG_MAX_WAVES), max_lds, s, t);
        nb = 0;
        return check_launch("linear_wgrad_partial_many");
    };
    for (int j = 0; j < n_jobs; j++) {
        gsvc_wgrad_partial_job &q = jobs[j];
        GSVC_REQUIRE(q.M > 0 && q.K > 0 && q.N > 0 && q.G && q.X && q.workspace, "linear_wgrad_partial_many: bad job %d", j);
        q.slots_used = 0;
        const int vg = vec_of(q.G, q.N), vx = vec_of(q.X, q.K);
        const int mt_g = vg == 4 ? 4 : 3, mt_x = vx == 4 ? 4 : 3;
        const int BN = ((q.N + 15) / 16 + mt_g - 1) / mt_g, BK = ((q.K + 15) / 16 + mt_x - 1) / mt_x, pairs = BN * BK;
        const long long per_slot = (long long)q.N * q.K + (q.want_db ? q.N : 0);
        const bool fits = q.N <= LIN_NT_MAX * 16 && q.K <= LIN_NT_MAX * 16 && pairs <= WG_MAX_WAVES && WG_MAX_WAVES % pairs == 0 &&
                          (size_t)pairs * wgt_block_floats(q.N, BN) * sizeof(float) <= 150 * 1024 &&
                          q.workspace_floats >= per_slot && !getenv("GSVC_WGRAD_BLOCK64");
        if (single || !fits) {
            if (int rc = flush()) return rc;
            int used = 0;
            if (int rc = wgrad_launch(q.G, q.X, nullptr, nullptr, q.want_db != 0, q.M, q.N, q.K, q.workspace, q.workspace_floats, s,
                                      "linear_wgrad_partial_many", &used))
                return rc;
            q.slots_used = used;
            continue;
        }
        // cost of the product on one CU: the larger of its MFMA time (tiles x 4 steps x 32 cycles per 16-row chunk, 4 SIMDs) and its
        // streaming time (a CU takes in ~10 bytes per cycle: a thin product — 100 x 10 — is bytes, not MFMAs, and dealt by MFMA
        // work alone it got 8 workgroups for 88 MB and ran 1.5x longer than the whole rest of its batch)
        const double chunks = (double)((q.M + 15) / 16);
        const double mfma_cycles = (double)((q.N + 15) / 16) * ((q.K + 15) / 16) * chunks * 4.0 * 32.0 / 4.0;
        const double byte_cycles = (double)q.M * (q.N + q.K) * 4.0 / 10.0;
        batch[nb++] = Plan{j, BN, BK, WG_MAX_WAVES / pairs, pairs, mfma_cycles > byte_cycles ? mfma_cycles : byte_cycles};
        if (nb == WG_MANY)
            if (int rc = flush()) return rc;
    }
    return flush();
}

extern "C" int gsvc_linear_wgrad_reduce_many(const gsvc_wgrad_reduce_job *jobs, int32_t n_jobs, void *stream)
{
    GSVC_REQUIRE(jobs && n_jobs >= 0, "linear_wgrad_reduce_many: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    for (int j0 = 0; j0 < n_jobs; j0 += WGR_JOBS) {
        WgReduceJobs t;
        const int nj = n_jobs - j0 < WGR_JOBS ? n_jobs - j0 : WGR_JOBS;
        int blocks = 0;
        for (int j = 0; j < nj; j++) {
            const gsvc_wgrad_reduce_job &q = jobs[j0 + j];
            GSVC_REQUIRE(q.partial && q.dW && q.slots > 0 && q.N > 0 && q.K > 0, "linear_wgrad_reduce_many: bad job %d", j0 + j);
            t.part[j] = q.partial; t.dW[j] = q.dW; t.db[j] = q.db; t.slots[j] = q.slots;
            t.nk[j] = q.N * q.K; t.n[j] = t.nk[j] + (q.db ? q.N : 0);
            t.first_block[j] = blocks;
            blocks += (t.n[j] + 63) / 64;
        }
        t.first_block[nj] = blocks;
        ProfScope _prof("k_linear_wgrad_reduce", s);
        hipLaunchKernelGGL(k_linear_wgrad_reduce_many, dim3(blocks), dim3(64 * WGR_WAVES), 0, s, t, nj);
    }
    return check_launch("linear_wgrad_reduce_many");
}

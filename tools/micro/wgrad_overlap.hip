// Microbenchmark: the weight-gradient kernel's loop (csrc/linear_wgrad.hip wg_main_t: dW = G^T X over M rows, a wave owns a
// (16 TN) x (16 TK) block and a share of the 16-row chunks) for N = K = 100 (2 x 2 blocks of 4 + 3 column tiles, 3 row splits)
// with its halves switchable: MEM (fragment loads) and MATH (MFMAs).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-slp-vectorize -I gsvc_amd/csrc tools/micro/wgrad_overlap.hip -o tools/micro/wgrad_overlap
#include "linear_ws.h"
#include <cstdio>
using namespace gsvc;

template <int T>
__device__ __forceinline__ void load_t(__amdgpu_buffer_rsrc_t rs, int off, float (&dst)[4])
{
    if (T == 4) {
        const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
        dst[0] = __uint_as_float(t.x); dst[1] = __uint_as_float(t.y); dst[2] = __uint_as_float(t.z); dst[3] = __uint_as_float(t.w);
    } else {
        typedef unsigned int u32x3 __attribute__((ext_vector_type(3)));
        const u32x3 t = __builtin_amdgcn_raw_buffer_load_b96(rs, off, 0, 0);
        dst[0] = __uint_as_float(t.x); dst[1] = __uint_as_float(t.y); dst[2] = __uint_as_float(t.z);
    }
}

template <int TN, int TK, bool MEM, bool MATH>
__device__ __forceinline__ void body(const float *G, const float *X, long long M, int n0, int k0, long long rb, long long workers, int j,
                                     int mq, v4f (&acc)[4][4])
{
    constexpr int N = 100, K = 100;
    const long long RB = (M + 15) >> 4;
    const int ca = n0 + TN * j, cb = k0 + TK * j;
    const bool va = ca < N, vb = cb < K;
    float fa[4][4], fb[4][4];
#pragma unroll
    for (int s = 0; s < 4; s++)
#pragma unroll
        for (int t = 0; t < 4; t++) { fa[s][t] = 0.5f + 0.1f * t + 1e-3f * j; fb[s][t] = 0.25f + 0.2f * t + 1e-3f * mq; }
    if (MEM) {
        const __amdgpu_buffer_rsrc_t rg = ws_block_rsrc(G, rb, RB, M, N), rx = ws_block_rsrc(X, rb, RB, M, K);
#pragma unroll
        for (int s = 0; s < 4; s++) {
            load_t<TN>(rg, va ? ((4 * s + mq) * N + ca) * 4 : BUF_OOB, fa[s]);
            load_t<TK>(rx, vb ? ((4 * s + mq) * K + cb) * 4 : BUF_OOB, fb[s]);
        }
    }
#pragma unroll
    for (int s = 0; s < 4; s++)
#pragma unroll
        for (int t = 0; t < 4; t++) asm volatile("" : "+v"(fa[s][t]), "+v"(fb[s][t]));
    for (; rb < RB; rb += workers) {
        const __amdgpu_buffer_rsrc_t rg = ws_block_rsrc(G, rb + workers, RB, M, N), rx = ws_block_rsrc(X, rb + workers, RB, M, K);
#pragma unroll
        for (int s = 0; s < 4; s++) {
            if (MATH) {
#pragma unroll
                for (int tn = 0; tn < TN; tn++)
#pragma unroll
                    for (int tk = 0; tk < TK; tk++)
                        acc[tn][tk] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[s][tn], fb[s][tk], acc[tn][tk], 0, 0, 0);
            } else {
#pragma unroll
                for (int t = 0; t < 4; t++) acc[0][0][t] += fa[s][t] + fb[s][t];
            }
            if (MEM) {
                load_t<TN>(rg, va ? ((4 * s + mq) * N + ca) * 4 : BUF_OOB, fa[s]);
                load_t<TK>(rx, vb ? ((4 * s + mq) * K + cb) * 4 : BUF_OOB, fb[s]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

template <bool MEM, bool MATH>
__global__ void __launch_bounds__(768) k(const float *__restrict__ G, const float *__restrict__ X, float *__restrict__ out, long long M)
{
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rs = wave / 4, pair = (wave + rs) % 4, bn = pair / 2, bk = pair & 1;
    const int j = lane & 15, mq = lane >> 4;
    const int n0 = bn ? 64 : 0, k0 = bk ? 64 : 0;
    const long long workers = (long long)gridDim.x * 3, rb = (long long)rs * gridDim.x + blockIdx.x;
    v4f acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) acc[a][b] = (v4f){0.f, 0.f, 0.f, 0.f};
    switch (bn * 2 + bk) {
        case 0: body<4, 4, MEM, MATH>(G, X, M, n0, k0, rb, workers, j, mq, acc); break;
        case 1: body<4, 3, MEM, MATH>(G, X, M, n0, k0, rb, workers, j, mq, acc); break;
        case 2: body<3, 4, MEM, MATH>(G, X, M, n0, k0, rb, workers, j, mq, acc); break;
        default: body<3, 3, MEM, MATH>(G, X, M, n0, k0, rb, workers, j, mq, acc); break;
    }
    float s = 0.f;
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) s += acc[a][b][0] + acc[a][b][1] + acc[a][b][2] + acc[a][b][3];
    if (s == 123.456f) out[0] = s;
}

template <bool MEM, bool MATH>
static void run(const char *name, const float *G, const float *X, float *out, long long M)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<MEM, MATH><<<256, 768>>>(G, X, out, M);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 20; r++) k<MEM, MATH><<<256, 768>>>(G, X, out, M);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 20;
    printf("%-40s %7.1f us\n", name, ms * 1e3);
}

int main()
{
    const long long M = 200000;
    float *G, *X, *out;
    (void)hipMalloc(&G, M * 100 * 4); (void)hipMalloc(&X, M * 100 * 4); (void)hipMalloc(&out, 64);
    (void)hipMemset(G, 0, M * 100 * 4); (void)hipMemset(X, 0, M * 100 * 4);
    run<true, false>("fragment loads only", G, X, out, M);
    run<false, true>("MFMAs only", G, X, out, M);
    run<true, true>("both (= k_linear_wgrad_t's loop)", G, X, out, M);
    return 0;
}

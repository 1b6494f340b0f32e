// Microbenchmark: do the fp32 MFMA stream and the memory stream of the weight-stationary row-block GEMM (csrc/linear_ws.h
// k_linear_ws) overlap on gfx950?  The kernel below is k_linear_ws's main loop for K = N = 100 (NT = 7 column tiles, 7 k-groups,
// 16-byte fragment loads through block descriptors with the hardware range check, weight fragments from LDS, register-neutral
// prefetch, 16-byte stores) with its two halves switchable: MEM (fragment loads + stores) and MATH (LDS reads + MFMAs).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I gsvc_amd/csrc tools/micro/gemm_overlap.hip -o tools/micro/gemm_overlap
#include "linear_ws.h"
#include <cstdio>
using namespace gsvc;

template <bool MEM, bool MATH, int THREADS>
__global__ void __launch_bounds__(THREADS) k(const float *__restrict__ X, float *__restrict__ Y, long long M, long long wrap)
{
    constexpr int NT = 7, KGM = 7, K = 100, N = 100;
    extern __shared__ float lds[];
    constexpr int WAVES = THREADS / 64;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, kq = lane >> 4;
    const int ld = ws_ld(K);
    const long long RB = (M + 15) >> 4, stride = (long long)gridDim.x * WAVES;
    long long rb = (long long)blockIdx.x * WAVES + wave;
    auto at = [&](long long b) { return wrap > 0 && b < RB ? b % wrap : b; };      // wrap > 0: the same few blocks (L2-resident)
    for (int i = tid; i < NT * 16 * ld; i += THREADS) lds[i] = 1e-3f * (float)(i % 97);
    __syncthreads();
    float4 a[KGM];
    const int voff = fr * K * 4 + 16 * kq;
    {
        const __amdgpu_buffer_rsrc_t rx = ws_block_rsrc(X, MEM ? at(rb) : RB, RB, M, K);
#pragma unroll
        for (int g = 0; g < KGM; g++) a[g] = ws_load_group<4, KGM>(rx, voff, kq, K, g);
    }
#pragma unroll
    for (int g = 0; g < KGM; g++) asm volatile("" : "+v"(a[g].x), "+v"(a[g].y), "+v"(a[g].z), "+v"(a[g].w));
    const float *wb = lds + fr * ld + 4 * kq;
    const int c00 = 4 * kq, yoff = (fr * N + c00) * 4;
    for (; rb < RB; rb += stride) {
        const __amdgpu_buffer_rsrc_t rx = ws_block_rsrc(X, MEM ? at(rb + stride) : RB, RB, M, K);
        v4f acc[NT];
#pragma unroll
        for (int t = 0; t < NT; t++) acc[t] = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < KGM; g++) {
            const float4 ag = a[g];
            if (MATH) {
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
            } else {
                acc[g][0] += ag.x + ag.y + ag.z + ag.w;
            }
            a[g] = ws_load_group<4, KGM>(rx, voff, kq, K, g);
            __builtin_amdgcn_sched_barrier(0);
        }
        const __amdgpu_buffer_rsrc_t ry = ws_block_rsrc(Y, MEM ? at(rb) : RB, RB, M, N);
#pragma unroll
        for (int t = 0; t < NT; t++) {
            float v[4] = {acc[t][0], acc[t][1], acc[t][2], acc[t][3]};
            ws_store4(ry, yoff + 64 * t, c00 + 16 * t, N, 4, v);
        }
    }
}

// variant: TWO 16-row blocks per wave and iteration share every weight fragment read from LDS (half the LDS bytes per MFMA)
template <bool MEM, bool MATH, int THREADS>
__global__ void __launch_bounds__(THREADS) k2(const float *__restrict__ X, float *__restrict__ Y, long long M, long long wrap)
{
    constexpr int NT = 7, KGM = 7, K = 100, N = 100;
    extern __shared__ float lds[];
    constexpr int WAVES = THREADS / 64;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, kq = lane >> 4;
    const int ld = ws_ld(K);
    const long long RB = (M + 15) >> 4, stride = 2 * (long long)gridDim.x * WAVES;
    long long rb = 2 * ((long long)blockIdx.x * WAVES + wave);
    auto at = [&](long long b) { return wrap > 0 && b < RB ? b % wrap : b; };
    for (int i = tid; i < NT * 16 * ld; i += THREADS) lds[i] = 1e-3f * (float)(i % 97);
    __syncthreads();
    float4 a0[KGM], a1[KGM];
    const int voff = fr * K * 4 + 16 * kq;
    {
        const __amdgpu_buffer_rsrc_t r0 = ws_block_rsrc(X, MEM ? at(rb) : RB, RB, M, K), r1 = ws_block_rsrc(X, MEM ? at(rb + 1) : RB, RB, M, K);
#pragma unroll
        for (int g = 0; g < KGM; g++) { a0[g] = ws_load_group<4, KGM>(r0, voff, kq, K, g); a1[g] = ws_load_group<4, KGM>(r1, voff, kq, K, g); }
    }
#pragma unroll
    for (int g = 0; g < KGM; g++) {
        asm volatile("" : "+v"(a0[g].x), "+v"(a0[g].y), "+v"(a0[g].z), "+v"(a0[g].w));
        asm volatile("" : "+v"(a1[g].x), "+v"(a1[g].y), "+v"(a1[g].z), "+v"(a1[g].w));
    }
    const float *wb = lds + fr * ld + 4 * kq;
    const int c00 = 4 * kq, yoff = (fr * N + c00) * 4;
    for (; rb < RB; rb += stride) {
        const __amdgpu_buffer_rsrc_t r0 = ws_block_rsrc(X, MEM ? at(rb + stride) : RB, RB, M, K);
        const __amdgpu_buffer_rsrc_t r1 = ws_block_rsrc(X, MEM ? at(rb + stride + 1) : RB, RB, M, K);
        v4f acc0[NT], acc1[NT];
#pragma unroll
        for (int t = 0; t < NT; t++) { acc0[t] = (v4f){0.f, 0.f, 0.f, 0.f}; acc1[t] = acc0[t]; }
#pragma unroll
        for (int g = 0; g < KGM; g++) {
            const float4 ag0 = a0[g], ag1 = a1[g];
            if (MATH) {
#pragma unroll
                for (int t0 = 0; t0 < NT; t0 += 4) {
                    float4 b[4];
#pragma unroll
                    for (int tt = 0; tt < 4; tt++)
                        if (t0 + tt < NT) b[tt] = *reinterpret_cast<const float4 *>(wb + (t0 + tt) * 16 * ld + 16 * g);
#define STEP(c)                                                                                                              \
    _Pragma("unroll") for (int tt = 0; tt < 4; tt++) if (t0 + tt < NT) {                                                    \
        acc0[t0 + tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[tt].c, ag0.c, acc0[t0 + tt], 0, 0, 0);                        \
        acc1[t0 + tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[tt].c, ag1.c, acc1[t0 + tt], 0, 0, 0);                        \
    }
                    STEP(x) STEP(y) STEP(z) STEP(w)
#undef STEP
                }
            } else {
                acc0[g][0] += ag0.x + ag0.y + ag0.z + ag0.w;
                acc1[g][0] += ag1.x + ag1.y + ag1.z + ag1.w;
            }
            a0[g] = ws_load_group<4, KGM>(r0, voff, kq, K, g);
            a1[g] = ws_load_group<4, KGM>(r1, voff, kq, K, g);
            __builtin_amdgcn_sched_barrier(0);
        }
        const __amdgpu_buffer_rsrc_t y0 = ws_block_rsrc(Y, MEM ? at(rb) : RB, RB, M, N), y1 = ws_block_rsrc(Y, MEM ? at(rb + 1) : RB, RB, M, N);
#pragma unroll
        for (int t = 0; t < NT; t++) {
            float v[4] = {acc0[t][0], acc0[t][1], acc0[t][2], acc0[t][3]};
            ws_store4(y0, yoff + 64 * t, c00 + 16 * t, N, 4, v);
            float w[4] = {acc1[t][0], acc1[t][1], acc1[t][2], acc1[t][3]};
            ws_store4(y1, yoff + 64 * t, c00 + 16 * t, N, 4, w);
        }
    }
}

template <bool MEM, bool MATH, int THREADS>
static void run2(const char *name, const float *X, float *Y, long long M, long long wrap = 0)
{
    const size_t lds = (size_t)7 * 16 * (ws_ld(100) + 1) * sizeof(float);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k2<MEM, MATH, THREADS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k2<MEM, MATH, THREADS><<<256, THREADS, lds>>>(X, Y, M, wrap);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 10; r++) k2<MEM, MATH, THREADS><<<256, THREADS, lds>>>(X, Y, M, wrap);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    printf("%-52s %7.1f us\n", name, ms * 1e3);
}

// variant: the weight fragments of the NEXT sub-group of column tiles are read from LDS before the current sub-group's MFMAs
// are issued (two sets of fragment registers), across k-groups and row blocks: no MFMA waits for an LDS read issued just before it
template <bool MEM, int THREADS, bool NOLDS = false>
__global__ void __launch_bounds__(THREADS) k3(const float *__restrict__ X, float *__restrict__ Y, long long M, long long wrap)
{
    constexpr int NT = 7, KGM = 7, K = 100, N = 100;
    extern __shared__ float lds[];
    constexpr int WAVES = THREADS / 64;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, kq = lane >> 4;
    const int ld = ws_ld(K);
    const long long RB = (M + 15) >> 4, stride = (long long)gridDim.x * WAVES;
    // the remainder blocks (RB mod total waves) go to wave 0 of as many workgroups, not to all waves of the first few
    long long rb = (long long)wave * gridDim.x + blockIdx.x;
    auto at = [&](long long b) { return wrap > 0 && b < RB ? b % wrap : b; };
    for (int i = tid; i < NT * 16 * ld; i += THREADS) lds[i] = 1e-3f * (float)(i % 97);
    __syncthreads();
    float4 a[KGM];
    const int voff = fr * K * 4 + 16 * kq;
    {
        const __amdgpu_buffer_rsrc_t rx = ws_block_rsrc(X, MEM ? at(rb) : RB, RB, M, K);
#pragma unroll
        for (int g = 0; g < KGM; g++) a[g] = ws_load_group<4, KGM>(rx, voff, kq, K, g);
    }
#pragma unroll
    for (int g = 0; g < KGM; g++) asm volatile("" : "+v"(a[g].x), "+v"(a[g].y), "+v"(a[g].z), "+v"(a[g].w));
    const float *wb = lds + fr * ld + 4 * kq;
    const int c00 = 4 * kq, yoff = (fr * N + c00) * 4;
    // sub-group s of k-group g: tiles [4 s, min(4 s + 4, NT))
    float4 b[2][4];
    float4 breg[2][4];      // NOLDS: the "weights" are eight registers (timing only)
#pragma unroll
    for (int u = 0; u < 8; u++) {
        breg[u >> 2][u & 3] = make_float4(0.1f * u + lane * 1e-3f, 0.2f * u, 0.3f * u, 0.4f * u);
        asm volatile("" : "+v"(breg[u >> 2][u & 3].x), "+v"(breg[u >> 2][u & 3].y), "+v"(breg[u >> 2][u & 3].z), "+v"(breg[u >> 2][u & 3].w));
    }
    auto read_b = [&](int g, int sgrp, float4 (&dst)[4]) {
#pragma unroll
        for (int tt = 0; tt < 4; tt++)
            if (4 * sgrp + tt < NT) dst[tt] = NOLDS ? breg[sgrp][tt] : *reinterpret_cast<const float4 *>(wb + (4 * sgrp + tt) * 16 * ld + 16 * g);
    };
    read_b(0, 0, b[0]);
    for (; rb < RB; rb += stride) {
        const __amdgpu_buffer_rsrc_t rx = ws_block_rsrc(X, MEM ? at(rb + stride) : RB, RB, M, K);
        v4f acc[NT];
#pragma unroll
        for (int t = 0; t < NT; t++) acc[t] = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < KGM; g++) {
            const float4 ag = a[g];
#pragma unroll
            for (int sgrp = 0; sgrp < 2; sgrp++) {
                const int cur = (2 * g + sgrp) & 1, nxt = cur ^ 1;
                // next sub-group's fragments (the first of the next block after the last): in flight during these MFMAs
                if (sgrp == 0) read_b(g, 1, b[nxt]);
                else read_b(g + 1 < KGM ? g + 1 : 0, 0, b[nxt]);
#define STEP(c)                                                                                                              \
    _Pragma("unroll") for (int tt = 0; tt < 4; tt++) if (4 * sgrp + tt < NT)                                                \
        acc[4 * sgrp + tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[cur][tt].c, ag.c, acc[4 * sgrp + tt], 0, 0, 0);
                STEP(x) STEP(y) STEP(z) STEP(w)
#undef STEP
                __builtin_amdgcn_sched_barrier(0);
            }
            if (MEM) a[g] = ws_load_group<4, KGM>(rx, voff, kq, K, g);
            else a[g].x += 1e-9f;
            __builtin_amdgcn_sched_barrier(0);
        }
        const __amdgpu_buffer_rsrc_t ry = ws_block_rsrc(Y, MEM ? at(rb) : RB, RB, M, N);
#pragma unroll
        for (int t = 0; t < NT; t++) {
            float v[4] = {acc[t][0], acc[t][1], acc[t][2], acc[t][3]};
            if (MEM) ws_store4(ry, yoff + 64 * t, c00 + 16 * t, N, 4, v);
            else if (v[0] == 123.456f) Y[t] = v[1] + v[2] + v[3];
        }
    }
}

template <bool MEM, int THREADS, bool NOLDS = false>
static void run3(const char *name, const float *X, float *Y, long long M, long long wrap = 0)
{
    const size_t lds = (size_t)7 * 16 * (ws_ld(100) + 1) * sizeof(float);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k3<MEM, THREADS, NOLDS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k3<MEM, THREADS, NOLDS><<<256, THREADS, lds>>>(X, Y, M, wrap);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 10; r++) k3<MEM, THREADS, NOLDS><<<256, THREADS, lds>>>(X, Y, M, wrap);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    (void)hipEventRecord(e0);
    for (int r = 0; r < 400; r++) k3<MEM, THREADS, NOLDS><<<256, THREADS, lds>>>(X, Y, M, wrap);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms2; (void)hipEventElapsedTime(&ms2, e0, e1); ms2 /= 400;
    printf("%-52s %7.1f us  (400 launches back to back: %.1f us)\n", name, ms * 1e3, ms2 * 1e3);
}

template <bool MEM, bool MATH, int THREADS>
static void run(const char *name, const float *X, float *Y, long long M, long long wrap = 0)
{
    const size_t lds = (size_t)7 * 16 * (ws_ld(100) + 1) * sizeof(float);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k<MEM, MATH, THREADS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<MEM, MATH, THREADS><<<256, THREADS, lds>>>(X, Y, M, wrap);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 10; r++) k<MEM, MATH, THREADS><<<256, THREADS, lds>>>(X, Y, M, wrap);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    printf("%-52s %7.1f us\n", name, ms * 1e3);
}

int main()
{
    const long long M = 200000;
    float *X, *Y;
    (void)hipMalloc(&X, M * 100 * 4); (void)hipMalloc(&Y, M * 100 * 4);
    (void)hipMemset(X, 0, M * 100 * 4);
    run<true, false, 1024>("fragment loads + stores only, 1024 threads", X, Y, M);
    run<false, true, 1024>("LDS reads + MFMAs only, 1024 threads", X, Y, M);
    run<true, true, 1024>("both (= k_linear_ws), 1024 threads", X, Y, M);
    run<true, false, 1024>("loads + stores from / to L2 (2048 rows)", X, Y, M, 128);
    run<true, true, 1024>("both, rows from / to L2", X, Y, M, 128);
    run3<false, 1024>("weight fragments one sub-group ahead: LDS + MFMAs", X, Y, M);
    run3<true, 1024>("weight fragments one sub-group ahead: both", X, Y, M);
    run3<false, 1024, true>("MFMAs with register weights (no LDS reads), 1024 thr", X, Y, M);
    run3<true, 1024, true>("the same + fragment loads + stores", X, Y, M);
    run3<false, 512>("the same, 512 threads: LDS + MFMAs", X, Y, M);
    run3<true, 512>("the same, 512 threads: both", X, Y, M);
    run2<false, true, 512>("two row blocks per wave: LDS + MFMAs only, 512 thr", X, Y, M);
    run2<true, true, 512>("two row blocks per wave: both, 512 threads", X, Y, M);
    run<true, false, 512>("fragment loads + stores only, 512 threads", X, Y, M);
    run<false, true, 512>("LDS reads + MFMAs only, 512 threads", X, Y, M);
    run<true, true, 512>("both, 512 threads", X, Y, M);
    return 0;
}

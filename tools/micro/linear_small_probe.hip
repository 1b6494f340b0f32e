// A few-rows form of the linear kernel, measured and NOT kept (round 6, HISTORY section 11): one workgroup per 16-row block, its
// waves split the column tiles and read their weight fragments straight from L2 (no LDS image) — for the ~10 k-row products of the
// rate sample, where the weight-stationary kernel (csrc/linear_ws.h) fills 40 CUs.  What bounds it: the loop with pieces switched
// off (DBG bits):
//   1 no weight loads   2 no X loads   4 no MFMAs (one add per fragment keeps the loads alive)   8 no stores
// and the workgroup shape varied (waves per row block).  10 000 x 192 -> 160 on MI355X: everything 17.6 us, without the weight
// loads 11.0, without any load 9.9, without the MFMAs 17.1, nothing but the stores 3.3 (k_linear_ws: 30): every
// row block re-reads its 123 KB of weights from L2 — 90 MB per launch through the L2 -> L1 path, which is what the 7 us of weight
// loads are (the fragment pattern itself streams at the contiguous rate: tools/micro/frag_access.hip).  In the fitting step the form was slower, not
// faster (6.81 / 7.15 -> 7.06 / 7.85 ms): spread over every CU it takes L2 and issue slots from the rasterizer's streams.  hipcc -O3 --offload-arch=gfx950 -I gsvc_amd/csrc
// tools/micro/linear_small_probe.hip -o tools/micro/linear_small_probe ; usage: linear_small_probe [M K N]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "linear_ws.h"

using namespace gsvc;

template <int TPW, int KGM, int DBG, int WAVES>
__global__ void __launch_bounds__(64 * WAVES) k_probe(const float *__restrict__ X, const float *__restrict__ W, float *__restrict__ Y,
                                                      long long M, int K, int N)
{
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, kq = lane >> 4;
    const int NT = (N + 15) >> 4, KG = (K + 15) >> 4;
    const int t0 = wave * TPW;
    if (t0 >= NT) return;
    const long long rb = blockIdx.x, RB = gridDim.x;
    const __amdgpu_buffer_rsrc_t rx = ws_block_rsrc(X, rb, RB, M, K);
    const unsigned wlo = __builtin_amdgcn_readfirstlane((unsigned)(reinterpret_cast<uintptr_t>(W) & 0xffffffffu));
    const unsigned whi = __builtin_amdgcn_readfirstlane((unsigned)(reinterpret_cast<uintptr_t>(W) >> 32));
    const __amdgpu_buffer_rsrc_t rw =
        __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(((uintptr_t)whi << 32) | wlo), 0, N * K * 4, 0x00020000);
    const int voff = fr * K * 4 + 16 * kq;
    float4 a[KGM], b[KGM][TPW];
#pragma unroll
    for (int g = 0; g < KGM; g++) {
        if (DBG & 2) a[g] = make_float4(1.f + lane, 2.f, 3.f, 4.f);
        else a[g] = ws_load_a<4, true>(rx, voff + 64 * g, 16 * g + 4 * kq, K);
#pragma unroll
        for (int tt = 0; tt < TPW; tt++) {
            const int n = 16 * (t0 + tt) + fr, k0 = 16 * g + 4 * kq;
            if (DBG & 1) b[g][tt] = make_float4(0.5f, 0.25f + lane, 1.f, 2.f);
            else b[g][tt] = ws_load_a<4, true>(rw, n < N ? (n * K + k0) * 4 : 0x7fff0000, n < N ? k0 : K, K);
        }
    }
    v4f acc[TPW];
#pragma unroll
    for (int tt = 0; tt < TPW; tt++) acc[tt] = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < KGM; g++) {
        if (g < KG) {
            if (DBG & 4) {
#pragma unroll
                for (int tt = 0; tt < TPW; tt++) acc[tt][0] += b[g][tt].x * a[g].x + b[g][tt].y * a[g].y + b[g][tt].z * a[g].z + b[g][tt].w * a[g].w;
            } else {
#pragma unroll
                for (int tt = 0; tt < TPW; tt++) acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[g][tt].x, a[g].x, acc[tt], 0, 0, 0);
#pragma unroll
                for (int tt = 0; tt < TPW; tt++) acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[g][tt].y, a[g].y, acc[tt], 0, 0, 0);
#pragma unroll
                for (int tt = 0; tt < TPW; tt++) acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[g][tt].z, a[g].z, acc[tt], 0, 0, 0);
#pragma unroll
                for (int tt = 0; tt < TPW; tt++) acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[g][tt].w, a[g].w, acc[tt], 0, 0, 0);
            }
        }
    }
    const int c00 = 4 * kq + 16 * t0, yoff = (fr * N + c00) * 4;
    const __amdgpu_buffer_rsrc_t ry = ws_block_rsrc(Y, rb, RB, M, N);
#pragma unroll
    for (int tt = 0; tt < TPW; tt++) {
        const float v[4] = {acc[tt][0], acc[tt][1], acc[tt][2], acc[tt][3]};
        if (!(DBG & 8) || v[0] == 123.456f) ws_store4(ry, yoff + 64 * tt, c00 + 16 * tt, N, 4, v);
    }
}

template <int TPW, int KGM, int DBG, int WAVES>
static void run(const char *what, const float *X, const float *W, float *Y, long long M, int K, int N)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const unsigned grid = (unsigned)((M + 15) / 16);
    for (int i = 0; i < 5; i++) hipLaunchKernelGGL((k_probe<TPW, KGM, DBG, WAVES>), dim3(grid), dim3(64 * WAVES), 0, 0, X, W, Y, M, K, N);
    hipEventRecord(e0, 0);
    const int R = 50;
    for (int i = 0; i < R; i++) hipLaunchKernelGGL((k_probe<TPW, KGM, DBG, WAVES>), dim3(grid), dim3(64 * WAVES), 0, 0, X, W, Y, M, K, N);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-64s %7.2f us per launch\n", what, ms / R * 1e3);
    fflush(stdout);
}

__global__ void k_empty(float *y) { if (y == nullptr) y[0] = 0; }

int main(int argc, char **argv)
{
    const long long M = argc > 1 ? atoll(argv[1]) : 10000;
    const int K = argc > 2 ? atoi(argv[2]) : 192, N = argc > 3 ? atoi(argv[3]) : 160;
    float *X, *W, *Y;
    hipMalloc(&X, M * K * 4);
    hipMalloc(&W, (size_t)N * K * 4);
    hipMalloc(&Y, M * N * 4);
    std::vector<float> h(M * K, 0.5f), w((size_t)N * K, 0.25f);
    hipMemcpy(X, h.data(), M * K * 4, hipMemcpyHostToDevice);
    hipMemcpy(W, w.data(), (size_t)N * K * 4, hipMemcpyHostToDevice);
    printf("M = %lld, K = %d, N = %d\n", M, K, N);
    {
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        for (int i = 0; i < 5; i++) hipLaunchKernelGGL(k_empty, dim3(625), dim3(256), 0, 0, Y);
        hipEventRecord(e0, 0);
        for (int i = 0; i < 50; i++) hipLaunchKernelGGL(k_empty, dim3(625), dim3(256), 0, 0, Y);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("%-64s %7.2f us per launch\n", "empty kernel, 625 x 256 threads", ms / 50 * 1e3);
    }
    run<3, 12, 0, 4>("4 waves x 3 tiles, everything", X, W, Y, M, K, N);
    run<3, 12, 1, 4>("4 waves x 3 tiles, no weight loads", X, W, Y, M, K, N);
    run<3, 12, 2, 4>("4 waves x 3 tiles, no X loads", X, W, Y, M, K, N);
    run<3, 12, 3, 4>("4 waves x 3 tiles, no loads", X, W, Y, M, K, N);
    run<3, 12, 4, 4>("4 waves x 3 tiles, no MFMAs", X, W, Y, M, K, N);
    run<3, 12, 8, 4>("4 waves x 3 tiles, no stores", X, W, Y, M, K, N);
    run<3, 12, 7, 4>("4 waves x 3 tiles, no loads, no MFMAs", X, W, Y, M, K, N);
    run<2, 12, 0, 5>("5 waves x 2 tiles, everything", X, W, Y, M, K, N);
    run<1, 12, 0, 10>("10 waves x 1 tile, everything", X, W, Y, M, K, N);
    run<1, 12, 1, 10>("10 waves x 1 tile, no weight loads", X, W, Y, M, K, N);
    run<1, 12, 3, 10>("10 waves x 1 tile, no loads", X, W, Y, M, K, N);
    run<5, 12, 0, 2>("2 waves x 5 tiles, everything", X, W, Y, M, K, N);
    return 0;
}

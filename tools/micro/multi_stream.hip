// Microbenchmark: a chain kernel's memory side alone — per 16-row block a wave reads R and writes W matrices of [M][100] floats,
// either as 16 rows x 64-byte pieces (lane (fr, kq): row fr, columns 16 t + 4 kq .. + 3: the MFMA accumulator layout stored
// row-major) or as lane-linear contiguous 1 KiB per instruction (a blocked private layout), 8 or 16 waves per CU.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/multi_stream.hip -o tools/micro/multi_stream
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE, int THREADS, int R, int W>
__global__ void __launch_bounds__(THREADS) k(const float4 *__restrict__ X, float4 *__restrict__ Y, long long M)
{
    constexpr int N4 = 25, WAVES = THREADS / 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long RB = M / 16, stride = (long long)gridDim.x * WAVES, S = M * N4;
    const int fr = lane & 15, kq = lane >> 4;
    for (long long rb = (long long)wave * gridDim.x + blockIdx.x; rb < RB; rb += stride) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int r = 0; r < R; r++) {
            if (MODE == 0) {
                const float4 *row = X + r * S + (rb * 16 + fr) * N4;
#pragma unroll
                for (int t = 0; t < 7; t++) if (4 * t + kq < N4) { const float4 v = row[4 * t + kq]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
            } else {
                const float4 *blk = X + r * S + rb * 16 * N4;
#pragma unroll
                for (int t = 0; t < 7; t++) if (lane + 64 * t < 16 * N4) { const float4 v = blk[lane + 64 * t]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
            }
        }
        for (int w = 0; w < W; w++) {
            if (MODE == 0) {
                float4 *row = Y + w * S + (rb * 16 + fr) * N4;
#pragma unroll
                for (int t = 0; t < 7; t++) if (4 * t + kq < N4) row[4 * t + kq] = acc;
            } else {
                float4 *blk = Y + w * S + rb * 16 * N4;
#pragma unroll
                for (int t = 0; t < 7; t++) if (lane + 64 * t < 16 * N4) blk[lane + 64 * t] = acc;
            }
        }
    }
}

template <int MODE, int THREADS, int R, int W>
static void run(const char *name, const float4 *X, float4 *Y, long long M)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<MODE, THREADS, R, W><<<256, THREADS>>>(X, Y, M);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 10; r++) k<MODE, THREADS, R, W><<<256, THREADS>>>(X, Y, M);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    printf("R=%d W=%d threads=%4d %-26s %7.1f us  %5.2f TB/s\n", R, W, THREADS, name, ms * 1e3, (double)M * (R + W) * 400 / (ms * 1e-3) / 1e12);
}

int main()
{
    const long long M = 200000;
    float4 *X, *Y;
    (void)hipMalloc(&X, M * 400 * 4); (void)hipMalloc(&Y, M * 400 * 4);
    (void)hipMemset(X, 0, M * 400 * 4);
    run<0, 512, 2, 4>("64-byte pieces", X, Y, M);
    run<1, 512, 2, 4>("contiguous 1 KiB", X, Y, M);
    run<0, 1024, 2, 4>("64-byte pieces", X, Y, M);
    run<1, 1024, 2, 4>("contiguous 1 KiB", X, Y, M);
    run<0, 512, 3, 4>("64-byte pieces", X, Y, M);
    run<1, 512, 3, 4>("contiguous 1 KiB", X, Y, M);
    run<0, 512, 1, 1>("64-byte pieces", X, Y, M);
    run<1, 512, 1, 1>("contiguous 1 KiB", X, Y, M);
    run<0, 512, 0, 4>("64-byte pieces", X, Y, M);
    run<1, 512, 0, 4>("contiguous 1 KiB", X, Y, M);
    run<0, 512, 4, 0>("64-byte pieces", X, Y, M);
    run<1, 512, 4, 0>("contiguous 1 KiB", X, Y, M);
    return 0;
}

// Microbenchmark: bandwidth of the MLP kernels' access pattern on gfx950 — a wave reads a 16-row block of a [M][K] float
// matrix as 16-byte pieces (lane (fr, kq): row fr, columns 16 g + 4 kq .. + 3, one load per k-group g) and writes a [M][N]
// block the same way — against the same bytes moved as contiguous 16-byte-per-lane streams.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/row_block_copy.hip -o tools/micro/row_block_copy
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ void __launch_bounds__(1024) k(const float4 *__restrict__ X, float4 *__restrict__ Y, long long M, int K4, int N4)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long RB = M / 16, stride = (long long)gridDim.x * 16;
    const int fr = lane & 15, kq = lane >> 4;
    for (long long rb = (long long)blockIdx.x * 16 + wave; rb < RB; rb += stride) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (MODE == 0) {            // fragment pattern: 16 rows x 64 bytes per wave instruction
            const float4 *row = X + (rb * 16 + fr) * K4;
            for (int g = 0; 4 * g + kq < K4; g++) { const float4 v = row[4 * g + kq]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
            float4 *out = Y + (rb * 16 + fr) * N4;
            for (int t = 0; 4 * t + kq < N4; t++) out[4 * t + kq] = acc;
        } else {                    // the same block as one contiguous stream
            const float4 *blk = X + rb * 16 * K4;
            for (int i = lane; i < 16 * K4; i += 64) { const float4 v = blk[i]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
            float4 *out = Y + rb * 16 * N4;
            for (int i = lane; i < 16 * N4; i += 64) out[i] = acc;
        }
    }
}

template <int MODE>
static void run(const char *name, const float4 *X, float4 *Y, long long M, int K, int N)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<MODE><<<256, 1024>>>(X, Y, M, K / 4, N / 4);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 10; r++) k<MODE><<<256, 1024>>>(X, Y, M, K / 4, N / 4);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    printf("K=%3d N=%3d %-28s %7.1f us  %5.2f TB/s\n", K, N, name, ms * 1e3, (double)M * (K + N) * 4 / (ms * 1e-3) / 1e12);
}

int main()
{
    const long long M = 200000;
    float4 *X, *Y;
    (void)hipMalloc(&X, M * 192 * 4); (void)hipMalloc(&Y, M * 192 * 4);
    (void)hipMemset(X, 0, M * 192 * 4);
    const int shapes[][2] = {{100, 100}, {64, 64}, {128, 128}, {52, 100}, {100, 12}, {192, 152}};
    for (auto &s : shapes) {
        run<0>("16 rows x 64-byte pieces", X, Y, M, s[0], s[1]);
        run<1>("contiguous block", X, Y, M, s[0], s[1]);
    }
    return 0;
}

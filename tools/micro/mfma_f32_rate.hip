// Microbenchmark: sustained rate of v_mfma_f32_16x16x4_f32 on gfx950 with every SIMD busy (1 / 2 / 4 waves per SIMD, 4
// independent accumulators per wave, operands in registers) — what "100 % MFMA" means for the fp32 MLP kernels.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_f32_rate.hip -o tools/micro/mfma_f32_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(1024) k(int iters, float *out)
{
    v4f a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    float x = threadIdx.x * 1e-3f, y = 1.0f + threadIdx.x * 1e-6f;
    for (int i = 0; i < iters; i++) {
        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, x, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, y, a3, 0, 0, 0);
    }
    const float s = a0[0] + a1[1] + a2[2] + a3[3];
    if (s == 123.456f) out[0] = s;
}

int main()
{
    float *d; (void)hipMalloc(&d, 64);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int waves_per_simd = 1; waves_per_simd <= 4; waves_per_simd *= 2) {
        const int threads = 256 * waves_per_simd, iters = 20000;
        k<<<256, threads>>>(iters, d);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        for (int r = 0; r < 5; r++) k<<<256, threads>>>(iters, d);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        const double mfmas = 256.0 * (threads / 64) * iters * 4;
        printf("%d wave(s) per SIMD: %.1f TFLOP/s fp32, %.1f ns per MFMA and SIMD (32 cycles at 2.4 GHz = 13.3 ns)\n", waves_per_simd,
               mfmas * 2048 / (ms * 1e-3) / 1e12, ms * 1e6 / (mfmas / 1024.0));
    }
    return 0;
}

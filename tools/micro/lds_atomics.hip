// Microbenchmark: rate of LDS atomics on gfx950 (random addresses in a 128 KiB array, 1024-thread workgroups, one per CU).
//   hipcc -O3 --offload-arch=gfx950 tools/micro/lds_atomics.hip -o tools/micro/lds_atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int MODE>
__global__ void __launch_bounds__(1024) k(int iters, float *out)
{
    extern __shared__ float sm[];
    constexpr int WORDS = 32768;
    for (int i = threadIdx.x; i < WORDS; i += 1024) sm[i] = 0.f;
    __syncthreads();
    uint32_t s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    for (int it = 0; it < iters; it++) {
        s = s * 1664525u + 1013904223u;
        const uint32_t row = (s >> 10) & 4095u;          // random row of 8 floats
#pragma unroll
        for (int ch = 0; ch < 8; ch++) {
            const uint32_t a = MODE >= 10 ? row * 9 + ch : row * 8 + ch;
            const float v = (float)(ch + 1);
            if (MODE % 10 == 0) atomicAdd(sm + a, v);                                                  // ds_add_f32
            if (MODE % 10 == 1) atomicAdd(reinterpret_cast<unsigned int *>(sm) + a, (unsigned)ch + 1);      // ds_add_u32
            if (MODE % 10 == 2) atomicAdd(reinterpret_cast<unsigned long long *>(sm) + (a >> 1), (unsigned long long)ch + 1);   // ds_add_u64
            if (MODE % 10 == 3) sm[a] = v;                                                             // ds_write_b32
            if (MODE % 10 == 4) sm[a] += v;                                                            // read + write (racy; timing only)
        }
    }
    __syncthreads();
    float acc = 0.f;
    for (int i = threadIdx.x; i < WORDS; i += 1024) acc += sm[i];
    if (acc == 123.456f) out[0] = acc;
}

template <int MODE>
static void run(const char *name, float *d)
{
    const int iters = 256;
    hipFuncSetAttribute(reinterpret_cast<const void *>(&k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 147456);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<256, 1024, 147456>>>(iters, d);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; r++) k<MODE><<<256, 1024, 147456>>>(iters, d);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double ops = 256.0 * 1024 * iters * 8;
    printf("%-34s %8.1f us  %6.2f lane-ops/cycle/CU (2.4 GHz)\n", name, ms * 1e3, ops / 256 / (ms * 1e-3 * 2.4e9));
}

int main()
{
    float *d; hipMalloc(&d, 64);
    run<0>("ds_add_f32 stride 8", d);
    run<10>("ds_add_f32 stride 9", d);
    run<1>("ds_add_u32 stride 8", d);
    run<11>("ds_add_u32 stride 9", d);
    run<2>("ds_add_u64 stride 8", d);
    run<12>("ds_add_u64 stride 9", d);
    run<3>("ds_write_b32 stride 8", d);
    run<13>("ds_write_b32 stride 9", d);
    run<4>("read+write stride 8", d);
    run<14>("read+write stride 9", d);
    return 0;
}

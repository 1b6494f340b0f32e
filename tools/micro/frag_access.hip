// What does the FRAGMENT access pattern of the MLP kernels cost at the memory pipeline?  A wave that reads a 16-row block of a row-major
// [M][K] fp32 matrix straight into MFMA fragments has lane (fr = lane % 16, kq = lane / 16) read 16 bytes of row fr at column
// 16 g + 4 kq: every 16-lane quarter of the instruction touches 16 different rows.  The same bytes can be read as the block's
// contiguous 16 K floats (lane l reads float4 number l + 64 i): every quarter touches 256 contiguous bytes.  Both forms move the same
// block per wave and iteration; this program times them (loads and stores, HBM-sized and cache-sized matrices).
//   hipcc -O3 --offload-arch=gfx950 tools/micro/frag_access.hip -o tools/micro/frag_access ; usage: frag_access [K=100]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const float *p, long long bytes)
{
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(reinterpret_cast<uintptr_t>(p) & 0xffffffffu));
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(reinterpret_cast<uintptr_t>(p) >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(((uintptr_t)hi << 32) | lo), 0,
                                             __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000);
}

// MODE 0: fragment loads   1: contiguous loads   2: fragment stores   3: contiguous stores
template <int MODE, int KG>
__global__ void __launch_bounds__(512) k_access(float *X, long long M, int K, float *out)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 15, kq = lane >> 4;
    const long long RB = M >> 4, stride = (long long)gridDim.x * 8;
    float acc = 0.f;
    for (long long rb = (long long)wave * gridDim.x + blockIdx.x; rb < RB; rb += stride) {
        const __amdgpu_buffer_rsrc_t r = rsrc(X + rb * 16 * K, 16LL * K * 4);
        u32x4 v[KG];
        if (MODE == 0 || MODE == 2) {
#pragma unroll
            for (int g = 0; g < KG; g++) {
                const int col = 16 * g + 4 * kq;
                const int off = col < K ? (fr * K + col) * 4 : 0x7fff0000;
                if (MODE == 0) v[g] = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
                else __builtin_amdgcn_raw_buffer_store_b128((u32x4){(unsigned)lane, (unsigned)g, 1u, 2u}, r, off, 0, 0);
            }
        } else {
#pragma unroll
            for (int g = 0; g < KG; g++) {
                const int idx = lane + 64 * g;      // float4 number inside the block's 4 K float4s
                const int off = idx < 4 * K ? idx * 16 : 0x7fff0000;
                if (MODE == 1) v[g] = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
                else __builtin_amdgcn_raw_buffer_store_b128((u32x4){(unsigned)lane, (unsigned)g, 1u, 2u}, r, off, 0, 0);
            }
        }
        if (MODE < 2) {
#pragma unroll
            for (int g = 0; g < KG; g++) acc += __uint_as_float(v[g].x) + __uint_as_float(v[g].w);
        }
    }
    if (acc == 123.456f) out[0] = acc;
}

template <int MODE, int KG>
static void run(const char *what, float *X, long long M, int K, float *out)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL((k_access<MODE, KG>), dim3(256), dim3(512), 0, 0, X, M, K, out);
    (void)hipEventRecord(e0, 0);
    const int R = 20;
    for (int i = 0; i < R; i++) hipLaunchKernelGGL((k_access<MODE, KG>), dim3(256), dim3(512), 0, 0, X, M, K, out);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double us = ms / R * 1e3, gb = (double)M * K * 4 / 1e9;
    printf("%-34s M = %8lld K = %3d: %8.1f us  %7.2f TB/s\n", what, M, K, us, gb / (us * 1e-6) / 1e3);
    fflush(stdout);
}

template <int KG>
static void all(float *X, float *out, int K)
{
    for (long long M : {3200000LL, 204800LL, 16384LL}) {
        run<0, KG>("fragment loads", X, M, K, out);
        run<1, KG>("contiguous loads", X, M, K, out);
        run<2, KG>("fragment stores", X, M, K, out);
        run<3, KG>("contiguous stores", X, M, K, out);
    }
}

int main(int argc, char **argv)
{
    const int K = argc > 1 ? atoi(argv[1]) : 100;
    float *X, *out;
    (void)hipMalloc(&X, 3200000LL * 192 * 4);
    (void)hipMalloc(&out, 64);
    (void)hipMemset(X, 0, 3200000LL * 192 * 4);
    const int KG = (K + 15) / 16;
    if (KG <= 4) all<4>(X, out, K);
    else if (KG <= 7) all<7>(X, out, K);
    else all<12>(X, out, K);
    return 0;
}

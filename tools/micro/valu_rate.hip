// Microbenchmark: what a gfx950 SIMD sustains per vector instruction with 1 / 2 / 4 / 8 resident waves — plain v_fma_f32, packed
// v_pk_fma_f32, the transcendentals, DPP adds and the permlane swaps the compositing-backward kernel is made of.  Every wave
// runs 8 independent chains (no dependent-latency stalls), shader cycles from s_memtime, one workgroup of 256 * W threads
// per CU (W waves per SIMD).
//   hipcc -O3 --offload-arch=gfx950 tools/micro/valu_rate.hip -o tools/micro/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int KIND>
__global__ void __launch_bounds__(1024) k(int iters, unsigned long long *cyc, float *out)
{
    float a[8], b[8];
    for (int i = 0; i < 8; i++) { a[i] = threadIdx.x * 1e-3f + i; b[i] = 1.0f + i * 1e-6f; }
    const float x = 1.0000001f, y = 1e-9f;
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f p[8], px = {x, x}, py = {y, y};
    for (int i = 0; i < 8; i++) p[i] = {a[i], b[i]};
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
        if (KIND == 0) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(x), "v"(y));
            REP8(X) REP8(X)
#undef X
        } else if (KIND == 1) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(px), "v"(py));
            REP8(X) REP8(X)
#undef X
        } else if (KIND == 2) {
#define X(i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
            REP8(X) REP8(X)
#undef X
        } else if (KIND == 3) {
#define X(i) asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
            REP8(X) REP8(X)
#undef X
        } else if (KIND == 4) {
#define X(i) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[i]), "+v"(b[i]));
            REP8(X) REP8(X)
#undef X
        } else if (KIND == 5) {          // the replay's mix: 3 fma : 1 cndmask/cmp-like (v_max) : exp + rcp per 16
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(x), "v"(y));
            REP8(X)
#undef X
            asm volatile("v_exp_f32 %0, %0" : "+v"(b[0]));
            asm volatile("v_rcp_f32 %0, %0" : "+v"(b[1]));
#define X(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(x));
            REP8(X)
#undef X
        } else if (KIND == 6) {
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(px));
            REP8(X) REP8(X)
#undef X
        } else if (KIND == 7) {
#define X(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(x));
            REP8(X) REP8(X)
#undef X
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 8; i++) s += a[i] + b[i] + p[i].x + p[i].y;
    if (s == 123.456f) out[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int KIND>
void run(const char *name, unsigned long long *dc, float *d)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int NI = KIND == 5 ? 18 : 16;
    for (int w = 1; w <= 4; w *= 2) {
        const int threads = 256 * w, iters = 20000;
        k<KIND><<<256, threads>>>(iters, dc, d);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        k<KIND><<<256, threads>>>(iters, dc, d);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        unsigned long long c; (void)hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
        const double insts_per_simd = (double)iters * NI * w;          // wave-instructions one SIMD executed
        printf("%-34s %d wave(s)/SIMD: %6.2f ns per wave-instruction and SIMD = %5.2f cycles at 2.4 GHz; s_memtime ticks per "
               "instruction of one wave %6.2f\n", name, w, ms * 1e6 / insts_per_simd, ms * 1e6 / insts_per_simd * 2.4,
               (double)c / (iters * (double)NI));
    }
    // 8 waves per SIMD: two workgroups of 1024 per CU
    {
        const int iters = 20000;
        k<KIND><<<512, 1024>>>(iters, dc, d);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        k<KIND><<<512, 1024>>>(iters, dc, d);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double insts_per_simd = (double)iters * NI * 8;
        printf("%-34s 8 wave(s)/SIMD: %6.2f ns = %5.2f cycles at 2.4 GHz\n", name, ms * 1e6 / insts_per_simd,
               ms * 1e6 / insts_per_simd * 2.4);
    }
}

int main()
{
    float *d; (void)hipMalloc(&d, 64);
    unsigned long long *dc; (void)hipMalloc(&dc, 64);
    run<0>("v_fma_f32", dc, d);
    run<1>("v_pk_fma_f32 (2 fma per lane)", dc, d);
    run<6>("v_pk_mul_f32", dc, d);
    run<2>("v_exp_f32", dc, d);
    run<3>("v_add_f32_dpp quad_perm", dc, d);
    run<4>("v_permlane32_swap_b32", dc, d);
    run<7>("v_cndmask_b32 (vcc)", dc, d);
    run<5>("mix 8 fma + exp + rcp + 8 mul (18)", dc, d);
    return 0;
}

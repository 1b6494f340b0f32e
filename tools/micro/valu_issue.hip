// Microbenchmark (round 5, VERDICT r04 "weak" 4): is one wave64 vector instruction per 2 cycles per SIMD — the rate behind the
// 157.3 TFLOP/s vector peak of MI355X_MICROARCH.md — reachable on gfx950, and under which conditions?  Settles the price list of
// bench.py's `roofline_valu` with measurements whose answer is known: every kernel below executes an exactly known number of
// vector instructions per wave, so  ns per wave-instruction and SIMD,  the shader clock actually held during the run
// (s_memtime ticks against the 100 MHz s_memrealtime), and — run under `rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU
// SQ_BUSY_CYCLES SQ_WAVE_CYCLES ...` — the unit of the SQ counters follow.
//
// Kinds (each 16 independent chains per wave, so no dependent-latency stall from 1 wave per SIMD upwards):
//   0  v_fma_f32 v, v, v, v     three VGPR sources (asm volatile; the form tools/micro/valu_rate.hip measured)
//   1  v_fma_f32 v, v, s, v     one SGPR source
//   2  v_fmac / v_mul / v_add   two-source VOP2 forms: v_mul_f32 v, s, v
//   3  compiler-scheduled       plain C++ fmaf chains on 16 accumulators with uniform multiplier / addend: hipcc -O3 SLP-packs
//                               them into 8 v_pk_fma_f32 v[..], v[..], s[..], v[..] per iteration (checked in the -S output)
//   4  v_pk_fma_f32             packed: two fma per lane and instruction
//   5  the compositing backward's mix per 4 entries: 74 DPP / permlane reduction instructions, 4 x (18.5
//      plain + 2 transcendental) replayed entries -> approximated as 96 fma + 8 v_exp_f32 + 48 v_add_f32_dpp + 8 v_permlane32_swap
//      + 16 v_cndmask per iteration (176 instructions)
//   hipcc -O3 --offload-arch=gfx950 tools/micro/valu_issue.hip -o tools/micro/valu_issue
//   usage: valu_issue [workgroups-per-launch-factor]   (default: every CU busy; "8" = 8 workgroups only: an almost idle chip)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int KIND>
__global__ void __launch_bounds__(1024) k_issue(int iters, float xs, float ys, unsigned long long *ticks, float *out)
{
    float a[16];
    for (int i = 0; i < 16; i++) a[i] = threadIdx.x * 1e-3f + i;
    float b[8];
    for (int i = 0; i < 8; i++) b[i] = 1.0f + i * 1e-6f;
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f p[16], px = {xs, xs}, py = {ys, ys};
    for (int i = 0; i < 16; i++) p[i] = {a[i], a[i] + 1.0f};
    const float xv = xs + threadIdx.x * 0.0f, yv = ys + threadIdx.x * 0.0f;      // the same numbers held in VGPRs
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();           // s_memtime
    const unsigned long long r0 = wall_clock64();                         // s_memrealtime, 100 MHz
    for (int it = 0; it < iters; it++) {
        if (KIND == 0) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(xv), "v"(yv));
            REP16(X)
#undef X
        } else if (KIND == 1) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "s"(xs), "v"(yv));
            REP16(X)
#undef X
        } else if (KIND == 2) {
#define X(i) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i]) : "s"(xs));
            REP16(X)
#undef X
        } else if (KIND == 3) {
#pragma unroll
            for (int i = 0; i < 16; i++) a[i] = __builtin_fmaf(a[i], xs, ys);
        } else if (KIND == 4) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(px), "v"(py));
            REP16(X)
#undef X
        } else if (KIND == 6) {          // inline constant 0 as the addend (what "s = fmaf(w, d, 0)" compiles to)
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, 0" : "+v"(a[i]) : "v"(xv));
            REP16(X)
#undef X
        } else if (KIND == 7) {          // 32-bit literal operand
#define X(i) asm volatile("v_min_f32 %0, 0x3f7d70a4, %0" : "+v"(a[i]));
            REP16(X)
#undef X
        } else if (KIND == 8) {          // two-source VOP2, registers only
#define X(i) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i]) : "v"(xv));
            REP16(X)
#undef X
        } else if (KIND == 9) {          // v_fmac (VOP2 with the destination as third source), registers only
#define X(i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(xv), "v"(yv));
            REP16(X)
#undef X
        } else if (KIND == 10) {         // compare against an SGPR into VCC, then select with an inline 0 (the replay's validity test)
#define X(i) asm volatile("v_cmp_ngt_f32 vcc, %1, %0\n v_cndmask_b32 %0, 0, %0, vcc" : "+v"(a[i]) : "s"(ys) : "vcc");
            REP16(X)
#undef X
        } else if (KIND == 11) {
#define X(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
            REP16(X)
#undef X
        } else if (KIND == 12) {
#define X(i) asm volatile("v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xc" : "+v"(a[i]));
            REP16(X)
#undef X
        } else if (KIND == 13) {
            for (int i = 0; i < 8; i++) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[i]), "+v"(a[i + 8]));
            for (int i = 0; i < 8; i++) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a[i]), "+v"(a[i + 8]));
        } else if (KIND == 14) {         // compare, registers only, into VCC
#define X(i) asm volatile("v_cmp_ngt_f32 vcc, %1, %0" : : "v"(a[i]), "v"(yv) : "vcc");
            REP16(X)
#undef X
        } else if (KIND == 15) {         // compare against an SGPR into VCC
#define X(i) asm volatile("v_cmp_ngt_f32 vcc, %1, %0" : : "v"(a[i]), "s"(ys) : "vcc");
            REP16(X)
#undef X
        } else if (KIND == 16) {         // compare, registers only, into an SGPR pair (VOP3 form)
#define X(i) asm volatile("v_cmp_ngt_f32 s[20:21], %1, %0" : : "v"(a[i]), "v"(yv) : "s20", "s21");
            REP16(X)
#undef X
        } else if (KIND == 17) {         // select on VCC, registers only
#define X(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(xv));
            REP16(X)
#undef X
        } else if (KIND == 18) {         // min against a register (instead of a literal)
#define X(i) asm volatile("v_min_f32 %0, %1, %0" : "+v"(a[i]) : "v"(xv));
            REP16(X)
#undef X
        } else if (KIND == 5) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "s"(xs), "v"(yv));
            REP16(X) REP16(X) REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
#define X(i) asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
            REP16(X) REP16(X) REP16(X)
#undef X
#define X(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(xv));
            REP16(X)
#undef X
            for (int i = 0; i < 8; i++) asm volatile("v_exp_f32 %0, %0" : "+v"(b[i]));
            for (int i = 0; i < 8; i++) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[i]), "+v"(a[i + 8]));
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    const unsigned long long r1 = wall_clock64();
    float s = 0;
    for (int i = 0; i < 16; i++) s += a[i] + p[i].x + p[i].y;
    for (int i = 0; i < 8; i++) s += b[i];
    if (s == 123.456f) out[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { ticks[0] = t1 - t0; ticks[1] = r1 - r0; }
}

template <int KIND>
void run(const char *name, int insts_per_iter, int wg_override, unsigned long long *dc, float *d)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = KIND == 5 ? 3000 : 20000;
    for (int w = 1; w <= 8; w *= 2) {
        // w waves per SIMD: one workgroup of 256 w threads per CU (two of 1024 for w = 8)
        const int threads = w == 8 ? 1024 : 256 * w;
        const int wgs = wg_override > 0 ? wg_override : (w == 8 ? 512 : 256);
        k_issue<KIND><<<wgs, threads>>>(iters, 1.0000001f, 1e-9f, dc, d);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        k_issue<KIND><<<wgs, threads>>>(iters, 1.0000001f, 1e-9f, dc, d);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        unsigned long long c[2]; (void)hipMemcpy(c, dc, 16, hipMemcpyDeviceToHost);
        const double per_simd = (double)iters * insts_per_iter * w;                 // wave-instructions one SIMD executed
        const double mhz = c[1] ? (double)c[0] / ((double)c[1] / 100.0) : 0.0;      // s_memtime ticks per microsecond
        const double ns = ms * 1e6 / per_simd;
        printf("%-44s %d wave(s)/SIMD %3d WGs: %6.3f ns per wave-instruction and SIMD; s_memtime %7.1f MHz -> %5.2f ticks per "
               "instruction and SIMD; ticks per instruction of ONE wave %6.2f\n", name, w, wgs, ns, mhz, ns * mhz * 1e-3,
               (double)c[0] / ((double)iters * insts_per_iter));
    }
}

int main(int argc, char **argv)
{
    const int wg = argc > 1 ? atoi(argv[1]) : 0;
    float *d; (void)hipMalloc(&d, 64);
    unsigned long long *dc; (void)hipMalloc(&dc, 64);
    run<0>("v_fma_f32 v,v,v,v (3 VGPR sources)", 16, wg, dc, d);
    run<1>("v_fma_f32 v,v,s,v (1 SGPR source)", 16, wg, dc, d);
    run<2>("v_mul_f32 v,s,v (VOP2)", 16, wg, dc, d);
    run<3>("fmaf chains by hipcc -O3 (8 v_pk_fma / 16 fma)", 8, wg, dc, d);
    run<4>("v_pk_fma_f32 (2 fma per lane)", 16, wg, dc, d);
    run<6>("v_fma_f32 v,v,v,0 (inline constant)", 16, wg, dc, d);
    run<7>("v_min_f32 v,literal,v", 16, wg, dc, d);
    run<8>("v_mul_f32 v,v,v (VOP2)", 16, wg, dc, d);
    run<9>("v_fmac_f32 v,v,v (VOP2)", 16, wg, dc, d);
    run<10>("v_cmp_ngt_f32 vcc,s,v + v_cndmask v,0,v,vcc (2)", 32, wg, dc, d);
    run<11>("v_rcp_f32", 16, wg, dc, d);
    run<12>("v_add_f32_dpp row_ror:8 bank_mask:0xc", 16, wg, dc, d);
    run<13>("v_permlane32_swap / v_permlane16_swap", 16, wg, dc, d);
    run<14>("v_cmp_ngt_f32 vcc,v,v", 16, wg, dc, d);
    run<15>("v_cmp_ngt_f32 vcc,s,v", 16, wg, dc, d);
    run<16>("v_cmp_ngt_f32 s[..],v,v (VOP3)", 16, wg, dc, d);
    run<17>("v_cndmask_b32 v,v,v,vcc", 16, wg, dc, d);
    run<18>("v_min_f32 v,v,v", 16, wg, dc, d);
    run<5>("backward mix (96 fma 48 dpp 16 cnd 8 exp 8 swap)", 176, wg, dc, d);
    return 0;
}

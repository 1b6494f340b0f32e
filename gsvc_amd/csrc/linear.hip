// gsvc_amd/csrc/linear.hip — tall-skinny fp32 linear layer Y[M,N] = X[M,K] W[N,K]^T + b on MFMA, gfx950.
//
// The generator / deformation / entropy-parameter MLPs of GSVC (reference scene/gaussian_model.py:150-232,
// 411-501) are chains of nn.Linear with K, N <= 192 applied to ~50k-200k anchor rows per step.  Those GEMMs are
// tall and skinny (arithmetic intensity ~25 FLOP/B): they are HBM-bound, and the library GEMM reached only ~7 TF
// (520 us for 196k x 100 x 100).  This kernel streams X once and writes Y once:
//   one workgroup = 4 waves = 64 rows; W and X are staged through LDS in K-chunks of 64 (row stride 66 dwords:
//   conflict-free for the MFMA fragment reads); each wave owns 16 rows x all N columns as N/16 accumulators of
//   v_mfma_f32_16x16x4_f32 (exact fp32 products, fp32 accumulate — same numerics class as the reference's fp32
//   GEMM, different summation order).
#include "common.h"

namespace gsvc {

typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int LIN_KC = 64;          // K chunk staged in LDS
constexpr int LIN_LD = LIN_KC + 2;  // LDS row stride in dwords
constexpr int LIN_ROWS = 64;        // rows per workgroup
constexpr int LIN_NT_MAX = 12;      // N <= 192

// Stage `rows` x `cols_pad` floats of a row-major matrix (leading dimension ld_src, valid columns [c0, c0+cols),
// valid rows < nrows_valid) into LDS with row stride ld_dst; everything outside is zero-filled.  8-byte loads when
// the source rows are 8-byte aligned (ld_src and c0 even), 4-byte loads otherwise.
__device__ __forceinline__ void stage_tile(const float *__restrict__ src, long long row0, long long nrows_valid, int rows,
                                           int c0, int cols, int cols_pad, int ld_src, float *__restrict__ dst, int ld_dst,
                                           int tid)
{
    if (((ld_src | c0) & 1) == 0) {
        const int half = cols_pad >> 1;                 // float2 per row
        for (int i = tid; i < rows * half; i += 256) {
            const int r = i / half, c = 2 * (i - r * half);
            const long long gr = row0 + r;
            float2 v = make_float2(0.f, 0.f);
            if (gr < nrows_valid) {
                if (c + 1 < cols) v = *reinterpret_cast<const float2 *>(src + gr * ld_src + c0 + c);
                else if (c < cols) v.x = src[gr * ld_src + c0 + c];
            }
            *reinterpret_cast<float2 *>(dst + r * ld_dst + c) = v;
        }
    } else {
        for (int i = tid; i < rows * cols_pad; i += 256) {
            const int r = i / cols_pad, c = i - r * cols_pad;
            const long long gr = row0 + r;
            dst[r * ld_dst + c] = (gr < nrows_valid && c < cols) ? src[gr * ld_src + c0 + c] : 0.f;
        }
    }
}

template <int NT>
__global__ void __launch_bounds__(256) k_linear_fwd(const float *__restrict__ X, const float *__restrict__ W,
                                                    const float *__restrict__ bias, float *__restrict__ Y, long long M,
                                                    int K, int N)
{
    extern __shared__ float lds[];
    float *sX = lds;                        // [64][LIN_LD]
    float *sW = lds + LIN_ROWS * LIN_LD;    // [NT*16][LIN_LD]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long row0 = (long long)blockIdx.x * LIN_ROWS;
    const int frag_r = lane & 15, frag_k = lane >> 4;
    v4f acc[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) acc[t] = (v4f){0.f, 0.f, 0.f, 0.f};

    for (int k0 = 0; k0 < K; k0 += LIN_KC) {
        const int kc = min(LIN_KC, K - k0);
        __syncthreads();
        // stage X[row0:row0+64, k0:k0+kc] and W[0:N, k0:k0+kc]; zero-fill the K tail and the N padding
        stage_tile(X, row0, M, LIN_ROWS, k0, kc, LIN_KC, K, sX, LIN_LD, tid);
        stage_tile(W, 0, N, NT * 16, k0, kc, LIN_KC, K, sW, LIN_LD, tid);
        __syncthreads();
        const float *xa = sX + (wave * 16 + frag_r) * LIN_LD + frag_k;
        const float *wb = sW + frag_r * LIN_LD + frag_k;
        const int kend = (kc + 3) & ~3;
        for (int kk = 0; kk < kend; kk += 4) {
            const float a = xa[kk];
#pragma unroll
            for (int t = 0; t < NT; t++) {
                const float b = wb[t * 16 * LIN_LD + kk];
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0);
            }
        }
    }
    // D fragment: lane holds rows 4*(lane/16)+i (i=0..3) of column lane%16
    const int col_in = lane & 15, rbase = 4 * (lane >> 4);
#pragma unroll
    for (int t = 0; t < NT; t++) {
        const int col = t * 16 + col_in;
        if (col < N) {
            const float bv = bias ? bias[col] : 0.f;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const long long gr = row0 + wave * 16 + rbase + i;
                if (gr < M) Y[gr * N + col] = acc[t][i] + bv;
            }
        }
    }
}

template <int NT>
static void launch_linear(const float *X, const float *W, const float *b, float *Y, long long M, int K, int N,
                          hipStream_t s)
{
    const size_t lds = (size_t)(LIN_ROWS + NT * 16) * LIN_LD * sizeof(float);
    static bool attr_set = false;
    if (lds > 48 * 1024 && !attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_linear_fwd<NT>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    ProfScope _prof("k_linear_fwd", s);
    hipLaunchKernelGGL((k_linear_fwd<NT>), dim3((unsigned)((M + LIN_ROWS - 1) / LIN_ROWS)), dim3(256), lds, s, X, W, b, Y,
                       M, K, N);
}

// ---------------------------------------------------------------------------------------------------------
// Weight gradient dW[N,K] += G[M,N]^T X[M,K]: a reduction over the M (anchor) dimension with a tiny output.  The
// library GEMM tiles the OUTPUT (32x32 tiles -> a dozen workgroups on a 256-CU chip, 520 us at M = 196k); here the
// workgroups split M instead: each walks its share of the rows in chunks of 64 staged in LDS, keeps the whole
// N x K product in MFMA accumulators (tiles dealt round-robin to the 4 waves), and adds its partial to dW with
// contiguous float atomics (64-byte segments).
constexpr int WG_ROWS = 64;

template <int TPW>  // accumulator tiles (16x16) per wave
__global__ void __launch_bounds__(256) k_linear_wgrad(const float *__restrict__ G, const float *__restrict__ X,
                                                      float *__restrict__ dW, long long M, int N, int K, int ldg, int ldx)
{
    extern __shared__ float lds[];
    float *sG = lds;                 // [64][ldg]
    float *sX = lds + WG_ROWS * ldg;  // [64][ldx]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nt = (N + 15) / 16, kt = (K + 15) / 16, tiles = nt * kt;
    const int frag_c = lane & 15, frag_r = lane >> 4;
    v4f acc[TPW];
#pragma unroll
    for (int t = 0; t < TPW; t++) acc[t] = (v4f){0.f, 0.f, 0.f, 0.f};

    const long long chunks = (M + WG_ROWS - 1) / WG_ROWS;
    for (long long ch = blockIdx.x; ch < chunks; ch += gridDim.x) {
        const long long row0 = ch * WG_ROWS;
        __syncthreads();
        stage_tile(G, row0, M, WG_ROWS, 0, N, nt * 16, N, sG, ldg, tid);
        stage_tile(X, row0, M, WG_ROWS, 0, K, kt * 16, K, sX, ldx, tid);
        __syncthreads();
#pragma unroll
        for (int t = 0; t < TPW; t++) {
            const int tile = wave + 4 * t;          // wave-uniform
            if (tile < tiles) {
                const int tn = tile / kt, tk = tile - tn * kt;
                const float *ga = sG + frag_r * ldg + tn * 16 + frag_c;
                const float *xb = sX + frag_r * ldx + tk * 16 + frag_c;
                v4f c = acc[t];
#pragma unroll 4
                for (int m4 = 0; m4 < WG_ROWS; m4 += 4)
                    c = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[m4 * ldg], xb[m4 * ldx], c, 0, 0, 0);
                acc[t] = c;
            }
        }
    }
    const int col_in = lane & 15, rbase = 4 * (lane >> 4);
#pragma unroll
    for (int t = 0; t < TPW; t++) {
        const int tile = wave + 4 * t;
        if (tile < tiles) {
            const int tn = tile / kt, tk = tile - tn * kt;
            const int k = tk * 16 + col_in;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int n = tn * 16 + rbase + i;
                if (n < N && k < K) atomicAdd(dW + (size_t)n * K + k, acc[t][i]);
            }
        }
    }
}

template <int TPW>
static void launch_wgrad(const float *G, const float *X, float *dW, long long M, int N, int K, hipStream_t s)
{
    const int nt = (N + 15) / 16, kt = (K + 15) / 16;
    const int ldg = nt * 16 + ((nt & 1) ? 0 : 16);  // row stride = 16 (mod 32) dwords: conflict-free fragment reads
    const int ldx = kt * 16 + ((kt & 1) ? 0 : 16);
    const size_t lds = (size_t)WG_ROWS * (ldg + ldx) * sizeof(float);
    static bool attr_set = false;
    if (lds > 48 * 1024 && !attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_linear_wgrad<TPW>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    const long long chunks = (M + WG_ROWS - 1) / WG_ROWS;
    const unsigned grid = (unsigned)(chunks < 512 ? chunks : 512);
    ProfScope _prof("k_linear_wgrad", s);
    hipLaunchKernelGGL((k_linear_wgrad<TPW>), dim3(grid), dim3(256), lds, s, G, X, dW, M, N, K, ldg, ldx);
}

}  // namespace gsvc

using namespace gsvc;

extern "C" int gsvc_linear_forward(const float *X, const float *W, const float *bias, float *Y, int64_t M, int32_t K,
                                   int32_t N, void *stream)
{
    GSVC_REQUIRE(M >= 0 && K > 0 && N > 0, "linear_forward: bad shape");
    if (N > LIN_NT_MAX * 16) {
        set_error("linear_forward: N=%d exceeds %d", N, LIN_NT_MAX * 16);
        return GSVC_E_UNSUPPORTED;
    }
    if (M == 0) return GSVC_OK;
    GSVC_REQUIRE(X && W && Y, "linear_forward: NULL pointer");
    hipStream_t s = (hipStream_t)stream;
    const int nt = (N + 15) / 16;
    switch (nt) {
        case 1: launch_linear<1>(X, W, bias, Y, M, K, N, s); break;
        case 2: launch_linear<2>(X, W, bias, Y, M, K, N, s); break;
        case 3: launch_linear<3>(X, W, bias, Y, M, K, N, s); break;
        case 4: launch_linear<4>(X, W, bias, Y, M, K, N, s); break;
        case 5: launch_linear<5>(X, W, bias, Y, M, K, N, s); break;
        case 6: launch_linear<6>(X, W, bias, Y, M, K, N, s); break;
        case 7: launch_linear<7>(X, W, bias, Y, M, K, N, s); break;
        case 8: launch_linear<8>(X, W, bias, Y, M, K, N, s); break;
        case 9: launch_linear<9>(X, W, bias, Y, M, K, N, s); break;
        case 10: launch_linear<10>(X, W, bias, Y, M, K, N, s); break;
        case 11: launch_linear<11>(X, W, bias, Y, M, K, N, s); break;
        default: launch_linear<12>(X, W, bias, Y, M, K, N, s); break;
    }
    return check_launch("linear_forward");
}

extern "C" int gsvc_linear_wgrad(const float *G, const float *X, float *dW, int64_t M, int32_t N, int32_t K, void *stream)
{
    GSVC_REQUIRE(M >= 0 && K > 0 && N > 0, "linear_wgrad: bad shape");
    if (N > LIN_NT_MAX * 16 || K > LIN_NT_MAX * 16) {
        set_error("linear_wgrad: N=%d / K=%d exceed %d", N, K, LIN_NT_MAX * 16);
        return GSVC_E_UNSUPPORTED;
    }
    if (M == 0) return GSVC_OK;
    GSVC_REQUIRE(G && X && dW, "linear_wgrad: NULL pointer");
    hipStream_t s = (hipStream_t)stream;
    const int tiles = ((N + 15) / 16) * ((K + 15) / 16);
    const int tpw = (tiles + 3) / 4;
    if (tpw <= 4) launch_wgrad<4>(G, X, dW, M, N, K, s);
    else if (tpw <= 9) launch_wgrad<9>(G, X, dW, M, N, K, s);
    else if (tpw <= 16) launch_wgrad<16>(G, X, dW, M, N, K, s);
    else if (tpw <= 25) launch_wgrad<25>(G, X, dW, M, N, K, s);
    else launch_wgrad<36>(G, X, dW, M, N, K, s);
    return check_launch("linear_wgrad");
}

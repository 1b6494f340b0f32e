// gsvc_amd/csrc/linear.hip — tall-skinny fp32 linear layer Y[M,N] = X[M,K] W[N,K]^T + b on MFMA, gfx950.
//
// The generator / deformation / entropy-parameter MLPs of GSVC (reference scene/gaussian_model.py:150-232,
// 411-501) are chains of nn.Linear with K, N <= 192 applied to ~50k-200k anchor rows per step.  Those GEMMs are
// tall and skinny (arithmetic intensity 12-48 FLOP/B): between the HBM and the fp32-MFMA roofs, and the library
// GEMM reached only 7-30 TF on them.  v_mfma_f32_16x16x4_f32: exact fp32 products, fp32 accumulate — same
// numerics class as the reference's fp32 GEMM, different summation order.
#include "linear_ws.h"

namespace gsvc {
// the EPI instantiations live in their own translation units (linear_epi_{a,b,c,d}.hip: 1-4, 5-7, 8-10, 11-12 column tiles)
void launch_ws_epi_a(const float *X, const float *W, const float *b, float *Y, long long M, int K, int N, int w_in_out, hipStream_t s, const LinEpi &ep);
void launch_ws_epi_b(const float *X, const float *W, const float *b, float *Y, long long M, int K, int N, int w_in_out, hipStream_t s, const LinEpi &ep);
void launch_ws_epi_c(const float *X, const float *W, const float *b, float *Y, long long M, int K, int N, int w_in_out, hipStream_t s, const LinEpi &ep);
void launch_ws_epi_d(const float *X, const float *W, const float *b, float *Y, long long M, int K, int N, int w_in_out, hipStream_t s, const LinEpi &ep);
}  // namespace gsvc

using namespace gsvc;

extern "C" int gsvc_linear_forward(const float *X, const float *W, const float *bias, float *Y, int64_t M, int32_t K,
                                      int32_t N, int32_t w_in_out, int32_t relu, void *stream)
{
    GSVC_REQUIRE(M >= 0 && K > 0 && N > 0, "linear_forward: bad shape");
    if (N > LIN_NT_MAX * 16 || K > LIN_NT_MAX * 16) {
        set_error("linear_forward: N=%d / K=%d exceed %d", N, K, LIN_NT_MAX * 16);
        return GSVC_E_UNSUPPORTED;
    }
    if (M == 0) return GSVC_OK;
    GSVC_REQUIRE(X && W && Y, "linear_forward: NULL pointer");
    hipStream_t s = (hipStream_t)stream;
    launch_ws_n<false>(X, W, bias, Y, M, K, N, w_in_out, relu, s, LinEpi{0, nullptr, nullptr, nullptr, nullptr});
    return check_launch("linear_forward");
}

extern "C" int gsvc_linear_forward_ex(const float *X, const float *W, const float *bias, float *Y, int64_t M, int32_t K,
                                      int32_t N, int32_t w_in_out, int32_t epilogue, const float *aux1, const float *aux2,
                                      float *Y2, float *Y3, void *stream)
{
    GSVC_REQUIRE(M >= 0 && K > 0 && N > 0, "linear_forward_ex: bad shape");
    GSVC_REQUIRE(epilogue >= GSVC_LIN_NONE && epilogue <= GSVC_LIN_ADD, "linear_forward_ex: unknown epilogue %d", epilogue);
    if (N > LIN_NT_MAX * 16 || K > LIN_NT_MAX * 16) {
        set_error("linear_forward_ex: N=%d / K=%d exceed %d", N, K, LIN_NT_MAX * 16);
        return GSVC_E_UNSUPPORTED;
    }
    if (M == 0) return GSVC_OK;
    GSVC_REQUIRE(X && W && Y, "linear_forward_ex: NULL pointer");
    const bool films = epilogue == GSVC_LIN_FILM || epilogue == GSVC_LIN_FILM_GRAD;
    const bool need1 = epilogue >= GSVC_LIN_MUL_GELU_GRAD, need2 = films;
    const bool out2 = epilogue == GSVC_LIN_GELU_DUAL || films, out3 = epilogue == GSVC_LIN_FILM_GRAD;
    GSVC_REQUIRE((!need1 || aux1) && (!need2 || aux2) && (!out2 || Y2) && (!out3 || Y3),
                 "linear_forward_ex: epilogue %d misses an operand", epilogue);
    hipStream_t s = (hipStream_t)stream;
    if (epilogue == GSVC_LIN_NONE || epilogue == GSVC_LIN_RELU)
        launch_ws_n<false>(X, W, bias, Y, M, K, N, w_in_out, epilogue == GSVC_LIN_RELU, s, LinEpi{0, nullptr, nullptr, nullptr, nullptr});
    else {
        const int nt = (N + 15) / 16;
        (nt <= 4 ? launch_ws_epi_a : (nt <= 7 ? launch_ws_epi_b : (nt <= 10 ? launch_ws_epi_c : launch_ws_epi_d)))(
            X, W, bias, Y, M, K, N, w_in_out, s, LinEpi{epilogue, aux1, aux2, Y2, Y3});
    }
    return check_launch("linear_forward_ex");
}


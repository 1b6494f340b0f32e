// gsvc_amd/csrc/linear_epi_b.hip — instantiations of k_linear_ws with an epilogue program (gsvc_linear_forward_ex) for
// 5 .. 7 column tiles (N in 65 .. 112); see linear_ws.h.
#include "linear_ws.h"

namespace gsvc {
void launch_ws_epi_b(const float *X, const float *W, const float *b, float *Y, long long M, int K, int N, int w_in_out, hipStream_t s,
                     const LinEpi &ep)
{
    launch_ws_n<true, 5, 7>(X, W, b, Y, M, K, N, w_in_out, 0, s, ep);
}
}  // namespace gsvc

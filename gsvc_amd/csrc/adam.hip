// gsvc_amd/csrc/adam.hip — Adam update of all parameter tensors of a step in one launch, gfx950.
//
// GSVC optimises 15 parameter groups (reference scene/gaussian_model.py:1034-1058: torch.optim.Adam, eps 1e-15, no weight
// decay, no amsgrad) — ~26 M floats at 220 k anchors, almost all in the five per-anchor tensors.  The update is a pure
// 28-byte-per-element stream (read p, g, m, v; write p, m, v); one multi-tensor launch walks up to 64 tensors with
// 16-byte accesses, the block -> (tensor, chunk) map is a prefix table passed by value.
//   m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g^2;  p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
#include "common.h"

namespace gsvc {

constexpr int ADAM_MAX_TENSORS = 64;
constexpr int ADAM_CHUNK = 4096;        // elements per workgroup (256 threads x 4 x float4)

struct AdamBatch {
    float *p[ADAM_MAX_TENSORS];
    const float *g[ADAM_MAX_TENSORS];
    float *m[ADAM_MAX_TENSORS];
    float *v[ADAM_MAX_TENSORS];
    long long n[ADAM_MAX_TENSORS];
    float step_size[ADAM_MAX_TENSORS];      // lr / (1 - b1^t)
    float inv_sqrt_bc2[ADAM_MAX_TENSORS];   // 1 / sqrt(1 - b2^t)
    int first_block[ADAM_MAX_TENSORS + 1];
    int count;
    const int32_t *guard[4];                // device words; a non-zero one turns the launch into a no-op (gsvc_adam_step_guarded)
    int n_guards;
};

// omb1 / omb2 = 1 - beta rounded from double (as torch does), not 1.0f - beta: that differs by 5e-5 relative for beta2 = 0.999
__device__ __forceinline__ void adam1(float &p, float g, float &m, float &v, float b1, float b2, float omb1, float omb2, float eps,
                                      float step_size, float inv_sqrt_bc2)
{
    m = b1 * m + omb1 * g;
    v = b2 * v + omb2 * g * g;
    p -= step_size * (m / (sqrtf(v) * inv_sqrt_bc2 + eps));
}

__global__ void __launch_bounds__(256) k_adam(AdamBatch b, float b1, float b2, float omb1, float omb2, float eps)
{
    for (int k = 0; k < b.n_guards; k++)
        if (*b.guard[k] != 0) return;       // uniform over the grid
    int t = 0;
    {   // which tensor does this block belong to (first_block is ascending): binary search over <= 64 entries
        int lo = 0, hi = b.count;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if ((int)blockIdx.x >= b.first_block[mid]) lo = mid; else hi = mid;
        }
        t = lo;
    }
    const long long base = (long long)((int)blockIdx.x - b.first_block[t]) * ADAM_CHUNK;
    const long long n = b.n[t];
    float *p = b.p[t], *m = b.m[t], *v = b.v[t];
    const float *g = b.g[t];
    const float ss = b.step_size[t], ib = b.inv_sqrt_bc2[t];
    const bool vec = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                       reinterpret_cast<uintptr_t>(v)) & 15) == 0;
    // a whole chunk inside the tensor and 16-byte aligned (all but a tensor's last workgroup): the four pieces' 16 loads are
    // issued before the first is used (piece by piece each thread made four dependent memory round trips)
    if (vec && base + ADAM_CHUNK <= n) {
        float4 P[4], M[4], V[4], G[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const long long i = base + (long long)(u * 256 + threadIdx.x) * 4;
            P[u] = *reinterpret_cast<const float4 *>(p + i); M[u] = *reinterpret_cast<const float4 *>(m + i);
            V[u] = *reinterpret_cast<const float4 *>(v + i); G[u] = *reinterpret_cast<const float4 *>(g + i);
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const long long i = base + (long long)(u * 256 + threadIdx.x) * 4;
            adam1(P[u].x, G[u].x, M[u].x, V[u].x, b1, b2, omb1, omb2, eps, ss, ib);
            adam1(P[u].y, G[u].y, M[u].y, V[u].y, b1, b2, omb1, omb2, eps, ss, ib);
            adam1(P[u].z, G[u].z, M[u].z, V[u].z, b1, b2, omb1, omb2, eps, ss, ib);
            adam1(P[u].w, G[u].w, M[u].w, V[u].w, b1, b2, omb1, omb2, eps, ss, ib);
            *reinterpret_cast<float4 *>(p + i) = P[u];
            *reinterpret_cast<float4 *>(m + i) = M[u];
            *reinterpret_cast<float4 *>(v + i) = V[u];
        }
        return;
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const long long i = base + (long long)(u * 256 + threadIdx.x) * 4;
        if (i >= n) break;
        if (vec && i + 4 <= n) {
            float4 P = *reinterpret_cast<float4 *>(p + i), M = *reinterpret_cast<float4 *>(m + i), V = *reinterpret_cast<float4 *>(v + i);
            const float4 G = *reinterpret_cast<const float4 *>(g + i);
            adam1(P.x, G.x, M.x, V.x, b1, b2, omb1, omb2, eps, ss, ib);
            adam1(P.y, G.y, M.y, V.y, b1, b2, omb1, omb2, eps, ss, ib);
            adam1(P.z, G.z, M.z, V.z, b1, b2, omb1, omb2, eps, ss, ib);
            adam1(P.w, G.w, M.w, V.w, b1, b2, omb1, omb2, eps, ss, ib);
            *reinterpret_cast<float4 *>(p + i) = P;
            *reinterpret_cast<float4 *>(m + i) = M;
            *reinterpret_cast<float4 *>(v + i) = V;
        } else {
            for (long long k = i; k < n && k < i + 4; k++) {
                float P = p[k], M = m[k], V = v[k];
                adam1(P, g[k], M, V, b1, b2, omb1, omb2, eps, ss, ib);
                p[k] = P; m[k] = M; v[k] = V;
            }
        }
    }
}

}  // namespace gsvc

using namespace gsvc;

static int adam_step_impl(int32_t n_tensors, const gsvc_adam_tensor *tensors_host, double beta1_d, double beta2_d, double eps_d,
                          const int32_t *const *guards, int32_t n_guards, void *stream)
{
    const float beta1 = (float)beta1_d, beta2 = (float)beta2_d, eps = (float)eps_d;
    GSVC_REQUIRE(n_tensors >= 0 && (n_tensors == 0 || tensors_host), "adam_step: bad arguments");
    GSVC_REQUIRE(n_guards >= 0 && n_guards <= 4 && (n_guards == 0 || guards), "adam_step: at most 4 guard words");
    hipStream_t s = (hipStream_t)stream;
    int done = 0;
    while (done < n_tensors) {
        AdamBatch b;
        b.count = 0;
        b.n_guards = n_guards;
        for (int k = 0; k < 4; k++) b.guard[k] = k < n_guards ? guards[k] : nullptr;
        for (int k = 0; k < n_guards; k++) GSVC_REQUIRE(guards[k], "adam_step: NULL guard word");
        int blocks = 0;
        while (done < n_tensors && b.count < ADAM_MAX_TENSORS) {
            const gsvc_adam_tensor &t = tensors_host[done++];
            if (t.n <= 0) continue;
            GSVC_REQUIRE(t.param && t.grad && t.exp_avg && t.exp_avg_sq, "adam_step: NULL pointer");
            GSVC_REQUIRE(t.bias_correction1 > 0.f && t.bias_correction2 > 0.f, "adam_step: bias corrections must be positive");
            const int k = b.count++;
            b.p[k] = t.param; b.g[k] = t.grad; b.m[k] = t.exp_avg; b.v[k] = t.exp_avg_sq; b.n[k] = t.n;
            b.step_size[k] = t.lr / t.bias_correction1;
            b.inv_sqrt_bc2[k] = 1.0f / sqrtf(t.bias_correction2);
            b.first_block[k] = blocks;
            blocks += (int)((t.n + ADAM_CHUNK - 1) / ADAM_CHUNK);
        }
        if (b.count == 0) break;
        b.first_block[b.count] = blocks;
        ProfScope _p("k_adam", s);
        hipLaunchKernelGGL(k_adam, dim3(blocks), dim3(256), 0, s, b, beta1, beta2, (float)(1.0 - (double)beta1_d), (float)(1.0 - (double)beta2_d), eps);
    }
    return check_launch("adam_step");
}

extern "C" int gsvc_adam_step(int32_t n_tensors, const gsvc_adam_tensor *tensors_host, double beta1_d, double beta2_d, double eps_d,
                              void *stream)
{
    return adam_step_impl(n_tensors, tensors_host, beta1_d, beta2_d, eps_d, nullptr, 0, stream);
}

extern "C" int gsvc_adam_step_guarded(int32_t n_tensors, const gsvc_adam_tensor *tensors_host, double beta1_d, double beta2_d,
                                      double eps_d, const int32_t *const *guards_host, int32_t n_guards, void *stream)
{
    return adam_step_impl(n_tensors, tensors_host, beta1_d, beta2_d, eps_d, guards_host, n_guards, stream);
}

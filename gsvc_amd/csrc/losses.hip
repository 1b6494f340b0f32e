// gsvc_amd/csrc/losses.hip — the per-Gaussian loss terms of the fitting step as fused kernels, gfx950.
//
//  * optical-flow consistency (reference utils/loss_utils.py:76-153, calc_optical_loss_one_frame): Gaussians alive
//    in both renders of an adjacent-frame pair (same anchor, same offset slot) should move by the optical flow
//    sampled at the frame-t pixel.  The reference builds two A*K boolean tables, intersects them, compacts both
//    renders by boolean masks (host synchronisations) and runs ~45 elementwise launches; here:
//      k_optical_mark : table[slot] = row+1 for the Gaussians of render t+1 with opacity > 0
//      k_optical_fwd  : per Gaussian of render t: partner lookup, pixel, bounds test, |d - uv| summed over the wave,
//                       two atomics per wave; remembers (partner row, sign x, sign y) for the backward
//      k_optical_bwd  : +-sign * g / (2 n) into both renders' position gradients (the pixel lookup is piecewise
//                       constant: no gradient through uv, as in the reference where it goes through .round().long())
//  * scaling / opacity regularisers of the 4 renders of a step (reference pipeline/train.py:417-424):
//      sum_r mean_{i in r, opacity_i > 0}(prod_c scaling_ic)   and   sum_r mean_{i in r}(1 - neural_opacity_i)
//    over un-compacted per-render segments: one reduction launch + a one-thread finalize, one backward launch.
#include "common.h"

namespace gsvc {

constexpr int MAX_RENDERS = 8;

struct SegOffsets {
    int64_t off[MAX_RENDERS + 1];   // Gaussian index where render r starts
    int R;
};

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// sum of v over the 256 threads of a block, returned to thread 0 (red: 4 floats of LDS per value in flight)
__device__ __forceinline__ float block_sum256(float v, float *red)
{
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// out[c] = sum over `count` partial rows of part[row*stride + c], c < ncol; one block of 256 threads (no atomics:
// thousands of atomics onto the same two or three addresses serialise)
template <int NCOL>
__device__ __forceinline__ void reduce_partials(const float *__restrict__ part, int count, float *__restrict__ out, float *red)
{
    float acc[NCOL];
#pragma unroll
    for (int c = 0; c < NCOL; c++) acc[c] = 0.f;
    for (int b = threadIdx.x; b < count; b += 256)
#pragma unroll
        for (int c = 0; c < NCOL; c++) acc[c] += part[(size_t)b * NCOL + c];
#pragma unroll
    for (int c = 0; c < NCOL; c++) {
        const float t = block_sum256(acc[c], red);
        if (threadIdx.x == 0) out[c] = t;
    }
}

// ------------------------------------------------------------------------------------------------ optical flow
__global__ void __launch_bounds__(256) k_optical_mark(const uint8_t *__restrict__ mask2, const int64_t *__restrict__ vis2,
                                                      int64_t n2, int K, int32_t *__restrict__ table)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n2 || !mask2[i]) return;
    const int64_t row = i / K;
    table[vis2[row] * K + (i - row * K)] = (int32_t)(i + 1);
}

__global__ void __launch_bounds__(256) k_optical_fwd(const float *__restrict__ world1, const uint8_t *__restrict__ mask1,
                                                     const int64_t *__restrict__ vis1, int64_t n1,
                                                     const float *__restrict__ world2, const int32_t *__restrict__ table,
                                                     const float *__restrict__ flow, int K, float x_min, float y_min, float scale,
                                                     int x_pix_max, int y_pix_max, int flow_h, int flow_w,
                                                     int32_t *__restrict__ partner, float *__restrict__ part)
{
    __shared__ float red[4];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    float err = 0.f, cnt = 0.f;
    int32_t code = 0;
    if (i < n1 && mask1[i]) {
        const int64_t row = i / K;
        const int32_t r = table[vis1[row] * K + (i - row * K)];
        if (r) {
            const float x1 = world1[3 * i], y1 = world1[3 * i + 1];
            // torch.round (half to even) of (xy - min) * scale, as the reference computes the pixel
            const float fx = rintf((x1 - x_min) * scale), fy = rintf((y1 - y_min) * scale);
            if (fx >= 0.f && fy >= 0.f && fx < (float)x_pix_max && fy < (float)y_pix_max) {
                const int px = (int)fx, py = (int)fy;
                // optical_flow[c, h, w] indexed [.., py, px]  (reference: flow.permute(2, 1, 0)[px, py])
                const float u = flow[(size_t)py * flow_w + px] / scale;
                const float v = flow[(size_t)flow_h * flow_w + (size_t)py * flow_w + px] / scale;
                const float x2 = world2[3 * (int64_t)(r - 1)], y2 = world2[3 * (int64_t)(r - 1) + 1];
                const float ex = (x2 - x1) - u, ey = (y2 - y1) - v;
                err = fabsf(ex) + fabsf(ey);
                cnt = 1.f;
                const int sx = (ex > 0.f) - (ex < 0.f), sy = (ey > 0.f) - (ey < 0.f);
                code = (r << 4) | ((sx + 1) << 2) | (sy + 1);      // r < 2^27 checked on the host
            }
        }
    }
    if (i < n1) partner[i] = code;
    err = block_sum256(err, red);
    cnt = block_sum256(cnt, red);
    if (threadIdx.x == 0) {
        part[2 * blockIdx.x] = err;
        part[2 * blockIdx.x + 1] = cnt;
    }
}

__global__ void __launch_bounds__(256) k_optical_finalize(const float *__restrict__ part, int blocks, float *__restrict__ sums)
{
    __shared__ float red[4];
    reduce_partials<2>(part, blocks, sums, red);
}

// grad_out = dL/d(loss) (device scalar); loss = sums[0] / (2 sums[1])
__global__ void __launch_bounds__(256) k_optical_bwd(const int32_t *__restrict__ partner, int64_t n1, const float *__restrict__ sums,
                                                     const float *__restrict__ grad_out, float *__restrict__ g1,
                                                     float *__restrict__ g2)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n1) return;
    const int32_t code = partner[i];
    if (!code) return;
    const float k = grad_out[0] / (2.0f * sums[1]);
    const float sx = (float)(((code >> 2) & 3) - 1) * k, sy = (float)((code & 3) - 1) * k;
    const int64_t r = (code >> 4) - 1;
    g1[3 * i] = -sx;
    g1[3 * i + 1] = -sy;
    g2[3 * r] = sx;          // a partner row is referenced by exactly one Gaussian of render 1 (same slot)
    g2[3 * r + 1] = sy;
}

// The same loss for several frame pairs whose renders are row ranges of ONE set of tensors (a fitting step: renders f1, b1, f2,
// b2 concatenated; pairs f1 -> f2 and b1 -> b2) — four launches forward and one backward for all pairs, no per-Gaussian table,
// no zero-filled gradients, the gradient written straight in the concatenated layout:
//   k_optical_rows      : table[r][anchor] = row + 1 of the anchor in render r (the pairing is by anchor and offset slot)
//   k_optical_fwd_many  : grid.y = pair; a source Gaussian looks its partner up through the destination render's table
//   k_optical_fin_many  : per-pair sums, out[0] = sum_p sums[p][0] / (2 sums[p][1])
//   k_optical_bwd_many  : every Gaussian of every render writes its own gradient (a destination Gaussian finds the source
//                         through the source render's table and reads that one's sign code)
struct OpticalBatch {
    int64_t goff[MAX_RENDERS + 1];      // Gaussian offsets of the renders
    int R, P;
    int src[MAX_RENDERS / 2], dst[MAX_RENDERS / 2];
    int pair_of[MAX_RENDERS];           // render -> pair (or -1), and whether it is the pair's source
    int is_src[MAX_RENDERS];
};

__global__ void __launch_bounds__(256) k_optical_rows(const int64_t *__restrict__ vis, OpticalBatch ob, int K, int64_t A,
                                                      int32_t *__restrict__ table)
{
    const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (row * K >= ob.goff[ob.R]) return;
    int r = 0;
    while (r + 1 < ob.R && row * K >= ob.goff[r + 1]) r++;
    table[(int64_t)r * A + vis[row]] = (int32_t)(row - ob.goff[r] / K) + 1;
}

__global__ void __launch_bounds__(256) k_optical_fwd_many(const float *__restrict__ world, const uint8_t *__restrict__ mask,
                                                          const int64_t *__restrict__ vis, OpticalBatch ob, int K, int64_t A,
                                                          const int32_t *__restrict__ table, const float *__restrict__ flow, float x_min,
                                                          float y_min, float scale, int x_pix_max, int y_pix_max, int flow_h, int flow_w,
                                                          int32_t *__restrict__ partner, float *__restrict__ part)
{
    __shared__ float red[4];
    const int p = blockIdx.y, rs = ob.src[p], rd = ob.dst[p];
    const int64_t i = ob.goff[rs] + (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool in = i < ob.goff[rs + 1];
    float err = 0.f, cnt = 0.f;
    int32_t code = 0;
    if (in && mask[i]) {
        const int64_t row = i / K;
        const int k = (int)(i - row * K);
        const int32_t r2 = table[(int64_t)rd * A + vis[row]];
        const int64_t j = ob.goff[rd] + (int64_t)(r2 - 1) * K + k;
        if (r2 && mask[j]) {
            const float x1 = world[3 * i], y1 = world[3 * i + 1];
            const float fx = rintf((x1 - x_min) * scale), fy = rintf((y1 - y_min) * scale);
            if (fx >= 0.f && fy >= 0.f && fx < (float)x_pix_max && fy < (float)y_pix_max) {
                const int px = (int)fx, py = (int)fy;
                const float u = flow[(size_t)py * flow_w + px] / scale;
                const float v = flow[(size_t)flow_h * flow_w + (size_t)py * flow_w + px] / scale;
                const float ex = (world[3 * j] - x1) - u, ey = (world[3 * j + 1] - y1) - v;
                err = fabsf(ex) + fabsf(ey);
                cnt = 1.f;
                const int sx = (ex > 0.f) - (ex < 0.f), sy = (ey > 0.f) - (ey < 0.f);
                code = 16 | ((sx + 1) << 2) | (sy + 1);
            }
        }
    }
    if (in) partner[i] = code;
    err = block_sum256(err, red);
    cnt = block_sum256(cnt, red);
    if (threadIdx.x == 0) {
        float *d = part + 2 * ((size_t)p * gridDim.x + blockIdx.x);
        d[0] = err;
        d[1] = cnt;
    }
}

__global__ void __launch_bounds__(256) k_optical_fin_many(const float *__restrict__ part, int blocks, int P, float *__restrict__ sums,
                                                          float *__restrict__ out)
{
    __shared__ float red[4];
    for (int p = 0; p < P; p++) reduce_partials<2>(part + 2 * (size_t)p * blocks, blocks, sums + 2 * p, red);
    __syncthreads();
    if (threadIdx.x == 0) {
        float a = 0.f;
        for (int p = 0; p < P; p++) a += sums[2 * p] / (2.0f * sums[2 * p + 1]);      // 0 / 0 = nan: the mean over no pairs
        out[0] = a;
    }
}

__global__ void __launch_bounds__(256) k_optical_bwd_many(const int32_t *__restrict__ partner, const int64_t *__restrict__ vis,
                                                          OpticalBatch ob, int K, int64_t A, const int32_t *__restrict__ table,
                                                          const float *__restrict__ sums, const float *__restrict__ grad_out,
                                                          float *__restrict__ g)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= ob.goff[ob.R]) return;
    int r = 0;
    while (r + 1 < ob.R && i >= ob.goff[r + 1]) r++;
    float gx = 0.f, gy = 0.f;
    const int p = ob.pair_of[r];
    if (p >= 0) {
        int32_t code = 0;
        float sign = -1.f;
        if (ob.is_src[r]) {
            code = partner[i];
        } else {
            const int64_t row = i / K;
            const int32_t r1 = table[(int64_t)ob.src[p] * A + vis[row]];
            if (r1) code = partner[ob.goff[ob.src[p]] + (int64_t)(r1 - 1) * K + (i - row * K)];
            sign = 1.f;
        }
        if (code) {
            const float k = sign * grad_out[0] / (2.0f * sums[2 * p + 1]);
            gx = (float)(((code >> 2) & 3) - 1) * k;
            gy = (float)((code & 3) - 1) * k;
        }
    }
    g[3 * i] = gx;
    g[3 * i + 1] = gy;
    g[3 * i + 2] = 0.f;
}

// ------------------------------------------------------------------------------------------------ regularisers
// part[(r*nbx + bx)*3 + {0,1,2}] = block sums of (m_i prod_c s_ic, m_i, 1 - o_i) over render r's Gaussians
__global__ void __launch_bounds__(256) k_regs_fwd(const float *__restrict__ scaling, const float *__restrict__ opacity,
                                                  const uint8_t *__restrict__ mask, SegOffsets seg, float *__restrict__ part)
{
    __shared__ float red[4];
    const int r = blockIdx.y;
    const int64_t lo = seg.off[r], hi = seg.off[r + 1];
    float p = 0.f, m = 0.f, q = 0.f;
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const int64_t i = lo + (int64_t)blockIdx.x * 1024 + u * 256 + threadIdx.x;
        if (i < hi) {
            const float mi = mask[i] ? 1.f : 0.f;
            m += mi;
            p += mi * scaling[3 * i] * scaling[3 * i + 1] * scaling[3 * i + 2];
            q += 1.f - opacity[i];
        }
    }
    p = block_sum256(p, red);
    m = block_sum256(m, red);
    q = block_sum256(q, red);
    if (threadIdx.x == 0) {
        float *d = part + ((size_t)r * gridDim.x + blockIdx.x) * 3;
        d[0] = p; d[1] = m; d[2] = q;
    }
}

// sums[3r..] = per-render totals; out[0] = sum_r P_r / C_r, out[1] = sum_r Q_r / n_r
__global__ void __launch_bounds__(256) k_regs_finalize(const float *__restrict__ part, int nbx, SegOffsets seg, float *__restrict__ sums,
                                                       float *__restrict__ out)
{
    __shared__ float red[4];
    for (int r = 0; r < seg.R; r++) reduce_partials<3>(part + (size_t)r * nbx * 3, nbx, sums + 3 * r, red);
    __syncthreads();
    if (threadIdx.x == 0) {
        float a = 0.f, b = 0.f;
        for (int r = 0; r < seg.R; r++) {
            a += sums[3 * r] / sums[3 * r + 1];                     // 0/0 = nan, like the mean of an empty selection
            b += sums[3 * r + 2] / (float)(seg.off[r + 1] - seg.off[r]);
        }
        out[0] = a;
        out[1] = b;
    }
}

__global__ void __launch_bounds__(256) k_regs_bwd(const float *__restrict__ scaling, const uint8_t *__restrict__ mask, SegOffsets seg,
                                                  const float *__restrict__ sums, const float *__restrict__ grad_out,
                                                  float *__restrict__ g_scaling, float *__restrict__ g_opacity)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= seg.off[seg.R]) return;
    int r = 0;
    while (r + 1 < seg.R && i >= seg.off[r + 1]) r++;
    const float s0 = scaling[3 * i], s1 = scaling[3 * i + 1], s2 = scaling[3 * i + 2];
    const float k = mask[i] ? grad_out[0] / sums[3 * r + 1] : 0.f;
    g_scaling[3 * i] = k * s1 * s2;
    g_scaling[3 * i + 1] = k * s0 * s2;
    g_scaling[3 * i + 2] = k * s0 * s1;
    g_opacity[i] = -grad_out[1] / (float)(seg.off[r + 1] - seg.off[r]);
}

static bool fill_seg(const int64_t *seg_offsets_host, int R, SegOffsets &seg)
{
    if (R < 1 || R > MAX_RENDERS) return false;
    seg.R = R;
    for (int r = 0; r <= R; r++) seg.off[r] = seg_offsets_host[r];
    for (int r = R + 1; r <= MAX_RENDERS; r++) seg.off[r] = seg_offsets_host[R];
    return true;
}

}  // namespace gsvc

using namespace gsvc;

extern "C" int gsvc_optical_forward(const float *world1, const uint8_t *mask1, const int64_t *vis1, int64_t n1,
                                    const float *world2, const uint8_t *mask2, const int64_t *vis2, int64_t n2, int32_t K,
                                    int64_t anchors, const float *flow, int32_t flow_h, int32_t flow_w, float x_min, float y_min,
                                    float scale, int32_t x_pix_max, int32_t y_pix_max, int32_t *table, int32_t *partner,
                                    float *sums, float *partial, void *stream)
{
    GSVC_REQUIRE(K > 0 && n1 >= 0 && n2 >= 0 && n1 % K == 0 && n2 % K == 0, "optical_forward: Gaussian counts must be multiples of K");
    GSVC_REQUIRE(n2 < (1 << 27), "optical_forward: more than 2^27 Gaussians per render");
    GSVC_REQUIRE(x_pix_max <= flow_w && y_pix_max <= flow_h, "optical_forward: pixel bounds exceed the flow field");
    GSVC_REQUIRE(table && sums && partial && (n1 == 0 || (world1 && mask1 && vis1 && partner)) && (n2 == 0 || (world2 && mask2 && vis2)) && flow,
                 "optical_forward: NULL pointer");
    hipStream_t s = (hipStream_t)stream;
    (void)hipMemsetAsync(table, 0, sizeof(int32_t) * (size_t)anchors * K, s);
    (void)hipMemsetAsync(sums, 0, 2 * sizeof(float), s);      // stays (0, 0) when render 1 is empty
    if (n2) {
        ProfScope _p("k_optical_mark", s);
        hipLaunchKernelGGL(k_optical_mark, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, s, mask2, vis2, n2, K, table);
    }
    if (n1) {
        ProfScope _p("k_optical_fwd", s);
        const int blocks = (int)((n1 + 255) / 256);
        hipLaunchKernelGGL(k_optical_fwd, dim3(blocks), dim3(256), 0, s, world1, mask1, vis1, n1, world2, table, flow, K, x_min, y_min,
                           scale, x_pix_max, y_pix_max, flow_h, flow_w, partner, partial);
        hipLaunchKernelGGL(k_optical_finalize, dim3(1), dim3(256), 0, s, partial, blocks, sums);
    }
    return check_launch("optical_forward");
}

extern "C" int gsvc_optical_backward(const int32_t *partner, int64_t n1, int64_t n2, const float *sums, const float *grad_out,
                                     float *grad_world1, float *grad_world2, void *stream)
{
    GSVC_REQUIRE(sums && grad_out && (n1 == 0 || (partner && grad_world1)) && (n2 == 0 || grad_world2), "optical_backward: NULL pointer");
    hipStream_t s = (hipStream_t)stream;
    if (n1) (void)hipMemsetAsync(grad_world1, 0, sizeof(float) * 3 * (size_t)n1, s);
    if (n2) (void)hipMemsetAsync(grad_world2, 0, sizeof(float) * 3 * (size_t)n2, s);
    if (n1) {
        ProfScope _p("k_optical_bwd", s);
        hipLaunchKernelGGL(k_optical_bwd, dim3((unsigned)((n1 + 255) / 256)), dim3(256), 0, s, partner, n1, sums, grad_out, grad_world1,
                           grad_world2);
    }
    return check_launch("optical_backward");
}

extern "C" int64_t gsvc_regs_partial_floats(const int64_t *seg_offsets_host, int32_t R)
{
    int64_t longest = 0;
    for (int r = 0; r < R; r++) longest = seg_offsets_host[r + 1] - seg_offsets_host[r] > longest ? seg_offsets_host[r + 1] - seg_offsets_host[r] : longest;
    const int64_t nbx = (longest + 1023) / 1024 > 0 ? (longest + 1023) / 1024 : 1;
    return 3 * (int64_t)R * nbx;
}

extern "C" int gsvc_regs_forward(const float *scaling, const float *neural_opacity, const uint8_t *mask,
                                 const int64_t *seg_offsets_host, int32_t R, float *sums, float *partial, float *out, void *stream)
{
    SegOffsets seg;
    GSVC_REQUIRE(seg_offsets_host && fill_seg(seg_offsets_host, R, seg), "regs_forward: 1..8 renders");
    GSVC_REQUIRE(sums && out && partial && (seg.off[R] == 0 || (scaling && neural_opacity && mask)), "regs_forward: NULL pointer");
    hipStream_t s = (hipStream_t)stream;
    const int nbx = (int)(gsvc_regs_partial_floats(seg_offsets_host, R) / (3 * R));
    {
        ProfScope _p("k_regs_fwd", s);
        hipLaunchKernelGGL(k_regs_fwd, dim3(nbx, R), dim3(256), 0, s, scaling, neural_opacity, mask, seg, partial);
    }
    hipLaunchKernelGGL(k_regs_finalize, dim3(1), dim3(256), 0, s, partial, nbx, seg, sums, out);
    return check_launch("regs_forward");
}

extern "C" int gsvc_regs_backward(const float *scaling, const uint8_t *mask, const int64_t *seg_offsets_host, int32_t R,
                                  const float *sums, const float *grad_out, float *grad_scaling, float *grad_opacity, void *stream)
{
    SegOffsets seg;
    GSVC_REQUIRE(seg_offsets_host && fill_seg(seg_offsets_host, R, seg), "regs_backward: 1..8 renders");
    const int64_t n = seg.off[R];
    GSVC_REQUIRE(sums && grad_out && (n == 0 || (scaling && mask && grad_scaling && grad_opacity)), "regs_backward: NULL pointer");
    hipStream_t s = (hipStream_t)stream;
    if (n) {
        ProfScope _p("k_regs_bwd", s);
        hipLaunchKernelGGL(k_regs_bwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, scaling, mask, seg, sums, grad_out, grad_scaling,
                           grad_opacity);
    }
    return check_launch("regs_backward");
}

static bool fill_optical(const int64_t *gaussian_offsets_host, int R, const int32_t *pair_src_host, const int32_t *pair_dst_host, int P, int K,
                         OpticalBatch &ob)
{
    if (!gaussian_offsets_host || !pair_src_host || !pair_dst_host || R < 2 || R > MAX_RENDERS || P < 1 || P > MAX_RENDERS / 2 || K < 1)
        return false;
    ob.R = R; ob.P = P;
    for (int r = 0; r <= MAX_RENDERS; r++) ob.goff[r] = gaussian_offsets_host[r <= R ? r : R];
    for (int r = 0; r < R; r++)
        if (ob.goff[r + 1] < ob.goff[r] || ob.goff[r] % K) return false;
    if (ob.goff[R] % K || ob.goff[R] >= ((int64_t)1 << 31)) return false;
    for (int r = 0; r < MAX_RENDERS; r++) { ob.pair_of[r] = -1; ob.is_src[r] = 0; }
    for (int p = 0; p < MAX_RENDERS / 2; p++) {
        const int q = p < P ? p : 0;
        ob.src[p] = pair_src_host[q]; ob.dst[p] = pair_dst_host[q];
        if (p >= P) continue;
        if (ob.src[p] < 0 || ob.src[p] >= R || ob.dst[p] < 0 || ob.dst[p] >= R || ob.src[p] == ob.dst[p]) return false;
        if (ob.pair_of[ob.src[p]] >= 0 || ob.pair_of[ob.dst[p]] >= 0) return false;      // a render belongs to one pair
        ob.pair_of[ob.src[p]] = p; ob.is_src[ob.src[p]] = 1;
        ob.pair_of[ob.dst[p]] = p;
    }
    return true;
}

static int optical_blocks(const OpticalBatch &ob)
{
    int64_t longest = 1;
    for (int p = 0; p < ob.P; p++) {
        const int64_t n = ob.goff[ob.src[p] + 1] - ob.goff[ob.src[p]];
        longest = n > longest ? n : longest;
    }
    return (int)((longest + 255) / 256);
}

extern "C" int64_t gsvc_optical_many_partial_floats(const int64_t *gaussian_offsets_host, int32_t R, const int32_t *pair_src_host,
                                                    const int32_t *pair_dst_host, int32_t P, int32_t K)
{
    OpticalBatch ob;
    if (!fill_optical(gaussian_offsets_host, R, pair_src_host, pair_dst_host, P, K, ob)) return -1;
    return 2 * (int64_t)P * optical_blocks(ob);
}

extern "C" int gsvc_optical_many_forward(const float *world, const uint8_t *mask, const int64_t *vis, const int64_t *gaussian_offsets_host,
                                         int32_t R, const int32_t *pair_src_host, const int32_t *pair_dst_host, int32_t P, int32_t K,
                                         int64_t anchors, const float *flow, int32_t flow_h, int32_t flow_w, float x_min, float y_min,
                                         float scale, int32_t x_pix_max, int32_t y_pix_max, int32_t *table, int32_t *partner, float *sums,
                                         float *partial, float *loss, void *stream)
{
    OpticalBatch ob;
    GSVC_REQUIRE(fill_optical(gaussian_offsets_host, R, pair_src_host, pair_dst_host, P, K, ob) && anchors > 0,
                 "optical_many_forward: 2..8 renders of whole rows, 1..4 disjoint pairs, fewer than 2^31 Gaussians");
    GSVC_REQUIRE(x_pix_max <= flow_w && y_pix_max <= flow_h, "optical_many_forward: pixel bounds exceed the flow field");
    GSVC_REQUIRE(table && partner && sums && partial && loss && flow && (ob.goff[R] == 0 || (world && mask && vis)),
                 "optical_many_forward: NULL pointer");
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(table, 0, sizeof(int32_t) * (size_t)R * anchors, s) != hipSuccess) {
        set_error("optical_many_forward: hipMemsetAsync failed");
        return GSVC_E_LAUNCH;
    }
    const int64_t rows = ob.goff[R] / K;
    const int blocks = optical_blocks(ob);
    ProfScope _p("k_optical_fwd", s);
    if (rows) hipLaunchKernelGGL(k_optical_rows, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s, vis, ob, (int)K, anchors, table);
    hipLaunchKernelGGL(k_optical_fwd_many, dim3(blocks, P), dim3(256), 0, s, world, mask, vis, ob, (int)K, anchors, table, flow, x_min, y_min,
                       scale, x_pix_max, y_pix_max, flow_h, flow_w, partner, partial);
    hipLaunchKernelGGL(k_optical_fin_many, dim3(1), dim3(256), 0, s, partial, blocks, (int)P, sums, loss);
    return check_launch("optical_many_forward");
}

extern "C" int gsvc_optical_many_backward(const int32_t *partner, const int64_t *vis, const int64_t *gaussian_offsets_host, int32_t R,
                                          const int32_t *pair_src_host, const int32_t *pair_dst_host, int32_t P, int32_t K, int64_t anchors,
                                          const int32_t *table, const float *sums, const float *grad_out, float *grad_world, void *stream)
{
    OpticalBatch ob;
    GSVC_REQUIRE(fill_optical(gaussian_offsets_host, R, pair_src_host, pair_dst_host, P, K, ob) && anchors > 0, "optical_many_backward: bad arguments");
    if (ob.goff[R] == 0) return GSVC_OK;
    GSVC_REQUIRE(partner && vis && table && sums && grad_out && grad_world, "optical_many_backward: NULL pointer");
    hipStream_t s = (hipStream_t)stream;
    ProfScope _p("k_optical_bwd", s);
    hipLaunchKernelGGL(k_optical_bwd_many, dim3((unsigned)((ob.goff[R] + 255) / 256)), dim3(256), 0, s, partner, vis, ob, (int)K, anchors, table,
                       sums, grad_out, grad_world);
    return check_launch("optical_many_backward");
}

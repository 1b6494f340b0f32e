// gsvc_amd/csrc/mlp_chain.hip — the generator and deformation MLPs of GSVC as whole-network chain kernels, gfx950.
//
// Reference: scene/gaussian_model.py:150-196 (FiLM, GeneratorNet), :468-489 (mlp_deform), evaluated per anchor row at
// ortho_gaussian_renderer/guassian.py:225-273.  Layer by layer (csrc/linear_ws.h) every activation of these networks crosses
// HBM three times (forward, dX, dW) in 64-byte pieces and every layer is its own launch.  Here a wave keeps a 16-row block's
// activations in REGISTERS from the network's input to its output:
//
//   * MFMAs (v_mfma_f32_16x16x4_f32, exact fp32) are issued with the WEIGHT fragment as the A operand, so lane (fr, kq) of the
//     accumulator tile t holds Y[row fr][16 t + 4 kq .. + 3] — which is exactly the B fragment (k-group t) the next layer's
//     MFMAs want.  Bias, GELU / ReLU / FiLM / tanh / sigmoid and their derivatives are applied to the accumulators in place;
//     nothing is shuffled between layers.
//   * the weights of a kernel's layers sit zero-padded in LDS for the life of a persistent workgroup (one per CU), row stride
//     = 16 ceil(K / 16) + 8 floats (== 8 mod 16: the ds_read_b128 of a B fragment is conflict-free).  A generator's seven
//     matrices do not fit 160 KiB at once, so it is two kernels each way (FiLM conditioning nets | trunk); mlp_deform likewise
//     (layers 1-2 | 3-5).  gamma / beta and the deformation's second activation are the only tensors that cross HBM between
//     kernels — and the backward needs them anyway.
//   * the forward leaves behind exactly what the backward and the weight gradients read (pre-activations, activations); the
//     backward chain writes the gradient operand of every layer's dW = G^T X, which the row-split kernels of linear_wgrad.hip
//     then form (one batched slot reduce per network).
//   * the last k-group of a layer issues only the MFMA steps that carry a valid k (K = 66, 50: 2 of 4).
//
// Instantiated for the production widths (feat 50, condition 66, hidden 100, K = 10 offsets: outputs 10 / 30 / 70 / 30);
// other widths return GSVC_E_UNSUPPORTED and the caller keeps the layer-by-layer path.
#include <cstdlib>
#include <functional>
#include <mutex>
#include <set>
#include <vector>

#include "linear_ws.h"

namespace gsvc {
namespace {

constexpr int cl_kg(int K) { return (K + 15) / 16; }
constexpr int cl_ld(int K) { return cl_kg(K) * 16 + 8; }
constexpr int cl_steps(int K) { return K % 16 == 0 ? 4 : (K % 16 >= 4 ? 4 : K % 16); }      // MFMA steps of the last k-group
constexpr int cl_sv(int N) { return N % 4 == 0 ? 4 : (N % 2 == 0 ? 2 : 1); }

struct Lane {
    int tid, lane, wave, fr, kq;
    __device__ Lane()
    {
        tid = threadIdx.x;
        lane = tid & 63;
        wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        fr = lane & 15;
        kq = lane >> 4;
    }
};

// ---- LDS images ---------------------------------------------------------------------------------------------------------
// Copies the [n_cnt][k_cnt] block of a row-major matrix (row stride ldw, first column k_lo) into an image with row stride LD:
// plain:       img[n][col_off + k]      (forward: contraction over k, output n)
// transposed:  img[col_off + k][n]      (dX = G W: contraction over n, output k)
// Loads are 8-byte pieces along W's rows (every width here is even, every base 16-byte aligned), a batch of them in flight
// per thread before the first LDS write: one memory round trip per batch (an element-by-element loop made ~60 dependent ones:
// 30 us of prologue per launch).
template <bool TR>
__device__ __forceinline__ void stage_block(const float *__restrict__ W, int ldw, int n_cnt, int k_lo, int k_cnt, float *img, int LD,
                                            int col_off, int tid, int threads)
{
    constexpr int BATCH = 8;
    const int kp = k_cnt >> 1, total = n_cnt * kp;
    for (int base = tid; base < total; base += threads * BATCH) {
        float2 v[BATCH];
        int dst[BATCH];
#pragma unroll
        for (int u = 0; u < BATCH; u++) {
            const int i = base + u * threads;
            dst[u] = -1;
            if (i < total) {
                const int n = i / kp, k = 2 * (i - n * kp);
                v[u] = *reinterpret_cast<const float2 *>(W + (size_t)n * ldw + k_lo + k);
                dst[u] = TR ? (col_off + k) * LD + n : n * LD + col_off + k;
            }
        }
#pragma unroll
        for (int u = 0; u < BATCH; u++) {
            if (dst[u] < 0) continue;
            if (TR) {
                img[dst[u]] = v[u].x;
                img[dst[u] + LD] = v[u].y;
            } else {
                *reinterpret_cast<float2 *>(img + dst[u]) = v[u];
            }
        }
    }
}

__device__ __forceinline__ void stage_bias(const float *__restrict__ b, int n, float *sb, int padded, int tid, int threads)
{
    for (int i = tid; i < padded; i += threads) sb[i] = (b && i < n) ? b[i] : 0.f;
}

__device__ __forceinline__ void zero_lds(float *lds, int floats, int tid, int threads)
{
    float4 *p = reinterpret_cast<float4 *>(lds);      // every image size here is a multiple of 4 floats
    for (int i = tid; i < floats / 4; i += threads) p[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// ---- fragments ----------------------------------------------------------------------------------------------------------
// a[g] of lane (fr, kq) = X[row fr][16 g + 4 kq .. + 3] of the 16-row block rb (zeros past M and past K)
template <int K>
__device__ __forceinline__ void load_frags(v4f (&a)[cl_kg(K)], const float *X, long long rb, long long RB, long long M, const Lane &L)
{
    constexpr int KG = cl_kg(K), VEC = cl_sv(K);
    const __amdgpu_buffer_rsrc_t rx = ws_block_rsrc(X, rb, RB, M, K);
    const int voff = L.fr * K * 4 + 16 * L.kq;
#pragma unroll
    for (int g = 0; g < KG; g++) {
        const float4 v = ws_load_group<VEC, KG>(rx, voff, L.kq, K, g);
        a[g] = (v4f){v.x, v.y, v.z, v.w};
    }
}

// tile-layout access to a row-major [M][N] matrix: lane (fr, kq), tile t <-> row fr, columns 16 t + 4 kq .. + 3
template <int N>
struct Tiles {
    __amdgpu_buffer_rsrc_t r;
    int yoff, c00;
    __device__ __forceinline__ Tiles(const float *base, long long rb, long long RB, long long M, const Lane &L)
    {
        r = ws_block_rsrc(base, rb, RB, M, N);
        c00 = 4 * L.kq;
        yoff = (L.fr * N + c00) * 4;
    }
    __device__ __forceinline__ v4f load(int t) const
    {
        float v[4];
        ws_load4(r, yoff + 64 * t, c00 + 16 * t, N, cl_sv(N), v);
        return (v4f){v[0], v[1], v[2], v[3]};
    }
    __device__ __forceinline__ void store(int t, v4f x) const
    {
        const float v[4] = {x[0], x[1], x[2], x[3]};
        ws_store4(r, yoff + 64 * t, c00 + 16 * t, N, cl_sv(N), v);
    }
};

// Whole-matrix descriptor for accesses by a per-lane row index (rows gathered through a map); the byte offsets are 32-bit:
// the host refuses matrices of 2 GiB and more.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t whole_rsrc(const float *base, long long rows, int ld)
{
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(reinterpret_cast<uintptr_t>(base) & 0xffffffffu));
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(reinterpret_cast<uintptr_t>(base) >> 32));
    const int bytes = __builtin_amdgcn_readfirstlane((int)(base ? rows * ld * 4 : 0));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(((uintptr_t)hi << 32) | lo), 0, bytes, 0x00020000);
}

// Tile-layout LOADS of a row-major [rows][N] matrix where lane (fr, kq) reads row `row` (its own: -1 = none, zeros)
template <int N>
struct RowTiles {
    __amdgpu_buffer_rsrc_t r;
    int off, c00;
    bool ok;
    __device__ __forceinline__ RowTiles(const float *base, long long rows, int row, const Lane &L)
    {
        r = whole_rsrc(base, rows, N);
        c00 = 4 * L.kq;
        ok = row >= 0;
        off = (row * N + c00) * 4;
    }
    __device__ __forceinline__ v4f load(int t) const
    {
        float v[4];
        ws_load4(r, ok ? off + 64 * t : BUF_OOB, ok ? c00 + 16 * t : N, N, cl_sv(N), v);
        return (v4f){v[0], v[1], v[2], v[3]};
    }
};

// fragments of row `row` (per lane; -1 = zeros) of a row-major [rows][K] matrix
template <int K>
__device__ __forceinline__ void load_frags_row(v4f (&a)[cl_kg(K)], const float *X, long long rows, int row, const Lane &L)
{
    constexpr int KG = cl_kg(K), VEC = cl_sv(K);
    const __amdgpu_buffer_rsrc_t rx = whole_rsrc(X, rows, K);
    const int voff = row >= 0 ? row * K * 4 + 16 * L.kq : 0x7fff0000;      // past any descriptor's range: zeros
#pragma unroll
    for (int g = 0; g < KG; g++) {
        const float4 v = ws_load_group<VEC, KG>(rx, voff, L.kq, K, g);
        a[g] = (v4f){v.x, v.y, v.z, v.w};
    }
}

// the lane's row of a map (int32 per chain row; rows past M read as -1)
__device__ __forceinline__ int map_row(const int *__restrict__ map, long long rb, long long RB, long long M, const Lane &L)
{
    const long long r = rb * 16 + L.fr;
    return (rb < RB && r < M) ? map[r] : -1;
}

// acc[t] += sum_k W_img[16 t + fr'][k] X[row][k]: K = contraction length (a has cl_kg(K) groups), NT output tiles.
// wimg: image with row stride LD; the lane's view starts at row fr, column 4 kq.
template <int K, int NT, int LD>
__device__ __forceinline__ void chain_mm(v4f (&acc)[NT], const v4f (&a)[cl_kg(K)], const float *wimg, const Lane &L)
{
    constexpr int KG = cl_kg(K), LAST = cl_steps(K);
    const float *wb = wimg + L.fr * LD + 4 * L.kq;
#pragma unroll
    for (int g = 0; g < KG; g++) {
        const int steps = g == KG - 1 ? LAST : 4;
#pragma unroll
        for (int t0 = 0; t0 < NT; t0 += 4) {
            v4f b[4];
#pragma unroll
            for (int tt = 0; tt < 4; tt++)
                if (t0 + tt < NT) {
                    const float4 v = *reinterpret_cast<const float4 *>(wb + (t0 + tt) * 16 * LD + 16 * g);
                    b[tt] = (v4f){v.x, v.y, v.z, v.w};
                }
#pragma unroll
            for (int i = 0; i < 4; i++) {
                if (i >= steps) continue;
#pragma unroll
                for (int tt = 0; tt < 4; tt++)
                    if (t0 + tt < NT) acc[t0 + tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[tt][i], a[g][i], acc[t0 + tt], 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);      // keep the groups in order: hoisting every B read blows the VGPR budget
    }
}

template <int NT>
__device__ __forceinline__ void init_bias(v4f (&acc)[NT], const float *sbias, const Lane &L)
{
#pragma unroll
    for (int t = 0; t < NT; t++) {
        const float4 v = *reinterpret_cast<const float4 *>(sbias + 16 * t + 4 * L.kq);
        acc[t] = (v4f){v.x, v.y, v.z, v.w};
    }
}

template <int NT>
__device__ __forceinline__ void init_zero(v4f (&acc)[NT])
{
#pragma unroll
    for (int t = 0; t < NT; t++) acc[t] = (v4f){0.f, 0.f, 0.f, 0.f};
}

// GELU (exact, erf form: torch.nn.GELU()) with the normal cdf from Abramowitz & Stegun 7.1.26 evaluated on the tail side:
//   Q(|x|) = 1/2 erfc(|x| / sqrt 2) = 1/2 p(t) exp(-x^2 / 2),  t = 1 / (1 + 0.3275911 |x| / sqrt 2),  |error| <= 0.75e-7,
// cdf(x) = x >= 0 ? 1 - Q : Q (no cancellation in the negative tail).  One v_rcp_f32 + one v_exp_f32 + 8 FMA-class
// instructions per value, and GELU'(x) = cdf(x) + x pdf(x) reuses the exponential; libm's erff is ~50 instructions, which made
// the activations of a 16-row block cost as many issue cycles as its MFMAs.
struct GeluParts {
    float cdf, e;      // e = exp(-x^2 / 2)
};
__device__ __forceinline__ GeluParts gelu_parts(float x)
{
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752440f, ax, 1.0f));
    float p = fmaf(t, 1.061405429f, -1.453152027f);
    p = fmaf(t, p, 1.421413741f);
    p = fmaf(t, p, -0.284496736f);
    p = fmaf(t, p, 0.254829592f);
    p *= t;
    const float e = __builtin_amdgcn_exp2f(-0.5f * 1.44269504088896340736f * ax * ax);
    const float q = 0.5f * p * e;
    return GeluParts{x >= 0.f ? 1.0f - q : q, e};
}
__device__ __forceinline__ float c_gelu(float x) { return x * gelu_parts(x).cdf; }
__device__ __forceinline__ float c_gelu_grad(float x)
{
    const GeluParts g = gelu_parts(x);
    return fmaf(x * 0.39894228040143267794f, g.e, g.cdf);
}
__device__ __forceinline__ v4f v_gelu(v4f z) { return (v4f){c_gelu(z[0]), c_gelu(z[1]), c_gelu(z[2]), c_gelu(z[3])}; }
__device__ __forceinline__ v4f v_gelu_grad(v4f z) { return (v4f){c_gelu_grad(z[0]), c_gelu_grad(z[1]), c_gelu_grad(z[2]), c_gelu_grad(z[3])}; }
__device__ __forceinline__ v4f v_relu(v4f z) { return (v4f){fmaxf(z[0], 0.f), fmaxf(z[1], 0.f), fmaxf(z[2], 0.f), fmaxf(z[3], 0.f)}; }

// 16-row blocks of one network are dealt wave-major over the nblk workgroups that serve it (linear_ws.h: even rounds on every
// SIMD); blk = this workgroup's index among them.  A launch may serve several networks at once (MAX_NETS below): workgroup b
// works for network b % n — one prologue and one partial last round for three networks' rows instead of one each.
template <int WAVES>
__device__ __forceinline__ void row_blocks(long long M, const Lane &L, int blk, int nblk, long long &rb, long long &RB, long long &stride)
{
    RB = (M + 15) >> 4;
    stride = (long long)nblk * WAVES;
    rb = (long long)L.wave * nblk + blk;
}
template <int WAVES>
__device__ __forceinline__ void row_blocks(long long M, const Lane &L, long long &rb, long long &RB, long long &stride)
{
    row_blocks<WAVES>(M, L, (int)blockIdx.x, (int)gridDim.x, rb, RB, stride);
}

constexpr int MAX_NETS = 3;
struct NetOfBlock {
    int net, blk, nblk;
    __device__ NetOfBlock(int n)
    {
        net = (int)blockIdx.x % n;
        blk = (int)blockIdx.x / n;
        nblk = ((int)gridDim.x - net + n - 1) / n;
    }
};
template <typename T>
__device__ __forceinline__ T pick(const T (&a)[MAX_NETS], int i) { return i == 0 ? a[0] : (i == 1 ? a[1] : a[2]); }

// ======================================================================================================= FiLM nets
// gamma = Wg1 relu(Wg0 c + bg0) + bg1, beta likewise (reference scene/gaussian_model.py:150-166); cg / cb = the hidden ReLU
// outputs (the backward's masks and the second layers' dW operands).
struct FilmW {
    const float *Wg0, *bg0, *Wg1, *bg1, *Wb0, *bb0, *Wb1, *bb1;
};

template <int COND, int HID>
struct FilmFwdLds {
    static constexpr int LD = cl_ld(COND), NT0 = cl_kg(COND), NT1 = cl_kg(HID);
    static constexpr int W0 = NT0 * 16 * LD, W1 = NT1 * 16 * LD;
    static constexpr int o_g0 = 0, o_g1 = W0, o_b0 = W0 + W1, o_b1 = 2 * W0 + W1, o_bias = 2 * W0 + 2 * W1;
    static constexpr int FLOATS = o_bias + 2 * NT0 * 16 + 2 * NT1 * 16;
};

template <int COND, int HID, int CHAIN_THREADS>
__device__ __forceinline__ void film_fwd_body(int blk, int nblk, const float *__restrict__ cond, FilmW w, float *__restrict__ cg,
                                                                 float *__restrict__ cb, float *__restrict__ gamma,
                                                                 float *__restrict__ beta, long long M)
{
    extern __shared__ float lds[];
    using S = FilmFwdLds<COND, HID>;
    const Lane L;
    zero_lds(lds, S::FLOATS, L.tid, CHAIN_THREADS);
    __syncthreads();
    stage_block<false>(w.Wg0, COND, COND, 0, COND, lds + S::o_g0, S::LD, 0, L.tid, CHAIN_THREADS);
    stage_block<false>(w.Wg1, COND, HID, 0, COND, lds + S::o_g1, S::LD, 0, L.tid, CHAIN_THREADS);
    stage_block<false>(w.Wb0, COND, COND, 0, COND, lds + S::o_b0, S::LD, 0, L.tid, CHAIN_THREADS);
    stage_block<false>(w.Wb1, COND, HID, 0, COND, lds + S::o_b1, S::LD, 0, L.tid, CHAIN_THREADS);
    float *sb_g0 = lds + S::o_bias, *sb_g1 = sb_g0 + S::NT0 * 16, *sb_b0 = sb_g1 + S::NT1 * 16, *sb_b1 = sb_b0 + S::NT0 * 16;
    stage_bias(w.bg0, COND, sb_g0, S::NT0 * 16, L.tid, CHAIN_THREADS);
    stage_bias(w.bg1, HID, sb_g1, S::NT1 * 16, L.tid, CHAIN_THREADS);
    stage_bias(w.bb0, COND, sb_b0, S::NT0 * 16, L.tid, CHAIN_THREADS);
    stage_bias(w.bb1, HID, sb_b1, S::NT1 * 16, L.tid, CHAIN_THREADS);
    __syncthreads();
    long long rb, RB, stride;
    row_blocks<CHAIN_THREADS / 64>(M, L, blk, nblk, rb, RB, stride);
    // memory and MFMA phases of a wave overlap: the next block's condition rows are requested as soon as this block's last use
    // of them has been issued, and nothing ever waits for a store
    v4f c[S::NT0];
    load_frags<COND>(c, cond, rb, RB, M, L);
    for (; rb < RB; rb += stride) {
#pragma unroll
        for (int net = 0; net < 2; net++) {
            const float *w0 = lds + (net ? S::o_b0 : S::o_g0), *w1 = lds + (net ? S::o_b1 : S::o_g1);
            v4f hid[S::NT0];
            init_bias(hid, net ? sb_b0 : sb_g0, L);
            chain_mm<COND, S::NT0, S::LD>(hid, c, w0, L);
            if (net == 1) {
                load_frags<COND>(c, cond, rb + stride, RB, M, L);
                __builtin_amdgcn_sched_barrier(0);
            }
            const Tiles<COND> th(net ? cb : cg, (net ? cb : cg) ? rb : RB, RB, M, L);      // NULL (inference): an empty descriptor drops the stores
#pragma unroll
            for (int t = 0; t < S::NT0; t++) {
                hid[t] = v_relu(hid[t]);
                th.store(t, hid[t]);
            }
            v4f out[S::NT1];
            init_bias(out, net ? sb_b1 : sb_g1, L);
            chain_mm<COND, S::NT1, S::LD>(out, hid, w1, L);
            const Tiles<HID> to(net ? beta : gamma, rb, RB, M, L);
#pragma unroll
            for (int t = 0; t < S::NT1; t++) to.store(t, out[t]);
        }
    }
}

// backward through the second layers and the ReLUs: gcg = (ggamma Wg1) * [cg > 0], gcb likewise (the first layers' dW
// operands; the condition itself — frame time and z embedding of detached anchors — needs no gradient)
template <int COND, int HID, int CHAIN_THREADS>
__device__ __forceinline__ void film_bwd_body(int blk, int nblk, const float *__restrict__ ggamma, const float *__restrict__ gbeta,
                                                                 const float *__restrict__ cg, const float *__restrict__ cb,
                                                                 const float *__restrict__ Wg1, const float *__restrict__ Wb1,
                                                                 float *__restrict__ gcg, float *__restrict__ gcb, long long M,
                                                                 const int *__restrict__ src_a, const int *__restrict__ src_b,
                                                                 long long src_rows, float *__restrict__ ggamma_sum,
                                                                 float *__restrict__ gbeta_sum)
{
    extern __shared__ float lds[];
    constexpr int LD = cl_ld(HID), NT = cl_kg(COND), IMG = NT * 16 * LD;
    const Lane L;
    zero_lds(lds, 2 * IMG, L.tid, CHAIN_THREADS);
    __syncthreads();
    stage_block<true>(Wg1, COND, HID, 0, COND, lds, LD, 0, L.tid, CHAIN_THREADS);
    stage_block<true>(Wb1, COND, HID, 0, COND, lds + IMG, LD, 0, L.tid, CHAIN_THREADS);
    __syncthreads();
    long long rb, RB, stride;
    row_blocks<CHAIN_THREADS / 64>(M, L, blk, nblk, rb, RB, stride);
    // src_a / src_b (shared FiLM rows): this kernel's row q is a (frame, anchor); ggamma / gbeta hold one row per (view, anchor) —
    // the frame's two opposite views — and the row's gradient is the sum of the two (either may be absent: -1).  The sums are
    // written out: they are the G operands of the second layers' weight gradients.
    v4f g[2][cl_kg(HID)], t2[2][cl_kg(HID)];
    int ra = -1, rb2 = -1, ra_n = -1, rb_n = -1;
    // the (summed) gradients of a block are requested one block ahead — network by network, as soon as this block's have been
    // summed into the product's operand —, the map entries two blocks ahead
    auto fetch = [&](int net, long long b, int a_row, int b_row) {
        const float *src = net ? gbeta : ggamma;
        if (src_a) {
            load_frags_row<HID>(g[net], src, src_rows, a_row, L);
            load_frags_row<HID>(t2[net], src, src_rows, b_row, L);
        } else {
            load_frags<HID>(g[net], src, b, RB, M, L);
        }
    };
    if (src_a) {
        ra = map_row(src_a, rb, RB, M, L);
        rb2 = map_row(src_b, rb, RB, M, L);
        ra_n = map_row(src_a, rb + stride, RB, M, L);
        rb_n = map_row(src_b, rb + stride, RB, M, L);
    }
    fetch(0, rb, ra, rb2);
    fetch(1, rb, ra, rb2);
    for (; rb < RB; rb += stride) {
        v4f m[2][NT];      // the ReLU outputs (masks) of this block
        {
            const Tiles<COND> t0(cg, rb, RB, M, L), t1(cb, rb, RB, M, L);
#pragma unroll
            for (int t = 0; t < NT; t++) { m[0][t] = t0.load(t); m[1][t] = t1.load(t); }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int net = 0; net < 2; net++) {
            v4f cur[cl_kg(HID)];
#pragma unroll
            for (int i = 0; i < cl_kg(HID); i++) cur[i] = src_a ? g[net][i] + t2[net][i] : g[net][i];
            fetch(net, rb + stride, ra_n, rb_n);
            __builtin_amdgcn_sched_barrier(0);
            if (src_a) {
                const Tiles<HID> ts(net ? gbeta_sum : ggamma_sum, rb, RB, M, L);
#pragma unroll
                for (int t = 0; t < cl_kg(HID); t++) ts.store(t, cur[t]);
            }
            v4f acc[NT];
            init_zero(acc);
            chain_mm<HID, NT, LD>(acc, cur, lds + net * IMG, L);
            if (net == 1 && src_a) {
                ra_n = map_row(src_a, rb + 2 * stride, RB, M, L);
                rb_n = map_row(src_b, rb + 2 * stride, RB, M, L);
            }
            const Tiles<COND> to(net ? gcb : gcg, rb, RB, M, L);
#pragma unroll
            for (int t = 0; t < NT; t++) {
                v4f o;
#pragma unroll
                for (int i = 0; i < 4; i++) o[i] = m[net][t][i] > 0.f ? acc[t][i] : 0.f;
                to.store(t, o);
            }
        }
    }
}

// ======================================================================================================= generator trunk
// y = act(W3 (gamma * (W2 gelu(W1 f + b1) + b2) + beta) + b3)   (reference scene/gaussian_model.py:168-196)
struct TrunkW {
    const float *W1, *b1, *W2, *b2, *W3, *b3;
};

enum { CH_ACT_NONE = 0, CH_ACT_TANH = 1, CH_ACT_SIGMOID = 2 };

template <int FEAT, int HID, int OUT>
struct TrunkFwdLds {
    static constexpr int LD1 = cl_ld(FEAT), LD2 = cl_ld(HID), NTH = cl_kg(HID), NTO = cl_kg(OUT);
    static constexpr int o_w1 = 0, o_w2 = NTH * 16 * LD1, o_w3 = o_w2 + NTH * 16 * LD2, o_bias = o_w3 + NTO * 16 * LD2;
    static constexpr int FLOATS = o_bias + 2 * NTH * 16 + NTO * 16;
};

template <int FEAT, int HID, int OUT, int CHAIN_THREADS>
__device__ __forceinline__ void trunk_fwd_body(int blk, int nblk, const float *__restrict__ feat, const float *__restrict__ gamma,
                                                             const float *__restrict__ beta, TrunkW w, int act,
                                                             float *__restrict__ z1, float *__restrict__ a1, float *__restrict__ h,
                                                             float *__restrict__ x3, float *__restrict__ y, long long M,
                                                             const int *__restrict__ film_row, long long film_rows)
{
    extern __shared__ float lds[];
    using S = TrunkFwdLds<FEAT, HID, OUT>;
    const Lane L;
    zero_lds(lds, S::FLOATS, L.tid, CHAIN_THREADS);
    __syncthreads();
    stage_block<false>(w.W1, FEAT, HID, 0, FEAT, lds + S::o_w1, S::LD1, 0, L.tid, CHAIN_THREADS);
    stage_block<false>(w.W2, HID, HID, 0, HID, lds + S::o_w2, S::LD2, 0, L.tid, CHAIN_THREADS);
    stage_block<false>(w.W3, HID, OUT, 0, HID, lds + S::o_w3, S::LD2, 0, L.tid, CHAIN_THREADS);
    float *sb1 = lds + S::o_bias, *sb2 = sb1 + S::NTH * 16, *sb3 = sb2 + S::NTH * 16;
    stage_bias(w.b1, HID, sb1, S::NTH * 16, L.tid, CHAIN_THREADS);
    stage_bias(w.b2, HID, sb2, S::NTH * 16, L.tid, CHAIN_THREADS);
    stage_bias(w.b3, OUT, sb3, S::NTO * 16, L.tid, CHAIN_THREADS);
    __syncthreads();
    long long rb, RB, stride;
    row_blocks<CHAIN_THREADS / 64>(M, L, blk, nblk, rb, RB, stride);
    v4f f[cl_kg(FEAT)];
    load_frags<FEAT>(f, feat, rb, RB, M, L);
    int fr_next = film_row ? map_row(film_row, rb, RB, M, L) : -1;      // the map entry travels one block ahead of its use
    for (; rb < RB; rb += stride) {
        // this block's gamma / beta travel while its first two products run; the next block's feature rows are requested as
        // soon as the first product has consumed this block's; no wait ever names a store
        v4f gam[S::NTH], bet[S::NTH];
        if (film_row) {
            // gamma / beta live once per (frame, anchor): the two opposite views of a frame share the condition (same camera z,
            // same anchor z), so the FiLM networks ran on those rows only and this row names its (film_row)
            const int fr_ = fr_next;
            fr_next = map_row(film_row, rb + stride, RB, M, L);
            const RowTiles<HID> tg(gamma, film_rows, fr_, L), tb(beta, film_rows, fr_, L);
#pragma unroll
            for (int t = 0; t < S::NTH; t++) { gam[t] = tg.load(t); bet[t] = tb.load(t); }
        } else {
            const Tiles<HID> tg(gamma, rb, RB, M, L), tb(beta, rb, RB, M, L);
#pragma unroll
            for (int t = 0; t < S::NTH; t++) { gam[t] = tg.load(t); bet[t] = tb.load(t); }
        }
        __builtin_amdgcn_sched_barrier(0);
        v4f u[S::NTH];
        init_bias(u, sb1, L);
        chain_mm<FEAT, S::NTH, S::LD1>(u, f, lds + S::o_w1, L);
        load_frags<FEAT>(f, feat, rb + stride, RB, M, L);
        __builtin_amdgcn_sched_barrier(0);
        {
            const Tiles<HID> tz(z1, z1 ? rb : RB, RB, M, L), ta(a1, a1 ? rb : RB, RB, M, L);      // NULL (inference): stores dropped
#pragma unroll
            for (int t = 0; t < S::NTH; t++) {
                tz.store(t, u[t]);
                u[t] = v_gelu(u[t]);
                ta.store(t, u[t]);
            }
        }
        v4f v[S::NTH];
        init_bias(v, sb2, L);
        chain_mm<HID, S::NTH, S::LD2>(v, u, lds + S::o_w2, L);
        {
            const Tiles<HID> th(h, h ? rb : RB, RB, M, L), tx(x3, x3 ? rb : RB, RB, M, L);
#pragma unroll
            for (int t = 0; t < S::NTH; t++) {
                th.store(t, v[t]);
#pragma unroll
                for (int i = 0; i < 4; i++) v[t][i] = fmaf(gam[t][i], v[t][i], bet[t][i]);
                tx.store(t, v[t]);
            }
        }
        v4f o[S::NTO];
        init_bias(o, sb3, L);
        chain_mm<HID, S::NTO, S::LD2>(o, v, lds + S::o_w3, L);
        const Tiles<OUT> ty(y, rb, RB, M, L);
#pragma unroll
        for (int t = 0; t < S::NTO; t++) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                if (act == CH_ACT_TANH) o[t][i] = tanhf(o[t][i]);
                else if (act == CH_ACT_SIGMOID) o[t][i] = 1.0f / (1.0f + expf(-o[t][i]));
            }
            ty.store(t, o[t]);
        }
    }
}

// go = gy * act'(y);  gx3 = go W3 (= d beta);  d gamma = gx3 * h;  gh = gx3 * gamma;  gz1 = (gh W2) * gelu'(z1);
// gfeat (+)= gz1 W1.  go, gx3, d gamma, gh, gz1 are written out: they are the G operands of the seven weight gradients.
template <int FEAT, int HID, int OUT>
struct TrunkBwdLds {
    static constexpr int LD3 = cl_ld(OUT), LDH = cl_ld(HID), NTH = cl_kg(HID), NTF = cl_kg(FEAT), LD1 = cl_ld(FEAT);
    static constexpr int o_w3 = 0, o_w2 = NTH * 16 * LD3, o_w1 = o_w2 + NTH * 16 * LDH;
    // + the forward image of W1 and b1: z1 = W1 feat + b1 is formed again here (the same products in the same order as the forward,
    // bit for bit) instead of travelling through HBM — 400 of the ~3.3 KB a row of a generator moved each way
    static constexpr int o_w1f = o_w1 + NTF * 16 * LDH, o_b1 = o_w1f + NTH * 16 * LD1;
    static constexpr int FLOATS = o_b1 + NTH * 16;
};

template <int FEAT, int HID, int OUT, int CHAIN_THREADS>
__device__ __forceinline__ void trunk_bwd_body(int blk, int nblk, const float *__restrict__ gy, const float *__restrict__ y, int act,
                                                             const float *__restrict__ h, const float *__restrict__ gamma,
                                                             const float *__restrict__ feat, TrunkW w, float *__restrict__ go,
                                                             float *__restrict__ gbeta, float *__restrict__ ggamma,
                                                             float *__restrict__ gh, float *__restrict__ gz1,
                                                             float *__restrict__ gfeat, int accumulate, long long M,
                                                             const int *__restrict__ film_row, long long film_rows)
{
    extern __shared__ float lds[];
    using S = TrunkBwdLds<FEAT, HID, OUT>;
    const Lane L;
    zero_lds(lds, S::FLOATS, L.tid, CHAIN_THREADS);
    __syncthreads();
    stage_block<true>(w.W3, HID, OUT, 0, HID, lds + S::o_w3, S::LD3, 0, L.tid, CHAIN_THREADS);
    stage_block<true>(w.W2, HID, HID, 0, HID, lds + S::o_w2, S::LDH, 0, L.tid, CHAIN_THREADS);
    stage_block<true>(w.W1, FEAT, HID, 0, FEAT, lds + S::o_w1, S::LDH, 0, L.tid, CHAIN_THREADS);
    stage_block<false>(w.W1, FEAT, HID, 0, FEAT, lds + S::o_w1f, S::LD1, 0, L.tid, CHAIN_THREADS);
    float *sb1 = lds + S::o_b1;
    stage_bias(w.b1, HID, sb1, S::NTH * 16, L.tid, CHAIN_THREADS);
    __syncthreads();
    long long rb, RB, stride;
    row_blocks<CHAIN_THREADS / 64>(M, L, blk, nblk, rb, RB, stride);
    constexpr int NTO = cl_kg(OUT);
    v4f f[cl_kg(FEAT)];
    load_frags<FEAT>(f, feat, rb, RB, M, L);
    v4f g0[NTO], yv[NTO];
    {
        const Tiles<OUT> tg(gy, rb, RB, M, L), ty(y, rb, RB, M, L);
#pragma unroll
        for (int t = 0; t < NTO; t++) { g0[t] = tg.load(t); yv[t] = ty.load(t); }
    }
    int fr_next = film_row ? map_row(film_row, rb, RB, M, L) : -1;
    for (; rb < RB; rb += stride) {
        // everything this block reads later (h, gamma, z1, the running feature gradient) is requested up front and lands while
        // the products run; the next block's (gy, y) are requested once this block's have been consumed
        v4f hh[S::NTH], gg[S::NTH], pf[S::NTF];
        {
            const Tiles<HID> th(h, rb, RB, M, L);
#pragma unroll
            for (int t = 0; t < S::NTH; t++) hh[t] = th.load(t);
            if (film_row) {
                const int fr_ = fr_next;
                fr_next = map_row(film_row, rb + stride, RB, M, L);
                const RowTiles<HID> tg(gamma, film_rows, fr_, L);
#pragma unroll
                for (int t = 0; t < S::NTH; t++) gg[t] = tg.load(t);
            } else {
                const Tiles<HID> tg(gamma, rb, RB, M, L);
#pragma unroll
                for (int t = 0; t < S::NTH; t++) gg[t] = tg.load(t);
            }
            const Tiles<FEAT> tf(gfeat, accumulate ? rb : RB, RB, M, L);      // empty descriptor (zeros) when not accumulating
#pragma unroll
            for (int t = 0; t < S::NTF; t++) pf[t] = tf.load(t);
        }
        __builtin_amdgcn_sched_barrier(0);
        v4f go_[NTO];
        {
            const Tiles<OUT> to(go, rb, RB, M, L);
#pragma unroll
            for (int t = 0; t < NTO; t++) {
                go_[t] = g0[t];
                if (act != CH_ACT_NONE) {
#pragma unroll
                    for (int i = 0; i < 4; i++)
                        go_[t][i] = act == CH_ACT_TANH ? g0[t][i] * (1.0f - yv[t][i] * yv[t][i]) : g0[t][i] * ((1.0f - yv[t][i]) * yv[t][i]);
                }
                to.store(t, go_[t]);
            }
            const Tiles<OUT> tg(gy, rb + stride, RB, M, L), ty(y, rb + stride, RB, M, L);
#pragma unroll
            for (int t = 0; t < NTO; t++) { g0[t] = tg.load(t); yv[t] = ty.load(t); }
        }
        __builtin_amdgcn_sched_barrier(0);
        v4f gx[S::NTH];
        init_zero(gx);
        chain_mm<OUT, S::NTH, S::LD3>(gx, go_, lds + S::o_w3, L);
        {
            const Tiles<HID> ob(gbeta, rb, RB, M, L), og(ggamma, rb, RB, M, L), oh(gh, rb, RB, M, L);
#pragma unroll
            for (int t = 0; t < S::NTH; t++) {
                ob.store(t, gx[t]);
                v4f dg;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    dg[i] = gx[t][i] * hh[t][i];
                    gx[t][i] = gx[t][i] * gg[t][i];
                }
                og.store(t, dg);
                oh.store(t, gx[t]);
            }
        }
        v4f ga[S::NTH];
        init_zero(ga);
        chain_mm<HID, S::NTH, S::LDH>(ga, gx, lds + S::o_w2, L);
        // z1 of this block, as the forward formed it; the next block's feature rows are requested behind it
        v4f zz[S::NTH];
        init_bias(zz, sb1, L);
        chain_mm<FEAT, S::NTH, S::LD1>(zz, f, lds + S::o_w1f, L);
        load_frags<FEAT>(f, feat, rb + stride, RB, M, L);
        {
            const Tiles<HID> oz(gz1, rb, RB, M, L);
#pragma unroll
            for (int t = 0; t < S::NTH; t++) {
                const v4f d = v_gelu_grad(zz[t]);
#pragma unroll
                for (int i = 0; i < 4; i++) ga[t][i] *= d[i];
                oz.store(t, ga[t]);
            }
        }
        chain_mm<HID, S::NTF, S::LDH>(pf, ga, lds + S::o_w1, L);      // on top of the running sum (or zeros)
        const Tiles<FEAT> of(gfeat, rb, RB, M, L);
#pragma unroll
        for (int t = 0; t < S::NTF; t++) of.store(t, pf[t]);
    }
}

// ---- one launch for up to MAX_NETS generators (workgroup b serves network b % n) ----------------------------------------------
struct FilmFwdBatch {
    int n;
    FilmW w[MAX_NETS];
    float *cg[MAX_NETS], *cb[MAX_NETS], *gamma[MAX_NETS], *beta[MAX_NETS];
};
template <int COND, int HID, int CHAIN_THREADS>
__global__ void __launch_bounds__(CHAIN_THREADS) k_film_nets_fwd(const float *__restrict__ cond, FilmFwdBatch b, long long M)
{
    const NetOfBlock nb(b.n);
    film_fwd_body<COND, HID, CHAIN_THREADS>(nb.blk, nb.nblk, cond, pick(b.w, nb.net), pick(b.cg, nb.net), pick(b.cb, nb.net),
                                            pick(b.gamma, nb.net), pick(b.beta, nb.net), M);
}

struct FilmBwdBatch {
    int n;
    const float *ggamma[MAX_NETS], *gbeta[MAX_NETS], *cg[MAX_NETS], *cb[MAX_NETS], *Wg1[MAX_NETS], *Wb1[MAX_NETS];
    float *gcg[MAX_NETS], *gcb[MAX_NETS];
    const int *src_a, *src_b;      // shared FiLM rows: the two (view, anchor) rows behind every (frame, anchor) row; NULL = one to one
    long long src_rows;
    float *ggamma_sum[MAX_NETS], *gbeta_sum[MAX_NETS];
};
template <int COND, int HID, int CHAIN_THREADS>
__global__ void __launch_bounds__(CHAIN_THREADS) k_film_nets_bwd(FilmBwdBatch b, long long M)
{
    const NetOfBlock nb(b.n);
    film_bwd_body<COND, HID, CHAIN_THREADS>(nb.blk, nb.nblk, pick(b.ggamma, nb.net), pick(b.gbeta, nb.net), pick(b.cg, nb.net),
                                            pick(b.cb, nb.net), pick(b.Wg1, nb.net), pick(b.Wb1, nb.net), pick(b.gcg, nb.net),
                                            pick(b.gcb, nb.net), M, b.src_a, b.src_b, b.src_rows, pick(b.ggamma_sum, nb.net),
                                            pick(b.gbeta_sum, nb.net));
}

struct TrunkFwdBatch {
    int n;
    int out[MAX_NETS], act[MAX_NETS];
    TrunkW w[MAX_NETS];
    const float *gamma[MAX_NETS], *beta[MAX_NETS];
    float *z1[MAX_NETS], *a1[MAX_NETS], *h[MAX_NETS], *x3[MAX_NETS], *y[MAX_NETS];
    const int *film_row;           // shared FiLM rows: chain row -> row of gamma / beta; NULL = the same row
    long long film_rows;
};
template <int FEAT, int HID, int CHAIN_THREADS>
__global__ void __launch_bounds__(CHAIN_THREADS) k_trunk_fwd(const float *__restrict__ feat, TrunkFwdBatch b, long long M)
{
    const NetOfBlock nb(b.n);
    const int i = nb.net;
#define GSVC_TRUNK_FWD(OUT) trunk_fwd_body<FEAT, HID, OUT, CHAIN_THREADS>(nb.blk, nb.nblk, feat, pick(b.gamma, i), pick(b.beta, i), pick(b.w, i), \
        pick(b.act, i), pick(b.z1, i), pick(b.a1, i), pick(b.h, i), pick(b.x3, i), pick(b.y, i), M, b.film_row, b.film_rows)
    switch (pick(b.out, i)) {
        case 10: GSVC_TRUNK_FWD(10); break;
        case 30: GSVC_TRUNK_FWD(30); break;
        default: GSVC_TRUNK_FWD(70); break;
    }
#undef GSVC_TRUNK_FWD
}

struct TrunkBwdBatch {
    int n;
    int out[MAX_NETS], act[MAX_NETS], accumulate[MAX_NETS];
    TrunkW w[MAX_NETS];
    const float *gy[MAX_NETS], *y[MAX_NETS], *h[MAX_NETS], *gamma[MAX_NETS];
    const float *feat;             // the generators' shared input: z1 is formed from it again
    float *go[MAX_NETS], *gbeta[MAX_NETS], *ggamma[MAX_NETS], *gh[MAX_NETS], *gz1[MAX_NETS], *gfeat[MAX_NETS];
    const int *film_row;
    long long film_rows;
};
template <int FEAT, int HID, int CHAIN_THREADS>
__global__ void __launch_bounds__(CHAIN_THREADS) k_trunk_bwd(TrunkBwdBatch b, long long M)
{
    const NetOfBlock nb(b.n);
    const int i = nb.net;
#define GSVC_TRUNK_BWD(OUT) trunk_bwd_body<FEAT, HID, OUT, CHAIN_THREADS>(nb.blk, nb.nblk, pick(b.gy, i), pick(b.y, i), pick(b.act, i), pick(b.h, i), \
        pick(b.gamma, i), b.feat, pick(b.w, i), pick(b.go, i), pick(b.gbeta, i), pick(b.ggamma, i), pick(b.gh, i), pick(b.gz1, i), \
        pick(b.gfeat, i), pick(b.accumulate, i), M, b.film_row, b.film_rows)
    switch (pick(b.out, i)) {
        case 10: GSVC_TRUNK_BWD(10); break;
        case 30: GSVC_TRUNK_BWD(30); break;
        default: GSVC_TRUNK_BWD(70); break;
    }
#undef GSVC_TRUNK_BWD
}

// ======================================================================================================= mlp_deform
// x = [feat | cond] -> L1 GELU -> L2 GELU | -> L3 GELU -> L4 GELU -> L5   (reference scene/gaussian_model.py:468-489).
// The two inputs stay two matrices: the image of W1 keeps the feature columns in k-groups 0 .. and the condition columns from
// the next 16-aligned column on (no concatenated [M, 116] copy, no split in the backward).
struct DeformW {
    const float *W[5], *b[5];
};

template <int FEAT, int COND, int HID>
struct DeformALds {
    static constexpr int KF = cl_kg(FEAT) * 16, KIN = KF + cl_kg(COND) * 16;      // padded input width of the image
    static constexpr int LD1 = KIN + 8, LD2 = cl_ld(HID), NTH = cl_kg(HID);
    static constexpr int o_w1 = 0, o_w2 = NTH * 16 * LD1, o_bias = o_w2 + NTH * 16 * LD2, FLOATS = o_bias + 2 * NTH * 16;
};

template <int FEAT, int COND, int HID, int CHAIN_THREADS>
__global__ void __launch_bounds__(CHAIN_THREADS) k_deform_a_fwd(const float *__restrict__ feat, const float *__restrict__ cond, DeformW w,
                                                                float *__restrict__ z1, float *__restrict__ a1, float *__restrict__ z2,
                                                                float *__restrict__ a2, long long M)
{
    extern __shared__ float lds[];
    using S = DeformALds<FEAT, COND, HID>;
    const Lane L;
    zero_lds(lds, S::FLOATS, L.tid, CHAIN_THREADS);
    __syncthreads();
    stage_block<false>(w.W[0], FEAT + COND, HID, 0, FEAT, lds + S::o_w1, S::LD1, 0, L.tid, CHAIN_THREADS);
    stage_block<false>(w.W[0], FEAT + COND, HID, FEAT, COND, lds + S::o_w1, S::LD1, S::KF, L.tid, CHAIN_THREADS);
    stage_block<false>(w.W[1], HID, HID, 0, HID, lds + S::o_w2, S::LD2, 0, L.tid, CHAIN_THREADS);
    float *sb1 = lds + S::o_bias, *sb2 = sb1 + S::NTH * 16;
    stage_bias(w.b[0], HID, sb1, S::NTH * 16, L.tid, CHAIN_THREADS);
    stage_bias(w.b[1], HID, sb2, S::NTH * 16, L.tid, CHAIN_THREADS);
    __syncthreads();
    long long rb, RB, stride;
    row_blocks<CHAIN_THREADS / 64>(M, L, rb, RB, stride);
    v4f f[cl_kg(FEAT)], c[cl_kg(COND)];
    load_frags<FEAT>(f, feat, rb, RB, M, L);
    load_frags<COND>(c, cond, rb, RB, M, L);
    for (; rb < RB; rb += stride) {
        v4f u[S::NTH];
        init_bias(u, sb1, L);
        chain_mm<FEAT, S::NTH, S::LD1>(u, f, lds + S::o_w1, L);
        chain_mm<COND, S::NTH, S::LD1>(u, c, lds + S::o_w1 + S::KF, L);
        load_frags<FEAT>(f, feat, rb + stride, RB, M, L);      // the next block's rows travel during the rest of this one
        load_frags<COND>(c, cond, rb + stride, RB, M, L);
        __builtin_amdgcn_sched_barrier(0);
        {
            const Tiles<HID> tz(z1, z1 ? rb : RB, RB, M, L), ta(a1, a1 ? rb : RB, RB, M, L);      // NULL (inference): stores dropped
#pragma unroll
            for (int t = 0; t < S::NTH; t++) {
                tz.store(t, u[t]);
                u[t] = v_gelu(u[t]);
                ta.store(t, u[t]);
            }
        }
        v4f v[S::NTH];
        init_bias(v, sb2, L);
        chain_mm<HID, S::NTH, S::LD2>(v, u, lds + S::o_w2, L);
        const Tiles<HID> tz(z2, z2 ? rb : RB, RB, M, L), ta(a2, rb, RB, M, L);
#pragma unroll
        for (int t = 0; t < S::NTH; t++) {
            tz.store(t, v[t]);
            ta.store(t, v_gelu(v[t]));
        }
    }
}

template <int HID, int OUT>
struct DeformBLds {
    static constexpr int LD = cl_ld(HID), NTH = cl_kg(HID), NTO = cl_kg(OUT);
    static constexpr int o_w3 = 0, o_w4 = NTH * 16 * LD, o_w5 = 2 * NTH * 16 * LD, o_bias = o_w5 + NTO * 16 * LD;
    static constexpr int FLOATS = o_bias + 2 * NTH * 16 + NTO * 16;
};

template <int HID, int OUT, int CHAIN_THREADS>
__global__ void __launch_bounds__(CHAIN_THREADS) k_deform_b_fwd(const float *__restrict__ a2, DeformW w, float *__restrict__ z3,
                                                                float *__restrict__ a3, float *__restrict__ z4, float *__restrict__ a4,
                                                                float *__restrict__ y, long long M)
{
    extern __shared__ float lds[];
    using S = DeformBLds<HID, OUT>;
    const Lane L;
    zero_lds(lds, S::FLOATS, L.tid, CHAIN_THREADS);
    __syncthreads();
    stage_block<false>(w.W[2], HID, HID, 0, HID, lds + S::o_w3, S::LD, 0, L.tid, CHAIN_THREADS);
    stage_block<false>(w.W[3], HID, HID, 0, HID, lds + S::o_w4, S::LD, 0, L.tid, CHAIN_THREADS);
    stage_block<false>(w.W[4], HID, OUT, 0, HID, lds + S::o_w5, S::LD, 0, L.tid, CHAIN_THREADS);
    float *sb3 = lds + S::o_bias, *sb4 = sb3 + S::NTH * 16, *sb5 = sb4 + S::NTH * 16;
    stage_bias(w.b[2], HID, sb3, S::NTH * 16, L.tid, CHAIN_THREADS);
    stage_bias(w.b[3], HID, sb4, S::NTH * 16, L.tid, CHAIN_THREADS);
    stage_bias(w.b[4], OUT, sb5, S::NTO * 16, L.tid, CHAIN_THREADS);
    __syncthreads();
    long long rb, RB, stride;
    row_blocks<CHAIN_THREADS / 64>(M, L, rb, RB, stride);
    v4f x[S::NTH];
    load_frags<HID>(x, a2, rb, RB, M, L);
    for (; rb < RB; rb += stride) {
        v4f u[S::NTH];
        init_bias(u, sb3, L);
        chain_mm<HID, S::NTH, S::LD>(u, x, lds + S::o_w3, L);
        load_frags<HID>(x, a2, rb + stride, RB, M, L);      // the next block's rows travel during the rest of this one
        __builtin_amdgcn_sched_barrier(0);
        {
            const Tiles<HID> tz(z3, z3 ? rb : RB, RB, M, L), ta(a3, a3 ? rb : RB, RB, M, L);      // NULL (inference): stores dropped
#pragma unroll
            for (int t = 0; t < S::NTH; t++) {
                tz.store(t, u[t]);
                u[t] = v_gelu(u[t]);
                ta.store(t, u[t]);
            }
        }
        v4f v[S::NTH];
        init_bias(v, sb4, L);
        chain_mm<HID, S::NTH, S::LD>(v, u, lds + S::o_w4, L);
        {
            const Tiles<HID> tz(z4, z4 ? rb : RB, RB, M, L), ta(a4, a4 ? rb : RB, RB, M, L);
#pragma unroll
            for (int t = 0; t < S::NTH; t++) {
                tz.store(t, v[t]);
                v[t] = v_gelu(v[t]);
                ta.store(t, v[t]);
            }
        }
        v4f o[S::NTO];
        init_bias(o, sb5, L);
        chain_mm<HID, S::NTO, S::LD>(o, v, lds + S::o_w5, L);
        const Tiles<OUT> ty(y, rb, RB, M, L);
#pragma unroll
        for (int t = 0; t < S::NTO; t++) ty.store(t, o[t]);
    }
}

// g4 = (gy W5) * gelu'(z4);  g3 = (g4 W4) * gelu'(z3);  g2 = (g3 W3) * gelu'(z2)
template <int HID, int OUT>
struct DeformBBwdLds {
    static constexpr int LD5 = cl_ld(OUT), LD = cl_ld(HID), NTH = cl_kg(HID);
    static constexpr int o_w5 = 0, o_w4 = NTH * 16 * LD5, o_w3 = o_w4 + NTH * 16 * LD, FLOATS = o_w3 + NTH * 16 * LD;
};

template <int HID, int OUT, int CHAIN_THREADS>
__global__ void __launch_bounds__(CHAIN_THREADS) k_deform_b_bwd(const float *__restrict__ gy, const float *__restrict__ z4,
                                                                const float *__restrict__ z3, const float *__restrict__ z2, DeformW w,
                                                                float *__restrict__ g4, float *__restrict__ g3, float *__restrict__ g2,
                                                                long long M)
{
    extern __shared__ float lds[];
    using S = DeformBBwdLds<HID, OUT>;
    const Lane L;
    zero_lds(lds, S::FLOATS, L.tid, CHAIN_THREADS);
    __syncthreads();
    stage_block<true>(w.W[4], HID, OUT, 0, HID, lds + S::o_w5, S::LD5, 0, L.tid, CHAIN_THREADS);
    stage_block<true>(w.W[3], HID, HID, 0, HID, lds + S::o_w4, S::LD, 0, L.tid, CHAIN_THREADS);
    stage_block<true>(w.W[2], HID, HID, 0, HID, lds + S::o_w3, S::LD, 0, L.tid, CHAIN_THREADS);
    __syncthreads();
    long long rb, RB, stride;
    row_blocks<CHAIN_THREADS / 64>(M, L, rb, RB, stride);
    v4f g[cl_kg(OUT)];
    load_frags<OUT>(g, gy, rb, RB, M, L);
    for (; rb < RB; rb += stride) {
        v4f d4[S::NTH], d3[S::NTH], d2[S::NTH];      // the three pre-activations of this block, requested up front
        {
            const Tiles<HID> t4(z4, rb, RB, M, L), t3(z3, rb, RB, M, L), t2(z2, rb, RB, M, L);
#pragma unroll
            for (int t = 0; t < S::NTH; t++) { d4[t] = t4.load(t); d3[t] = t3.load(t); d2[t] = t2.load(t); }
        }
        __builtin_amdgcn_sched_barrier(0);
        v4f p[S::NTH], q[S::NTH];
        init_zero(p);
        chain_mm<OUT, S::NTH, S::LD5>(p, g, lds + S::o_w5, L);
        load_frags<OUT>(g, gy, rb + stride, RB, M, L);
        __builtin_amdgcn_sched_barrier(0);
        {
            const Tiles<HID> to(g4, rb, RB, M, L);
#pragma unroll
            for (int t = 0; t < S::NTH; t++) {
                const v4f d = v_gelu_grad(d4[t]);
#pragma unroll
                for (int i = 0; i < 4; i++) p[t][i] *= d[i];
                to.store(t, p[t]);
            }
        }
        init_zero(q);
        chain_mm<HID, S::NTH, S::LD>(q, p, lds + S::o_w4, L);
        {
            const Tiles<HID> to(g3, rb, RB, M, L);
#pragma unroll
            for (int t = 0; t < S::NTH; t++) {
                const v4f d = v_gelu_grad(d3[t]);
#pragma unroll
                for (int i = 0; i < 4; i++) q[t][i] *= d[i];
                to.store(t, q[t]);
            }
        }
        init_zero(p);
        chain_mm<HID, S::NTH, S::LD>(p, q, lds + S::o_w3, L);
        const Tiles<HID> to(g2, rb, RB, M, L);
#pragma unroll
        for (int t = 0; t < S::NTH; t++) {
            const v4f d = v_gelu_grad(d2[t]);
#pragma unroll
            for (int i = 0; i < 4; i++) p[t][i] *= d[i];
            to.store(t, p[t]);
        }
    }
}

// g1 = (g2 W2) * gelu'(z1);  gfeat (+)= g1 W1[:, :FEAT]   (the condition columns of W1 need no input gradient)
template <int FEAT, int HID, int CHAIN_THREADS>
__global__ void __launch_bounds__(CHAIN_THREADS) k_deform_a_bwd(const float *__restrict__ g2, const float *__restrict__ z1, DeformW w,
                                                                int ldw1, float *__restrict__ g1, float *__restrict__ gfeat,
                                                                int accumulate, const float *__restrict__ add0,
                                                                const float *__restrict__ add1, const float *__restrict__ add2, long long M)
{
    extern __shared__ float lds[];
    constexpr int LD = cl_ld(HID), NTH = cl_kg(HID), NTF = cl_kg(FEAT), o_w1 = NTH * 16 * LD, FLOATS = o_w1 + NTF * 16 * LD;
    const Lane L;
    zero_lds(lds, FLOATS, L.tid, CHAIN_THREADS);
    __syncthreads();
    stage_block<true>(w.W[1], HID, HID, 0, HID, lds, LD, 0, L.tid, CHAIN_THREADS);
    stage_block<true>(w.W[0], ldw1, HID, 0, FEAT, lds + o_w1, LD, 0, L.tid, CHAIN_THREADS);
    __syncthreads();
    long long rb, RB, stride;
    row_blocks<CHAIN_THREADS / 64>(M, L, rb, RB, stride);
    v4f g[NTH];
    load_frags<HID>(g, g2, rb, RB, M, L);
    for (; rb < RB; rb += stride) {
        v4f zz[NTH], pf[NTF];
        {
            const Tiles<HID> tz(z1, rb, RB, M, L);
#pragma unroll
            for (int t = 0; t < NTH; t++) zz[t] = tz.load(t);
            // the feature matrix feeds four networks: the generators' trunk backward leaves one gradient each (add0..2), this
            // kernel sums them on top of its own product (a NULL addend reads as zeros through an empty descriptor)
            const Tiles<FEAT> tf(gfeat, accumulate ? rb : RB, RB, M, L), t0(add0, add0 ? rb : RB, RB, M, L), t1(add1, add1 ? rb : RB, RB, M, L),
                t2(add2, add2 ? rb : RB, RB, M, L);
#pragma unroll
            for (int t = 0; t < NTF; t++) {
                const v4f a = tf.load(t), b = t0.load(t), c = t1.load(t), d = t2.load(t);
#pragma unroll
                for (int i = 0; i < 4; i++) pf[t][i] = (a[i] + b[i]) + (c[i] + d[i]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        v4f p[NTH];
        init_zero(p);
        chain_mm<HID, NTH, LD>(p, g, lds, L);
        load_frags<HID>(g, g2, rb + stride, RB, M, L);
        __builtin_amdgcn_sched_barrier(0);
        {
            const Tiles<HID> to(g1, rb, RB, M, L);
#pragma unroll
            for (int t = 0; t < NTH; t++) {
                const v4f d = v_gelu_grad(zz[t]);
#pragma unroll
                for (int i = 0; i < 4; i++) p[t][i] *= d[i];
                to.store(t, p[t]);
            }
        }
        chain_mm<HID, NTF, LD>(pf, p, lds + o_w1, L);      // on top of the running sum (or zeros)
        const Tiles<FEAT> of(gfeat, rb, RB, M, L);
#pragma unroll
        for (int t = 0; t < NTF; t++) of.store(t, pf[t]);
    }
}

// ======================================================================================================= quant_step nets
// The three quant_step networks of the EntropyParamsNets (reference scene/gaussian_model.py:198-232: Linear(192 -> 50), GELU,
// Linear(50 -> 1)) read the same hash-grid feature row and produce one number each.  Layer by layer that was one multi-product
// launch for the first layers, three layer launches with a one-column output (a 16 x 16 MFMA tile for one useful column) and,
// backward, three more plus a multi-product launch — ~210 us per fitting step for 0.03 GFLOP per thousand rows.  Here a wave
// carries a 16-row block through all three networks: the feature fragments are read once, each first layer is 192 MFMAs against
// its weight image in LDS (all three resident: 3 x 51 KB), GELU on the accumulators, the second layer a dot product with the
// lane's 16 columns and two cross-lane adds.  The backward forms dz = dq w2 GELU'(z) in the same tile layout — which is the
// B fragment of dX += dz W1 — against the three transposed images (row stride 56: 3 x 43 KB) and writes dX once.
struct QuantFwd {
    const float *W1[3], *b1[3], *W2[3], *b2[3];
    float *z[3], *a[3], *q[3];
};
struct QuantBwd {
    const float *W1[3], *W2[3], *z[3], *dq[3];
    float *dz[3], *dX;
};

template <int IN, int HQ>
struct QuantLds {
    static constexpr int LD = cl_ld(IN), NT = cl_kg(HQ), IMG = NT * 16 * LD;
    static constexpr int o_b1 = 3 * IMG, o_w2 = o_b1 + 3 * NT * 16, FLOATS = o_w2 + 3 * NT * 16;
    static constexpr int LDT = 16 * (cl_kg(HQ) - 1) + 8, NTI = cl_kg(IN), IMGT = NTI * 16 * LDT;      // transposed: [IN][hidden], stride 56
    static constexpr int o_w2t = 3 * IMGT, FLOATS_T = o_w2t + 3 * NT * 16 + 16;
};

template <int IN, int HQ, int CHAIN_THREADS>
__global__ void __launch_bounds__(CHAIN_THREADS) k_quant_nets_fwd(const float *__restrict__ X, QuantFwd p, long long M)
{
    extern __shared__ float lds[];
    using S = QuantLds<IN, HQ>;
    const Lane L;
    zero_lds(lds, S::FLOATS, L.tid, CHAIN_THREADS);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 3; i++) {
        stage_block<false>(p.W1[i], IN, HQ, 0, IN, lds + i * S::IMG, S::LD, 0, L.tid, CHAIN_THREADS);
        stage_bias(p.b1[i], HQ, lds + S::o_b1 + i * S::NT * 16, S::NT * 16, L.tid, CHAIN_THREADS);
        stage_bias(p.W2[i], HQ, lds + S::o_w2 + i * S::NT * 16, S::NT * 16, L.tid, CHAIN_THREADS);
    }
    __syncthreads();
    long long rb, RB, stride;
    row_blocks<CHAIN_THREADS / 64>(M, L, rb, RB, stride);
    for (; rb < RB; rb += stride) {
        const long long row = rb * 16 + L.fr;
        // (a wave has one or two blocks: the block's fragments are requested here, not a block ahead — the workgroup's other
        // seven waves cover the round trip)
        v4f x[cl_kg(IN)];
        load_frags<IN>(x, X, rb, RB, M, L);
#pragma unroll
        for (int i = 0; i < 3; i++) {
            v4f u[S::NT];
            init_bias(u, lds + S::o_b1 + i * S::NT * 16, L);
            chain_mm<IN, S::NT, S::LD>(u, x, lds + i * S::IMG, L);
            const Tiles<HQ> tz(p.z[i], rb, RB, M, L), ta(p.a[i], rb, RB, M, L);
            const float *w2 = lds + S::o_w2 + i * S::NT * 16 + 4 * L.kq;
            float part = 0.f;
#pragma unroll
            for (int t = 0; t < S::NT; t++) {
                tz.store(t, u[t]);
                const v4f g = v_gelu(u[t]);
                ta.store(t, g);
                const float4 w = *reinterpret_cast<const float4 *>(w2 + 16 * t);
                part = fmaf(g[0], w.x, fmaf(g[1], w.y, fmaf(g[2], w.z, fmaf(g[3], w.w, part))));
            }
            // the row's 64 (padded) columns live in the four lanes fr, fr + 16, fr + 32, fr + 48
            part += __shfl_xor(part, 16, 64);
            part += __shfl_xor(part, 32, 64);
            if (L.kq == 0 && row < M) p.q[i][row] = part + p.b2[i][0];
        }
    }
}

template <int IN, int HQ, int CHAIN_THREADS>
__global__ void __launch_bounds__(CHAIN_THREADS) k_quant_nets_bwd(QuantBwd p, long long M)
{
    extern __shared__ float lds[];
    using S = QuantLds<IN, HQ>;
    const Lane L;
    zero_lds(lds, S::FLOATS_T, L.tid, CHAIN_THREADS);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 3; i++) {
        // imgT[k][n] = W1[n][k]: contraction over the hidden index n, output = the feature column k
        stage_block<true>(p.W1[i], IN, HQ, 0, IN, lds + i * S::IMGT, S::LDT, 0, L.tid, CHAIN_THREADS);
        stage_bias(p.W2[i], HQ, lds + S::o_w2t + i * S::NT * 16, S::NT * 16, L.tid, CHAIN_THREADS);
    }
    __syncthreads();
    long long rb, RB, stride;
    row_blocks<CHAIN_THREADS / 64>(M, L, rb, RB, stride);
    for (; rb < RB; rb += stride) {
        const long long row = rb * 16 + L.fr;
        v4f gx[S::NTI];
        init_zero(gx);
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const Tiles<HQ> tz(p.z[i], rb, RB, M, L), td(p.dz[i], rb, RB, M, L);
            const float dq = (p.dq[i] && row < M) ? p.dq[i][row] : 0.f;      // an output nobody used: zeros
            const float *w2 = lds + S::o_w2t + i * S::NT * 16 + 4 * L.kq;
            v4f d[S::NT];
#pragma unroll
            for (int t = 0; t < S::NT; t++) {
                const v4f gg = v_gelu_grad(tz.load(t));
                const float4 w = *reinterpret_cast<const float4 *>(w2 + 16 * t);      // zero past the hidden width: so is dz there
                d[t] = (v4f){dq * w.x * gg[0], dq * w.y * gg[1], dq * w.z * gg[2], dq * w.w * gg[3]};
                td.store(t, d[t]);
            }
            // the image's row stride (56) is shorter than the last k-group's 16-column read: the lanes kq > 0 of that group read into
            // the next row (finite numbers) against dz columns >= 52, which are zero
            chain_mm<HQ, S::NTI, S::LDT>(gx, d, lds + i * S::IMGT, L);
        }
        const Tiles<IN> tx(p.dX, rb, RB, M, L);
#pragma unroll
        for (int t = 0; t < S::NTI; t++) tx.store(t, gx[t]);
    }
}


// ---- host side ----------------------------------------------------------------------------------------------------------
// One launch of a chain kernel: persistent grid (one workgroup per CU), 512 threads = 2 waves per SIMD with up to 256 VGPRs each
// (the operands a wave keeps in flight across its products need 140-240 of them; 1024-thread builds of the first version,
// without those prefetches, were no faster: the kernels were serialising memory and MFMA phases, not short of waves).
constexpr int CHAIN_T = 512;

template <typename... KA, typename... A>
void chain_launch(const char *name, void (*k)(KA...), size_t lds, long long M, int min_grid, hipStream_t s, A... args)
{
    static std::mutex mu;
    static std::set<const void *> done;
    {
        std::lock_guard<std::mutex> g(mu);
        if (done.insert(reinterpret_cast<const void *>(k)).second)
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    const int waves = CHAIN_T / 64;
    const long long RB = (M + 15) / 16, want = (RB + waves - 1) / waves;
    const unsigned grid = (unsigned)(want < 256 ? (want < min_grid ? min_grid : want) : 256);      // every network of a batch gets a workgroup
    ProfScope _p(name, s);
    hipLaunchKernelGGL(k, dim3(grid), dim3(CHAIN_T), lds, s, static_cast<KA>(args)...);
}

constexpr int FEAT = 50, COND = 66, HID = 100;

bool aligned16(std::initializer_list<const void *> ps)
{
    for (const void *p : ps)
        if (reinterpret_cast<uintptr_t>(p) & 15) return false;
    return true;
}

// carve `count` floats (rounded up to a multiple of 4: every matrix starts 16-byte aligned) from a cursor
inline float *take(float *&cur, long long count)
{
    float *p = cur;
    cur += (count + 3) / 4 * 4;
    return p;
}

// per-row floats of the tensors a generator's forward leaves for its backward, in order: cg, cb [COND], gamma [HID],
// a1, h, x3 [HID] (z1 is not kept: the backward forms it again from the features); beta (a forward intermediate) lives behind them
struct GenSaved {
    float *cg, *cb, *gamma, *z1, *a1, *h, *x3, *beta;
    GenSaved(float *base, long long M, long long Mf, bool inference = false)      // Mf: rows of the FiLM networks (= M unless the views share them)
    {
        float *cur = base;
        gamma = take(cur, Mf * HID); beta = take(cur, Mf * HID);
        if (inference) {       // forward only: gamma / beta travel from the FiLM kernel to the trunk kernel, nothing else is kept
            cg = cb = z1 = a1 = h = x3 = nullptr;
            return;
        }
        cg = take(cur, Mf * COND); cb = take(cur, Mf * COND);
        z1 = nullptr;
        a1 = take(cur, M * HID); h = take(cur, M * HID); x3 = take(cur, M * HID);
    }
};
constexpr long long GEN_SAVED_PER_ROW = 3 * HID, GEN_SAVED_PER_FILM_ROW = 2 * COND + 2 * HID;

// backward scratch of a generator: go [OUT], gbeta, ggamma, gh, gz1 [HID] per chain row; gcg, gcb [COND] and (shared FiLM rows)
// the two views' summed gbeta / ggamma [HID] per FiLM row; then the wgrad partial sums
struct GenScratch {
    float *go, *gbeta, *ggamma, *gh, *gz1, *gcg, *gcb, *gbeta_sum, *ggamma_sum, *wg;
    GenScratch(float *base, long long M, long long Mf, int out, bool shared)
    {
        float *cur = base;
        go = take(cur, M * out); gbeta = take(cur, M * HID); ggamma = take(cur, M * HID); gh = take(cur, M * HID);
        gz1 = take(cur, M * HID); gcg = take(cur, Mf * COND); gcb = take(cur, Mf * COND);
        gbeta_sum = shared ? take(cur, Mf * HID) : gbeta;
        ggamma_sum = shared ? take(cur, Mf * HID) : ggamma;
        wg = cur;
    }
};

long long gen_wgrad_floats(int out)
{
    return gsvc_linear_wgrad_workspace(HID, FEAT) + gsvc_linear_wgrad_workspace(HID, HID) + gsvc_linear_wgrad_workspace(out, HID) +
           2 * gsvc_linear_wgrad_workspace(COND, COND) + 2 * gsvc_linear_wgrad_workspace(HID, COND);
}

struct FilmRows {      // resolved gsvc_film_rows (rows == M and NULL maps when the views do not share the FiLM rows)
    long long rows;
    const float *cond;
    const int *row_of, *src_a, *src_b;
    bool shared;
    FilmRows(const gsvc_film_rows *f, long long M, const float *cond_rows)
    {
        shared = f != nullptr && f->rows > 0;
        rows = shared ? f->rows : M;
        cond = shared ? f->cond : cond_rows;
        row_of = shared ? f->row_of : nullptr;
        src_a = shared ? f->src_a : nullptr;
        src_b = shared ? f->src_b : nullptr;
    }
};

int generators_forward(const gsvc_generator_net *nets, int n, const float *feat, const float *cond, long long M, const gsvc_film_rows *film,
                       float *const *saved, float *const *y, hipStream_t s, bool inference = false)
{
    const FilmRows fr(film, M, cond);
    FilmFwdBatch fb;
    TrunkFwdBatch tb;
    fb.n = tb.n = n;
    tb.film_row = fr.row_of;
    tb.film_rows = fr.rows;
    for (int i = 0; i < MAX_NETS; i++) {
        const gsvc_generator_net &g = nets[i < n ? i : 0];
        const GenSaved sv(saved[i < n ? i : 0], M, fr.rows, inference);
        fb.w[i] = FilmW{g.Wg0, g.bg0, g.Wg1, g.bg1, g.Wb0, g.bb0, g.Wb1, g.bb1};
        fb.cg[i] = sv.cg; fb.cb[i] = sv.cb; fb.gamma[i] = sv.gamma; fb.beta[i] = sv.beta;
        tb.out[i] = g.out_dim; tb.act[i] = g.out_act;
        tb.w[i] = TrunkW{g.W1, g.b1, g.W2, g.b2, g.W3, g.b3};
        tb.gamma[i] = sv.gamma; tb.beta[i] = sv.beta; tb.z1[i] = sv.z1; tb.a1[i] = sv.a1; tb.h[i] = sv.h; tb.x3[i] = sv.x3;
        tb.y[i] = y[i < n ? i : 0];
    }
    chain_launch("k_film_nets_fwd", &k_film_nets_fwd<COND, HID, CHAIN_T>, FilmFwdLds<COND, HID>::FLOATS * 4, fr.rows * n, n, s, fr.cond, fb, fr.rows);
    chain_launch("k_trunk_fwd", &k_trunk_fwd<FEAT, HID, CHAIN_T>, TrunkFwdLds<FEAT, HID, 70>::FLOATS * 4, M * n, n, s, feat, tb, M);
    return check_launch("generators_forward");
}

// The weight gradients of a backward pass read what its chain kernels wrote and nothing downstream reads THEM before the optimizer:
// with gsvc_set_wgrad_stream they are queued on that stream behind an event of the caller's, and the caller's stream goes on with
// the feature gradient (the quantisers', gathers' and hash grid's backward) while they run.
hipStream_t g_wgrad_stream = nullptr;

// gsvc_wgrad_hold(1): the entries keep their weight-gradient launches back (operands by value) until gsvc_wgrad_flush — so that a
// caller with several chain backward passes in a row (the generators', then mlp_deform's) gets every chain kernel onto its stream
// BEFORE the first product starts beside them: a chain workgroup needs its whole CU, a product queued in between made the second
// network's chain kernels wait for the first network's products to retire.
struct DeferredWgrad {
    std::vector<gsvc_wgrad_partial_job> part;
    std::vector<gsvc_wgrad_reduce_job> red;
    std::function<void(hipStream_t)> tail;
};
bool g_wgrad_hold = false;
std::vector<DeferredWgrad> g_wgrad_deferred;

int run_wgrad(gsvc_wgrad_partial_job *part, gsvc_wgrad_reduce_job *red, int n, hipStream_t s)
{
    if (int rc = gsvc_linear_wgrad_partial_many(part, n, s)) return rc;
    for (int i = 0; i < n; i++) red[i].slots = part[i].slots_used;
    return gsvc_linear_wgrad_reduce_many(red, n, s);
}

hipStream_t wgrad_stream_behind(hipStream_t s)
{
    if (g_wgrad_stream == nullptr || g_wgrad_stream == s) return s;
    constexpr int POOL = 32;
    static hipEvent_t pool[POOL];
    static int at = -1;
    if (at < 0) {
        for (int i = 0; i < POOL; i++)
            if (hipEventCreateWithFlags(&pool[i], hipEventDisableTiming) != hipSuccess) return s;
        at = 0;
    }
    hipEvent_t e = pool[at];
    at = (at + 1) % POOL;
    if (hipEventRecord(e, s) != hipSuccess || hipStreamWaitEvent(g_wgrad_stream, e, 0) != hipSuccess) return s;
    return g_wgrad_stream;
}

int generators_backward(const gsvc_generator_net *nets, int n, const float *feat, const float *cond, long long M, const gsvc_film_rows *film,
                        const float *const *saved, const float *const *y, const float *const *gy, float *scratch, float *const *gfeat,
                        const int *accumulate, const gsvc_generator_grads *grads, hipStream_t s)
{
    const FilmRows fr(film, M, cond);
    TrunkBwdBatch tb;
    FilmBwdBatch fb;
    tb.n = fb.n = n;
    tb.feat = feat;
    tb.film_row = fr.row_of;
    tb.film_rows = fr.rows;
    fb.src_a = fr.src_a; fb.src_b = fr.src_b; fb.src_rows = M;
    float *sbase[MAX_NETS];
    {
        float *cur = scratch;
        for (int i = 0; i < n; i++) {
            sbase[i] = cur;
            cur += (gsvc_generator_scratch_floats(&nets[i], M, fr.shared ? fr.rows : 0) + 3) / 4 * 4;
        }
    }
    for (int i = 0; i < MAX_NETS; i++) {
        const int j = i < n ? i : 0;
        const gsvc_generator_net &g = nets[j];
        const GenSaved sv(const_cast<float *>(saved[j]), M, fr.rows);
        const GenScratch sc(sbase[j], M, fr.rows, g.out_dim, fr.shared);
        tb.out[i] = g.out_dim; tb.act[i] = g.out_act; tb.accumulate[i] = accumulate ? accumulate[j] : 0;
        tb.w[i] = TrunkW{g.W1, g.b1, g.W2, g.b2, g.W3, g.b3};
        tb.gy[i] = gy[j]; tb.y[i] = y[j]; tb.h[i] = sv.h; tb.gamma[i] = sv.gamma;
        tb.go[i] = sc.go; tb.gbeta[i] = sc.gbeta; tb.ggamma[i] = sc.ggamma; tb.gh[i] = sc.gh; tb.gz1[i] = sc.gz1; tb.gfeat[i] = gfeat[j];
        fb.ggamma[i] = sc.ggamma; fb.gbeta[i] = sc.gbeta; fb.cg[i] = sv.cg; fb.cb[i] = sv.cb; fb.Wg1[i] = g.Wg1; fb.Wb1[i] = g.Wb1;
        fb.gcg[i] = sc.gcg; fb.gcb[i] = sc.gcb; fb.ggamma_sum[i] = sc.ggamma_sum; fb.gbeta_sum[i] = sc.gbeta_sum;
    }
    chain_launch("k_trunk_bwd", &k_trunk_bwd<FEAT, HID, CHAIN_T>, TrunkBwdLds<FEAT, HID, 70>::FLOATS * 4, M * n, n, s, tb, M);
    chain_launch("k_film_nets_bwd", &k_film_nets_bwd<COND, HID, CHAIN_T>, (size_t)2 * cl_kg(COND) * 16 * cl_ld(HID) * 4, fr.rows * n, n, s, fb,
                 fr.rows);
    if (int rc = check_launch("generators_backward")) return rc;
    // the seven weight gradients dW = G^T X (+ db) of every network: row-split partial sums, batched slot reduces.  The FiLM
    // networks' four run over the FiLM rows (the summed gradients of the two views when they share them)
    struct Job { const float *G, *X; float *dW, *db; int N, K; long long rows; };
    gsvc_wgrad_partial_job part[7 * MAX_NETS];
    gsvc_wgrad_reduce_job red[7 * MAX_NETS];
    int nred = 0;
    for (int i = 0; i < n; i++) {
        const gsvc_generator_net &g = nets[i];
        const gsvc_generator_grads &gr = grads[i];
        const GenSaved sv(const_cast<float *>(saved[i]), M, fr.rows);
        const GenScratch sc(sbase[i], M, fr.rows, g.out_dim, fr.shared);
        const Job jobs[7] = {
            {sc.gz1, feat, gr.W1, gr.b1, HID, FEAT, M},        {sc.gh, sv.a1, gr.W2, gr.b2, HID, HID, M},
            {sc.go, sv.x3, gr.W3, gr.b3, g.out_dim, HID, M},   {sc.gcg, fr.cond, gr.Wg0, gr.bg0, COND, COND, fr.rows},
            {sc.ggamma_sum, sv.cg, gr.Wg1, gr.bg1, HID, COND, fr.rows}, {sc.gcb, fr.cond, gr.Wb0, gr.bb0, COND, COND, fr.rows},
            {sc.gbeta_sum, sv.cb, gr.Wb1, gr.bb1, HID, COND, fr.rows}};
        float *ws = sc.wg;
        for (const Job &j : jobs) {
            if (!j.dW) continue;
            const long long need = gsvc_linear_wgrad_workspace(j.N, j.K);
            part[nred] = gsvc_wgrad_partial_job{j.G, j.X, ws, j.rows, need, j.db != nullptr, j.N, j.K, 0};
            red[nred++] = gsvc_wgrad_reduce_job{ws, j.dW, j.db, 0, j.N, j.K};
            ws += need;
        }
    }
    if (nred) {
        if (g_wgrad_hold && g_wgrad_stream != nullptr) {
            g_wgrad_deferred.push_back(DeferredWgrad{std::vector<gsvc_wgrad_partial_job>(part, part + nred),
                                                     std::vector<gsvc_wgrad_reduce_job>(red, red + nred), nullptr});
            return GSVC_OK;
        }
        return run_wgrad(part, red, nred, wgrad_stream_behind(s));
    }
    return GSVC_OK;
}

constexpr int DEF_OUT = 30;
constexpr long long DEF_SAVED_PER_ROW = 8 * HID;      // z1 a1 z2 a2 z3 a3 z4 a4
constexpr long long DEF_SCRATCH_PER_ROW = 4 * HID;    // g1 g2 g3 g4

long long deform_wgrad_floats()
{
    return gsvc_linear_wgrad_workspace(HID, FEAT) + gsvc_linear_wgrad_workspace(HID, COND) + 3 * gsvc_linear_wgrad_workspace(HID, HID) +
           gsvc_linear_wgrad_workspace(DEF_OUT, HID);
}

}  // namespace
}  // namespace gsvc

using namespace gsvc;

static int gen_supported(const gsvc_generator_net *n, const char *what)
{
    GSVC_REQUIRE(n, "%s: NULL network", what);
    if (n->feat_dim != FEAT || n->cond_dim != COND || n->hidden_dim != HID || (n->out_dim != 10 && n->out_dim != 30 && n->out_dim != 70) ||
        n->out_act < CH_ACT_NONE || n->out_act > CH_ACT_SIGMOID) {
        set_error("%s: widths feat %d / cond %d / hidden %d / out %d (act %d) have no chain-kernel instantiation", what, n->feat_dim,
                  n->cond_dim, n->hidden_dim, n->out_dim, n->out_act);
        return GSVC_E_UNSUPPORTED;
    }
    GSVC_REQUIRE(n->W1 && n->b1 && n->W2 && n->b2 && n->W3 && n->b3 && n->Wg0 && n->bg0 && n->Wg1 && n->bg1 && n->Wb0 && n->bb0 && n->Wb1 &&
                     n->bb1, "%s: NULL weight pointer", what);
    return GSVC_OK;
}

extern "C" int64_t gsvc_generator_saved_floats(const gsvc_generator_net *n, int64_t M, int64_t film_rows)
{
    if (!n || M < 0 || film_rows < 0) return -1;
    return GEN_SAVED_PER_ROW * M + GEN_SAVED_PER_FILM_ROW * (film_rows > 0 ? film_rows : M) + 64;
}

extern "C" int64_t gsvc_generator_scratch_floats(const gsvc_generator_net *n, int64_t M, int64_t film_rows)
{
    if (!n || M < 0 || film_rows < 0) return -1;
    const int64_t Mf = film_rows > 0 ? film_rows : M;
    return (int64_t)(n->out_dim + 4 * HID) * M + (int64_t)(2 * COND + (film_rows > 0 ? 2 * HID : 0)) * Mf + gen_wgrad_floats(n->out_dim) + 96;
}

static int gens_supported(const gsvc_generator_net *nets, int32_t n, const char *what)
{
    GSVC_REQUIRE(nets && n >= 1 && n <= MAX_NETS, "%s: 1 .. %d networks per call", what, MAX_NETS);
    for (int i = 0; i < n; i++)
        if (int rc = gen_supported(&nets[i], what)) return rc;
    return GSVC_OK;
}

static int film_rows_ok(const gsvc_film_rows *f, int64_t M, const char *what)
{
    if (!f || f->rows <= 0) return GSVC_OK;
    GSVC_REQUIRE(f->cond && f->row_of && f->src_a && f->src_b && aligned16({f->cond}), "%s: shared FiLM rows need cond, row_of, src_a, src_b", what);
    GSVC_REQUIRE((M > f->rows ? M : f->rows) * (int64_t)HID * 4 < ((int64_t)1 << 31), "%s: %lld rows exceed the 2 GiB reach of the row maps", what,
                 (long long)(M > f->rows ? M : f->rows));
    return GSVC_OK;
}

extern "C" int gsvc_generators_forward(const gsvc_generator_net *nets, int32_t n_nets, const float *feat, const float *cond, int64_t M,
                                       const gsvc_film_rows *film, float *const *saved, float *const *y, void *stream)
{
    if (int rc = gens_supported(nets, n_nets, "generators_forward")) return rc;
    GSVC_REQUIRE(M >= 0, "generators_forward: bad row count");
    if (M == 0) return GSVC_OK;
    GSVC_REQUIRE(feat && cond && saved && y && aligned16({feat, cond}), "generators_forward: NULL or unaligned pointer");
    if (int rc = film_rows_ok(film, M, "generators_forward")) return rc;
    for (int i = 0; i < n_nets; i++)
        GSVC_REQUIRE(saved[i] && y[i] && aligned16({saved[i], y[i]}), "generators_forward: NULL or unaligned pointer (network %d)", i);
    return generators_forward(nets, n_nets, feat, cond, M, film, saved, y, (hipStream_t)stream);
}

extern "C" int64_t gsvc_generator_inference_floats(const gsvc_generator_net *n, int64_t M, int64_t film_rows)
{
    if (!n || M < 0 || film_rows < 0) return -1;
    return 2 * (int64_t)HID * (film_rows > 0 ? film_rows : M) + 64;
}

// forward only: nothing the backward would read is written (the trunk of a generator stores four [M, 100] activations per
// network for it: 60 % of the forward's traffic)
extern "C" int gsvc_generators_forward_inference(const gsvc_generator_net *nets, int32_t n_nets, const float *feat, const float *cond, int64_t M,
                                                 const gsvc_film_rows *film, float *const *scratch, float *const *y, void *stream)
{
    if (int rc = gens_supported(nets, n_nets, "generators_forward_inference")) return rc;
    GSVC_REQUIRE(M >= 0, "generators_forward_inference: bad row count");
    if (M == 0) return GSVC_OK;
    GSVC_REQUIRE(feat && cond && scratch && y && aligned16({feat, cond}), "generators_forward_inference: NULL or unaligned pointer");
    if (int rc = film_rows_ok(film, M, "generators_forward_inference")) return rc;
    for (int i = 0; i < n_nets; i++)
        GSVC_REQUIRE(scratch[i] && y[i] && aligned16({scratch[i], y[i]}), "generators_forward_inference: NULL or unaligned pointer (network %d)", i);
    return generators_forward(nets, n_nets, feat, cond, M, film, scratch, y, (hipStream_t)stream, true);
}

extern "C" int gsvc_generators_backward(const gsvc_generator_net *nets, int32_t n_nets, const float *feat, const float *cond, int64_t M,
                                        const gsvc_film_rows *film, const float *const *saved, const float *const *y,
                                        const float *const *gy, float *scratch, float *const *gfeat, const gsvc_generator_grads *grads,
                                        void *stream)
{
    if (int rc = gens_supported(nets, n_nets, "generators_backward")) return rc;
    GSVC_REQUIRE(M >= 0 && grads, "generators_backward: bad arguments");
    if (M == 0) return GSVC_OK;
    GSVC_REQUIRE(feat && cond && saved && y && gy && scratch && gfeat && aligned16({feat, cond, scratch}),
                 "generators_backward: NULL or unaligned pointer");
    if (int rc = film_rows_ok(film, M, "generators_backward")) return rc;
    for (int i = 0; i < n_nets; i++) {
        GSVC_REQUIRE(saved[i] && y[i] && gy[i] && gfeat[i] && aligned16({saved[i], y[i], gy[i], gfeat[i]}),
                     "generators_backward: NULL or unaligned pointer (network %d)", i);
        for (int j = 0; j < i; j++) GSVC_REQUIRE(gfeat[i] != gfeat[j], "generators_backward: the networks' feature gradients must be distinct buffers");
    }
    return generators_backward(nets, n_nets, feat, cond, M, film, saved, y, gy, scratch, gfeat, nullptr, grads, (hipStream_t)stream);
}

extern "C" int gsvc_generator_forward(const gsvc_generator_net *n, const float *feat, const float *cond, int64_t M, float *saved, float *y,
                                      void *stream)
{
    return gsvc_generators_forward(n, 1, feat, cond, M, nullptr, &saved, &y, stream);
}

extern "C" int gsvc_generator_backward(const gsvc_generator_net *n, const float *feat, const float *cond, int64_t M, const float *saved,
                                       const float *y, const float *gy, float *scratch, float *gfeat, int32_t accumulate_gfeat,
                                       const gsvc_generator_grads *grads, void *stream)
{
    if (int rc = gens_supported(n, 1, "generator_backward")) return rc;
    GSVC_REQUIRE(M >= 0 && grads, "generator_backward: bad arguments");
    if (M == 0) return GSVC_OK;
    GSVC_REQUIRE(feat && cond && saved && y && gy && scratch && gfeat, "generator_backward: NULL pointer");
    GSVC_REQUIRE(aligned16({feat, cond, saved, y, gy, scratch, gfeat}), "generator_backward: operands must be 16-byte aligned");
    const int acc = accumulate_gfeat;
    return generators_backward(n, 1, feat, cond, M, nullptr, &saved, &y, &gy, scratch, &gfeat, &acc, grads, (hipStream_t)stream);
}

extern "C" int gsvc_set_wgrad_stream(void *stream)
{
    gsvc::g_wgrad_stream = (hipStream_t)stream;
    return GSVC_OK;
}

extern "C" int gsvc_wgrad_hold(int32_t on)
{
    gsvc::g_wgrad_hold = on != 0;
    return GSVC_OK;
}

extern "C" int gsvc_wgrad_flush(void *stream)
{
    if (gsvc::g_wgrad_deferred.empty()) return GSVC_OK;
    hipStream_t ws = gsvc::g_wgrad_stream != nullptr ? gsvc::wgrad_stream_behind((hipStream_t)stream) : (hipStream_t)stream;
    int rc = GSVC_OK;
    for (auto &d : gsvc::g_wgrad_deferred) {
        if (rc == GSVC_OK) rc = gsvc::run_wgrad(d.part.data(), d.red.data(), (int)d.part.size(), ws);
        if (rc == GSVC_OK && d.tail) d.tail(ws);
    }
    gsvc::g_wgrad_deferred.clear();
    return rc != GSVC_OK ? rc : check_launch("wgrad_flush");
}

static int deform_supported(const gsvc_deform_net *n, const char *what)
{
    GSVC_REQUIRE(n, "%s: NULL network", what);
    if (n->feat_dim != FEAT || n->cond_dim != COND || n->hidden_dim != HID || n->out_dim != DEF_OUT) {
        set_error("%s: widths feat %d / cond %d / hidden %d / out %d have no chain-kernel instantiation", what, n->feat_dim, n->cond_dim,
                  n->hidden_dim, n->out_dim);
        return GSVC_E_UNSUPPORTED;
    }
    for (int i = 0; i < 5; i++) GSVC_REQUIRE(n->W[i] && n->b[i], "%s: NULL weight pointer (layer %d)", what, i);
    return GSVC_OK;
}

extern "C" int64_t gsvc_deform_saved_floats(const gsvc_deform_net *n, int64_t M)
{
    if (!n || M < 0) return -1;
    return DEF_SAVED_PER_ROW * M;
}

extern "C" int64_t gsvc_deform_scratch_floats(const gsvc_deform_net *n, int64_t M)
{
    if (!n || M < 0) return -1;
    return DEF_SCRATCH_PER_ROW * M + deform_wgrad_floats() + 2 * (int64_t)HID * (FEAT + COND) + 64;
}

extern "C" int gsvc_deform_forward(const gsvc_deform_net *n, const float *feat, const float *cond, int64_t M, float *saved, float *y,
                                   void *stream)
{
    if (int rc = deform_supported(n, "deform_forward")) return rc;
    GSVC_REQUIRE(M >= 0, "deform_forward: bad row count");
    if (M == 0) return GSVC_OK;
    GSVC_REQUIRE(feat && cond && saved && y, "deform_forward: NULL pointer");
    GSVC_REQUIRE(aligned16({feat, cond, saved, y}), "deform_forward: operands must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    DeformW w;
    for (int i = 0; i < 5; i++) { w.W[i] = n->W[i]; w.b[i] = n->b[i]; }
    float *z1 = saved, *a1 = z1 + M * HID, *z2 = a1 + M * HID, *a2 = z2 + M * HID, *z3 = a2 + M * HID, *a3 = z3 + M * HID,
          *z4 = a3 + M * HID, *a4 = z4 + M * HID;
    chain_launch("k_deform_a_fwd", &k_deform_a_fwd<FEAT, COND, HID, CHAIN_T>, DeformALds<FEAT, COND, HID>::FLOATS * 4, M, 1, s, feat, cond, w, z1, a1, z2, a2, M);
    chain_launch("k_deform_b_fwd", &k_deform_b_fwd<HID, DEF_OUT, CHAIN_T>, DeformBLds<HID, DEF_OUT>::FLOATS * 4, M, 1, s,
                 a2, w, z3, a3, z4, a4, y, M);
    return check_launch("deform_forward");
}

// forward only: scratch = HID * M floats (the activation that travels from the first kernel to the second)
extern "C" int gsvc_deform_forward_inference(const gsvc_deform_net *n, const float *feat, const float *cond, int64_t M, float *scratch, float *y,
                                             void *stream)
{
    if (int rc = deform_supported(n, "deform_forward_inference")) return rc;
    GSVC_REQUIRE(M >= 0, "deform_forward_inference: bad row count");
    if (M == 0) return GSVC_OK;
    GSVC_REQUIRE(feat && cond && scratch && y && aligned16({feat, cond, scratch, y}), "deform_forward_inference: NULL or unaligned pointer");
    hipStream_t s = (hipStream_t)stream;
    DeformW w;
    for (int i = 0; i < 5; i++) { w.W[i] = n->W[i]; w.b[i] = n->b[i]; }
    float *none = nullptr;
    chain_launch("k_deform_a_fwd", &k_deform_a_fwd<FEAT, COND, HID, CHAIN_T>, DeformALds<FEAT, COND, HID>::FLOATS * 4, M, 1, s, feat, cond, w, none, none,
                 none, scratch, M);
    chain_launch("k_deform_b_fwd", &k_deform_b_fwd<HID, DEF_OUT, CHAIN_T>, DeformBLds<HID, DEF_OUT>::FLOATS * 4, M, 1, s,
                 (const float *)scratch, w, none, none, none, none, y, M);
    return check_launch("deform_forward_inference");
}

extern "C" int gsvc_deform_backward(const gsvc_deform_net *n, const float *feat, const float *cond, int64_t M, const float *saved,
                                    const float *gy, float *scratch, float *gfeat, int32_t accumulate_gfeat,
                                    const float *const *gfeat_addends, int32_t n_addends, const gsvc_deform_grads *grads, void *stream)
{
    if (int rc = deform_supported(n, "deform_backward")) return rc;
    GSVC_REQUIRE(M >= 0 && grads && n_addends >= 0 && n_addends <= 3 && (n_addends == 0 || gfeat_addends), "deform_backward: bad arguments");
    const float *add[3] = {nullptr, nullptr, nullptr};
    for (int i = 0; i < n_addends; i++) {
        GSVC_REQUIRE(gfeat_addends[i] && aligned16({gfeat_addends[i]}) && gfeat_addends[i] != gfeat, "deform_backward: bad addend %d", i);
        add[i] = gfeat_addends[i];
    }
    if (M == 0) return GSVC_OK;
    GSVC_REQUIRE(feat && cond && saved && gy && scratch && gfeat, "deform_backward: NULL pointer");
    GSVC_REQUIRE(aligned16({feat, cond, saved, gy, scratch, gfeat}),
                 "deform_backward: operands must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    DeformW w;
    for (int i = 0; i < 5; i++) { w.W[i] = n->W[i]; w.b[i] = n->b[i]; }
    const float *z1 = saved, *a1 = z1 + M * HID, *z2 = a1 + M * HID, *a2 = z2 + M * HID, *z3 = a2 + M * HID, *a3 = z3 + M * HID,
                *z4 = a3 + M * HID, *a4 = z4 + M * HID;
    float *g1 = scratch, *g2 = g1 + M * HID, *g3 = g2 + M * HID, *g4 = g3 + M * HID, *ws = g4 + M * HID;
    ws = reinterpret_cast<float *>((reinterpret_cast<uintptr_t>(ws) + 15) & ~uintptr_t(15));
    chain_launch("k_deform_b_bwd", &k_deform_b_bwd<HID, DEF_OUT, CHAIN_T>, DeformBBwdLds<HID, DEF_OUT>::FLOATS * 4,
                 M, 1, s, gy, z4, z3, z2, w, g4, g3, g2, M);
    chain_launch("k_deform_a_bwd", &k_deform_a_bwd<FEAT, HID, CHAIN_T>, (size_t)(cl_kg(HID) + cl_kg(FEAT)) * 16 * cl_ld(HID) * 4, M, 1, s, g2, z1, w, FEAT + COND, g1, gfeat, (int)accumulate_gfeat, add[0], add[1], add[2], M);
    if (int rc = check_launch("deform_backward")) return rc;
    // weight gradients; layer 1 = [g1^T feat | g1^T cond] formed as two products into a staging area, interleaved by the caller's
    // layout (grads->W[0] is [HID][FEAT + COND]): the reduce writes contiguous [N][K] blocks, so the two halves go to scratch
    // and a strided copy puts them side by side
    struct Job { const float *G, *X; float *dW, *db; int N, K; };
    float *stage_f = ws, *stage_c = stage_f + (size_t)HID * FEAT;
    ws = stage_c + (size_t)HID * COND;
    ws = reinterpret_cast<float *>((reinterpret_cast<uintptr_t>(ws) + 15) & ~uintptr_t(15));
    const bool want1 = grads->W[0] != nullptr;
    const Job jobs[6] = {{g1, feat, want1 ? stage_f : nullptr, grads->b[0], HID, FEAT}, {g1, cond, want1 ? stage_c : nullptr, nullptr, HID, COND},
                         {g2, a1, grads->W[1], grads->b[1], HID, HID},                 {g3, a2, grads->W[2], grads->b[2], HID, HID},
                         {g4, a3, grads->W[3], grads->b[3], HID, HID},                 {gy, a4, grads->W[4], grads->b[4], DEF_OUT, HID}};
    gsvc_wgrad_partial_job part[6];
    gsvc_wgrad_reduce_job red[6];
    int nred = 0;
    for (const Job &j : jobs) {
        if (!j.dW) continue;
        const long long need = gsvc_linear_wgrad_workspace(j.N, j.K);
        part[nred] = gsvc_wgrad_partial_job{j.G, j.X, ws, M, need, j.db != nullptr, j.N, j.K, 0};
        red[nred++] = gsvc_wgrad_reduce_job{ws, j.dW, j.db, 0, j.N, j.K};
        ws += need;
    }
    float *const w0 = grads->W[0];
    auto tail = [=](hipStream_t wst) {
        if (!want1) return;
        (void)hipMemcpy2DAsync(w0, (size_t)(FEAT + COND) * 4, stage_f, (size_t)FEAT * 4, (size_t)FEAT * 4, HID, hipMemcpyDeviceToDevice, wst);
        (void)hipMemcpy2DAsync(w0 + FEAT, (size_t)(FEAT + COND) * 4, stage_c, (size_t)COND * 4, (size_t)COND * 4, HID, hipMemcpyDeviceToDevice, wst);
    };
    if (nred && g_wgrad_hold && g_wgrad_stream != nullptr) {
        g_wgrad_deferred.push_back(DeferredWgrad{std::vector<gsvc_wgrad_partial_job>(part, part + nred),
                                                 std::vector<gsvc_wgrad_reduce_job>(red, red + nred), tail});
        return check_launch("deform_backward");
    }
    hipStream_t wst = nred ? wgrad_stream_behind(s) : s;
    if (nred)
        if (int rc = run_wgrad(part, red, nred, wst)) return rc;
    tail(wst);
    return check_launch("deform_backward");
}

// ---- quant_step nets ----------------------------------------------------------------------------------------------------------
namespace gsvc {
namespace {
constexpr int Q_IN = 192, Q_HID = 50;
}
}

extern "C" int gsvc_quant_step_nets_forward(const gsvc_quant_step_net *nets, const float *X, int64_t M, int32_t in_dim, int32_t hidden,
                                            float *const *z, float *const *a, float *const *q, void *stream)
{
    GSVC_REQUIRE(nets && z && a && q && M >= 0, "quant_step_nets_forward: bad arguments");
    if (in_dim != Q_IN || hidden != Q_HID) {
        set_error("quant_step_nets: instantiated for %d -> %d -> 1 (got %d -> %d)", Q_IN, Q_HID, in_dim, hidden);
        return GSVC_E_UNSUPPORTED;
    }
    if (M == 0) return GSVC_OK;
    GSVC_REQUIRE(X, "quant_step_nets_forward: NULL pointer");
    QuantFwd p;
    for (int i = 0; i < 3; i++) {
        GSVC_REQUIRE(nets[i].W1 && nets[i].b1 && nets[i].W2 && nets[i].b2 && z[i] && a[i] && q[i], "quant_step_nets_forward: NULL pointer (net %d)", i);
        if (!aligned16({nets[i].W1, X, z[i], a[i]}) || (reinterpret_cast<uintptr_t>(nets[i].W2) & 7)) {
            set_error("quant_step_nets_forward: matrices must be 16-byte aligned");
            return GSVC_E_UNSUPPORTED;
        }
        p.W1[i] = nets[i].W1; p.b1[i] = nets[i].b1; p.W2[i] = nets[i].W2; p.b2[i] = nets[i].b2;
        p.z[i] = z[i]; p.a[i] = a[i]; p.q[i] = q[i];
    }
    hipStream_t s = (hipStream_t)stream;
    chain_launch("k_quant_nets_fwd", &k_quant_nets_fwd<Q_IN, Q_HID, CHAIN_T>, (size_t)QuantLds<Q_IN, Q_HID>::FLOATS * 4, M, 1, s, X, p, (long long)M);
    return check_launch("quant_step_nets_forward");
}

extern "C" int gsvc_quant_step_nets_backward(const gsvc_quant_step_net *nets, const float *const *z, const float *const *dq, int64_t M,
                                             int32_t in_dim, int32_t hidden, float *const *dz, float *dX, void *stream)
{
    GSVC_REQUIRE(nets && z && dq && dz && M >= 0, "quant_step_nets_backward: bad arguments");
    if (in_dim != Q_IN || hidden != Q_HID) {
        set_error("quant_step_nets: instantiated for %d -> %d -> 1 (got %d -> %d)", Q_IN, Q_HID, in_dim, hidden);
        return GSVC_E_UNSUPPORTED;
    }
    if (M == 0) return GSVC_OK;
    GSVC_REQUIRE(dX, "quant_step_nets_backward: NULL pointer");
    QuantBwd p;
    for (int i = 0; i < 3; i++) {
        GSVC_REQUIRE(nets[i].W1 && nets[i].W2 && z[i] && dz[i], "quant_step_nets_backward: NULL pointer (net %d)", i);
        if (!aligned16({nets[i].W1, z[i], dz[i], dX}) || (reinterpret_cast<uintptr_t>(nets[i].W2) & 7)) {
            set_error("quant_step_nets_backward: matrices must be 16-byte aligned");
            return GSVC_E_UNSUPPORTED;
        }
        p.W1[i] = nets[i].W1; p.W2[i] = nets[i].W2; p.z[i] = z[i]; p.dq[i] = dq[i]; p.dz[i] = dz[i];
    }
    p.dX = dX;
    hipStream_t s = (hipStream_t)stream;
    chain_launch("k_quant_nets_bwd", &k_quant_nets_bwd<Q_IN, Q_HID, CHAIN_T>, (size_t)QuantLds<Q_IN, Q_HID>::FLOATS_T * 4, M, 1, s, p, (long long)M);
    return check_launch("quant_step_nets_backward");
}

// gsvc_amd/csrc/generate.hip — tail of the anchor -> neural-Gaussian generation as one kernel each way, gfx950.
//
// After the MLPs, reference ortho_gaussian_renderer/guassian.py:262-296 turns their raw outputs into the rasterizer's
// inputs with a dozen elementwise launches (and ~30 more in backward).  For un-compacted renders (all K slots of every
// visible anchor, n = rows*K Gaussians) this is, per Gaussian i of anchor row r = i / K:
//     neural_opacity = opacity_raw * offset_mask                 mask = neural_opacity > 0
//     scaling        = grid_scaling[r, 3:6] * sigmoid(scale_rot[i, 0:3])
//     rot            = scale_rot[i, 3:7] / max(|scale_rot[i, 3:7]|, 1e-12)            (F.normalize)
//     world          = anchor[r] + (grid_offsets[i] + neural_offset[i]) * grid_scaling[r, 0:3]
//     xyz            = clamp(world, bound_min, bound_max)
// One lane per Gaussian both ways; in backward a block takes whole anchor rows and the gradients of grid_scaling /
// anchor (sums over the row's K Gaussians) meet in LDS: no atomics, deterministic.
#include "common.h"

namespace gsvc {

struct Bounds3 {
    float lo[3], hi[3];
};

__global__ void __launch_bounds__(256) k_gen_tail_fwd(const float *__restrict__ op_raw, const float *__restrict__ offset_mask,
                                                      const float *__restrict__ grid_offsets, const float *__restrict__ neural_offset,
                                                      const float *__restrict__ scale_rot, const float *__restrict__ grid_scaling,
                                                      const float *__restrict__ anchor, Bounds3 bd, int64_t n, int K,
                                                      float *__restrict__ neural_opacity, uint8_t *__restrict__ mask,
                                                      float *__restrict__ scaling, float *__restrict__ rot,
                                                      float *__restrict__ world, float *__restrict__ xyz)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t r = i / K;
    const float no = op_raw[i] * offset_mask[i];
    neural_opacity[i] = no;
    mask[i] = no > 0.0f ? 1 : 0;
    const float *sr = scale_rot + 7 * i;
    const float *gs = grid_scaling + 6 * r;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float sg = 1.0f / (1.0f + expf(-sr[c]));
        scaling[3 * i + c] = gs[3 + c] * sg;
        const float w = anchor[3 * r + c] + (grid_offsets[3 * i + c] + neural_offset[3 * i + c]) * gs[c];
        world[3 * i + c] = w;
        xyz[3 * i + c] = fminf(fmaxf(w, bd.lo[c]), bd.hi[c]);
    }
    const float q0 = sr[3], q1 = sr[4], q2 = sr[5], q3 = sr[6];
    const float inv = 1.0f / fmaxf(sqrtf(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3), 1e-12f);
    rot[4 * i] = q0 * inv; rot[4 * i + 1] = q1 * inv; rot[4 * i + 2] = q2 * inv; rot[4 * i + 3] = q3 * inv;
}

// any of the incoming gradients may be NULL (that output was not used).  A block takes rpb = 256 / K whole anchor rows
// (rpb * K lanes, one per Gaussian, coalesced accesses); the per-row sums (grid_scaling / anchor gradients) meet in LDS.
__global__ void __launch_bounds__(256) k_gen_tail_bwd(const float *__restrict__ op_raw, const float *__restrict__ offset_mask,
                                                      const float *__restrict__ grid_offsets, const float *__restrict__ neural_offset,
                                                      const float *__restrict__ scale_rot, const float *__restrict__ grid_scaling,
                                                      const float *__restrict__ world, Bounds3 bd, int64_t rows, int K, int rpb,
                                                      const float *__restrict__ g_no, const float *__restrict__ g_scaling,
                                                      const float *__restrict__ g_rot, const float *__restrict__ g_world,
                                                      const float *__restrict__ g_xyz, float *__restrict__ d_op_raw,
                                                      float *__restrict__ d_offset_mask, float *__restrict__ d_offsets,
                                                      float *__restrict__ d_scale_rot, float *__restrict__ d_grid_scaling,
                                                      float *__restrict__ d_anchor)
{
    __shared__ float part[256][9];      // per Gaussian: gw*offset (3), g_scaling*sigmoid (3), gw (3)
    // the 7- and 3-float records of the block's Gaussians are contiguous in memory: they are collected here and stored as whole
    // segments (lane-per-component stores at a 28- / 12-byte stride wrote every line 7 / 3 times: 272 MB of write traffic per
    // launch for 95 MB of output, profiles/r03 PMC)
    __shared__ float s_dsr[256 * 7];
    __shared__ float s_dof[256 * 3];
    const int t = threadIdx.x;
    const int64_t row0 = (int64_t)blockIdx.x * rpb;
    const int lr = t / K;                                   // row inside the block
    const int64_t r = row0 + lr, i = r * K + (t - lr * K);
    const bool live = lr < rpb && r < rows;
    float c9[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (live) {
        const float gno = g_no ? g_no[i] : 0.f;
        d_op_raw[i] = gno * offset_mask[i];
        d_offset_mask[i] = gno * op_raw[i];
        const float *sr = scale_rot + 7 * i;
        const float *gs = grid_scaling + 6 * r;
        float *dsr = s_dsr + 7 * t;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float w = world[3 * i + c];
            float gw = g_world ? g_world[3 * i + c] : 0.f;
            if (g_xyz && w >= bd.lo[c] && w <= bd.hi[c]) gw += g_xyz[3 * i + c];
            s_dof[3 * t + c] = gw * gs[c];
            const float sg = 1.0f / (1.0f + expf(-sr[c]));
            const float gsc = g_scaling ? g_scaling[3 * i + c] : 0.f;
            dsr[c] = gsc * gs[3 + c] * sg * (1.0f - sg);
            c9[c] = gw * (grid_offsets[3 * i + c] + neural_offset[3 * i + c]);
            c9[3 + c] = gsc * sg;
            c9[6 + c] = gw;
        }
        const float q0 = sr[3], q1 = sr[4], q2 = sr[5], q3 = sr[6];
        const float nrm = sqrtf(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3);
        float g0 = 0.f, g1 = 0.f, g2 = 0.f, g3 = 0.f;
        if (g_rot) { g0 = g_rot[4 * i]; g1 = g_rot[4 * i + 1]; g2 = g_rot[4 * i + 2]; g3 = g_rot[4 * i + 3]; }
        if (nrm > 1e-12f) {
            const float inv = 1.0f / nrm;
            const float y0 = q0 * inv, y1 = q1 * inv, y2 = q2 * inv, y3 = q3 * inv;
            const float dot = y0 * g0 + y1 * g1 + y2 * g2 + y3 * g3;
            dsr[3] = (g0 - y0 * dot) * inv; dsr[4] = (g1 - y1 * dot) * inv;
            dsr[5] = (g2 - y2 * dot) * inv; dsr[6] = (g3 - y3 * dot) * inv;
        } else {
            dsr[3] = g0 * 1e12f; dsr[4] = g1 * 1e12f; dsr[5] = g2 * 1e12f; dsr[6] = g3 * 1e12f;
        }
    }
#pragma unroll
    for (int c = 0; c < 9; c++) part[t][c] = c9[c];
    __syncthreads();
    {
        const int64_t i0 = row0 * K;
        const int64_t left = rows - row0;
        const int n_live = (int)(left < rpb ? left : rpb) * K;      // live lanes are 0 .. n_live - 1 (t = lr * K + k)
        for (int v = t; v < n_live * 7; v += 256) d_scale_rot[7 * i0 + v] = s_dsr[v];
        for (int v = t; v < n_live * 3; v += 256) d_offsets[3 * i0 + v] = s_dof[v];
    }
    // lane (row, component): 9 components x rpb rows
    for (int v = t; v < rpb * 9; v += 256) {
        const int rr = v / 9, c = v - rr * 9;
        const int64_t gr = row0 + rr;
        if (gr >= rows) continue;
        float a = 0.f;
        for (int k = 0; k < K; k++) a += part[rr * K + k][c];
        if (c < 6) d_grid_scaling[6 * gr + c] = a;
        else if (d_anchor) d_anchor[3 * gr + (c - 6)] = a;
    }
}

// Conditioning of the generator / deformation networks for the rows of R renders (reference guassian.py:225-230 +
// utils/time_util.py:7-55): pe[row] = [embed(cam_z of the row's render) | embed(anchor_z - cam_z)], embed(x) = [x, sin(2^0 x),
// cos(2^0 x), ..., sin(2^(F-1) x), cos(2^(F-1) x)].  One lane per output element (coalesced store); was a dozen launches.
struct EmbedSegs {
    long long bound[17];      // row offsets of the renders, bound[R] = rows
    float cam_z[16];
};

// One lane per (row, frequency): sin and cos of an argument come from ONE sincosf, the camera half of a row — the same 2 F + 1
// numbers for every row of a render — from a table the workgroup fills once, and the rows leave through LDS as whole segments
// (one lane per output element called sinf / cosf 4 F + 2 times per row, half of them on the render's constant, and stored 4 bytes
// at a time: 40 us for 198 k rows).  256 / F rows per workgroup.
constexpr int EMBED_MAX_F = 24, EMBED_OUT_FLOATS = 1536;      // (256 / F) * 2 (2 F + 1) <= 1536 for F >= 1
__global__ void __launch_bounds__(256) k_embed_pe(const float *__restrict__ anchor, EmbedSegs sg, int R, int F, int64_t rows,
                                                  float *__restrict__ pe)
{
    __shared__ float s_cam[16][2 * EMBED_MAX_F + 1];
    __shared__ float s_out[EMBED_OUT_FLOATS];
    __shared__ long long s_bound[17];      // (a lane-varying index into the by-value struct would go through scratch)
    const int W = 2 * F + 1, rpb = 256 / F, t = threadIdx.x;
    if (t < 17) s_bound[t] = sg.bound[t];
    const int64_t r0 = (int64_t)blockIdx.x * rpb;
    const int n_rows = (int)(rows - r0 < rpb ? rows - r0 : rpb);
    // renders of the workgroup's first and last row
    int r_lo = 0, r_hi = 0;
    while (r_lo + 1 < R && r0 >= sg.bound[r_lo + 1]) r_lo++;
    r_hi = r_lo;
    while (r_hi + 1 < R && r0 + n_rows - 1 >= sg.bound[r_hi + 1]) r_hi++;
    for (int e = t; e < (r_hi - r_lo + 1) * (F + 1); e += 256) {
        const int rr = e / (F + 1), k = e - rr * (F + 1);
        const float cz = sg.cam_z[r_lo + rr];
        if (k == F) s_cam[rr][0] = cz;
        else {
            float sn, cs;
            sincosf(cz * (float)(1u << k), &sn, &cs);
            s_cam[rr][1 + 2 * k] = sn;
            s_cam[rr][2 + 2 * k] = cs;
        }
    }
    __syncthreads();
    const int lr = t / F, k = t - lr * F;
    if (lr < n_rows) {
        const int64_t row = r0 + lr;
        int r = r_lo;
        while (r < r_hi && row >= s_bound[r + 1]) r++;
        const float xa = anchor[3 * row + 2] - s_cam[r - r_lo][0];
        float sn, cs;
        sincosf(xa * (float)(1u << k), &sn, &cs);
        float *o = s_out + lr * 2 * W;
        o[1 + 2 * k] = s_cam[r - r_lo][1 + 2 * k];
        o[2 + 2 * k] = s_cam[r - r_lo][2 + 2 * k];
        o[W + 1 + 2 * k] = sn;
        o[W + 2 + 2 * k] = cs;
        if (k == 0) { o[0] = s_cam[r - r_lo][0]; o[W] = xa; }
    }
    __syncthreads();
    float *dst = pe + r0 * 2 * W;
    for (int e = t; e < n_rows * 2 * W; e += 256) dst[e] = s_out[e];
}

// Per-anchor parameters of the batch's rows in one pass each way (reference guassian.py:160-176 gathers them by boolean mask and
// re-applies the getters' activations per render): feat = _anchor_feat[v], offsets = _offset[v], scaling = exp(_scaling[v]),
// mask = straight-through threshold of sigmoid(_mask[v]) (forward value ((s > 0.01) - s) + s, gradient of s), v = vis[row].
// Backward: the rows of one anchor (it is visible in up to R renders) meet in float atomics on the zero-filled dense gradients.
struct GatherDims {
    int F, K3, S, K;       // columns of feat, offsets (3 K), scaling, mask
};

// 32 lanes per row (8 rows per workgroup), a lane takes columns l, l + 32, ... of the row's F + 3 K + S + K values: the row index
// is read once per lane (not once per element behind a 64-bit division), and a lane's loads are independent of each other
__global__ void __launch_bounds__(256) k_gather_rows_fwd(const float *__restrict__ pf, const float *__restrict__ po,
                                                         const float *__restrict__ ps, const float *__restrict__ pm,
                                                         const long long *__restrict__ vis, long long rows, GatherDims d, int decoded,
                                                         float *__restrict__ feat, float *__restrict__ off, float *__restrict__ scal,
                                                         float *__restrict__ mask)
{
#pragma clang fp contract(off)
    const int CT = d.F + d.K3 + d.S + d.K;
    const long long row = (long long)blockIdx.x * 8 + (threadIdx.x >> 5);
    if (row >= rows) return;
    const long long v = vis[row];
    for (int c0 = threadIdx.x & 31; c0 < CT; c0 += 32) {
        int c = c0;
        if (c < d.F) { feat[row * d.F + c] = pf[v * d.F + c]; continue; }
        c -= d.F;
        if (c < d.K3) { off[row * d.K3 + c] = po[v * d.K3 + c]; continue; }
        c -= d.K3;
        if (c < d.S) {
            const float r = ps[v * d.S + c];
            scal[row * d.S + c] = decoded ? r : 1.0f * expf(r);
            continue;
        }
        c -= d.S;
        const float r = pm[v * d.K + c];
        if (decoded) { mask[row * d.K + c] = r; continue; }
        const float sg = 1.0f / (1.0f + expf(-r));
        const float h = sg > 0.01f ? 1.0f : 0.0f;
        const float diff = h - sg;
        mask[row * d.K + c] = diff + sg;
    }
}

__global__ void __launch_bounds__(256) k_gather_rows_bwd(const float *__restrict__ ps, const float *__restrict__ pm,
                                                         const long long *__restrict__ vis, long long rows, GatherDims d, int decoded,
                                                         const float *__restrict__ gf, const float *__restrict__ go,
                                                         const float *__restrict__ gs, const float *__restrict__ gm,
                                                         float *__restrict__ df, float *__restrict__ dof, float *__restrict__ ds,
                                                         float *__restrict__ dm)
{
    const int CT = d.F + d.K3 + d.S + d.K;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * CT) return;
    const long long row = i / CT;
    int c = (int)(i - row * CT);
    const long long v = vis[row];
    if (c < d.F) { if (gf && df) atomicAdd(df + v * d.F + c, gf[row * d.F + c]); return; }
    c -= d.F;
    if (c < d.K3) { if (go && dof) atomicAdd(dof + v * d.K3 + c, go[row * d.K3 + c]); return; }
    c -= d.K3;
    if (c < d.S) {
        if (gs && ds) {
            const float g = gs[row * d.S + c];
            atomicAdd(ds + v * d.S + c, decoded ? g : g * expf(ps[v * d.S + c]));
        }
        return;
    }
    c -= d.S;
    if (gm && dm) {
        const float g = gm[row * d.K + c];
        float w = 1.0f;
        if (!decoded) {
            const float sg = 1.0f / (1.0f + expf(-pm[v * d.K + c]));
            w = sg * (1.0f - sg);
        }
        atomicAdd(dm + v * d.K + c, g * w);
    }
}

// The same backward WITHOUT atomics, for rows that are the concatenation of R ascending index lists (the R views of a
// fitting step): seen[r * A + a] says whether view r holds anchor a, rank[r * A + a] (inclusive scan of seen) - 1 is its row.
// One thread per (anchor, channel) adds the <= R rows in view order — the same bits every run, no zero fill before it, and
// not bound by the ~0.13 T float atomics per second the chip retires (19 M of them per step).
constexpr int RANKED_ANCHORS = 64, RANKED_VIEWS = 8;      // anchors per workgroup; views held in LDS (R <= 8)
__global__ void __launch_bounds__(256) k_gather_rows_bwd_ranked(const float *__restrict__ ps, const float *__restrict__ pm,
                                                                const unsigned char *__restrict__ seen,
                                                                const long long *__restrict__ rank, int R, long long A, GatherDims d,
                                                                int decoded, const float *__restrict__ gf, const float *__restrict__ go,
                                                                const float *__restrict__ gs, const float *__restrict__ gm,
                                                                float *__restrict__ df, float *__restrict__ dof, float *__restrict__ ds,
                                                                float *__restrict__ dm)
{
    // the row of every (anchor, view) pair of the workgroup's 64 anchors is looked up ONCE (one coalesced load per thread) and
    // shared through LDS; every (anchor, channel) then adds its <= R rows.  (With each (anchor, channel) thread reading the
    // mask / rank words itself the kernel was bound by those ~50-fold redundant load instructions: 80-110 us.)
    __shared__ long long s_row[RANKED_VIEWS][RANKED_ANCHORS];
    const long long a0 = (long long)blockIdx.x * RANKED_ANCHORS;
    for (int t = threadIdx.x; t < RANKED_VIEWS * RANKED_ANCHORS; t += 256) {
        const int v = t / RANKED_ANCHORS, u = t - v * RANKED_ANCHORS;
        const long long a = a0 + u, j = (long long)v * A + a;
        const bool in = v < R && a < A;
        const unsigned char sn = in ? seen[j] : (unsigned char)0;
        const long long rk = in ? rank[j] : 0;
        s_row[v][u] = sn ? rk - 1 : -1;
    }
    __syncthreads();
    const int CT = d.F + d.K3 + d.S + d.K;
    const int n_anchor = (int)min((long long)RANKED_ANCHORS, A - a0);
    for (int item = threadIdx.x; item < n_anchor * CT; item += 256) {
        const int u = item / CT;
        int c = item - u * CT;
        const float *g;
        float *out;
        int W, kind;
        if (c < d.F) { g = gf; out = df; W = d.F; kind = 0; }
        else if ((c -= d.F) < d.K3) { g = go; out = dof; W = d.K3; kind = 0; }
        else if ((c -= d.K3) < d.S) { g = gs; out = ds; W = d.S; kind = 1; }
        else { c -= d.S; g = gm; out = dm; W = d.K; kind = 2; }
        if (!g || !out) continue;
        float val[RANKED_VIEWS];
#pragma unroll
        for (int v = 0; v < RANKED_VIEWS; v++) {
            const long long row = s_row[v][u];
            val[v] = row >= 0 ? g[row * W + c] : 0.f;
        }
        float acc = 0.f;
#pragma unroll
        for (int v = 0; v < RANKED_VIEWS; v++) acc += val[v];
        const long long a = a0 + u;
        if (!decoded && kind == 1) acc *= expf(ps[a * W + c]);
        if (!decoded && kind == 2) {
            const float sg = 1.0f / (1.0f + expf(-pm[a * W + c]));
            acc *= sg * (1.0f - sg);
        }
        out[a * W + c] = acc;
    }
}

// Masks of a step plan in one pass over the anchors (gsvc_amd.generate.StepPlan): the R views' visibility masks side by side,
// "some view sees the anchor", and the rate sample = visible & live (an offset of the anchor is kept: sigmoid(mask) > 0.01,
// reference scene/gaussian_model.py get_mask_anchor) & (u <= rate) with u drawn by the caller.  Was ~15 PyTorch launches.
struct PlanViews {
    const unsigned char *vis[16];
};

__global__ void __launch_bounds__(256) k_plan_masks(PlanViews pv, int R, long long A, const float *__restrict__ mask_raw, int K,
                                                    int decoded, const float *__restrict__ u, float rate,
                                                    unsigned char *__restrict__ M, unsigned char *__restrict__ present,
                                                    unsigned char *__restrict__ chosen)
{
    const long long a = (long long)blockIdx.x * 256 + threadIdx.x;
    if (a >= A) return;
    bool live = false;
    if (chosen) {
        float sum = 0.f;
        for (int k = 0; k < K; k++) {
            const float m = mask_raw[a * K + k];
            if (decoded) sum += m;
            else live |= 1.0f / (1.0f + expf(-m)) > 0.01f;
        }
        if (decoded) live = sum > 0.f;
    }
    bool any = false;
    for (int r = 0; r < R; r++) {
        const bool v = pv.vis[r][a] != 0;
        any |= v;
        M[(long long)r * A + a] = v;
        if (chosen) chosen[(long long)r * A + a] = v && live && u[(long long)r * A + a] <= rate;
    }
    present[a] = any;
}

// Tail of an EntropyParamsNet (reference scene/gaussian_model.py:1586-1596): params [n, 2 C] = [mean | scale], q [n] ->
// scale_c = max(scale, 1e-9), adj = exp(clamp(q, -10, 10)); backward assembles d params = [g_mean | g_scale * (scale >= 1e-9)]
// and d q = g_adj * adj * (|q| <= 10) — was 9 launches forward and ~30 backward for the three networks.
__global__ void __launch_bounds__(256) k_ctx_post_fwd(const float *__restrict__ params, const float *__restrict__ q, long long n, int C,
                                                      float *__restrict__ mean, float *__restrict__ scale, float *__restrict__ adj)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n * (C + 1)) return;
    const long long row = i / (C + 1);
    const int c = (int)(i - row * (C + 1));
    if (c == C) { adj[row] = expf(fminf(fmaxf(q[row], -10.0f), 10.0f)); return; }
    mean[row * C + c] = params[row * 2 * C + c];
    scale[row * C + c] = fmaxf(params[row * 2 * C + C + c], 1e-9f);
}

__global__ void __launch_bounds__(256) k_ctx_post_bwd(const float *__restrict__ params, const float *__restrict__ q,
                                                      const float *__restrict__ adj, long long n, int C, const float *__restrict__ g_mean,
                                                      const float *__restrict__ g_scale, const float *__restrict__ g_adj,
                                                      float *__restrict__ dparams, float *__restrict__ dq)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n * (C + 1)) return;
    const long long row = i / (C + 1);
    const int c = (int)(i - row * (C + 1));
    if (c == C) {
        const float qv = q[row];
        dq[row] = (g_adj && qv >= -10.0f && qv <= 10.0f) ? g_adj[row] * adj[row] : 0.0f;
        return;
    }
    dparams[row * 2 * C + c] = g_mean ? g_mean[row * C + c] : 0.0f;
    dparams[row * 2 * C + C + c] = (g_scale && params[row * 2 * C + C + c] >= 1e-9f) ? g_scale[row * C + c] : 0.0f;
}

// FiLM combine and its product rule as streaming elementwise kernels (16-byte accesses, whole lines): y = gamma * h + beta;
// d gamma = g * h, d h = g * gamma.  (As GEMM epilogues — gsvc_linear_forward_ex FILM / FILM_GRAD — the same traffic moves in
// 64-byte pieces of 400-byte rows at 2.5-3.2 TB/s and costs 25-35 us more per call than these.)
__global__ void __launch_bounds__(256) k_film_fwd(const float4 *__restrict__ gamma, const float4 *__restrict__ h,
                                                  const float4 *__restrict__ beta, float4 *__restrict__ y, long long n4)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const float4 a = gamma[i], b = h[i], c = beta[i];
        y[i] = make_float4(fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y), fmaf(a.z, b.z, c.z), fmaf(a.w, b.w, c.w));
    }
}

__global__ void __launch_bounds__(256) k_film_bwd(const float4 *__restrict__ g, const float4 *__restrict__ h,
                                                  const float4 *__restrict__ gamma, float4 *__restrict__ dgamma,
                                                  float4 *__restrict__ dh, long long n4)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const float4 a = g[i], b = h[i], c = gamma[i];
        dgamma[i] = make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w);
        dh[i] = make_float4(a.x * c.x, a.y * c.y, a.z * c.z, a.w * c.w);
    }
}

// Row maps of the generators' shared FiLM rows (gsvc_amd.generate._film_rows): the R views of a step come in opposite pairs
// (2 f, 2 f + 1) that share their condition, so the FiLM networks run once per (frame f, distinct visible anchor d).
//   row_of[i]  = FiLM row of chain row i = (view of i / 2) * D + pos[vis[i]]            (pos: anchor -> index in the distinct list)
//   src_a/b[f D + d] = chain row of anchor distinct[d] in view 2 f / 2 f + 1 (inclusive scan of the flattened [R, A] view masks
//                      - 1), -1 where that view does not see it
struct FilmBounds { long long b[17]; };
__global__ void __launch_bounds__(256) k_film_rows(const long long *__restrict__ vis, FilmBounds rb, int R, long long rows,
                                                   const long long *__restrict__ pos, long long D, long long A,
                                                   const uint8_t *__restrict__ mflat, const long long *__restrict__ scan,
                                                   const long long *__restrict__ distinct, int *__restrict__ row_of,
                                                   int *__restrict__ src_a, int *__restrict__ src_b)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < rows) {
        int r = 0;
        while (r + 1 < R && i >= rb.b[r + 1]) r++;
        row_of[i] = (int)(pos[vis[i]] + (long long)(r >> 1) * D);
    }
    if (i < (long long)(R >> 1) * D) {
        const long long f = i / D, d = i - f * D;
        const long long p0 = 2 * f * A + distinct[d], p1 = p0 + A;
        src_a[i] = mflat[p0] ? (int)(scan[p0] - 1) : -1;
        src_b[i] = mflat[p1] ? (int)(scan[p1] - 1) : -1;
    }
}

// Per-row quantisation steps of the three attribute groups: Q_g[i] = base_g * adj_g[ctx_row[i]] (reference
// gaussian_renderer/guassian.py:250-262: Q_feat * Q_feat_adj etc.; the entropy context is evaluated once per DISTINCT anchor, so
// its step adjustments are gathered by ctx_row).  One launch instead of three gathers and three products; the backward
// scatter-adds the rows' gradients onto the distinct anchors (one launch instead of three products, three fills, three index_adds).
// raw != 0: a0..a2 are the quant_step networks' RAW outputs q and the adjustment exp(clamp(q, -10, 10)) (reference
// scene/gaussian_model.py:1586-1596) is applied here, its derivative adj * [|q| <= 10] in the backward — the training step then
// never materialises the three adjustment vectors.
__device__ __forceinline__ float q_adj(float q) { return expf(fminf(fmaxf(q, -10.0f), 10.0f)); }

__global__ void __launch_bounds__(256) k_q_rows_fwd(const float *__restrict__ a0, const float *__restrict__ a1, const float *__restrict__ a2,
                                                    const long long *__restrict__ ctx_row, float q0, float q1, float q2, long long rows,
                                                    int raw, float *__restrict__ out)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows) return;
    const long long d = ctx_row ? ctx_row[i] : i;
    const float v0 = a0[d], v1 = a1[d], v2 = a2[d];
    out[i] = q0 * (raw ? q_adj(v0) : v0);
    out[rows + i] = q1 * (raw ? q_adj(v1) : v1);
    out[2 * rows + i] = q2 * (raw ? q_adj(v2) : v2);
}

__global__ void __launch_bounds__(256) k_q_rows_bwd(const float *__restrict__ g0, const float *__restrict__ g1, const float *__restrict__ g2,
                                                    const long long *__restrict__ ctx_row, float q0, float q1, float q2, long long rows,
                                                    long long D, const float *__restrict__ a0, const float *__restrict__ a1,
                                                    const float *__restrict__ a2, int raw, float *__restrict__ gadj)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows) return;
    const long long d = ctx_row ? ctx_row[i] : i;
    float f0 = q0, f1 = q1, f2 = q2;
    if (raw) {
        const float v0 = a0[d], v1 = a1[d], v2 = a2[d];
        f0 = (v0 >= -10.0f && v0 <= 10.0f) ? q0 * expf(v0) : 0.0f;
        f1 = (v1 >= -10.0f && v1 <= 10.0f) ? q1 * expf(v1) : 0.0f;
        f2 = (v2 >= -10.0f && v2 <= 10.0f) ? q2 * expf(v2) : 0.0f;
    }
    if (ctx_row) {
        if (g0) atomicAdd(gadj + d, f0 * g0[i]);
        if (g1) atomicAdd(gadj + D + d, f1 * g1[i]);
        if (g2) atomicAdd(gadj + 2 * D + d, f2 * g2[i]);
    } else {
        gadj[i] = g0 ? f0 * g0[i] : 0.f;
        gadj[D + i] = g1 ? f1 * g1[i] : 0.f;
        gadj[2 * D + i] = g2 ? f2 * g2[i] : 0.f;
    }
}

// The scans and compactions of a step plan (gsvc_amd.generate.StepPlan) in three launches: inclusive scan c of the flattened
// view masks M [R A] with the list of the anchors (position mod A) at its set positions, the rate sample's list (rows c - 1 of the chosen (view, anchor)
// pairs, chosen a subset of M) and, over the anchors, pos = scan of `present` - 1 with the list of distinct anchors.  The tensor
// form was three cumsums (each preceded by a bool -> int64 copy), slicing / differences for the counts and three compactions.
//   phase 1: set bytes per 4096-element chunk;  phase 2: one workgroup scans the chunk counts;  phase 3: chunk-local scans + writes.
constexpr int PS_CHUNK = 4096;

__device__ __forceinline__ int ps_count16(const uint8_t *__restrict__ m, long long i0, long long n, uint8_t (&v)[16])
{
    int c = 0;
    if (m && i0 + 16 <= n && ((reinterpret_cast<uintptr_t>(m) + i0) & 15) == 0) {
        const uint4 u = *reinterpret_cast<const uint4 *>(m + i0);
        const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int q = 0; q < 16; q++) { v[q] = (w[q >> 2] >> (8 * (q & 3))) & 0xff ? 1 : 0; c += v[q]; }
    } else {
#pragma unroll
        for (int q = 0; q < 16; q++) { v[q] = (m && i0 + q < n && m[i0 + q]) ? 1 : 0; c += v[q]; }
    }
    return c;
}

// exclusive prefix of `c` over the 256 threads of the workgroup (thread order), total in *total
__device__ __forceinline__ int ps_block_excl(int c, int *sm, int *total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(inc, d, 64);
        if (lane >= d) inc += o;
    }
    if (lane == 63) sm[wave] = inc;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; w++) base += sm[w];
    *total = sm[0] + sm[1] + sm[2] + sm[3];
    __syncthreads();
    return base + inc - c;
}

__global__ void __launch_bounds__(256) k_plan_scan_count(const uint8_t *__restrict__ M, const uint8_t *__restrict__ chosen, long long nM,
                                                         const uint8_t *__restrict__ present, long long A, int nbM, int *__restrict__ counts)
{
    __shared__ int sm[4];
    const int b = blockIdx.x;
    uint8_t v[16];
    int total;
    if (b < nbM) {
        const long long i0 = (long long)b * PS_CHUNK + 16 * threadIdx.x;
        ps_block_excl(ps_count16(M, i0, nM, v), sm, &total);
        if (threadIdx.x == 0) counts[b] = total;
        ps_block_excl(ps_count16(chosen, i0, nM, v), sm, &total);
        if (threadIdx.x == 0) counts[nbM + b] = total;
    } else {
        const long long i0 = (long long)(b - nbM) * PS_CHUNK + 16 * threadIdx.x;
        ps_block_excl(ps_count16(present, i0, A, v), sm, &total);
        if (threadIdx.x == 0) counts[nbM + b] = total;           // = 2 nbM + (b - nbM)
    }
}

// offsets[k] = exclusive scan of counts within each of the three arrays; totals[0..2] = their sums (M, chosen, present)
// wave j scans array j, 64 chunk counts per round (a one-thread loop over ~240 counts took 20 us of dependent loads)
__global__ void __launch_bounds__(192) k_plan_scan_offsets(const int *__restrict__ counts, int nbM, int nbP, long long *__restrict__ offsets,
                                                           long long *__restrict__ totals)
{
    const int j = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int first = j == 0 ? 0 : (j == 1 ? nbM : 2 * nbM), n = j == 2 ? nbP : nbM;
    long long run = 0;
    for (int k0 = 0; k0 < n; k0 += 64) {
        const int k = k0 + lane;
        const long long v = k < n ? counts[first + k] : 0;
        long long inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const long long o = __shfl_up(inc, d, 64);
            if (lane >= d) inc += o;
        }
        if (k < n) offsets[first + k] = run + inc - v;
        run += __shfl(inc, 63, 64);
    }
    if (lane == 0) totals[j] = run;
}

__global__ void __launch_bounds__(256) k_plan_scan_write(const uint8_t *__restrict__ M, const uint8_t *__restrict__ chosen, long long nM,
                                                         const uint8_t *__restrict__ present, long long A, int nbM,
                                                         const long long *__restrict__ offsets, long long *__restrict__ c,
                                                         long long *__restrict__ flat, long long *__restrict__ sel_rows,
                                                         long long *__restrict__ pos, long long *__restrict__ distinct,
                                                         long long *__restrict__ view_ends)
{
    __shared__ int sm[4];
    const int b = blockIdx.x;
    uint8_t v[16], w[16];
    int total;
    if (b < nbM) {
        const long long i0 = (long long)b * PS_CHUNK + 16 * threadIdx.x;
        long long cm = offsets[b] + ps_block_excl(ps_count16(M, i0, nM, v), sm, &total);
        long long cc = chosen ? offsets[nbM + b] + ps_block_excl(ps_count16(chosen, i0, nM, w), sm, &total) : 0;
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const long long i = i0 + q;
            if (i >= nM) break;
            cm += v[q];
            c[i] = cm;
            if (v[q]) flat[cm - 1] = i % A;          // the anchor: view r's list is the r-th segment of this one list
            if (chosen && w[q]) sel_rows[cc++] = cm - 1;
            if ((i + 1) % A == 0) view_ends[(i + 1) / A - 1] = cm;
        }
    } else {
        const long long i0 = (long long)(b - nbM) * PS_CHUNK + 16 * threadIdx.x;
        long long cp = offsets[nbM + b] + ps_block_excl(ps_count16(present, i0, A, v), sm, &total);
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const long long i = i0 + q;
            if (i >= A) break;
            cp += v[q];
            pos[i] = cp - 1;
            if (v[q]) distinct[cp - 1] = i;
        }
    }
}

}  // namespace gsvc

using namespace gsvc;

extern "C" int gsvc_gen_tail_forward(const float *op_raw, const float *offset_mask, const float *grid_offsets,
                                     const float *neural_offset, const float *scale_rot, const float *grid_scaling,
                                     const float *anchor, const float *bound_min3_host, const float *bound_max3_host, int64_t rows,
                                     int32_t K, float *neural_opacity, uint8_t *mask, float *scaling, float *rot, float *world,
                                     float *xyz, void *stream)
{
    GSVC_REQUIRE(rows >= 0 && K > 0 && bound_min3_host && bound_max3_host, "gen_tail_forward: bad arguments");
    if (rows == 0) return GSVC_OK;
    GSVC_REQUIRE(op_raw && offset_mask && grid_offsets && neural_offset && scale_rot && grid_scaling && anchor && neural_opacity &&
                 mask && scaling && rot && world && xyz, "gen_tail_forward: NULL pointer");
    Bounds3 bd;
    for (int c = 0; c < 3; c++) { bd.lo[c] = bound_min3_host[c]; bd.hi[c] = bound_max3_host[c]; }
    hipStream_t s = (hipStream_t)stream;
    const int64_t n = rows * K;
    ProfScope _p("k_gen_tail_fwd", s);
    hipLaunchKernelGGL(k_gen_tail_fwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, op_raw, offset_mask, grid_offsets,
                       neural_offset, scale_rot, grid_scaling, anchor, bd, n, K, neural_opacity, mask, scaling, rot, world, xyz);
    return check_launch("gen_tail_forward");
}

extern "C" int gsvc_gen_tail_backward(const float *op_raw, const float *offset_mask, const float *grid_offsets,
                                      const float *neural_offset, const float *scale_rot, const float *grid_scaling,
                                      const float *world, const float *bound_min3_host, const float *bound_max3_host, int64_t rows,
                                      int32_t K, const float *g_neural_opacity, const float *g_scaling, const float *g_rot,
                                      const float *g_world, const float *g_xyz, float *d_op_raw, float *d_offset_mask,
                                      float *d_offsets, float *d_scale_rot, float *d_grid_scaling, float *d_anchor, void *stream)
{
    GSVC_REQUIRE(rows >= 0 && K > 0 && bound_min3_host && bound_max3_host, "gen_tail_backward: bad arguments");
    if (rows == 0) return GSVC_OK;
    GSVC_REQUIRE(op_raw && offset_mask && grid_offsets && neural_offset && scale_rot && grid_scaling && world && d_op_raw &&
                 d_offset_mask && d_offsets && d_scale_rot && d_grid_scaling, "gen_tail_backward: NULL pointer");
    Bounds3 bd;
    for (int c = 0; c < 3; c++) { bd.lo[c] = bound_min3_host[c]; bd.hi[c] = bound_max3_host[c]; }
    hipStream_t s = (hipStream_t)stream;
    GSVC_REQUIRE(K <= 256, "gen_tail_backward: K > 256");
    const int rpb = 256 / K;
    ProfScope _p("k_gen_tail_bwd", s);
    hipLaunchKernelGGL(k_gen_tail_bwd, dim3((unsigned)((rows + rpb - 1) / rpb)), dim3(256), 0, s, op_raw, offset_mask, grid_offsets,
                       neural_offset, scale_rot, grid_scaling, world, bd, rows, K, rpb, g_neural_opacity, g_scaling, g_rot, g_world,
                       g_xyz, d_op_raw, d_offset_mask, d_offsets, d_scale_rot, d_grid_scaling, d_anchor);
    return check_launch("gen_tail_backward");
}

extern "C" int gsvc_embed_pe(const float *anchor, const int64_t *row_bounds, const float *cam_z, int32_t renders, int32_t freqs,
                             float *pe, void *stream)
{
    GSVC_REQUIRE(renders >= 1 && renders <= 16 && freqs >= 1 && freqs <= 24 && row_bounds && cam_z, "embed_pe: bad arguments");
    gsvc::EmbedSegs sg;
    for (int r = 0; r <= renders; r++) sg.bound[r] = row_bounds[r];
    for (int r = 0; r < renders; r++) sg.cam_z[r] = cam_z[r];
    const int64_t rows = row_bounds[renders];
    if (rows == 0) return GSVC_OK;
    GSVC_REQUIRE(anchor && pe, "embed_pe: NULL pointer");
    const int rpb = 256 / freqs;
    gsvc::ProfScope _prof("k_embed_pe", (hipStream_t)stream);
    hipLaunchKernelGGL(gsvc::k_embed_pe, dim3((unsigned)((rows + rpb - 1) / rpb)), dim3(256), 0, (hipStream_t)stream, anchor, sg, renders,
                       freqs, rows, pe);
    return gsvc::check_launch("embed_pe");
}

extern "C" int gsvc_gather_rows_forward(const float *feat_p, const float *offset_p, const float *scaling_p, const float *mask_p,
                                        const int64_t *vis, int64_t rows, int32_t F, int32_t K, int32_t S, int32_t decoded, float *feat,
                                        float *offsets, float *scaling, float *mask, void *stream)
{
    GSVC_REQUIRE(rows >= 0 && F >= 0 && K >= 0 && S >= 0 && F + K + S > 0, "gather_rows_forward: bad shape");
    if (rows == 0) return GSVC_OK;
    GSVC_REQUIRE(vis && (F == 0 || (feat_p && feat)) && (K == 0 || (offset_p && mask_p && offsets && mask)) && (S == 0 || (scaling_p && scaling)),
                 "gather_rows_forward: NULL pointer");
    const gsvc::GatherDims d{F, 3 * K, S, K};
    gsvc::ProfScope _prof("k_gather_rows", (hipStream_t)stream);
    hipLaunchKernelGGL(gsvc::k_gather_rows_fwd, dim3((unsigned)((rows + 7) / 8)), dim3(256), 0, (hipStream_t)stream, feat_p, offset_p,
                       scaling_p, mask_p, (const long long *)vis, (long long)rows, d, decoded, feat, offsets, scaling, mask);
    return gsvc::check_launch("gather_rows_forward");
}

extern "C" int gsvc_gather_rows_backward(const float *scaling_p, const float *mask_p, const int64_t *vis, int64_t rows, int32_t F,
                                         int32_t K, int32_t S, int32_t decoded, const float *g_feat, const float *g_offsets,
                                         const float *g_scaling, const float *g_mask, float *d_feat, float *d_offset, float *d_scaling,
                                         float *d_mask, void *stream)
{
    GSVC_REQUIRE(rows >= 0 && F >= 0 && K >= 0 && S >= 0 && F + K + S > 0, "gather_rows_backward: bad shape");
    if (rows == 0) return GSVC_OK;
    GSVC_REQUIRE(vis && (S == 0 || scaling_p) && (K == 0 || mask_p), "gather_rows_backward: NULL pointer");
    const gsvc::GatherDims d{F, 3 * K, S, K};
    const int64_t n = rows * (F + 3 * K + S + K);
    gsvc::ProfScope _prof("k_gather_rows_bwd", (hipStream_t)stream);
    hipLaunchKernelGGL(gsvc::k_gather_rows_bwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, scaling_p, mask_p,
                       (const long long *)vis, (long long)rows, d, decoded, g_feat, g_offsets, g_scaling, g_mask, d_feat, d_offset,
                       d_scaling, d_mask);
    return gsvc::check_launch("gather_rows_backward");
}

extern "C" int gsvc_gather_rows_backward_ranked(const float *scaling_p, const float *mask_p, const uint8_t *seen, const int64_t *rank,
                                                int32_t R, int64_t A, int32_t F, int32_t K, int32_t S, int32_t decoded,
                                                const float *g_feat, const float *g_offsets, const float *g_scaling, const float *g_mask,
                                                float *d_feat, float *d_offset, float *d_scaling, float *d_mask, void *stream)
{
    GSVC_REQUIRE(R >= 0 && A >= 0 && F >= 0 && K >= 0 && S >= 0 && F + K + S > 0, "gather_rows_backward_ranked: bad shape");
    if (A == 0) return GSVC_OK;
    GSVC_REQUIRE(R <= gsvc::RANKED_VIEWS, "gather_rows_backward_ranked: at most 8 views");
    GSVC_REQUIRE((R == 0 || (seen && rank)) && (S == 0 || scaling_p) && (K == 0 || mask_p), "gather_rows_backward_ranked: NULL pointer");
    const gsvc::GatherDims d{F, 3 * K, S, K};
    const int64_t blocks = (A + gsvc::RANKED_ANCHORS - 1) / gsvc::RANKED_ANCHORS;
    gsvc::ProfScope _prof("k_gather_rows_bwd", (hipStream_t)stream);
    hipLaunchKernelGGL(gsvc::k_gather_rows_bwd_ranked, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, scaling_p,
                       mask_p, seen, (const long long *)rank, R, (long long)A, d, decoded, g_feat, g_offsets, g_scaling, g_mask, d_feat,
                       d_offset, d_scaling, d_mask);
    return gsvc::check_launch("gather_rows_backward_ranked");
}

extern "C" int gsvc_plan_masks(const uint8_t *const *visible_host, int32_t R, int64_t A, const float *mask_raw, int32_t K, int32_t decoded,
                               const float *u, float rate, uint8_t *M, uint8_t *present, uint8_t *chosen, void *stream)
{
    GSVC_REQUIRE(R >= 1 && R <= 16 && A >= 0 && K >= 0, "plan_masks: bad shape (at most 16 views)");
    if (A == 0) return GSVC_OK;
    GSVC_REQUIRE(visible_host && M && present && (!chosen || (mask_raw && u)), "plan_masks: NULL pointer");
    gsvc::PlanViews pv;
    for (int r = 0; r < 16; r++) pv.vis[r] = r < R ? visible_host[r] : nullptr;
    for (int r = 0; r < R; r++) GSVC_REQUIRE(pv.vis[r], "plan_masks: NULL view mask");
    gsvc::ProfScope _prof("k_plan_masks", (hipStream_t)stream);
    hipLaunchKernelGGL(gsvc::k_plan_masks, dim3((unsigned)((A + 255) / 256)), dim3(256), 0, (hipStream_t)stream, pv, R, (long long)A,
                       mask_raw, K, decoded, u, rate, M, present, chosen);
    return gsvc::check_launch("plan_masks");
}

// Compaction of a mask by its inclusive scan: out[scan[i] - 1] = (value ? value[i] : i) for every set mask[i] — one
// elementwise pass (torch.nonzero_static re-scans: 38 us for the step plan's 1 M-entry view mask, three of them per step).
// The tail of `out` (capacity - count entries) is left untouched: callers cut the list to the count they read back.
namespace gsvc {
__global__ void __launch_bounds__(256) k_compact_by_scan(const uint8_t *__restrict__ mask, const long long *__restrict__ scan,
                                                         const long long *__restrict__ value, long long value_bias, long long n,
                                                         long long *__restrict__ out)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n && mask[i]) out[scan[i] - 1] = value ? value[i] + value_bias : i;
}
}  // namespace gsvc

extern "C" int gsvc_compact_by_scan(const uint8_t *mask, const int64_t *scan, const int64_t *value, int64_t value_bias, int64_t n,
                                    int64_t *out, void *stream)
{
    GSVC_REQUIRE(n >= 0, "compact_by_scan: bad size");
    if (n == 0) return GSVC_OK;
    GSVC_REQUIRE(mask && scan && out, "compact_by_scan: NULL pointer");
    gsvc::ProfScope _prof("k_compact_by_scan", (hipStream_t)stream);
    hipLaunchKernelGGL(gsvc::k_compact_by_scan, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, mask,
                       (const long long *)scan, (const long long *)value, (long long)value_bias, (long long)n, (long long *)out);
    return gsvc::check_launch("compact_by_scan");
}

extern "C" int gsvc_ctx_post_forward(const float *params, const float *q, int64_t n, int32_t C, float *mean, float *scale, float *adj,
                                     void *stream)
{
    GSVC_REQUIRE(n >= 0 && C > 0, "ctx_post_forward: bad shape");
    if (n == 0) return GSVC_OK;
    GSVC_REQUIRE(params && q && mean && scale && adj, "ctx_post_forward: NULL pointer");
    const int64_t t = n * (C + 1);
    gsvc::ProfScope _prof("k_ctx_post", (hipStream_t)stream);
    hipLaunchKernelGGL(gsvc::k_ctx_post_fwd, dim3((unsigned)((t + 255) / 256)), dim3(256), 0, (hipStream_t)stream, params, q,
                       (long long)n, C, mean, scale, adj);
    return gsvc::check_launch("ctx_post_forward");
}

extern "C" int gsvc_ctx_post_backward(const float *params, const float *q, const float *adj, int64_t n, int32_t C, const float *g_mean,
                                      const float *g_scale, const float *g_adj, float *dparams, float *dq, void *stream)
{
    GSVC_REQUIRE(n >= 0 && C > 0, "ctx_post_backward: bad shape");
    if (n == 0) return GSVC_OK;
    GSVC_REQUIRE(params && q && adj && dparams && dq, "ctx_post_backward: NULL pointer");
    const int64_t t = n * (C + 1);
    gsvc::ProfScope _prof("k_ctx_post_bwd", (hipStream_t)stream);
    hipLaunchKernelGGL(gsvc::k_ctx_post_bwd, dim3((unsigned)((t + 255) / 256)), dim3(256), 0, (hipStream_t)stream, params, q, adj,
                       (long long)n, C, g_mean, g_scale, g_adj, dparams, dq);
    return gsvc::check_launch("ctx_post_backward");
}

extern "C" int gsvc_film_forward(const float *gamma, const float *h, const float *beta, float *y, int64_t n, void *stream)
{
    GSVC_REQUIRE(n >= 0 && n % 4 == 0, "film_forward: the element count must be a multiple of 4");
    if (n == 0) return GSVC_OK;
    GSVC_REQUIRE(gamma && h && beta && y, "film_forward: NULL pointer");
    const long long n4 = n / 4;
    long long blocks = (n4 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    gsvc::ProfScope _prof("k_film", (hipStream_t)stream);
    hipLaunchKernelGGL(gsvc::k_film_fwd, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const float4 *)gamma, (const float4 *)h,
                       (const float4 *)beta, (float4 *)y, n4);
    return gsvc::check_launch("film_forward");
}

extern "C" int gsvc_film_backward(const float *g, const float *h, const float *gamma, float *dgamma, float *dh, int64_t n, void *stream)
{
    GSVC_REQUIRE(n >= 0 && n % 4 == 0, "film_backward: the element count must be a multiple of 4");
    if (n == 0) return GSVC_OK;
    GSVC_REQUIRE(g && h && gamma && dgamma && dh, "film_backward: NULL pointer");
    const long long n4 = n / 4;
    long long blocks = (n4 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    gsvc::ProfScope _prof("k_film_bwd", (hipStream_t)stream);
    hipLaunchKernelGGL(gsvc::k_film_bwd, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const float4 *)g, (const float4 *)h,
                       (const float4 *)gamma, (float4 *)dgamma, (float4 *)dh, n4);
    return gsvc::check_launch("film_backward");
}

namespace gsvc {
// out[j][c] = g[src_a[j]][c] + g[src_b[j]][c] (a source of -1: zeros); one lane per output element, consecutive lanes along c
__global__ void __launch_bounds__(256) k_pair_rows_sum(const float *__restrict__ g, const int32_t *__restrict__ src_a,
                                                       const int32_t *__restrict__ src_b, long long n, int C, float *__restrict__ out)
{
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    const long long j = e / C;
    const int c = (int)(e - j * C);
    const int a = src_a[j], b = src_b[j];
    const float va = a >= 0 ? g[(long long)a * C + c] : 0.f;
    const float vb = b >= 0 ? g[(long long)b * C + c] : 0.f;
    out[e] = va + vb;
}
}  // namespace gsvc

extern "C" int gsvc_pair_rows_sum(const float *g, const int32_t *src_a, const int32_t *src_b, int64_t rows_u, int32_t C, float *out, void *stream)
{
    GSVC_REQUIRE(rows_u >= 0 && C > 0 && rows_u * (int64_t)C < ((int64_t)1 << 40), "pair_rows_sum: bad shape");
    if (rows_u == 0) return GSVC_OK;
    GSVC_REQUIRE(g && src_a && src_b && out, "pair_rows_sum: NULL pointer");
    const long long n = rows_u * (long long)C;
    gsvc::ProfScope _prof("k_pair_rows_sum", (hipStream_t)stream);
    hipLaunchKernelGGL(gsvc::k_pair_rows_sum, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, g, src_a, src_b, n, (int)C, out);
    return gsvc::check_launch("pair_rows_sum");
}

namespace gsvc {
// Scatter-add of rows WITHOUT atomics (GSVC_DETERMINISTIC=1): the n source rows sorted by their target (order = stable argsort of
// the targets, sorted_idx = the targets in that order); the thread of (run head p, channel c) adds the run's rows in list order
// and writes (or adds to) the target row once — the same bits every run whatever the launch order of the workgroups.
__global__ void __launch_bounds__(256) k_segment_rows_sum(const float *__restrict__ src, const long long *__restrict__ order,
                                                          const long long *__restrict__ sorted_idx, long long n, int C,
                                                          float *__restrict__ dst, int accumulate)
{
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= n * C) return;
    const long long p = e / C;
    const int c = (int)(e - p * C);
    const long long t = sorted_idx[p];
    if (p > 0 && sorted_idx[p - 1] == t) return;
    float acc = 0.f;
    for (long long j = p; j < n && sorted_idx[j] == t; j++) acc += src[order[j] * C + c];
    float *o = dst + t * C + c;
    *o = accumulate ? *o + acc : acc;
}
}  // namespace gsvc

extern "C" int gsvc_segment_rows_sum(const float *src, const int64_t *order, const int64_t *sorted_idx, int64_t n, int32_t C, float *dst,
                                     int32_t accumulate, void *stream)
{
    GSVC_REQUIRE(n >= 0 && C > 0 && n * (int64_t)C < ((int64_t)1 << 40), "segment_rows_sum: bad shape");
    if (n == 0) return GSVC_OK;
    GSVC_REQUIRE(src && order && sorted_idx && dst, "segment_rows_sum: NULL pointer");
    const long long e = n * (long long)C;
    gsvc::ProfScope _prof("k_segment_rows_sum", (hipStream_t)stream);
    hipLaunchKernelGGL(gsvc::k_segment_rows_sum, dim3((unsigned)((e + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src,
                       (const long long *)order, (const long long *)sorted_idx, (long long)n, (int)C, dst, (int)accumulate);
    return gsvc::check_launch("segment_rows_sum");
}

extern "C" int gsvc_film_row_maps(const int64_t *vis, const int64_t *row_bounds_host, int32_t R, const int64_t *pos, int64_t D, int64_t A,
                              const uint8_t *view_masks, const int64_t *scan, const int64_t *distinct, int32_t *row_of, int32_t *src_a,
                              int32_t *src_b, void *stream)
{
    GSVC_REQUIRE(row_bounds_host && R >= 2 && R <= 16 && R % 2 == 0 && D >= 0 && A > 0, "film_rows: an even number of views (2..16)");
    gsvc::FilmBounds rb;
    for (int r = 0; r <= 16; r++) rb.b[r] = row_bounds_host[r <= R ? r : R];
    const long long rows = rb.b[R], n = rows > (R / 2) * D ? rows : (R / 2) * D;
    GSVC_REQUIRE(rows < ((int64_t)1 << 31) && (R / 2) * D < ((int64_t)1 << 31), "film_rows: row numbers must fit int32");
    if (n == 0) return GSVC_OK;
    GSVC_REQUIRE(vis && pos && view_masks && scan && distinct && row_of && src_a && src_b, "film_rows: NULL pointer");
    hipLaunchKernelGGL(gsvc::k_film_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const long long *)vis, rb,
                       (int)R, rows, (const long long *)pos, (long long)D, (long long)A, view_masks, (const long long *)scan,
                       (const long long *)distinct, row_of, src_a, src_b);
    return gsvc::check_launch("film_rows");
}

extern "C" int gsvc_q_rows_forward(const float *adj_feat, const float *adj_scaling, const float *adj_offsets, const int64_t *ctx_row,
                                   float q_feat, float q_scaling, float q_offsets, int64_t rows, int32_t raw, float *out3, void *stream)
{
    GSVC_REQUIRE(rows >= 0, "q_rows_forward: bad shape");
    if (rows == 0) return GSVC_OK;
    GSVC_REQUIRE(adj_feat && adj_scaling && adj_offsets && out3, "q_rows_forward: NULL pointer");
    gsvc::ProfScope _prof("k_q_rows", (hipStream_t)stream);
    hipLaunchKernelGGL(gsvc::k_q_rows_fwd, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, adj_feat, adj_scaling,
                       adj_offsets, (const long long *)ctx_row, q_feat, q_scaling, q_offsets, (long long)rows, (int)raw, out3);
    return gsvc::check_launch("q_rows_forward");
}

extern "C" int gsvc_q_rows_backward(const float *g_feat, const float *g_scaling, const float *g_offsets, const int64_t *ctx_row,
                                    float q_feat, float q_scaling, float q_offsets, int64_t rows, int64_t D, const float *adj_feat,
                                    const float *adj_scaling, const float *adj_offsets, int32_t raw, float *grad_adj3, void *stream)
{
    GSVC_REQUIRE(rows >= 0 && D >= 0 && (ctx_row || D == rows), "q_rows_backward: bad shape");
    if (D == 0) return GSVC_OK;
    GSVC_REQUIRE(grad_adj3 && (!raw || (adj_feat && adj_scaling && adj_offsets)), "q_rows_backward: NULL pointer");
    hipStream_t s = (hipStream_t)stream;
    if (ctx_row && hipMemsetAsync(grad_adj3, 0, sizeof(float) * 3 * (size_t)D, s) != hipSuccess) {
        gsvc::set_error("q_rows_backward: hipMemsetAsync failed");
        return GSVC_E_LAUNCH;
    }
    if (rows == 0) return GSVC_OK;
    gsvc::ProfScope _prof("k_q_rows_bwd", s);
    hipLaunchKernelGGL(gsvc::k_q_rows_bwd, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s, g_feat, g_scaling, g_offsets,
                       (const long long *)ctx_row, q_feat, q_scaling, q_offsets, (long long)rows, (long long)D, adj_feat, adj_scaling,
                       adj_offsets, (int)raw, grad_adj3);
    return gsvc::check_launch("q_rows_backward");
}

extern "C" int64_t gsvc_plan_scans_scratch_bytes(int32_t R, int64_t A)
{
    if (R < 1 || A < 1) return -1;
    const int64_t nbM = (R * A + gsvc::PS_CHUNK - 1) / gsvc::PS_CHUNK, nbP = (A + gsvc::PS_CHUNK - 1) / gsvc::PS_CHUNK;
    return (2 * nbM + nbP) * (int64_t)(sizeof(int) + sizeof(long long)) + 16;
}

extern "C" int gsvc_plan_scans(const uint8_t *view_masks, const uint8_t *chosen, const uint8_t *present, int32_t R, int64_t A, void *scratch,
                               int64_t *scan, int64_t *flat, int64_t *sel_rows, int64_t *pos, int64_t *distinct, int64_t *counts,
                               void *stream)
{
    GSVC_REQUIRE(R >= 1 && A >= 1 && R * A < ((int64_t)1 << 40), "plan_scans: bad shape");
    GSVC_REQUIRE(view_masks && present && scratch && scan && flat && pos && distinct && counts && (!chosen || sel_rows), "plan_scans: NULL pointer");
    const long long nM = (long long)R * A;
    const int nbM = (int)((nM + gsvc::PS_CHUNK - 1) / gsvc::PS_CHUNK), nbP = (int)((A + gsvc::PS_CHUNK - 1) / gsvc::PS_CHUNK);
    GSVC_REQUIRE(nbM <= 65536, "plan_scans: too many chunks for the one-workgroup offset pass");
    // scratch: [offsets: (2 nbM + nbP) int64][counts: (2 nbM + nbP) int]
    long long *offsets = (long long *)scratch;
    int *bc = (int *)(offsets + 2 * nbM + nbP);
    hipStream_t s = (hipStream_t)stream;
    gsvc::ProfScope _prof("k_plan_scans", s);
    hipLaunchKernelGGL(gsvc::k_plan_scan_count, dim3(nbM + nbP), dim3(256), 0, s, view_masks, chosen, nM, present, (long long)A, nbM, bc);
    // counts: [0, R) the scan at each view's end, [R] chosen pairs, [R + 1] distinct anchors: the totals (M, chosen, present) land on
    // counts[R - 1 ..] (the first repeats the last view's end)
    hipLaunchKernelGGL(gsvc::k_plan_scan_offsets, dim3(1), dim3(192), 0, s, bc, nbM, nbP, offsets, (long long *)counts + R - 1);
    hipLaunchKernelGGL(gsvc::k_plan_scan_write, dim3(nbM + nbP), dim3(256), 0, s, view_masks, chosen, nM, present, (long long)A, nbM, offsets,
                       (long long *)scan, (long long *)flat, (long long *)sel_rows, (long long *)pos, (long long *)distinct,
                       (long long *)counts);
    return gsvc::check_launch("plan_scans");
}

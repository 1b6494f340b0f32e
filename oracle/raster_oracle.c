/*
 * oracle/raster_oracle.c — CPU ORACLE (TEST INFRASTRUCTURE ONLY).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may link or call this file.
 * The product path (gsvc_amd/) never does: it fails loudly when the HIP library is missing.
 *
 * What it restates: the orthographic sliding-window tile rasterizer that GSVC calls as
 *   diff_gaussian_rasterization.cuda_ortho_gaussian_rasterizer.{GaussianRasterizer,GaussianRasterizationSettings}
 * (call sites: reference ortho_gaussian_renderer/renderer.py:63-98 and preprocess.py:58-104).
 *
 * PARITY UNPINNED: the rasterizer source is an un-pinned external git dependency
 * (github.com/actcwlf/ortho_diff_gaussian_rasterization, reference README.md:52) that is NOT present under
 * /root/reference, and the reference holds no tests or golden vectors for it.  This file therefore follows
 *   - the call-site contract (settings fields, argument layouts, three return values),
 *   - the camera / cube conventions of reference frame_cube/frame.py:18-43,92-101,156-190,
 *   - the pixel mapping implied by reference utils/loss_utils.py:122-123,
 *   - and the published 3D Gaussian Splatting tile-rasterizer algorithm the dependency forks
 *     (16x16 tiles, 3-sigma radius, alpha=min(.99,o*exp(p)), skip alpha<1/255, stop when T<1e-4, out=C+T*bg),
 * with every free choice written down in DESIGN.md ("Raster spec").  All arithmetic that decides an integer
 * result (radius, tile rectangle, sort key) is plain IEEE-754 binary32 with no FMA contraction
 * (build with -ffp-contract=off), in a fixed operation order that the HIP kernels repeat, so that
 * radii, tile lists and num_rendered are bit-exact between this file and the GPU.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define TILE 16
#define LOWPASS 0.3f
#define ALPHA_MIN (1.0f / 255.0f)
#define ALPHA_MAX 0.99f
#define T_MIN 0.0001f

typedef struct {
    int32_t image_height;
    int32_t image_width;
    float x_min;
    float y_min;
    float scale;
    float threshold;
    float scale_modifier;
    float bg[3];
    float viewmatrix[16]; /* row-major M: p_view = M[:3,:3] p + M[:3,3] */
    uint32_t flags;       /* convention switches, same bits as include/gsvc_hip.h GSVC_RASTER_* */
    float low_pass;       /* 0 = default 0.3 */
} oracle_raster_settings;

#define F_SLAB_ONE_SIDED 1u
#define F_PIXEL_CORNER 2u
#define F_DEPTH_DESCENDING 4u
#define F_MEANS2D_PIXEL_UNITS 8u
#define F_CLAMP_STOPS_GRADIENT 16u
#define F_NO_LOW_PASS 32u

static inline float low_pass_of(const oracle_raster_settings *st)
{
    return (st->flags & F_NO_LOW_PASS) ? 0.0f : (st->low_pass != 0.0f ? st->low_pass : LOWPASS);
}

typedef struct {
    float u, v, depth;
    float conic[3];   /* A, B, C of the inverse 2-D covariance */
    float cov2[3];    /* a, b, c (low-pass included) */
    float cov3[6];    /* xx xy xz yy yz zz */
    int32_t radius;
    int32_t rect[4];  /* x0 y0 x1 y1 (tile units, x1/y1 exclusive) */
} pre_t;

static inline uint32_t order_bits(float f)
{
    uint32_t b;
    memcpy(&b, &f, 4);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

static inline int tile_clamp(float t, int g)
{
    /* clamp in float first so the int conversion is defined for huge/NaN inputs; same result as
       min(g, max(0, (int)t)) for every finite t */
    t = fmaxf(t, -1.0f);
    t = fminf(t, (float)g + 1.0f);
    int i = (int)t;
    if (i < 0) i = 0;
    if (i > g) i = g;
    return i;
}

/* Steps 1-7 of the raster spec.  Returns radius (0 = culled). */
static int preprocess_one(const oracle_raster_settings *st, const float *p, const float *s, const float *q,
                          int gx, int gy, pre_t *o)
{
    const float *M = st->viewmatrix;
    memset(o, 0, sizeof(*o));
    float xv = M[0] * p[0] + M[1] * p[1] + M[2] * p[2] + M[3];
    float yv = M[4] * p[0] + M[5] * p[1] + M[6] * p[2] + M[7];
    float zv = M[8] * p[0] + M[9] * p[1] + M[10] * p[2] + M[11];
    if (st->flags & F_SLAB_ONE_SIDED) {
        if (!(zv <= 0.0f && zv >= -st->threshold)) return 0;
    } else if (!(fabsf(zv) <= st->threshold)) return 0; /* slab cull, NaN culled */

    /* rotation from quaternion (r,x,y,z) used as given (reference utils/general_utils.py:98-119 convention) */
    float r = q[0], x = q[1], y = q[2], z = q[3];
    float R[3][3];
    R[0][0] = 1.f - 2.f * (y * y + z * z);
    R[0][1] = 2.f * (x * y - r * z);
    R[0][2] = 2.f * (x * z + r * y);
    R[1][0] = 2.f * (x * y + r * z);
    R[1][1] = 1.f - 2.f * (x * x + z * z);
    R[1][2] = 2.f * (y * z - r * x);
    R[2][0] = 2.f * (x * z - r * y);
    R[2][1] = 2.f * (y * z + r * x);
    R[2][2] = 1.f - 2.f * (x * x + y * y);
    float S[3] = {st->scale_modifier * s[0], st->scale_modifier * s[1], st->scale_modifier * s[2]};
    float L[3][3];
    for (int a = 0; a < 3; a++)
        for (int k = 0; k < 3; k++) L[a][k] = R[a][k] * S[k];
    float C3[3][3];
    for (int a = 0; a < 3; a++)
        for (int b = a; b < 3; b++) {
            float v = L[a][0] * L[b][0] + L[a][1] * L[b][1] + L[a][2] * L[b][2];
            C3[a][b] = v;
            C3[b][a] = v;
        }
    o->cov3[0] = C3[0][0]; o->cov3[1] = C3[0][1]; o->cov3[2] = C3[0][2];
    o->cov3[3] = C3[1][1]; o->cov3[4] = C3[1][2]; o->cov3[5] = C3[2][2];

    /* T = scale * M[:2,:3];  Sigma2 = T Sigma3 T^T + h I */
    float T0[3] = {st->scale * M[0], st->scale * M[1], st->scale * M[2]};
    float T1[3] = {st->scale * M[4], st->scale * M[5], st->scale * M[6]};
    float U0[3], U1[3]; /* U = Sigma3 T^T columns */
    for (int a = 0; a < 3; a++) {
        U0[a] = C3[a][0] * T0[0] + C3[a][1] * T0[1] + C3[a][2] * T0[2];
        U1[a] = C3[a][0] * T1[0] + C3[a][1] * T1[1] + C3[a][2] * T1[2];
    }
    const float lowpass = low_pass_of(st);
    float ca = T0[0] * U0[0] + T0[1] * U0[1] + T0[2] * U0[2] + lowpass;
    float cb = T0[0] * U1[0] + T0[1] * U1[1] + T0[2] * U1[2];
    float cc = T1[0] * U1[0] + T1[1] * U1[1] + T1[2] * U1[2] + lowpass;
    float det = ca * cc - cb * cb;
    if (!(det != 0.0f)) return 0; /* det==0 or NaN */
    float det_inv = 1.0f / det;
    o->cov2[0] = ca; o->cov2[1] = cb; o->cov2[2] = cc;
    o->conic[0] = cc * det_inv;
    o->conic[1] = -cb * det_inv;
    o->conic[2] = ca * det_inv;
    float mid = 0.5f * (ca + cc);
    float disc = sqrtf(fmaxf(0.1f, mid * mid - det));
    float lam = fmaxf(mid + disc, mid - disc);
    float rad_f = ceilf(3.0f * sqrtf(lam));
    if (!(rad_f >= 0.0f)) return 0; /* NaN */
    if (rad_f > 1.0e9f) rad_f = 1.0e9f;
    int radius = (int)rad_f;

    const float pix_off = (st->flags & F_PIXEL_CORNER) ? 0.0f : 0.5f;
    float u = (xv - st->x_min) * st->scale - pix_off;
    float v = (yv - st->y_min) * st->scale - pix_off;
    float rf = (float)radius;
    int x0 = tile_clamp((u - rf) / (float)TILE, gx);
    int x1 = tile_clamp((u + rf + (float)(TILE - 1)) / (float)TILE, gx);
    int y0 = tile_clamp((v - rf) / (float)TILE, gy);
    int y1 = tile_clamp((v + rf + (float)(TILE - 1)) / (float)TILE, gy);
    if ((x1 - x0) * (y1 - y0) <= 0) return 0;
    o->u = u; o->v = v; o->depth = zv;
    o->radius = radius;
    o->rect[0] = x0; o->rect[1] = y0; o->rect[2] = x1; o->rect[3] = y1;
    return radius;
}

/* ------------------------------------------------------------------------------------------------ */
/* visible_filter (reference call: ortho_gaussian_renderer/preprocess.py:99-104): radii for all anchors.
   Also returns tiles_touched so a caller can size the instance list. */
int64_t gsvc_oracle_raster_preprocess(const oracle_raster_settings *st, int64_t P, const float *means3D,
                                      const float *scales, const float *rotations, int32_t *radii,
                                      int32_t *tiles_touched /* may be NULL */)
{
    int gx = (st->image_width + TILE - 1) / TILE, gy = (st->image_height + TILE - 1) / TILE;
    int64_t total = 0;
    for (int64_t i = 0; i < P; i++) {
        pre_t o;
        int r = preprocess_one(st, means3D + 3 * i, scales + 3 * i, rotations + 4 * i, gx, gy, &o);
        radii[i] = r;
        int t = r ? (o.rect[2] - o.rect[0]) * (o.rect[3] - o.rect[1]) : 0;
        if (tiles_touched) tiles_touched[i] = t;
        total += t;
    }
    return total;
}

typedef struct {
    uint32_t tile;
    uint32_t depth_bits;
    uint32_t id;
} inst_t;

static int inst_cmp(const void *pa, const void *pb)
{
    const inst_t *a = (const inst_t *)pa, *b = (const inst_t *)pb;
    if (a->tile != b->tile) return a->tile < b->tile ? -1 : 1;
    if (a->depth_bits != b->depth_bits) return a->depth_bits < b->depth_bits ? -1 : 1;
    if (a->id != b->id) return a->id < b->id ? -1 : 1;
    return 0;
}

/*
 * Forward.  Outputs (all caller-allocated):
 *   image[3,H,W], radii[P], final_T[H,W], n_contrib[H,W] (1-based position of the last contributor in the
 *   tile list, 0 = none), tile_ranges[gx*gy,2] (start,end into point_list), point_list[list_capacity]
 *   (sorted Gaussian ids), geom[P,8] = (u, v, depth, A, B, C, opacity, tiles_touched-as-float) for
 *   intermediate parity, borderline[H,W] (1 where some threshold decision of this pixel was within float
 *   rounding of its boundary; such pixels are excluded from the 1e-4 pixel-parity claim).
 * Returns num_rendered (sum of tiles touched) or -1 if list_capacity is too small.
 */
int64_t gsvc_oracle_raster_forward(const oracle_raster_settings *st, int64_t P, const float *means3D,
                                   const float *colors, const float *opacities, const float *scales,
                                   const float *rotations, float *image, int32_t *radii, float *final_T,
                                   int32_t *n_contrib, int32_t *tile_ranges, int64_t list_capacity,
                                   int32_t *point_list, float *geom, uint8_t *borderline, int num_threads)
{
    const int H = st->image_height, W = st->image_width;
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    pre_t *pre = (pre_t *)malloc(sizeof(pre_t) * (size_t)(P > 0 ? P : 1));
    int64_t total = 0;
#ifdef _OPENMP
    if (num_threads > 0) omp_set_num_threads(num_threads);
#endif
    (void)num_threads;
    /* every Gaussian on its own: the same numbers whatever the number of threads */
#pragma omp parallel for schedule(static) reduction(+ : total)
    for (int64_t i = 0; i < P; i++) {
        /* opacity <= 0 can never reach alpha >= 1/255: culled (radius 0), so un-compacted sets may be passed */
        int r = 0;
        if (opacities[i] > 0.0f) r = preprocess_one(st, means3D + 3 * i, scales + 3 * i, rotations + 4 * i, gx, gy, &pre[i]);
        else memset(&pre[i], 0, sizeof(pre_t));
        radii[i] = r;
        int t = r ? (pre[i].rect[2] - pre[i].rect[0]) * (pre[i].rect[3] - pre[i].rect[1]) : 0;
        total += t;
        if (geom) {
            float *g = geom + 8 * i;
            g[0] = pre[i].u; g[1] = pre[i].v; g[2] = pre[i].depth;
            g[3] = pre[i].conic[0]; g[4] = pre[i].conic[1]; g[5] = pre[i].conic[2];
            g[6] = opacities[i]; g[7] = (float)t;
        }
    }
    if (total > list_capacity) { free(pre); return -1; }

    /* The instances ordered by (tile, depth bits, Gaussian index): bucketed by tile (two linear passes), then every tile's
       bucket sorted on its own, in parallel — the same total order as one sort of all instances (the key is unique). */
    const int n_tiles = gx * gy;
    inst_t *inst = (inst_t *)malloc(sizeof(inst_t) * (size_t)(total > 0 ? total : 1));
    int64_t *cursor = (int64_t *)calloc((size_t)n_tiles + 1, sizeof(int64_t));
    for (int64_t i = 0; i < P; i++) {
        if (!radii[i]) continue;
        for (int ty = pre[i].rect[1]; ty < pre[i].rect[3]; ty++)
            for (int tx = pre[i].rect[0]; tx < pre[i].rect[2]; tx++) cursor[ty * gx + tx + 1]++;
    }
    for (int t = 0; t < n_tiles; t++) {
        cursor[t + 1] += cursor[t];
        tile_ranges[2 * t] = cursor[t + 1] > cursor[t] ? (int32_t)cursor[t] : 0;
        tile_ranges[2 * t + 1] = cursor[t + 1] > cursor[t] ? (int32_t)cursor[t + 1] : 0;
    }
    const int64_t n = cursor[n_tiles];
    {
        int64_t *at = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n_tiles > 0 ? n_tiles : 1));
        memcpy(at, cursor, sizeof(int64_t) * (size_t)n_tiles);
        for (int64_t i = 0; i < P; i++) {
            if (!radii[i]) continue;
            uint32_t db = order_bits((st->flags & F_DEPTH_DESCENDING) ? -pre[i].depth : pre[i].depth);
            for (int ty = pre[i].rect[1]; ty < pre[i].rect[3]; ty++)
                for (int tx = pre[i].rect[0]; tx < pre[i].rect[2]; tx++) {
                    inst_t *e = &inst[at[ty * gx + tx]++];
                    e->tile = (uint32_t)(ty * gx + tx);
                    e->depth_bits = db;
                    e->id = (uint32_t)i;
                }
        }
        free(at);
    }
#pragma omp parallel for schedule(dynamic, 16)
    for (int t = 0; t < n_tiles; t++)
        if (cursor[t + 1] - cursor[t] > 1) qsort(inst + cursor[t], (size_t)(cursor[t + 1] - cursor[t]), sizeof(inst_t), inst_cmp);
#pragma omp parallel for schedule(static)
    for (int64_t k = 0; k < n; k++) point_list[k] = (int32_t)inst[k].id;
    free(cursor);
    free(inst);

#pragma omp parallel for schedule(dynamic, 4)
    for (int t = 0; t < gx * gy; t++) {
        int tx = t % gx, ty = t / gx;
        int s0 = tile_ranges[2 * t], s1 = tile_ranges[2 * t + 1];
        for (int ly = 0; ly < TILE; ly++)
            for (int lx = 0; lx < TILE; lx++) {
                int px = tx * TILE + lx, py = ty * TILE + ly;
                if (px >= W || py >= H) continue;
                float T = 1.0f, C[3] = {0, 0, 0};
                int last = 0, contributor = 0;
                uint8_t bl = 0;
                for (int k = s0; k < s1; k++) {
                    contributor++;
                    int id = point_list[k];
                    const pre_t *g = &pre[id];
                    float dx = g->u - (float)px, dy = g->v - (float)py;
                    float power = -0.5f * (g->conic[0] * dx * dx + g->conic[2] * dy * dy) - g->conic[1] * dx * dy;
                    if (fabsf(power) < 1e-6f) bl = 1;
                    if (power > 0.0f) continue;
                    float o = opacities[id];
                    float alpha = fminf(ALPHA_MAX, o * expf(power));
                    if (fabsf(alpha - ALPHA_MIN) < 2e-5f * ALPHA_MIN) bl = 1;
                    if (alpha < ALPHA_MIN) continue;
                    float test_T = T * (1.0f - alpha);
                    if (fabsf(test_T - T_MIN) < 1e-4f * T_MIN) bl = 1;
                    if (test_T < T_MIN) break;
                    const float *c = colors + 3 * id;
                    float w = alpha * T;
                    C[0] += c[0] * w; C[1] += c[1] * w; C[2] += c[2] * w;
                    T = test_T;
                    last = contributor;
                }
                int pix = py * W + px;
                final_T[pix] = T;
                n_contrib[pix] = last;
                if (borderline) borderline[pix] = bl;
                for (int ch = 0; ch < 3; ch++) image[ch * H * W + pix] = C[ch] + T * st->bg[ch];
            }
    }
    free(pre);
    return total;
}

/* ------------------------------------------------------------------------------------------------ */
/*
 * Backward.  Inputs: the forward's inputs, its point_list / tile_ranges / final_T / n_contrib, and
 * dL_dimage[3,H,W].  Outputs (caller-allocated, overwritten): dL_dmeans3D[P,3], dL_dmeans2D[P,3]
 * (NDC-scaled screen gradient: (dL/du*0.5*W, dL/dv*0.5*H, 0)), dL_dcolors[P,3], dL_dopacity[P,1],
 * dL_dscales[P,3], dL_drotations[P,4].  Per-Gaussian sums are accumulated in double.
 */
/* num_threads <= 1: the serial statement (one accumulator per Gaussian, contributions added tile after tile, pixel after
 * pixel: the numbers every fixture was generated with).  num_threads > 1 (bench.py's CPU baseline): the tiles are processed in
 * parallel into one accumulator per (tile, Gaussian) INSTANCE, and a Gaussian's instances are then added in tile order — the
 * same contributions in the same order within a tile, associated per tile instead of one by one (differences of a few 1e-16
 * relative in the double sums, the same result whatever the number of threads). */
void gsvc_oracle_raster_backward_mt(const oracle_raster_settings *st, int64_t P, const float *means3D,
                                    const float *colors, const float *opacities, const float *scales,
                                    const float *rotations, const int32_t *radii, const int32_t *tile_ranges,
                                    const int32_t *point_list, const float *final_T, const int32_t *n_contrib,
                                    const float *dL_dimage, float *dL_dmeans3D, float *dL_dmeans2D,
                                    float *dL_dcolors, float *dL_dopacity, float *dL_dscales, float *dL_drotations,
                                    int num_threads)
{
    const int H = st->image_height, W = st->image_width;
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    const int mt = num_threads > 1;
#ifdef _OPENMP
    const int threads_before = omp_get_max_threads();      /* restored below: the setting is process-wide */
    omp_set_num_threads(mt ? num_threads : 1);
#endif
    pre_t *pre = (pre_t *)malloc(sizeof(pre_t) * (size_t)(P > 0 ? P : 1));
    /* per-Gaussian accumulators: du dv dA dB dC dopacity dcol[3] */
    double *acc = (double *)calloc((size_t)(P > 0 ? P : 1) * 9, sizeof(double));
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < P; i++)
        preprocess_one(st, means3D + 3 * i, scales + 3 * i, rotations + 4 * i, gx, gy, &pre[i]);
    int64_t n_inst = 0;
    for (int t = 0; t < gx * gy; t++)
        if (tile_ranges[2 * t + 1] > n_inst) n_inst = tile_ranges[2 * t + 1];
    double *iacc = mt ? (double *)calloc((size_t)(n_inst > 0 ? n_inst : 1) * 9, sizeof(double)) : NULL;

#pragma omp parallel for schedule(dynamic, 4)
    for (int t = 0; t < gx * gy; t++) {
        int tx = t % gx, ty = t / gx;
        int s0 = tile_ranges[2 * t];
        for (int ly = 0; ly < TILE; ly++)
            for (int lx = 0; lx < TILE; lx++) {
                int px = tx * TILE + lx, py = ty * TILE + ly;
                if (px >= W || py >= H) continue;
                int pix = py * W + px;
                const float Tf = final_T[pix];
                float T = Tf;
                int last = n_contrib[pix];
                float dpix[3] = {dL_dimage[pix], dL_dimage[H * W + pix], dL_dimage[2 * H * W + pix]};
                float bg_dot = st->bg[0] * dpix[0] + st->bg[1] * dpix[1] + st->bg[2] * dpix[2];
                float behind[3] = {0, 0, 0}; /* colour accumulated behind the current Gaussian */
                float last_alpha = 0.0f, last_col[3] = {0, 0, 0};
                for (int k = s0 + last - 1; k >= s0; k--) {
                    int id = point_list[k];
                    const pre_t *g = &pre[id];
                    float dx = g->u - (float)px, dy = g->v - (float)py;
                    float power = -0.5f * (g->conic[0] * dx * dx + g->conic[2] * dy * dy) - g->conic[1] * dx * dy;
                    if (power > 0.0f) continue;
                    float o = opacities[id];
                    float G = expf(power);
                    float alpha = fminf(ALPHA_MAX, o * G);
                    if (alpha < ALPHA_MIN) continue;
                    T = T / (1.0f - alpha);
                    float w = alpha * T;
                    const float *c = colors + 3 * id;
                    float dL_dalpha = 0.0f;
                    double *a = mt ? iacc + 9 * (int64_t)k : acc + 9 * (int64_t)id;
                    for (int ch = 0; ch < 3; ch++) {
                        behind[ch] = last_alpha * last_col[ch] + (1.0f - last_alpha) * behind[ch];
                        last_col[ch] = c[ch];
                        dL_dalpha += (c[ch] - behind[ch]) * dpix[ch];
                        a[6 + ch] += (double)(w * dpix[ch]);
                    }
                    dL_dalpha *= T;
                    last_alpha = alpha;
                    dL_dalpha += (-Tf / (1.0f - alpha)) * bg_dot;
                    /* the 0.99 clamp is not special-cased (gradient flows as if unclamped), as in the
                       published algorithm */
                    float dL_dG = o * dL_dalpha;
                    if ((st->flags & F_CLAMP_STOPS_GRADIENT) && o * G > ALPHA_MAX) {
                        dL_dG = 0.0f;      /* alpha = 0.99 there: it depends neither on G nor on the opacity */
                        dL_dalpha = 0.0f;  /* (only its use for dL/dopacity below: the recurrences are done with it) */
                    }
                    float gdx = G * dx, gdy = G * dy;
                    float dG_du = -gdx * g->conic[0] - gdy * g->conic[1];
                    float dG_dv = -gdy * g->conic[2] - gdx * g->conic[1];
                    a[0] += (double)(dL_dG * dG_du);
                    a[1] += (double)(dL_dG * dG_dv);
                    a[2] += (double)(-0.5f * gdx * dx * dL_dG);
                    a[3] += (double)(-gdx * dy * dL_dG); /* true d/dB (cross term counted in full) */
                    a[4] += (double)(-0.5f * gdy * dy * dL_dG);
                    a[5] += (double)(G * dL_dalpha);
                }
            }
    }

    if (mt) {
        /* a Gaussian's instances in list order = tile order: CSR of the instance indices per Gaussian, then one sum each */
        int64_t *off = (int64_t *)calloc((size_t)P + 2, sizeof(int64_t));
        for (int64_t k = 0; k < n_inst; k++) off[point_list[k] + 2]++;
        for (int64_t i = 0; i < P; i++) off[i + 2] += off[i + 1];
        int64_t *idx = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n_inst > 0 ? n_inst : 1));
        for (int64_t k = 0; k < n_inst; k++) idx[off[point_list[k] + 1]++] = k;
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < P; i++)
            for (int64_t j = off[i]; j < off[i + 1]; j++)
                for (int c = 0; c < 9; c++) acc[9 * i + c] += iacc[9 * idx[j] + c];
        free(idx);
        free(off);
        free(iacc);
    }

    const float *M = st->viewmatrix;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < P; i++) {
        float *gm3 = dL_dmeans3D + 3 * i, *gm2 = dL_dmeans2D + 3 * i, *gc = dL_dcolors + 3 * i;
        float *gs = dL_dscales + 3 * i, *gq = dL_drotations + 4 * i;
        gm3[0] = gm3[1] = gm3[2] = 0; gm2[0] = gm2[1] = gm2[2] = 0; gc[0] = gc[1] = gc[2] = 0;
        gs[0] = gs[1] = gs[2] = 0; gq[0] = gq[1] = gq[2] = gq[3] = 0; dL_dopacity[i] = 0;
        if (!radii[i]) continue;
        const double *a = acc + 9 * i;
        double du = a[0], dv = a[1], dA = a[2], dB = a[3], dC = a[4];
        if (st->flags & F_MEANS2D_PIXEL_UNITS) {
            gm2[0] = (float)du;
            gm2[1] = (float)dv;
        } else {
            gm2[0] = (float)(du * 0.5 * W);
            gm2[1] = (float)(dv * 0.5 * H);
        }
        dL_dopacity[i] = (float)a[5];
        gc[0] = (float)a[6]; gc[1] = (float)a[7]; gc[2] = (float)a[8];
        /* means: u = (M0.p + M03 - x_min)*scale - .5 */
        for (int j = 0; j < 3; j++) gm3[j] = (float)((double)st->scale * ((double)M[j] * du + (double)M[4 + j] * dv));

        /* conic (A,B,C) = (c,-b,a)/det  ->  (a,b,c) */
        double ca = pre[i].cov2[0], cb = pre[i].cov2[1], cc = pre[i].cov2[2];
        double det = ca * cc - cb * cb;
        double inv2 = 1.0 / (det * det + 1e-7);
        /* dA/da = -c^2/det^2, dA/db = 2bc/det^2, dA/dc = (det - ac)/det^2 = -b^2/det^2
           dB/da = bc/det^2,   dB/db = -(det + 2b^2)/det^2, dB/dc = ab/det^2
           dC/da = -b^2/det^2, dC/db = 2ab/det^2, dC/dc = -a^2/det^2 */
        double dLa = inv2 * (-cc * cc * dA + cb * cc * dB + (det - ca * cc) * dC);
        double dLc = inv2 * (-ca * ca * dC + ca * cb * dB + (det - ca * cc) * dA);
        double dLb = inv2 * (2.0 * cb * cc * dA - (det + 2.0 * cb * cb) * dB + 2.0 * ca * cb * dC);
        /* Sigma2 = T Sigma3 T^T + hI, symmetric gradient matrix [[dLa, dLb/2],[dLb/2, dLc]] */
        double T0[3] = {(double)st->scale * M[0], (double)st->scale * M[1], (double)st->scale * M[2]};
        double T1[3] = {(double)st->scale * M[4], (double)st->scale * M[5], (double)st->scale * M[6]};
        double G3[3][3];
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++)
                G3[r][c] = T0[r] * dLa * T0[c] + 0.5 * dLb * (T0[r] * T1[c] + T1[r] * T0[c]) + T1[r] * dLc * T1[c];
        /* Sigma3 = L L^T, L = R diag(S):  dL/dL = 2 G3 L (G3 symmetric) */
        const float *q = rotations + 4 * i, *s = scales + 3 * i;
        double qr = q[0], qx = q[1], qy = q[2], qz = q[3];
        double R[3][3] = {{1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qr * qz), 2 * (qx * qz + qr * qy)},
                          {2 * (qx * qy + qr * qz), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qr * qx)},
                          {2 * (qx * qz - qr * qy), 2 * (qy * qz + qr * qx), 1 - 2 * (qx * qx + qy * qy)}};
        double S[3] = {(double)st->scale_modifier * s[0], (double)st->scale_modifier * s[1],
                       (double)st->scale_modifier * s[2]};
        double dLL[3][3], dR[3][3];
        for (int r = 0; r < 3; r++)
            for (int k = 0; k < 3; k++) {
                double v = 0;
                for (int c = 0; c < 3; c++) v += G3[r][c] * R[c][k] * S[k];
                dLL[r][k] = 2.0 * v;
            }
        for (int k = 0; k < 3; k++) {
            double v = 0;
            for (int r = 0; r < 3; r++) { v += dLL[r][k] * R[r][k]; dR[r][k] = dLL[r][k] * S[k]; }
            gs[k] = (float)(v * (double)st->scale_modifier);
        }
        /* R(q) polynomial derivatives (q not normalised here) */
        double dr = 2 * (-qz * dR[0][1] + qy * dR[0][2] + qz * dR[1][0] - qx * dR[1][2] - qy * dR[2][0] + qx * dR[2][1]);
        double dx = 2 * (qy * dR[0][1] + qz * dR[0][2] + qy * dR[1][0] - 2 * qx * dR[1][1] - qr * dR[1][2] + qz * dR[2][0] + qr * dR[2][1] - 2 * qx * dR[2][2]);
        double dy = 2 * (-2 * qy * dR[0][0] + qx * dR[0][1] + qr * dR[0][2] + qx * dR[1][0] + qz * dR[1][2] - qr * dR[2][0] + qz * dR[2][1] - 2 * qy * dR[2][2]);
        double dz = 2 * (-2 * qz * dR[0][0] - qr * dR[0][1] + qx * dR[0][2] + qr * dR[1][0] - 2 * qz * dR[1][1] + qy * dR[1][2] + qx * dR[2][0] + qy * dR[2][1]);
        gq[0] = (float)dr; gq[1] = (float)dx; gq[2] = (float)dy; gq[3] = (float)dz;
    }
    free(acc);
    free(pre);
#ifdef _OPENMP
    omp_set_num_threads(threads_before);
#endif
}

void gsvc_oracle_raster_backward(const oracle_raster_settings *st, int64_t P, const float *means3D,
                                 const float *colors, const float *opacities, const float *scales,
                                 const float *rotations, const int32_t *radii, const int32_t *tile_ranges,
                                 const int32_t *point_list, const float *final_T, const int32_t *n_contrib,
                                 const float *dL_dimage, float *dL_dmeans3D, float *dL_dmeans2D,
                                 float *dL_dcolors, float *dL_dopacity, float *dL_dscales, float *dL_drotations)
{
    gsvc_oracle_raster_backward_mt(st, P, means3D, colors, opacities, scales, rotations, radii, tile_ranges, point_list, final_T,
                                   n_contrib, dL_dimage, dL_dmeans3D, dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dscales,
                                   dL_drotations, 1);
}

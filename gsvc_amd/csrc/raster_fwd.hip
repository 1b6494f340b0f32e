// gsvc_amd/csrc/raster_fwd.hip — forward of the orthographic sliding-window tile rasterizer, gfx950.
//
// Replaces GaussianRasterizer.forward / .visible_filter of the external CUDA extension GSVC imports
// (call sites reference ortho_gaussian_renderer/renderer.py:90-98, preprocess.py:99-104).
//
// Pipeline (all on the caller's stream, no host sync, no allocation):
//   K1 preprocess   1024-lane workgroups, one lane per Gaussian: project, cull to slab/screen, conic, radius,
//                   tile rectangle, exact alpha>=1/255 pixel bounding box.  Instances are counted per tile in
//                   an LDS histogram of the whole tile grid (LDS atomics give each instance its rank inside
//                   the workgroup); the histogram is then flushed with CONTIGUOUS 256-B returning atomic
//                   wave-instructions (the shape that runs at the full memory-side atomic rate — scattered
//                   per-instance atomics are ~16x slower on MI355X), which also yields each workgroup's base
//                   inside every tile, so every instance leaves K1 knowing its slot in its tile's segment.
//   K2 scan_tiles   one workgroup: exclusive scan of the per-tile counts -> tile_offsets[T+1], counters
//   K3 scatter      one lane per Gaussian: writes (depth_bits<<32 | id) at tile_offsets[tile] + slot.  No atomics
//                   (Gaussians touching more than BIN_SLOTS tiles use a per-tile cursor for the remainder).
//   K4 sort_tiles   one WAVE per tile: rank sort in LDS (<=256 entries: every key is compared with every
//                   other, no barriers), bitonic in LDS (<=1024); whole workgroups for the rare longer lists.
//                   Orders each segment by (depth, id) -> point_list.  No device-wide radix sort: segments
//                   are independent, depth ordering is a wavefront-local problem, and the tile boundaries
//                   come for free from the scan.
//   K5 blend        one wave per 8x8 pixel quadrant of a 16x16 tile, no barriers.  Per chunk of 64 list entries
//                   one vector test of the entries' alpha bounding boxes against the wave's quadrant keeps
//                   only the Gaussians that can reach alpha >= 1/255 there (exact: skipped pairs contribute
//                   nothing in the spec either); survivors are compacted into a wave-private LDS strip and
//                   composited front to back from LDS broadcast reads by a branch-free inner loop.
#include "raster_common.h"

#include <cstdlib>

namespace gsvc {

__device__ __forceinline__ uint32_t pack_i16(int lo, int hi)
{
    return ((uint32_t)lo & 0xffffu) | ((uint32_t)hi << 16);
}

// Pixel bounding box outside which alpha = opacity*exp(power) < 1/255 for certain (margins cover the
// rounding of power in the blend loop).  Derived from the conic the blend loop itself uses.
__device__ __forceinline__ void alpha_bbox(float u, float v, float A, float B, float C, float opacity,
                                           uint32_t &bx, uint32_t &by)
{
    const float BIG = 30000.0f;
    float x0 = -BIG, x1 = BIG, y0 = -BIG, y1 = BIG;
    const float tau = logf(255.0f * opacity) + 1e-3f;
    const float detc = A * C - B * B;
    if (opacity <= 0.0f || tau < 0.0f) {
        x0 = 1.0f; x1 = 0.0f; y0 = 1.0f; y1 = 0.0f;  // can never reach 1/255: empty box
    } else if (detc > 1e-3f * A * C && tau == tau) {
        const float ex = sqrtf(2.0f * tau * C / detc) * 1.002f + 0.02f;
        const float ey = sqrtf(2.0f * tau * A / detc) * 1.002f + 0.02f;
        x0 = fmaxf(floorf(u - ex), -BIG); x1 = fminf(ceilf(u + ex), BIG);
        y0 = fmaxf(floorf(v - ey), -BIG); y1 = fminf(ceilf(v + ey), BIG);
    }
    bx = pack_i16((int)x0, (int)x1);
    by = pack_i16((int)y0, (int)y1);
}

// ------------------------------------------------------------------------------------------- K1 (filter)
__global__ void __launch_bounds__(256) k_visible_filter(RasterParams st, int P, const float *__restrict__ means3D,
                                                        const float *__restrict__ scales,
                                                        const float *__restrict__ rotations, int32_t *__restrict__ radii)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const float4 q = reinterpret_cast<const float4 *>(rotations)[i];
    PreOut o;
    radii[i] = preprocess_gaussian(st, means3D[3 * i], means3D[3 * i + 1], means3D[3 * i + 2], scales[3 * i],
                                   scales[3 * i + 1], scales[3 * i + 2], q.x, q.y, q.z, q.w, o);
}

// The visibility test of R views over the same anchors in one launch (a fitting step tests its four views): the anchor is read
// once, masks[r][i] = radius under view r > 0.  The per-anchor scales / rotations may be given raw — scales as their logarithm
// (``scale_exp``: scene/gaussian_model.py scaling_activation = exp), rows ``scale_stride`` floats apart (the first three of the
// six per-anchor scales are the ones tested), rotations un-normalised (``rot_normalise``: rotation_activation = normalize with
// eps 1e-12) — which folds the activations' elementwise passes into the test.
constexpr int VIS_MAX_VIEWS = 8;
struct VisViews { RasterParams p[VIS_MAX_VIEWS]; };

__global__ void __launch_bounds__(256) k_visible_masks(VisViews v, int R, int P, const float *__restrict__ means3D,
                                                       const float *__restrict__ scales, int scale_stride, int scale_exp,
                                                       const float *__restrict__ rotations, int rot_normalise, uint8_t *__restrict__ masks)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    float4 q = reinterpret_cast<const float4 *>(rotations)[i];
    if (rot_normalise) {
        const float n = fmaxf(sqrtf((q.x * q.x + q.y * q.y) + (q.z * q.z + q.w * q.w)), 1e-12f);
        q.x /= n; q.y /= n; q.z /= n; q.w /= n;
    }
    float sx = scales[(size_t)scale_stride * i], sy = scales[(size_t)scale_stride * i + 1], sz = scales[(size_t)scale_stride * i + 2];
    if (scale_exp) { sx = expf(sx); sy = expf(sy); sz = expf(sz); }
    const float x = means3D[3 * i], y = means3D[3 * i + 1], z = means3D[3 * i + 2];
#pragma unroll
    for (int r = 0; r < VIS_MAX_VIEWS; r++)
        if (r < R) {
            PreOut o;
            masks[(size_t)r * P + i] = preprocess_gaussian(v.p[r], x, y, z, sx, sy, sz, q.x, q.y, q.z, q.w, o) > 0 ? 1 : 0;
        }
}

// ------------------------------------------------------------------------------------------------- K2
// tile_offsets = exclusive scan of (tile_count + tile_extra); tile_extra is zeroed (K3's cursor for the
// instances beyond BIN_SLOTS, which sit behind the tile_count[t] slotted ones).  One workgroup; rounds of 8192
// tiles staged through LDS so that every global access is a coalesced 4-byte-per-lane stream.  Tiles whose list
// is too long for the one-wave sort are appended to big_list for the workgroup sort.
constexpr int SCAN_CHUNK = 8192;
constexpr int SORT_RANK_MAX = 256;    // entries one wave rank-sorts (4 keys per lane)
constexpr int SORT_WAVE_MAX = 1024;   // entries one wave sorts in LDS (8 KiB)
#ifdef GSVC_SORT_NO_BUCKET
constexpr bool SORT_DEBUG_NO_BUCKET = true;      // timing / A-B experiment: the round-2 paths only
#else
constexpr bool SORT_DEBUG_NO_BUCKET = false;
#endif

// s_v: SCAN_CHUNK ints of LDS; called by all 1024 threads of one workgroup.
__device__ __forceinline__ void scan_tiles_body(int T, const int32_t *__restrict__ tile_count, int32_t *__restrict__ tile_extra,
                                                int32_t *__restrict__ tile_offsets, int32_t *__restrict__ big_list,
                                                gsvc_raster_counters *__restrict__ counters, long long max_instances,
                                                int *__restrict__ s_v, int api_count_in_reserved2 = 0)
{
    __shared__ int wave_sum[16];
    __shared__ int wave_max[16];
    __shared__ int s_big;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_big = 0;
    int carry = 0, local_max = 0;
    for (int base = 0; base < T; base += SCAN_CHUNK) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int idx = base + k * 1024 + tid;
            s_v[k * 1024 + tid] = idx < T ? __hip_atomic_load(tile_count + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) +
                                                __hip_atomic_load(tile_extra + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                          : 0;
        }
        __syncthreads();
        int v[8];
        int sum = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            v[k] = s_v[tid * 8 + k];
            local_max = max(local_max, v[k]);
            sum += v[k];
            if (v[k] > SORT_WAVE_MAX) big_list[atomicAdd(&s_big, 1)] = base + tid * 8 + k;
        }
        int x = sum;  // inclusive wave scan of the per-lane sums
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int y = __shfl_up(x, d, 64);
            if (lane >= d) x += y;
        }
        if (lane == 63) wave_sum[wave] = x;
        __syncthreads();
        int prefix = carry, total = 0;
        for (int w = 0; w < 16; w++) {
            const int ws = wave_sum[w];
            if (w < wave) prefix += ws;
            total += ws;
        }
        int run = prefix + x - sum;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            s_v[tid * 8 + k] = run;
            run += v[k];
        }
        carry += total;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int idx = base + k * 1024 + tid;
            if (idx < T) {
                tile_offsets[idx] = s_v[k * 1024 + tid];
                tile_extra[idx] = 0;
            }
        }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) local_max = max(local_max, __shfl_xor(local_max, m, 64));
    if (lane == 0) wave_max[wave] = local_max;
    __syncthreads();
    if (tid == 0) {
        int mx = 0;
        for (int w = 0; w < 16; w++) mx = max(mx, wave_max[w]);
        tile_offsets[T] = carry;
        // GSVC_RASTER_TIGHT_BINNING: the lists are shorter than the API's count (the 3-sigma rectangles' tiles, summed by K1)
        counters->num_rendered = api_count_in_reserved2 ? __hip_atomic_load(&counters->reserved[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : carry;
        counters->overflow = ((long long)carry > max_instances) ? 1 : 0;
        counters->reserved[2] = carry;      // the LISTS' length (= num_rendered for the 3-sigma lists): what max_instances must hold
        counters->max_tile_len = mx;
        counters->num_big_tiles = s_big;
    }
}

__global__ void __launch_bounds__(1024) k_scan_tiles(int T, const int32_t *__restrict__ tile_count,
                                                     int32_t *__restrict__ tile_extra,
                                                     int32_t *__restrict__ tile_offsets,
                                                     int32_t *__restrict__ big_list,
                                                     gsvc_raster_counters *__restrict__ counters,
                                                     long long max_instances, int api_count_in_reserved2)
{
    __shared__ int s_v[SCAN_CHUNK];
    scan_tiles_body(T, tile_count, tile_extra, tile_offsets, big_list, counters, max_instances, s_v, api_count_in_reserved2);
}

// Walk the tiles j >= BIN_SLOTS ("extras") of every lane's binning rectangle (row-major, x0 / w / y0 / nt per lane; nt = 0:
// nothing).  A lane that walked a 60-tile rectangle alone would keep its 63 neighbours waiting (footprints are very uneven:
// most Gaussians of a fitting render are culled or small), so rectangles of COOP_MIN_TILES tiles or more are walked by the
// whole wave, 64 tiles per step; smaller ones by their own lane.  f(owner lane, ty, tx).
constexpr int COOP_MIN_TILES = 24;

template <typename OWN, typename F>
__device__ __forceinline__ void for_each_extra(int lane, int nt, int x0, int w, int y0, OWN &&owner, F &&f)
{
    // owner(src) fetches what f needs from the rectangle's owner lane; it runs with the whole wave active (cross-lane reads
    // inside the partially active tile loop could meet an inactive owner)
    unsigned long long big = __ballot(nt >= COOP_MIN_TILES);
    while (big != 0ull) {
        const int src = __builtin_ctzll(big);
        big &= big - 1ull;
        const int sx0 = __shfl(x0, src, 64), sw = __shfl(w, src, 64), sy0 = __shfl(y0, src, 64), snt = __shfl(nt, src, 64);
        const auto od = owner(src);
        for (int j = BIN_SLOTS + lane; j < snt; j += 64) {
            const int jy = j / sw, jx = j - jy * sw;
            f(od, sy0 + jy, sx0 + jx);
        }
    }
    const auto own = owner(lane);
    if (nt < COOP_MIN_TILES)
        for (int j = BIN_SLOTS; j < nt; j++) {
            const int jy = j / w, jx = j - jy * w;
            f(own, y0 + jy, x0 + jx);
        }
}

// ------------------------------------------------------------------------------------------------- K1
// USE_LDS: per-workgroup histogram of the whole tile grid in dynamic LDS (4*T bytes).  Otherwise (grids too
// large for LDS) every instance does its own global atomic.
// PAIR: bin for the two-view frame — a Gaussian is listed in the union of this view's tile rectangle and the
// opposite view's rectangle mirrored into this view's tile grid (the 3-sigma rectangle formula is not mirror
// symmetric, so the two differ by up to one tile column); K3 tags every instance with the views it belongs to.
template <bool USE_LDS, bool PAIR>
__global__ void __launch_bounds__(1024) k_preprocess(RasterParams st, int P, const float *__restrict__ means3D,
                                                     const float *__restrict__ colors,
                                                     const float *__restrict__ opacities,
                                                     const float *__restrict__ scales,
                                                     const float *__restrict__ rotations, int32_t *__restrict__ radii,
                                                     GeomRec *__restrict__ geom, BinRec *__restrict__ bins,
                                                     int32_t *__restrict__ tile_count, int32_t *__restrict__ tile_extra,
                                                     gsvc_raster_counters *__restrict__ counters,
                                                     int32_t *__restrict__ tile_offsets, int32_t *__restrict__ big_list,
                                                     int32_t *__restrict__ wg_extras, long long max_instances)
{
    extern __shared__ int hist[];
    __shared__ int s_wt[16];
    __shared__ int s_gbase;
    int row_excl = 0;
    const int T = st.gx * st.gy;
    const int tid = threadIdx.x;
    if (USE_LDS) {
        for (int t = tid; t < T; t += 1024) hist[t] = 0;
        __syncthreads();
    }
    const int i = blockIdx.x * 1024 + tid;
    int radius = 0;
    PreOut o;
    int slot[BIN_SLOTS] = {0, 0, 0, 0};
    int n_extra = 0;        // this Gaussian's instances beyond BIN_SLOTS
    bool heavy = false;     // workgroup-uniform: extras go through the LDS histogram
    int bx0 = 0, bx1 = 0;   // x range of the binning rectangle (pair mode: union of the two views)
    int by0 = 0, by1 = 0;   // its y range (the 3-sigma rectangle's, or what the alpha box leaves of it: GSVC_RASTER_TIGHT_BINNING)
    const bool tight = (st.flags & GSVC_RASTER_TIGHT_BINNING) != 0;
    uint32_t abx = pack_i16(1, 0), aby = pack_i16(1, 0);      // alpha box (empty)
    int n3 = 0;             // tiles of the 3-sigma rectangle: what num_rendered counts whatever is listed
    uint32_t pair_fx = 0u;  // pair mode: this view's x range as the record keeps it (0 = not in this view's lists)
    bool listed = false;
    int my_tiles = 0;       // instances of this Gaussian = rows it owns in the backward's partial-sum buffer
    float rec2_b = 0.f;
    uint32_t rec2_bx = 0u, rec2_by = 0u;
    uint32_t ub_bits = 0u;     // pair mode: the opposite view's pixel x of this Gaussian (float bits), kept in GeomRec::pad
    if (i < P) {
        // a Gaussian with opacity <= 0 can never reach alpha >= 1/255: culled here (radius 0), which lets callers
        // pass un-compacted Gaussian sets (GSVC's "opacity > 0" selection) without a host-side compaction
        const float op = opacities[i];
        o.radius_raw = 0;
        if (op > 0.0f) {
            const float4 q = reinterpret_cast<const float4 *>(rotations)[i];
            radius = preprocess_gaussian(st, means3D[3 * i], means3D[3 * i + 1], means3D[3 * i + 2], scales[3 * i],
                                         scales[3 * i + 1], scales[3 * i + 2], q.x, q.y, q.z, q.w, o);
        }
        radii[i] = radius;
        GeomRec rec;
        BinRec br;
        br.pad = 0u;
        listed = radius > 0;
        by0 = o.y0; by1 = o.y1;
        if (radius > 0) {
            alpha_bbox(o.u, o.v, o.A, o.B, o.C, op, abx, aby);
            n3 = (o.x1 - o.x0) * (o.y1 - o.y0);      // (pair mode: replaced below by the union's tiles, what its 3-sigma lists hold)
        }
        if (!PAIR) {
            bx0 = o.x0; bx1 = o.x1;
            if (tight && radius > 0) {
                // the tiles of the 3-sigma rectangle that the alpha box touches (bbox_hits_tile's rule: pixel interval [lo, hi] against
                // the tile's 16 pixels); an empty box (opacity too small to ever reach 1/255) lists nothing
                const int lx = (int)(int16_t)(abx & 0xffffu), hx = (int)(int16_t)(abx >> 16), ly = (int)(int16_t)(aby & 0xffffu), hy = (int)(int16_t)(aby >> 16);
                if (lx > hx || ly > hy) { bx0 = bx1 = by0 = by1 = 0; }
                else {
                    bx0 = max(o.x0, lx >> 4); bx1 = min(o.x1, (hx >> 4) + 1);
                    by0 = max(o.y0, ly >> 4); by1 = min(o.y1, (hy >> 4) + 1);
                    if (bx1 <= bx0 || by1 <= by0) { bx0 = bx1 = by0 = by1 = 0; }
                }
                listed = (bx1 - bx0) * (by1 - by0) > 0;
            }
        } else if (o.radius_raw > 0) {
            // opposite view: x_view' = -x_view, same y, same radius; rectangle by the same formula, then mirrored.  ub is
            // what preprocess_gaussian() computes for u under view_matrix_s (same operations, no FMA contraction): k_blend<PAIR>
            // composites the opposite view at it
            float ub;
            {
#pragma clang fp contract(off)
                ub = (-o.xv - st.x_min) * st.scale - st.pix_off;
            }
            ub_bits = __float_as_uint(ub);
            const float rf = (float)o.radius_raw;
            const int xb0 = tile_clamp((ub - rf) / (float)TILE, st.gx), xb1 = tile_clamp((ub + rf + (float)(TILE - 1)) / (float)TILE, st.gx);
            const bool vis_b = (xb1 - xb0) * (o.y1 - o.y0) > 0;
            int mx0 = st.gx - xb1, mx1 = st.gx - xb0;
            int fx0 = o.x0, fx1 = o.x1;
            bool in_f = radius > 0, in_b = vis_b;
            if (tight && (in_f || in_b)) {
                // both views' rectangles against the tiles the alpha box touches (the opposite view's box is this view's, mirrored into
                // this view's tile grid with the rectangle: one box serves both, as in k_blend<PAIR>); K3 forms the union from the two
                // clamped x ranges it finds in the record
                if (!in_f) alpha_bbox(o.u, o.v, o.A, o.B, o.C, op, abx, aby);
                const int lx = (int)(int16_t)(abx & 0xffffu), hx = (int)(int16_t)(abx >> 16), ly = (int)(int16_t)(aby & 0xffffu), hy = (int)(int16_t)(aby >> 16);
                if (lx > hx || ly > hy) in_f = in_b = false;
                else {
                    by0 = max(o.y0, ly >> 4); by1 = min(o.y1, (hy >> 4) + 1);
                    fx0 = max(fx0, lx >> 4); fx1 = min(fx1, (hx >> 4) + 1);
                    mx0 = max(mx0, lx >> 4); mx1 = min(mx1, (hx >> 4) + 1);
                    if (by1 <= by0) in_f = in_b = false;
                    in_f = in_f && fx1 > fx0;
                    in_b = in_b && mx1 > mx0;
                }
                if (!in_f && !in_b) by0 = by1 = 0;
            }
            pair_fx = in_f ? ((uint32_t)fx0 | ((uint32_t)fx1 << 16)) : 0u;
            if (in_b) br.pad = (uint32_t)mx0 | ((uint32_t)mx1 << 16);
            if (in_f && in_b) { bx0 = min(fx0, mx0); bx1 = max(fx1, mx1); }
            else if (in_f) { bx0 = fx0; bx1 = fx1; }
            else if (in_b) { bx0 = mx0; bx1 = mx1; }
            listed = in_f || in_b;
            {
                const int ux0 = (radius > 0 && vis_b) ? min(o.x0, st.gx - xb1) : (radius > 0 ? o.x0 : st.gx - xb1);
                const int ux1 = (radius > 0 && vis_b) ? max(o.x1, st.gx - xb0) : (radius > 0 ? o.x1 : st.gx - xb0);
                n3 = (radius > 0 || vis_b) ? (ux1 - ux0) * (o.y1 - o.y0) : 0;
            }
        }
        if (listed) {
            rec.u = o.u; rec.v = o.v; rec.A = o.A; rec.B = o.B;
            rec.C = o.C; rec.opacity = op;
            rec.r = colors[3 * i + 0]; rec.g = colors[3 * i + 1]; rec.b = colors[3 * i + 2];
            rec.depth = o.depth;
            if (PAIR && radius <= 0 && !tight) alpha_bbox(o.u, o.v, o.A, o.B, o.C, rec.opacity, abx, aby);      // listed for the opposite view only
            rec.bbox_x = abx; rec.bbox_y = aby;
            br.depth = (st.flags & GSVC_RASTER_DEPTH_DESCENDING) ? -o.depth : o.depth;      // the sort key
            br.rect_x = PAIR ? pair_fx : (radius > 0 ? ((uint32_t)bx0 | ((uint32_t)bx1 << 16)) : 0u);
            br.rect_y = (uint32_t)by0 | ((uint32_t)by1 << 16);
            rec.rect_x = br.rect_x; rec.rect_y = br.rect_y;
            my_tiles = (bx1 - bx0) * (by1 - by0);
            // the first BIN_SLOTS tiles of the rectangle (row-major): rank inside the workgroup from the LDS histogram
            // (histogram word of a tile: low half = slotted instances of this workgroup, high half = its extras in the
            // heavy case; at most 1024 each: no carry).  The remaining tiles ("extras") follow below.
            int j = 0;
            for (int ty = by0; ty < by1 && j < BIN_SLOTS; ty++)
                for (int tx = bx0; tx < bx1 && j < BIN_SLOTS; tx++, j++) {
                    const int t = ty * st.gx + tx;
                    const int r = USE_LDS ? (atomicAdd(&hist[t], 1) & 0xffff) : atomicAdd(&tile_count[t], 1);
                    if (j == 0) slot[0] = r;
                    if (j == 1) slot[1] = r;
                    if (j == 2) slot[2] = r;
                    if (j == 3) slot[3] = r;
                }
            n_extra = max((bx1 - bx0) * (by1 - by0) - BIN_SLOTS, 0);
        } else {
            rec.u = rec.v = rec.A = rec.B = rec.C = rec.opacity = rec.r = rec.g = rec.b = rec.depth = 0.f;
            rec.bbox_x = pack_i16(1, 0); rec.bbox_y = pack_i16(1, 0);
            rec.rect_x = rec.rect_y = 0u;
            br.depth = 0.f; br.rect_x = br.rect_y = 0u;
        }
        rec.goff = 0; rec.pad = PAIR ? ub_bits : 0u;
        // the record of a Gaussian that is in no list is never read (the sort, the compositing kernels and the backward reach
        // records through list entries or behind radii > 0): a fitting render culls ~70 % of what it submits (opacity <= 0), and
        // their 64 + 16 bytes of zeros were a quarter of this kernel's stores
        if (listed || radius > 0) {      // (radius > 0 but in no list — tight binning: the backward reads its empty rectangle behind radii > 0)
            float4 *dst = reinterpret_cast<float4 *>(geom + i);
            const float4 *src = reinterpret_cast<const float4 *>(&rec);
            dst[0] = src[0]; dst[1] = src[1]; dst[3] = src[3];
        }
        // word 2 (blue, alpha box, goff) is stored once the workgroup's row range is known (below)
        rec2_b = rec.b; rec2_bx = rec.bbox_x; rec2_by = rec.bbox_y;
        if (!USE_LDS && listed) {
            br.slot[0] = slot[0]; br.slot[1] = slot[1]; br.slot[2] = slot[2]; br.slot[3] = slot[3];
            float4 *bd = reinterpret_cast<float4 *>(bins + i);
            const float4 *bs = reinterpret_cast<const float4 *>(&br);
            bd[0] = bs[0]; bd[1] = bs[1];
        } else {
            // first half: depth + rectangles (all zero = in no list: K3 reads the slots only behind a non-empty rectangle)
            reinterpret_cast<float4 *>(bins + i)[0] = reinterpret_cast<const float4 *>(&br)[0];
        }
    }
    {
        // visible count: one atomic per workgroup; the workgroup's extras: one store (K3 picks its path by it)
        __shared__ int s_vis, s_ext, s_n3;
        if (tid == 0) { s_vis = 0; s_ext = 0; s_n3 = 0; }
        // rows of the backward's per-instance buffer: exclusive scan of the instance counts inside the workgroup; the
        // workgroup's base comes from ONE returning atomic on a global cursor (its order among workgroups is irrelevant: a
        // Gaussian's rows only have to be contiguous and its own)
        int incl = my_tiles;
        if (!PAIR) {
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int y = __shfl_up(incl, d, 64);
                if ((tid & 63) >= d) incl += y;
            }
            if ((tid & 63) == 63) s_wt[tid >> 6] = incl;
        }
        __syncthreads();
        const unsigned long long vm = __ballot(radius > 0);
        if ((tid & 63) == 0 && vm != 0ull) atomicAdd(&s_vis, __popcll(vm));
        int ne = n_extra;
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) ne += __shfl_xor(ne, m, 64);
        if ((tid & 63) == 0 && ne != 0) atomicAdd(&s_ext, ne);
        if (tight) {
            int t3 = n3;
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) t3 += __shfl_xor(t3, m, 64);
            if ((tid & 63) == 0 && t3 != 0) atomicAdd(&s_n3, t3);
        }
        __syncthreads();
        if (tid == 0 && s_vis != 0) atomicAdd(&counters->num_visible, s_vis);
        if (tid == 0 && tight && s_n3 != 0) atomicAdd(&counters->reserved[2], s_n3);
        if (tid == 0) wg_extras[blockIdx.x] = s_ext;
        if (!PAIR) {
            int before = 0;
            const int my_wave = __builtin_amdgcn_readfirstlane(tid >> 6);
            for (int w = 0; w < my_wave; w++) before += s_wt[w];
            row_excl = before + incl - my_tiles;
            // the atomic's round trip hides under the histogram flush; s_gbase is read behind the next barrier
            if (tid == 1023) {
                const int total = before + incl;
                s_gbase = total != 0 ? atomicAdd(&counters->reserved[1], total) : 0;
            }
        }
        // extras: a workgroup with many of them (large footprints) counts them in the high half of the LDS histogram and
        // adds them to the tiles' cursors with the contiguous flush below; otherwise one fire-and-forget global atomic each
        heavy = USE_LDS && s_ext > HEAVY_EXTRAS;
    }
    {
        const int nt = n_extra > 0 ? n_extra + BIN_SLOTS : 0;
        for_each_extra(tid & 63, nt, bx0, bx1 - bx0, by0, [](int) { return 0; }, [&](int, int ty, int tx) {
            const int t = ty * st.gx + tx;
            if (heavy) atomicAdd(&hist[t], 0x10000);
            else atomicAdd(&tile_extra[t], 1);
        });
    }
    auto store_word2 = [&]() {
        if (i < P && (listed || radius > 0)) {
            GeomRec r2;
            r2.b = rec2_b; r2.bbox_x = rec2_bx; r2.bbox_y = rec2_by;
            r2.goff = PAIR ? 0 : s_gbase + row_excl;
            reinterpret_cast<float4 *>(geom + i)[2] = reinterpret_cast<const float4 *>(&r2)[2];
        }
    };
    if (!USE_LDS) {
        __syncthreads();
        store_word2();
        return;
    }
    if (heavy) __syncthreads();
    // flush: 64 consecutive tiles per wave-instruction = 256 contiguous bytes of returning atomics; a wave's
    // (up to 8) groups are issued back to back and waited for once — a memory-side atomic takes microseconds
    {
        const int lane = tid & 63, wave = tid >> 6;
        for (int g0 = wave * 64; g0 < T; g0 += 8 * 1024) {
            int v[8], base[8];
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const int t = g0 + r * 1024 + lane;
                v[r] = t < T ? hist[t] : 0;
            }
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const int t = g0 + r * 1024 + lane;
                base[r] = 0;
                const int lo = v[r] & 0xffff, hi = (int)((unsigned)v[r] >> 16);
                if (__ballot(lo != 0) != 0ull && t < T) base[r] = atomicAdd(&tile_count[t], lo);
                if (heavy && __ballot(hi != 0) != 0ull && t < T) atomicAdd(&tile_extra[t], hi);      // no return value needed
            }
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const int t = g0 + r * 1024 + lane;
                if (t < T) hist[t] = base[r];
            }
        }
    }
    __syncthreads();
    store_word2();
    if (i < P && listed) {
        int j = 0;
        for (int ty = by0; ty < by1 && j < BIN_SLOTS; ty++)
            for (int tx = bx0; tx < bx1 && j < BIN_SLOTS; tx++, j++) {
                const int base = hist[ty * st.gx + tx];
                if (j == 0) slot[0] += base;
                if (j == 1) slot[1] += base;
                if (j == 2) slot[2] += base;
                if (j == 3) slot[3] += base;
            }
        reinterpret_cast<int4 *>(bins + i)[1] = make_int4(slot[0], slot[1], slot[2], slot[3]);
    }
    // K2 fused: the workgroup that finishes last runs the tile scan (release / acquire through the ticket counter),
    // which saves a launch and the dependent-kernel gap; hist (>= SCAN_CHUNK ints) is free by now
    {
        // tile_count / tile_extra are only ever modified by device-scope atomics (performed at the memory side, never
        // dirty in an XCD's L2), so "release" = wait until this lane's atomics have been acknowledged, and the scan
        // reads them with agent-scope loads.  (A full __threadfence() here costs an L2 write-back per workgroup: 5x.)
        __shared__ int s_last;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) s_last = (atomicAdd(&counters->reserved[0], 1) == (int)gridDim.x - 1) ? 1 : 0;
        __syncthreads();
        if (!s_last) return;
        scan_tiles_body(T, tile_count, tile_extra, tile_offsets, big_list, counters, max_instances, hist, tight ? 1 : 0);
    }
}

// ------------------------------------------------------------------------------------------------- K3
template <bool PAIR>
__global__ void __launch_bounds__(256) k_scatter(int P, int gx, const BinRec *__restrict__ bins,
                                                 const int32_t *__restrict__ tile_offsets,
                                                 const int32_t *__restrict__ tile_count,
                                                 int32_t *__restrict__ tile_extra, uint64_t *__restrict__ keys,
                                                 const gsvc_raster_counters *__restrict__ counters)
{
    // the overflow flag (set by the scan) is only needed before the first store: its load travels with the record loads
    const int overflow = counters->overflow;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const float4 b0 = reinterpret_cast<const float4 *>(bins + i)[0];
    const uint32_t rx = __float_as_uint(b0.y), ry = __float_as_uint(b0.z), rb = __float_as_uint(b0.w);
    if (((PAIR ? (rx | rb) : rx) | ry) == 0u || ry == 0u) return;
    const int4 sl = reinterpret_cast<const int4 *>(bins + i)[1];
    const int fx0 = rx & 0xffff, fx1 = rx >> 16, y0 = ry & 0xffff, y1 = ry >> 16;
    int x0 = fx0, x1 = fx1;
    const int mx0 = rb & 0xffff, mx1 = rb >> 16;
    if (PAIR) {   // union of this view's x range and the mirrored opposite view's (either may be empty)
        if (rx == 0u) { x0 = mx0; x1 = mx1; }
        else if (rb != 0u) { x0 = min(fx0, mx0); x1 = max(fx1, mx1); }
    }
    // pair mode: low word = id << 2 | (in this view's rectangle) | (in the opposite view's rectangle) << 1
    const uint64_t key_hi = (uint64_t)order_bits(b0.x) << 32;
    auto make_key = [&](int tx) -> uint64_t {
        if (!PAIR) return key_hi | (uint32_t)i;
        const uint32_t fl = ((rx != 0u && tx >= fx0 && tx < fx1) ? 1u : 0u) | ((rb != 0u && tx >= mx0 && tx < mx1) ? 2u : 0u);
        return key_hi | (((uint32_t)i << 2) | fl);
    };
    const int w = x1 - x0, ntiles = w * (y1 - y0);
    // first BIN_SLOTS tiles: the four segment offsets are fetched together (independent loads), then stored
    int tl[BIN_SLOTS], off[BIN_SLOTS];
    const int slots[BIN_SLOTS] = {sl.x, sl.y, sl.z, sl.w};
#pragma unroll
    for (int j = 0; j < BIN_SLOTS; j++) {
        const int jy = j / w, jx = j - jy * w;     // row-major walk of the rectangle, as in K1
        tl[j] = (y0 + jy) * gx + x0 + jx;
        off[j] = j < ntiles ? tile_offsets[tl[j]] : 0;
    }
    if (overflow) return;
#pragma unroll
    for (int j = 0; j < BIN_SLOTS; j++)
        if (j < ntiles) keys[off[j] + slots[j]] = make_key(tl[j] % gx);
    for (int j = BIN_SLOTS; j < ntiles; j++) {
        const int jy = j / w, jx = j - jy * w;
        const int t = (y0 + jy) * gx + x0 + jx;
        keys[tile_offsets[t] + tile_count[t] + atomicAdd(&tile_extra[t], 1)] = make_key(x0 + jx);
    }
}

// K3 for tile grids that fit the LDS histogram: same 1024-Gaussian partition as K1.  A workgroup whose Gaussians own
// many instances beyond the slots (large footprints: a fitting step at 4K has ~60 tiles per visible Gaussian) places
// them the way K1 counts — tile histogram in LDS, ONE contiguous returning atomic wave-instruction per 64 tiles to
// reserve the workgroup's range behind every tile's slotted instances, then LDS atomics hand out the positions inside
// the range — instead of one scattered returning global atomic per instance (measured: 5.6 ms -> K3 of a 4K fitting
// render; per-instance atomics on a tile's cursor serialise).  Workgroups with few extras keep the per-instance path.
template <bool PAIR>
__global__ void __launch_bounds__(1024) k_scatter_lds(int P, int gx, int T, const BinRec *__restrict__ bins,
                                                      const int32_t *__restrict__ tile_offsets,
                                                      const int32_t *__restrict__ tile_count,
                                                      int32_t *__restrict__ tile_extra, const int32_t *__restrict__ wg_extras,
                                                      uint64_t *__restrict__ keys,
                                                      const gsvc_raster_counters *__restrict__ counters)
{
    extern __shared__ int hist[];
    if (counters->overflow) return;                      // uniform: the whole workgroup leaves
    const int tid = threadIdx.x;
    const bool heavy = wg_extras[blockIdx.x] > HEAVY_EXTRAS;
    if (heavy) {
        for (int t = tid; t < T; t += 1024) hist[t] = 0;
        __syncthreads();
    }
    const int i = blockIdx.x * 1024 + tid;
    bool live = false;
    int x0 = 0, x1 = 0, y0 = 0, y1 = 0, fx0 = 0, fx1 = 0, mx0 = 0, mx1 = 0;
    uint32_t rx = 0, rb = 0;
    uint64_t key_hi = 0;
    int w = 1, ntiles = 0;
    if (i < P) {
        const float4 b0 = reinterpret_cast<const float4 *>(bins + i)[0];
        rx = __float_as_uint(b0.y); rb = __float_as_uint(b0.w);
        const uint32_t ry = __float_as_uint(b0.z);
        live = !((((PAIR ? (rx | rb) : rx) | ry) == 0u) || ry == 0u);
        if (live) {
            fx0 = rx & 0xffff; fx1 = rx >> 16; y0 = ry & 0xffff; y1 = ry >> 16;
            x0 = fx0; x1 = fx1;
            mx0 = rb & 0xffff; mx1 = rb >> 16;
            if (PAIR) {
                if (rx == 0u) { x0 = mx0; x1 = mx1; }
                else if (rb != 0u) { x0 = min(fx0, mx0); x1 = max(fx1, mx1); }
            }
            key_hi = (uint64_t)order_bits(b0.x) << 32;
            w = x1 - x0; ntiles = w * (y1 - y0);
        }
    }
    auto make_key = [&](int tx) -> uint64_t {
        if (!PAIR) return key_hi | (uint32_t)i;
        const uint32_t fl = ((rx != 0u && tx >= fx0 && tx < fx1) ? 1u : 0u) | ((rb != 0u && tx >= mx0 && tx < mx1) ? 2u : 0u);
        return key_hi | (((uint32_t)i << 2) | fl);
    };
    if (live) {
        const int4 sl = reinterpret_cast<const int4 *>(bins + i)[1];
        const int slots[BIN_SLOTS] = {sl.x, sl.y, sl.z, sl.w};
        int tl[BIN_SLOTS], off[BIN_SLOTS];
#pragma unroll
        for (int j = 0; j < BIN_SLOTS; j++) {
            const int jy = j / w, jx = j - jy * w;
            tl[j] = (y0 + jy) * gx + x0 + jx;
            off[j] = j < ntiles ? tile_offsets[tl[j]] : 0;
        }
#pragma unroll
        for (int j = 0; j < BIN_SLOTS; j++)
            if (j < ntiles) keys[off[j] + slots[j]] = make_key(tl[j] % gx);
    }
    // what a tile's key needs from the Gaussian that owns the rectangle (fetched across the wave for shared rectangles)
    struct Owner { uint32_t khi, i, rx, rb; int fx0, fx1, mx0, mx1; };
    const int lane = tid & 63;
    auto owner = [&](int src) -> Owner {
        Owner o;
        o.khi = __shfl((uint32_t)(key_hi >> 32), src, 64); o.i = __shfl((uint32_t)i, src, 64);
        o.rx = o.rb = 0u; o.fx0 = o.fx1 = o.mx0 = o.mx1 = 0;
        if (PAIR) {
            o.rx = __shfl(rx, src, 64); o.rb = __shfl(rb, src, 64);
            o.fx0 = __shfl(fx0, src, 64); o.fx1 = __shfl(fx1, src, 64); o.mx0 = __shfl(mx0, src, 64); o.mx1 = __shfl(mx1, src, 64);
        }
        return o;
    };
    auto key_of = [&](const Owner &o, int tx) -> uint64_t {
        if (!PAIR) return ((uint64_t)o.khi << 32) | o.i;
        const uint32_t fl = ((o.rx != 0u && tx >= o.fx0 && tx < o.fx1) ? 1u : 0u) | ((o.rb != 0u && tx >= o.mx0 && tx < o.mx1) ? 2u : 0u);
        return ((uint64_t)o.khi << 32) | ((o.i << 2) | fl);
    };
    const int nt_x = live ? ntiles : 0;
    if (!heavy) {
        for_each_extra(lane, nt_x, x0, w, y0, owner, [&](const Owner &od, int ty, int tx) {
            const int t = ty * gx + tx;
            keys[tile_offsets[t] + tile_count[t] + atomicAdd(&tile_extra[t], 1)] = key_of(od, tx);
        });
        return;
    }
    // heavy: (1) count this workgroup's extras per tile
    for_each_extra(lane, nt_x, x0, w, y0, [](int) { return 0; }, [&](int, int ty, int tx) { atomicAdd(&hist[ty * gx + tx], 1); });
    __syncthreads();
    // (2) reserve the workgroup's range on every tile's cursor: contiguous returning atomics, as in K1's flush
    {
        const int lane = tid & 63, wave = tid >> 6;
        for (int g0 = wave * 64; g0 < T; g0 += 8 * 1024) {
            int v[8], base[8];
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const int t = g0 + r * 1024 + lane;
                v[r] = t < T ? hist[t] : 0;
            }
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const int t = g0 + r * 1024 + lane;
                base[r] = 0;
                if (__ballot(v[r] != 0) != 0ull && t < T) base[r] = atomicAdd(&tile_extra[t], v[r]);
            }
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const int t = g0 + r * 1024 + lane;
                if (t < T) hist[t] = base[r];
            }
        }
    }
    __syncthreads();
    // (3) positions inside the range from LDS atomics
    for_each_extra(lane, nt_x, x0, w, y0, owner, [&](const Owner &od, int ty, int tx) {
        const int t = ty * gx + tx;
        keys[tile_offsets[t] + tile_count[t] + atomicAdd(&hist[t], 1)] = key_of(od, tx);
    });
}

// ------------------------------------------------------------------------------------------------- K4
// Ascending-only ("flip") bitonic network: every compare-exchange moves the smaller key to the lower
// index, so positions >= n (virtual +inf padding) never move and n need not be a power of two.
__device__ __forceinline__ void cmpswap(uint64_t *a, int i, int j, int n)
{
    if (j < n) {
        const uint64_t x = a[i], y = a[j];
        if (y < x) { a[i] = y; a[j] = x; }
    }
}

template <bool WAVE_ONLY>
__device__ __forceinline__ void sort_sync()
{
    if (WAVE_ONLY) {   // a single wave owns the array: LDS/global ops of one wave complete in order
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
        __syncthreads();
    }
}

template <int NT>
__device__ __forceinline__ void bitonic_sort(uint64_t *a, int n, int tid)
{
    int N = 2;
    while (N < n) N <<= 1;
    for (int k = 2; k <= N; k <<= 1) {
        const int hk = k >> 1;
        for (int t = tid; t < (N >> 1); t += NT) {
            const int blk = t / hk, off = t - blk * hk;
            const int i = blk * k + off, j = blk * k + k - 1 - off;
            if (i < n) cmpswap(a, i, j, n);
        }
        sort_sync<NT == 64>();
        for (int jj = k >> 2; jj >= 1; jj >>= 1) {
            for (int t = tid; t < (N >> 1); t += NT) {
                const int i = (t / jj) * 2 * jj + (t % jj);
                if (i < n) cmpswap(a, i, i + jj, n);
            }
            sort_sync<NT == 64>();
        }
    }
}

// one wave per tile, segments of <= SORT_WAVE_MAX entries
// every sorted entry gets its Gaussian id (point_list) and that Gaussian's alpha bounding box (inst_bbox), so the
// blend kernels test 64 entries per wave-instruction without touching the Gaussian records
// Row of the backward's partial-sum buffer that belongs to the instance of Gaussian (words 2, 3 of its record) in tile
// (tx, ty): goff + row-major index of the tile inside the Gaussian's rectangle.
__device__ __forceinline__ int32_t row_of_instance(const float4 &f2, const float4 &f3, int tx, int ty)
{
    const uint32_t rx = __float_as_uint(f3.x), ry = __float_as_uint(f3.y);
    const int x0 = rx & 0xffff, w = (int)(rx >> 16) - x0, y0 = ry & 0xffff;
    return __float_as_int(f2.w) + (ty - y0) * w + (tx - x0);
}

__device__ __forceinline__ void emit_entry(int pos, uint64_t key, const GeomRec *__restrict__ geom,
                                           int32_t *__restrict__ point_list, uint2 *__restrict__ inst_bbox,
                                           int32_t *__restrict__ gslot, int tx, int ty, int id_shift)
{
    const uint32_t low = (uint32_t)key, id = low >> id_shift;   // pair mode keeps the two view flags in the low bits
    point_list[pos] = (int32_t)low;
    const float4 f2 = reinterpret_cast<const float4 *>(geom + id)[2];
    inst_bbox[pos] = make_uint2(__float_as_uint(f2.y), __float_as_uint(f2.z));
    if (gslot) gslot[pos] = row_of_instance(f2, reinterpret_cast<const float4 *>(geom + id)[3], tx, ty);
}

// Bucket sort of one tile's keys by one wave, 64 < n <= 64 KPL (round 3; replaces the LDS bitonic network for 257..1024 entries and
// the all-pairs ranking above 128): the keys' DEPTH word is mapped monotonically onto 256 buckets between the tile's smallest and
// largest depth (float scale: conversion, multiplication by a positive constant and truncation are all monotone), a histogram
// (LDS integer atomics: their order does not matter) and its scan give every bucket a range, the keys' indices are dropped into
// their bucket in whatever order the atomics hand out, and every key then ranks itself inside its bucket with exact 64-bit
// compares — so the result is the total order of the keys (depth, then Gaussian index) whatever the placement was.  ~n / 256
// entries per bucket: O(n) instead of O(n^2) or O(n log^2 n).  Returns false (nothing written) when some bucket holds more than
// BUCKET_MAX entries (many equal or tightly clustered depths): the caller falls back to the exact networks.
constexpr int BUCKET_MAX = 48;
template <int KPL>
__device__ __forceinline__ bool bucket_rank(const uint64_t *__restrict__ s, int n, int lane, uint32_t *scr /* 2 KiB */, int (&rank)[KPL])
{
    // the depth VALUE (the key's high word is order_bits(depth): a monotone but exponent-shaped image of it — bucketed by the bit
    // pattern, depths on both sides of zero pile up in a few buckets at the two ends)
    auto depth_of = [](uint64_t key) -> float {
        const uint32_t u = (uint32_t)(key >> 32);
        return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
    };
    float dmin = 3.0e38f, dmax = -3.0e38f;
#pragma unroll
    for (int q = 0; q < KPL; q++) {
        const int i = lane + 64 * q;
        if (i < n) {
            const float d = depth_of(s[i]);
            dmin = fminf(dmin, d); dmax = fmaxf(dmax, d);
        }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        dmin = fminf(dmin, __shfl_xor(dmin, m, 64));
        dmax = fmaxf(dmax, __shfl_xor(dmax, m, 64));
    }
    const float span = dmax - dmin;
    const float scale = span > 0.f ? 255.0f / span : 0.f;      // (NaN / inf depths never reach a tile list)
    uint32_t *cnt = scr;                                        // [256] counts; the same bytes hold the index list afterwards
    uint16_t *bidx = reinterpret_cast<uint16_t *>(scr);         // [1024] key indices grouped by bucket
#pragma unroll
    for (int q = 0; q < 4; q++) cnt[lane + 64 * q] = 0u;
    sort_sync<true>();
    uint32_t bs[KPL];          // bucket << 16 | slot inside the bucket
#pragma unroll
    for (int q = 0; q < KPL; q++) {
        const int i = lane + 64 * q;
        bs[q] = 0;
        if (i < n) {
            const uint32_t b = (uint32_t)fminf(255.0f, fmaxf(0.0f, (depth_of(s[i]) - dmin) * scale));      // monotone in the depth
            bs[q] = (b << 16) | atomicAdd(&cnt[b], 1u);
        }
    }
    sort_sync<true>();
    // exclusive scan of the 256 counts (lane l: buckets 4 l .. 4 l + 3), largest bucket
    uint32_t c[4];
#pragma unroll
    for (int q = 0; q < 4; q++) c[q] = cnt[4 * lane + q];
    uint32_t big = max(max(c[0], c[1]), max(c[2], c[3]));
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) big = max(big, (uint32_t)__shfl_xor((int)big, m, 64));
    if (big > (uint32_t)BUCKET_MAX) return false;
    const uint32_t tot = (c[0] + c[1]) + (c[2] + c[3]);
    uint32_t inc = tot;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)inc, d, 64);
        if (lane >= d) inc += o;
    }
    // the buckets' (start: 10 bits, count: 6 bits) stay in the registers of the lane that scanned them, two per register;
    // bucket b's pair is fetched from lane b / 4 (ds_bpermute: no memory)
    uint32_t run = inc - tot, t16[4];
#pragma unroll
    for (int q = 0; q < 4; q++) { t16[q] = run | (c[q] << 10); run += c[q]; }
    const uint32_t p01 = t16[0] | (t16[1] << 16), p23 = t16[2] | (t16[3] << 16);
    auto bucket_of = [&](uint32_t b, int &start, int &count) {
        const int src = (int)(b >> 2) << 2;                      // byte address of the source lane
        const uint32_t a = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)p01), z = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)p23);
        const uint32_t v = ((b & 2u) ? z : a) >> ((b & 1u) * 16u);
        start = (int)(v & 0x3ffu);
        count = (int)((v >> 10) & 0x3fu);
    };
    sort_sync<true>();          // every count has been read: the bytes become the index list
#pragma unroll
    for (int q = 0; q < KPL; q++) {
        const int i = lane + 64 * q;
        int start, count;
        bucket_of(bs[q] >> 16, start, count);                    // (all lanes take part in the exchange)
        if (i < n) bidx[start + (int)(bs[q] & 0xffffu)] = (uint16_t)i;
    }
    sort_sync<true>();
#pragma unroll
    for (int q = 0; q < KPL; q++) {
        const int i = lane + 64 * q;
        int start, count;
        bucket_of(bs[q] >> 16, start, count);
        rank[q] = 0;
        if (i < n) {
            const uint64_t ki = s[i];
            int r = start;
            for (int t = 0; t < count; t++) r += s[bidx[start + t]] < ki;
            rank[q] = r;
        }
    }
    return true;
}

// One launch sorts every tile: 256-lane workgroups, each wave takes one tile at a time (tile = 4*block + wave):
//   n <= 256   rank sort: every key is compared with every other from LDS (keys are unique: rank = position);
//              the bbox gather of the lane's own entries is issued BEFORE the ranking so its latency hides under it
//   n <= 1024  LDS bitonic by the wave
// Longer lists (listed by K2, normally none) are then handled by whole workgroups: LDS bitonic up to 4096 entries
// (the four waves' strips together), in place in global memory beyond.
__global__ void __launch_bounds__(256) k_sort_tiles(int T, const int32_t *__restrict__ tile_offsets,
                                                    const int32_t *__restrict__ big_list, uint64_t *__restrict__ keys,
                                                    const GeomRec *__restrict__ geom, int32_t *__restrict__ point_list,
                                                    uint2 *__restrict__ inst_bbox, int32_t *__restrict__ gslot, int gx,
                                                    const gsvc_raster_counters *__restrict__ counters, int id_shift)
{
    __shared__ uint64_t s_all[4 * SORT_WAVE_MAX];   // 32 KiB: one 8-KiB strip per wave
    // 2 KiB of scratch per wave.  Rank sort: the keys' depth words [256] + who claimed rank r [256 x 16 bit]; bucket sort: the
    // bucket counts [256], then the key indices grouped by bucket [1024 x 16 bit]
    __shared__ __attribute__((aligned(16))) uint32_t s_scr[4][512];
    const int overflow = counters->overflow;   // loaded together with the segment bounds (one scalar round trip, not two)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    uint64_t *s = s_all + wave * SORT_WAVE_MAX;
    const int t = blockIdx.x * 4 + wave;
    int beg = 0, n = 0;
    if (t < T) {
        beg = tile_offsets[t];
        n = tile_offsets[t + 1] - beg;
    }
    if (overflow) n = 0;
    const int ty = t / gx, tx = t - ty * gx;
    bool done = false;
    if (n > SORT_RANK_MAX && n <= SORT_WAVE_MAX && !(SORT_DEBUG_NO_BUCKET)) {
        for (int i = lane; i < n; i += 64) s[i] = keys[beg + i];
        sort_sync<true>();
        auto emit = [&](int i, int r) { emit_entry(beg + r, s[i], geom, point_list, inst_bbox, gslot, tx, ty, id_shift); };
        if (n <= 512) {
            int rk[8];
            done = bucket_rank<8>(s, n, lane, s_scr[wave], rk);
            if (done) {
#pragma unroll
                for (int q = 0; q < 8; q++) if (lane + 64 * q < n) emit(lane + 64 * q, rk[q]);
            }
        } else {
            int rk[16];
            done = bucket_rank<16>(s, n, lane, s_scr[wave], rk);
            if (done) {
#pragma unroll
                for (int q = 0; q < 16; q++) if (lane + 64 * q < n) emit(lane + 64 * q, rk[q]);
            }
        }
        sort_sync<true>();
    }
    if (done) {
    } else if (n > 0 && n <= SORT_RANK_MAX) {
        uint64_t k[4];
        uint2 bb[4];
        int32_t row[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int i = lane + 64 * q;
            k[q] = ~0ull;
            bb[q] = make_uint2(0u, 0u);
            row[q] = 0;
            if (i < n) {
                k[q] = keys[beg + i];
                s[i] = k[q];
            }
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
#ifndef GSVC_SORT_NO_GATHER      // timing experiment (tools/scratch): the sort without the per-instance record gather
            if (lane + 64 * q < n) {
                const float4 *rec = reinterpret_cast<const float4 *>(geom + ((uint32_t)k[q] >> id_shift));
                const float4 f2 = rec[2];
                bb[q] = make_uint2(__float_as_uint(f2.y), __float_as_uint(f2.z));
                if (gslot) row[q] = row_of_instance(f2, rec[3], tx, ty);
            }
#endif
        }
        // Rank = number of smaller keys.  First by the DEPTH word alone (32-bit compares, four staged depths per 16-byte LDS read:
        // a 64-bit compare costs several times a 32-bit one, and its result cannot be consumed by the next instruction); the
        // ranks are then checked for uniqueness through LDS — equal depths (the only way two entries can share a rank) are rare,
        // and a tile that has them takes the 64-bit loop, which breaks ties by the Gaussian's index exactly as the key says.
        uint32_t *sd = s_scr[wave];
        uint16_t *chk = reinterpret_cast<uint16_t *>(s_scr[wave] + SORT_RANK_MAX);
        uint32_t d4[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            d4[q] = (uint32_t)(k[q] >> 32);
            sd[lane + 64 * q] = lane + 64 * q < n ? d4[q] : 0xffffffffu;      // padding never counts as smaller
        }
        sort_sync<true>();
        int r0 = 0, r1 = 0, r2 = 0, r3 = 0;
        const int n4 = (n + 3) & ~3;
        if (n <= 64) {
            for (int j = 0; j < n4; j += 4) {
                const uint4 dj = *reinterpret_cast<const uint4 *>(sd + j);
                r0 += (dj.x < d4[0]) + (dj.y < d4[0]) + (dj.z < d4[0]) + (dj.w < d4[0]);
            }
        } else if (n <= 128) {
            for (int j = 0; j < n4; j += 4) {
                const uint4 dj = *reinterpret_cast<const uint4 *>(sd + j);
                r0 += (dj.x < d4[0]) + (dj.y < d4[0]) + (dj.z < d4[0]) + (dj.w < d4[0]);
                r1 += (dj.x < d4[1]) + (dj.y < d4[1]) + (dj.z < d4[1]) + (dj.w < d4[1]);
            }
        } else {
            for (int j = 0; j < n4; j += 4) {
                const uint4 dj = *reinterpret_cast<const uint4 *>(sd + j);
                r0 += (dj.x < d4[0]) + (dj.y < d4[0]) + (dj.z < d4[0]) + (dj.w < d4[0]);
                r1 += (dj.x < d4[1]) + (dj.y < d4[1]) + (dj.z < d4[1]) + (dj.w < d4[1]);
                r2 += (dj.x < d4[2]) + (dj.y < d4[2]) + (dj.z < d4[2]) + (dj.w < d4[2]);
                r3 += (dj.x < d4[3]) + (dj.y < d4[3]) + (dj.z < d4[3]) + (dj.w < d4[3]);
            }
        }
        {
            const int rr[4] = {r0, r1, r2, r3};
#pragma unroll
            for (int q = 0; q < 4; q++)
                if (lane + 64 * q < n) chk[rr[q]] = (uint16_t)(lane + 64 * q);
            sort_sync<true>();
            bool clash = false;
#pragma unroll
            for (int q = 0; q < 4; q++)
                if (lane + 64 * q < n) clash |= chk[rr[q]] != (uint16_t)(lane + 64 * q);
            if (__ballot(clash) != 0ull) {          // equal depths in this tile: the exact 64-bit ranks
                r0 = r1 = r2 = r3 = 0;
                for (int j = 0; j < n; j++) {
                    const uint64_t kj = s[j];
                    r0 += kj < k[0]; r1 += kj < k[1]; r2 += kj < k[2]; r3 += kj < k[3];
                }
            }
        }
#ifdef GSVC_SORT_NO_RANK         // timing experiment: no ranking (entries stay where they are)
        r0 = lane; r1 = lane + 64; r2 = lane + 128; r3 = lane + 192;
#endif
        const int r[4] = {r0, r1, r2, r3};
#pragma unroll
        for (int q = 0; q < 4; q++) {
            if (lane + 64 * q < n) {
                point_list[beg + r[q]] = (int32_t)(uint32_t)k[q];
                inst_bbox[beg + r[q]] = bb[q];
                if (gslot) gslot[beg + r[q]] = row[q];
            }
        }
    } else if (n > SORT_RANK_MAX && n <= SORT_WAVE_MAX) {
        for (int i = lane; i < n; i += 64) s[i] = keys[beg + i];
        sort_sync<true>();
        bitonic_sort<64>(s, n, lane);
        for (int i = lane; i < n; i += 64) emit_entry(beg + i, s[i], geom, point_list, inst_bbox, gslot, tx, ty, id_shift);
    }
    // long lists: whole workgroups (uniform loop bounds; usually zero iterations)
    const int nbig = overflow ? 0 : counters->num_big_tiles;
    for (int w = blockIdx.x; w < nbig; w += gridDim.x) {
        const int bt = big_list[w];
        const int bbeg = tile_offsets[bt], bn = tile_offsets[bt + 1] - bbeg;
        const int bty = bt / gx, btx = bt - bty * gx;
        __syncthreads();
        if (bn <= 4 * SORT_WAVE_MAX) {
            for (int i = tid; i < bn; i += 256) s_all[i] = keys[bbeg + i];
            __syncthreads();
            bitonic_sort<256>(s_all, bn, tid);
            for (int i = tid; i < bn; i += 256) emit_entry(bbeg + i, s_all[i], geom, point_list, inst_bbox, gslot, btx, bty, id_shift);
        } else {
            uint64_t *a = keys + bbeg;
            __threadfence_block();
            bitonic_sort<256>(a, bn, tid);  // __syncthreads() orders the workgroup's own global accesses
            for (int i = tid; i < bn; i += 256) emit_entry(bbeg + i, a[i], geom, point_list, inst_bbox, gslot, btx, bty, id_shift);
        }
    }
}

// ------------------------------------------------------------------------------------------------- K5
__device__ __forceinline__ int sext16(uint32_t v) { return (int)(int16_t)(v & 0xffffu); }

// Does the alpha bounding box (int16 pairs) touch the 8x8 quadrant at (qx0, qy0)?
__device__ __forceinline__ bool bbox_hits(uint2 bb, int qx0, int qy0)
{
    return !(sext16(bb.x) > qx0 + 7 || sext16(bb.x >> 16) < qx0 || sext16(bb.y) > qy0 + 7 || sext16(bb.y >> 16) < qy0);
}

constexpr float LOG2E = 1.44269504088896340736f;

// One workgroup per 16x16 tile, one wave per 8x8 quadrant, one pixel per lane, no barriers.
// (Two pixels per lane with packed v_pk_*_f32 math was measured 30 % SLOWER on MI355X: CDNA4 SIMDs are 32 lanes
// wide, packed f32 does not raise the FLOP rate, and the coarser 8x16 culling keeps more pairs; the file is built
// with -fno-slp-vectorize for the same reason.)
// Survivors of the quadrant test are staged with the conic pre-scaled by log2(e) (and the 1/2 folded in), so the
// per-pixel work is p = A' dx^2 + C' dy^2 + B' dx dy, G = exp2(-p): 6 mul/fma + one v_exp_f32.  The inner loop
// is branch-free: selects on SGPR masks instead of EXEC-mask branches.
//
// PAIR = true renders the two-view frame GSVC's evaluation and decoder actually output (reference utils/report_utils.py:
// 297-319, pipeline/train.py:368-375): image = (render(view) + flip_W(render(opposite view))) / 2 from ONE binning.  The
// opposite view (view_matrix_s) sees the same Gaussians mirrored in x in the reverse depth order, so its composite is a second
// pass over the SAME sorted list, walked from its end, with the opposite view's own numbers: its fp32 pixel coordinate
// u_b = (-x_view - x_min) scale - 1/2 (what preprocess_gaussian() computes under view_matrix_s, kept in GeomRec::pad by
// k_preprocess<PAIR>), its conic (A, -B, C) — bit-identical: the mirror only flips signs —, its quadrant (the mirrored one,
// lanes mirrored) and its own front-to-back recurrence with the alpha >= 1/255 and T < 1e-4 decisions.  Every pixel therefore
// carries what two separate launches give, decision for decision (round 2 composited the opposite view back to front at this
// view's coordinates without the T stop: 2.5e-3 on isolated pixels).
template <bool PAIR, bool BACK>
__device__ __forceinline__ void blend_pass(const RasterParams &st, int beg, int end, const int32_t *__restrict__ point_list,
                                           const uint2 *__restrict__ inst_bbox, const GeomRec *__restrict__ geom,
                                           float4 *w_f0, float4 *w_f1, float2 *w_f2, int lane, int qx0, int qy0, bool inside,
                                           float &T, float &C0, float &C1, float &C2, int &last)
{
    // the quadrant and the lane's pixel in THIS pass's view (BACK: mirrored; the image width is a multiple of the tile size)
    const int vqx0 = BACK ? st.W - 8 - qx0 : qx0;
    const int lx = BACK ? 7 - (lane & 7) : (lane & 7);
    const float fx = (float)(vqx0 + lx), fy = (float)(qy0 + (lane >> 3));
    // pixel position relative to the quadrant centre, and its products: the lane's side of the polynomial form
    const float xl = (float)lx - 3.5f, yl = (float)(lane >> 3) - 3.5f;
    const float xl2 = xl * xl, yl2 = yl * yl, xyl = xl * yl;
    const float qcx = (float)vqx0 + 3.5f, qcy = (float)qy0 + 3.5f;
    const int view_bit = BACK ? 2 : 1;
    unsigned long long done_m = __ballot(!inside);    // lanes that take no further entries (uniform mask, in SGPRs)
    for (int c0 = BACK ? beg + ((end - beg - 1) & ~63) : beg; BACK ? c0 >= beg : c0 < end; c0 += BACK ? -64 : 64) {
        // phase 1: 64 list entries per wave-instruction against this wave's quadrant
        const int k = c0 + lane;
        bool hit = false;
        int id = 0;
        if (k < end) {
            hit = bbox_hits(inst_bbox[k], qx0, qy0);      // the alpha box is this view's; mirrored it is the opposite view's
            id = point_list[k];
            if (PAIR) hit = hit && (id & view_bit);
        }
        if (__ballot(hit) == 0ull) continue;
        // second, exact test on the bbox survivors: does the alpha >= 1/255 ellipse reach this quadrant at all?
        float4 r0, r1;
        const float4 *rec = reinterpret_cast<const float4 *>(geom + (PAIR ? (id >> 2) : id));
        bool safe = true;
        if (hit) {
            r0 = rec[0];
            r1 = rec[1];
            if (BACK) { r0.x = rec[3].w; r0.w = -r0.w; }
            hit = ellipse_hits_quad(r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, (float)vqx0, (float)qy0);
            // conic positive definite with a margin and opacity > 0: "power > 0" (spec step 6) cannot happen for this
            // entry, in exact or in rounded arithmetic, so the composite loop need not test for it
            safe = r0.z > 0.f && r1.x > 0.f && r0.z * r1.x >= 1.002f * (r0.w * r0.w) && r1.y > 0.f;
        }
        const unsigned long long mask = __ballot(hit);
        if (mask == 0ull) continue;
        const bool generic = __ballot(hit && !safe) != 0ull;    // NaN / indefinite conics: the literal loop
        if (hit) {
            const int pos = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
            const float Ap = (0.5f * LOG2E) * r0.z, Bp = LOG2E * r0.w, Cp = (0.5f * LOG2E) * r1.x;
            const float tagf = __int_as_float(PAIR ? 1 : (k - beg + 1));      // 1-based list position (single view: kept for the backward)
            if (generic) {
                w_f0[pos] = make_float4(r0.x, r0.y, Ap, Bp);
                w_f1[pos] = make_float4(Cp, r1.y, r1.z, r1.w);
            } else {
                // -log2(alpha) of pixel (qcx + x, qcy + y) as a polynomial in the lane's (x, y), |x|, |y| <= 3.5:
                //   A'(U-x)^2 + C'(V-y)^2 + B'(U-x)(V-y) - log2(o),  U = u - qcx, V = v - qcy;  stored negated
                const float U = r0.x - qcx, V = r0.y - qcy;
                const float n0 = __builtin_amdgcn_logf(r1.y) - (Ap * U * U + Cp * V * V + Bp * U * V);
                w_f0[pos] = make_float4(n0, 2.0f * Ap * U + Bp * V, 2.0f * Cp * V + Bp * U, -Ap);
                w_f1[pos] = make_float4(-Cp, -Bp, r1.z, r1.w);
            }
            w_f2[pos] = make_float2(rec[2].x, tagf);
        }
        const int cnt = __popcll(mask);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // phase 2: composite the survivors front to back (BACK: from the chunk's last survivor to its first).  Lane predicates
        // live as uniform 64-bit masks (v_cmp results in SGPRs, combined by scalar instructions); 17 vector instructions per entry.
        if (!generic) {
            // alpha of entry j at this lane's pixel (independent of the pixel state: four of them are evaluated
            // together so that their dependent FMA -> exp chains interleave)
            auto alpha_of = [&](int j) -> float {
                const float4 a = w_f0[j];
                const float2 b = *reinterpret_cast<const float2 *>(&w_f1[j]);
                float t = fmaf(b.y, xyl, a.x);
                t = fmaf(a.y, xl, t);
                t = fmaf(a.z, yl, t);
                t = fmaf(a.w, xl2, t);
                t = fmaf(b.x, yl2, t);
                return fminf(ALPHA_MAX, __builtin_amdgcn_exp2f(t));
            };
            auto apply = [&](int j, float alpha) {
                const float2 rg = *(reinterpret_cast<const float2 *>(&w_f1[j]) + 1);
                const float2 c = w_f2[j];
                const float test_T = T - alpha * T;
                const unsigned long long small_m = __ballot(alpha < ALPHA_MIN), lt_m = __ballot(test_T < T_MIN);
                const int tagv = __float_as_int(c.y);
                const unsigned long long keep_m = ~done_m & ~small_m;
                const unsigned long long acc_m = keep_m & ~lt_m;
                done_m |= keep_m & lt_m;
                const bool acc = __builtin_amdgcn_inverse_ballot_w64(acc_m);
                const float w = acc ? alpha * T : 0.0f;
                C0 += rg.x * w; C1 += rg.y * w; C2 += c.x * w;
                T = acc ? test_T : T;
                last = acc ? tagv : last;
            };
            // unrolled by hand: the mask intrinsics are convergent, which rules out "#pragma unroll 4"
            if (!BACK) {
                int j = 0;
                for (; j + 4 <= cnt; j += 4) {
                    const float a0 = alpha_of(j), a1 = alpha_of(j + 1), a2 = alpha_of(j + 2), a3 = alpha_of(j + 3);
                    apply(j, a0); apply(j + 1, a1); apply(j + 2, a2); apply(j + 3, a3);
                }
                for (; j < cnt; j++) apply(j, alpha_of(j));
            } else {
                int j = cnt - 1;
                for (; j >= 3; j -= 4) {
                    const float a0 = alpha_of(j), a1 = alpha_of(j - 1), a2 = alpha_of(j - 2), a3 = alpha_of(j - 3);
                    apply(j, a0); apply(j - 1, a1); apply(j - 2, a2); apply(j - 3, a3);
                }
                for (; j >= 0; j--) apply(j, alpha_of(j));
            }
        } else {
            bool done = __builtin_amdgcn_inverse_ballot_w64(done_m);
            for (int jj = 0; jj < cnt; jj++) {
                const int j = BACK ? cnt - 1 - jj : jj;
                const float4 a = w_f0[j];
                const float4 b = w_f1[j];
                const float2 c = w_f2[j];
                const float dx = a.x - fx, dy = a.y - fy;
                const float p = a.z * dx * dx + b.x * dy * dy + a.w * dx * dy;   // = -power * log2(e)
                const float alpha = fminf(ALPHA_MAX, b.y * __builtin_amdgcn_exp2f(-p));
                const float test_T = T - alpha * T;
                const int tagv = __float_as_int(c.y);
                const bool contrib = !(p < 0.0f) && !(alpha < ALPHA_MIN);
                const bool keep = !done && contrib;
                const bool stop = keep && (test_T < T_MIN);
                const bool acc = keep && !stop;
                done |= stop;
                const float w = acc ? alpha * T : 0.0f;
                C0 += b.z * w; C1 += b.w * w; C2 += c.x * w;
                T = acc ? test_T : T;
                last = acc ? tagv : last;
            }
            done_m = __ballot(done);
        }
        __builtin_amdgcn_wave_barrier();
        if (done_m == ~0ull) break;
    }
}

template <bool PAIR>
__global__ void __launch_bounds__(256) k_blend(RasterParams st, const int32_t *__restrict__ tile_offsets,
                                               const int32_t *__restrict__ point_list,
                                               const uint2 *__restrict__ inst_bbox, const GeomRec *__restrict__ geom,
                                               float *__restrict__ image, float *__restrict__ final_T,
                                               int32_t *__restrict__ n_contrib,
                                               const gsvc_raster_counters *__restrict__ counters)
{
    // per-wave staging of the entries that survive the quadrant test (no cross-wave sharing, no barriers)
    __shared__ float4 s_f0[4][64];  // u v A' B'
    __shared__ float4 s_f1[4][64];  // C' opacity r g
    __shared__ float2 s_f2[4][64];  // b, 1-based list position (as int bits)
    if (counters->overflow) return;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int qx0 = blockIdx.x * TILE + 8 * (wave & 1), qy0 = blockIdx.y * TILE + 8 * (wave >> 1);
    const int px = qx0 + (lane & 7), py = qy0 + (lane >> 3);
    const bool inside = px < st.W && py < st.H;
    const int tile = blockIdx.y * st.gx + blockIdx.x;
    const int beg = tile_offsets[tile], end = tile_offsets[tile + 1];

    float T = 1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f;
    int last = 0;
    blend_pass<PAIR, false>(st, beg, end, point_list, inst_bbox, geom, s_f0[wave], s_f1[wave], s_f2[wave], lane, qx0, qy0, inside,
                            T, C0, C1, C2, last);
    if (!PAIR) {
        // per-pixel state for the backward, TILE-MAJOR (tile, quadrant, lane): one coalesced 256-B store per wave here and
        // one coalesced load per quadrant there; pixels outside the image hold T = 1, last = 0 (nothing contributes)
        const int sidx = (tile * 4 + wave) * 64 + lane;
        final_T[sidx] = T;
        n_contrib[sidx] = last;
    }
    float Tb = 1.0f, B0 = 0.f, B1 = 0.f, B2 = 0.f;
    if (PAIR) {
        int last_b = 0;
        blend_pass<PAIR, true>(st, beg, end, point_list, inst_bbox, geom, s_f0[wave], s_f1[wave], s_f2[wave], lane, qx0, qy0, inside,
                               Tb, B0, B1, B2, last_b);
    }
    if (inside) {
        const int HW = st.H * st.W, pix = py * st.W + px;
        if (PAIR) {
            image[pix] = 0.5f * ((C0 + T * st.bg0) + (B0 + Tb * st.bg0));
            image[HW + pix] = 0.5f * ((C1 + T * st.bg1) + (B1 + Tb * st.bg1));
            image[2 * HW + pix] = 0.5f * ((C2 + T * st.bg2) + (B2 + Tb * st.bg2));
        } else {
            image[pix] = C0 + T * st.bg0;
            image[HW + pix] = C1 + T * st.bg1;
            image[2 * HW + pix] = C2 + T * st.bg2;
        }
    }
}

static int check_settings(const gsvc_raster_settings *s, int64_t P)
{
    GSVC_REQUIRE(s != nullptr, "raster: settings is NULL");
    GSVC_REQUIRE(s->image_height > 0 && s->image_width > 0, "raster: image size must be positive (got %d x %d)",
                 s->image_height, s->image_width);
    GSVC_REQUIRE(P >= 0 && P < (int64_t)1 << 31, "raster: P out of range (%lld)", (long long)P);
    GSVC_REQUIRE(s->image_width <= 16384 && s->image_height <= 16384, "raster: image larger than 16384 pixels a side");
    return GSVC_OK;
}

constexpr int LDS_HIST_MAX_TILES = 36 * 1024;  // 144 KiB of the 160 KiB LDS

}  // namespace gsvc

using namespace gsvc;

extern "C" int gsvc_raster_sizes_query(const gsvc_raster_settings *settings, int64_t P, int64_t max_instances,
                                       gsvc_raster_sizes *sizes)
{
    if (int rc = check_settings(settings, P)) return rc;
    GSVC_REQUIRE(sizes != nullptr && max_instances >= 0, "raster_sizes_query: bad arguments");
    const RasterLayout L = raster_layout(*settings, P, max_instances);
    sizes->geom_bytes = L.geom_bytes;
    sizes->binning_bytes = L.binning_bytes;
    sizes->image_bytes = L.image_bytes;
    return GSVC_OK;
}

extern "C" int gsvc_raster_binning_layout(const gsvc_raster_settings *settings, int64_t P, int64_t max_instances,
                                          uint64_t *tile_offsets_off, uint64_t *point_list_off)
{
    if (int rc = check_settings(settings, P)) return rc;
    const RasterLayout L = raster_layout(*settings, P, max_instances);
    if (tile_offsets_off) *tile_offsets_off = L.off_tile_offsets;
    if (point_list_off) *point_list_off = L.off_point_list;
    return GSVC_OK;
}

extern "C" int gsvc_raster_image_layout(const gsvc_raster_settings *settings, uint64_t *final_T_off,
                                        uint64_t *n_contrib_off)
{
    if (int rc = check_settings(settings, 0)) return rc;
    const RasterLayout L = raster_layout(*settings, 0, 0);
    if (final_T_off) *final_T_off = L.off_final_T;
    if (n_contrib_off) *n_contrib_off = L.off_n_contrib;
    return GSVC_OK;
}

extern "C" int gsvc_raster_visible_filter(const gsvc_raster_settings *settings, int64_t P, const float *means3D,
                                          const float *scales, const float *rotations, int32_t *radii, void *stream)
{
    if (int rc = check_settings(settings, P)) return rc;
    if (P == 0) return GSVC_OK;
    GSVC_REQUIRE(means3D && scales && rotations && radii, "visible_filter: NULL pointer");
    const RasterParams p = make_params(*settings);
    hipStream_t s = (hipStream_t)stream;
    {
        ProfScope _prof("k_visible_filter", s);
        hipLaunchKernelGGL(k_visible_filter, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, s, p, (int)P, means3D,
                           scales, rotations, radii);
    }
    return check_launch("visible_filter");
}

extern "C" int gsvc_raster_visible_masks(const gsvc_raster_settings *const *settings, int32_t views, int64_t P, const float *means3D,
                                         const float *scales, int32_t scale_stride, int32_t scale_exp, const float *rotations,
                                         int32_t rot_normalise, uint8_t *masks, void *stream)
{
    GSVC_REQUIRE(settings && views >= 1 && views <= VIS_MAX_VIEWS && scale_stride >= 3, "visible_masks: 1..8 views, scale rows of >= 3 floats");
    VisViews v;
    for (int r = 0; r < VIS_MAX_VIEWS; r++) {
        const gsvc_raster_settings *st = settings[r < views ? r : 0];
        if (int rc = check_settings(st, P)) return rc;
        v.p[r] = make_params(*st);
    }
    if (P == 0) return GSVC_OK;
    GSVC_REQUIRE(means3D && scales && rotations && masks, "visible_masks: NULL pointer");
    hipStream_t s = (hipStream_t)stream;
    ProfScope _prof("k_visible_filter", s);
    hipLaunchKernelGGL(k_visible_masks, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, s, v, (int)views, (int)P, means3D, scales,
                       (int)scale_stride, (int)scale_exp, rotations, (int)rot_normalise, masks);
    return check_launch("visible_masks");
}

static int raster_forward_impl(const gsvc_raster_settings *settings, int64_t P, int64_t max_instances,
                               const float *means3D, const float *colors, const float *opacities, const float *scales,
                               const float *rotations, float *image, int32_t *radii, void *geom, void *binning,
                               void *image_state, void *stream, bool pair)
{
    if (int rc = check_settings(settings, P)) return rc;
    GSVC_REQUIRE(max_instances >= 0 && max_instances < (int64_t)1 << 31, "raster_forward: max_instances out of range");
    GSVC_REQUIRE(image && geom && binning && image_state, "raster_forward: NULL output/state pointer");
    if (pair && (settings->image_width % TILE != 0 || P >= ((int64_t)1 << 29))) {
        set_error("raster_forward_pair: needs image_width %% 16 == 0 (tile grids of the two views must mirror) and P < 2^29");
        return GSVC_E_UNSUPPORTED;
    }
    if (pair && (settings->flags & (GSVC_RASTER_SLAB_ONE_SIDED | GSVC_RASTER_PIXEL_CORNER))) {
        // a one-sided slab shows the two views different Gaussians, and without the half-pixel offset the mirrored view's
        // pixel centres fall one pixel beside this view's: the opposite view is then not the reverse composite of this one
        set_error("raster_forward_pair: not defined under GSVC_RASTER_SLAB_ONE_SIDED / GSVC_RASTER_PIXEL_CORNER (render the two views)");
        return GSVC_E_UNSUPPORTED;
    }
    GSVC_REQUIRE(P == 0 || (means3D && colors && opacities && scales && rotations && radii),
                 "raster_forward: NULL input pointer");
    const RasterParams p = make_params(*settings);
    const RasterLayout L = raster_layout(*settings, P, max_instances);
    hipStream_t s = (hipStream_t)stream;
    char *bin = (char *)binning;
    auto *counters = (gsvc_raster_counters *)(bin + L.off_counters);
    auto *tile_offsets = (int32_t *)(bin + L.off_tile_offsets);
    auto *tile_count = (int32_t *)(bin + L.off_tile_count);
    auto *tile_extra = (int32_t *)(bin + L.off_tile_extra);
    auto *big_list = (int32_t *)(bin + L.off_big_list);
    auto *wg_extras = (int32_t *)(bin + L.off_wg_extras);
    auto *keys = (uint64_t *)(bin + L.off_keys);
    auto *point_list = (int32_t *)(bin + L.off_point_list);
    auto *inst_bbox = (uint2 *)(bin + L.off_inst_bbox);
    auto *gslot = (int32_t *)(bin + L.off_gslot);
    auto *grec = (GeomRec *)((char *)geom + L.off_geom);
    auto *brec = (BinRec *)((char *)geom + L.off_bin);
    auto *final_T = (float *)((char *)image_state + L.off_final_T);
    auto *n_contrib = (int32_t *)((char *)image_state + L.off_n_contrib);

    // counters + tile_offsets + tile_count + tile_extra are contiguous at the head of the blob
    if (hipMemsetAsync(bin, 0, L.off_big_list, s) != hipSuccess) {
        set_error("raster_forward: hipMemsetAsync failed");
        return GSVC_E_LAUNCH;
    }
    if (P > 0) {
        const unsigned blocks = (unsigned)((P + 1023) / 1024);
        ProfScope _prof("k_preprocess", s);
        const bool use_lds = L.tiles <= LDS_HIST_MAX_TILES;
        // the fused scan stages SCAN_CHUNK tile counts in the same LDS
        const size_t lds = use_lds ? (size_t)(L.tiles > SCAN_CHUNK ? L.tiles : SCAN_CHUNK) * sizeof(int) : 0;
        auto launch = [&](auto kernel) {
            if (lds > 48 * 1024)
                (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                          LDS_HIST_MAX_TILES * 4);
            hipLaunchKernelGGL(kernel, dim3(blocks), dim3(1024), lds, s, p, (int)P, means3D, colors, opacities, scales,
                               rotations, radii, grec, brec, tile_count, tile_extra, counters, tile_offsets, big_list, wg_extras,
                               (long long)max_instances);
        };
        if (use_lds && pair) launch(&k_preprocess<true, true>);
        else if (use_lds) launch(&k_preprocess<true, false>);
        else if (pair) launch(&k_preprocess<false, true>);
        else launch(&k_preprocess<false, false>);
    }
    if (P == 0 || L.tiles > LDS_HIST_MAX_TILES) {      // otherwise the scan ran inside k_preprocess (last workgroup)
        ProfScope _prof("k_scan_tiles", s);
        hipLaunchKernelGGL(k_scan_tiles, dim3(1), dim3(1024), 0, s, L.tiles, tile_count, tile_extra, tile_offsets,
                           big_list, counters, (long long)max_instances, (int)((p.flags & GSVC_RASTER_TIGHT_BINNING) ? 1 : 0));
    }
    if (P > 0) {
        {
            ProfScope _prof("k_scatter", s);
            if (L.tiles <= LDS_HIST_MAX_TILES) {
                const unsigned blocks = (unsigned)((P + BIN_WG - 1) / BIN_WG);
                const size_t lds = (size_t)L.tiles * sizeof(int);
                auto launch = [&](auto kernel) {
                    if (lds > 48 * 1024)
                        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                  LDS_HIST_MAX_TILES * 4);
                    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(1024), lds, s, (int)P, L.gx, L.tiles, brec, tile_offsets, tile_count,
                                       tile_extra, wg_extras, keys, counters);
                };
                if (pair) launch(&k_scatter_lds<true>);
                else launch(&k_scatter_lds<false>);
            } else if (pair)
                hipLaunchKernelGGL(k_scatter<true>, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, s, (int)P, L.gx, brec,
                                   tile_offsets, tile_count, tile_extra, keys, counters);
            else
                hipLaunchKernelGGL(k_scatter<false>, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, s, (int)P, L.gx, brec,
                                   tile_offsets, tile_count, tile_extra, keys, counters);
        }
        {
            ProfScope _prof("k_sort_tiles", s);
            hipLaunchKernelGGL(k_sort_tiles, dim3((unsigned)((L.tiles + 3) / 4)), dim3(256), 0, s, L.tiles, tile_offsets,
                               big_list, keys, grec, point_list, inst_bbox, pair ? (int32_t *)nullptr : gslot, L.gx, counters,
                               pair ? 2 : 0);
        }
    }
    {
        ProfScope _prof(pair ? "k_blend_pair" : "k_blend", s);
        if (pair)
            hipLaunchKernelGGL(k_blend<true>, dim3(L.gx, L.gy), dim3(256), 0, s, p, tile_offsets, point_list, inst_bbox,
                               grec, image, final_T, n_contrib, counters);
        else
            hipLaunchKernelGGL(k_blend<false>, dim3(L.gx, L.gy), dim3(256), 0, s, p, tile_offsets, point_list, inst_bbox,
                               grec, image, final_T, n_contrib, counters);
    }
    return check_launch("raster_forward");
}

extern "C" int gsvc_raster_forward(const gsvc_raster_settings *settings, int64_t P, int64_t max_instances,
                                   const float *means3D, const float *colors, const float *opacities,
                                   const float *scales, const float *rotations, float *image, int32_t *radii,
                                   void *geom, void *binning, void *image_state, void *stream)
{
    return raster_forward_impl(settings, P, max_instances, means3D, colors, opacities, scales, rotations, image, radii, geom,
                               binning, image_state, stream, false);
}

extern "C" int gsvc_raster_forward_pair(const gsvc_raster_settings *settings, int64_t P, int64_t max_instances,
                                        const float *means3D, const float *colors, const float *opacities,
                                        const float *scales, const float *rotations, float *image_pair, int32_t *radii,
                                        void *geom, void *binning, void *image_state, void *stream)
{
    return raster_forward_impl(settings, P, max_instances, means3D, colors, opacities, scales, rotations, image_pair, radii,
                               geom, binning, image_state, stream, true);
}

// gsvc_amd/csrc/raster_fwd.hip — forward of the orthographic sliding-window tile rasterizer, gfx950.
//
// Replaces GaussianRasterizer.forward / .visible_filter of the external CUDA extension GSVC imports
// (call sites reference ortho_gaussian_renderer/renderer.py:90-98, preprocess.py:99-104).
//
// Pipeline (all on the caller's stream, no host sync, no allocation):
//   K1 preprocess      one lane per Gaussian: project, cull to slab/screen, conic, radius, tile rectangle;
//                      writes the 48-B GeomRec and counts instances per tile (integer atomics)
//   K2 scan_tiles      one workgroup: exclusive scan of the per-tile counts -> tile_offsets[T+1], counters
//   K3 scatter         one lane per Gaussian: appends (depth_bits<<32 | id) to each touched tile's segment
//   K4 sort_tiles      one WAVE per tile (<=1024 entries, LDS bitonic) / one workgroup per tile (longer):
//                      orders each segment by (depth, id) -> point_list.  No device-wide radix sort: the
//                      per-tile segments are independent, so the depth ordering is a wavefront-local
//                      problem and the tile boundaries come for free from the scan.
//   K5 blend           one 256-lane workgroup per 16x16 tile: LDS-staged batches of 256 GeomRecs,
//                      front-to-back alpha compositing, early exit when every pixel is saturated
#include "raster_common.h"

namespace gsvc {

// ---------------------------------------------------------------------------------------------- K1
template <bool FILTER_ONLY>
__global__ void __launch_bounds__(256) k_preprocess(RasterParams st, int P, const float *__restrict__ means3D,
                                                    const float *__restrict__ colors,
                                                    const float *__restrict__ opacities,
                                                    const float *__restrict__ scales,
                                                    const float *__restrict__ rotations, int32_t *__restrict__ radii,
                                                    GeomRec *__restrict__ geom, int32_t *__restrict__ tile_count,
                                                    gsvc_raster_counters *__restrict__ counters)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const float px = means3D[3 * i + 0], py = means3D[3 * i + 1], pz = means3D[3 * i + 2];
    const float s0 = scales[3 * i + 0], s1 = scales[3 * i + 1], s2 = scales[3 * i + 2];
    const float4 q = reinterpret_cast<const float4 *>(rotations)[i];
    PreOut o;
    const int radius = preprocess_gaussian(st, px, py, pz, s0, s1, s2, q.x, q.y, q.z, q.w, o);
    radii[i] = radius;
    if (FILTER_ONLY) return;
    GeomRec rec;
    if (radius > 0) {
        rec.u = o.u; rec.v = o.v; rec.A = o.A; rec.B = o.B;
        rec.C = o.C; rec.opacity = opacities[i];
        rec.r = colors[3 * i + 0]; rec.g = colors[3 * i + 1]; rec.b = colors[3 * i + 2];
        rec.depth = o.depth;
        rec.rect_x = (uint32_t)o.x0 | ((uint32_t)o.x1 << 16);
        rec.rect_y = (uint32_t)o.y0 | ((uint32_t)o.y1 << 16);
        for (int ty = o.y0; ty < o.y1; ty++)
            for (int tx = o.x0; tx < o.x1; tx++) atomicAdd(&tile_count[ty * st.gx + tx], 1);
        atomicAdd(&counters->num_visible, 1);
    } else {
        rec.u = rec.v = rec.A = rec.B = rec.C = rec.opacity = rec.r = rec.g = rec.b = rec.depth = 0.f;
        rec.rect_x = rec.rect_y = 0u;
    }
    float4 *dst = reinterpret_cast<float4 *>(geom + i);
    const float4 *src = reinterpret_cast<const float4 *>(&rec);
    dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2];
}

// ---------------------------------------------------------------------------------------------- K2
// Exclusive scan of tile_count[T] -> tile_offsets[T+1]; tile_count is zeroed afterwards (K3 reuses it as
// the per-tile fill cursor).  One workgroup of 1024 lanes walks the array in chunks of 1024.
__global__ void __launch_bounds__(1024) k_scan_tiles(int T, int32_t *__restrict__ tile_count,
                                                     int32_t *__restrict__ tile_offsets,
                                                     gsvc_raster_counters *__restrict__ counters,
                                                     long long max_instances)
{
    __shared__ int wave_sum[16];
    __shared__ int carry_s;
    __shared__ int max_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) { carry_s = 0; max_s = 0; }
    __syncthreads();
    int local_max = 0;
    for (int base = 0; base < T; base += 1024) {
        const int idx = base + tid;
        const int v = idx < T ? tile_count[idx] : 0;
        local_max = v > local_max ? v : local_max;
        int x = v;  // inclusive wave scan
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            int y = __shfl_up(x, d, 64);
            if (lane >= d) x += y;
        }
        if (lane == 63) wave_sum[wave] = x;
        __syncthreads();
        int wave_prefix = 0;
        for (int w = 0; w < wave; w++) wave_prefix += wave_sum[w];
        const int carry = carry_s;
        if (idx < T) {
            tile_offsets[idx] = carry + wave_prefix + x - v;
            tile_count[idx] = 0;
        }
        __syncthreads();
        if (tid == 1023) carry_s = carry + wave_prefix + x;
        __syncthreads();
    }
    atomicMax(&max_s, local_max);
    __syncthreads();
    if (tid == 0) {
        const int total = carry_s;
        tile_offsets[T] = total;
        counters->num_rendered = total;
        counters->overflow = ((long long)total > max_instances) ? 1 : 0;
        counters->max_tile_len = max_s;
    }
}

// ---------------------------------------------------------------------------------------------- K3
__global__ void __launch_bounds__(256) k_scatter(int P, int gx, const GeomRec *__restrict__ geom,
                                                 const int32_t *__restrict__ tile_offsets,
                                                 int32_t *__restrict__ tile_fill, uint64_t *__restrict__ keys,
                                                 const gsvc_raster_counters *__restrict__ counters)
{
    if (counters->overflow) return;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const float4 r2 = reinterpret_cast<const float4 *>(geom + i)[2];
    const uint32_t rx = __float_as_uint(r2.z), ry = __float_as_uint(r2.w);
    if ((rx | ry) == 0u) return;
    const int x0 = rx & 0xffff, x1 = rx >> 16, y0 = ry & 0xffff, y1 = ry >> 16;
    const uint64_t key = ((uint64_t)order_bits(r2.y) << 32) | (uint32_t)i;
    for (int ty = y0; ty < y1; ty++)
        for (int tx = x0; tx < x1; tx++) {
            const int t = ty * gx + tx;
            const int slot = atomicAdd(&tile_fill[t], 1);
            keys[tile_offsets[t] + slot] = key;
        }
}

// ---------------------------------------------------------------------------------------------- K4
// Ascending-only ("flip") bitonic network: every compare-exchange moves the smaller key to the lower
// index, so positions >= n (virtual +inf padding) never move and n need not be a power of two.
__device__ __forceinline__ void cmpswap(uint64_t *a, int i, int j, int n)
{
    if (j < n) {
        const uint64_t x = a[i], y = a[j];
        if (y < x) { a[i] = y; a[j] = x; }
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
        __syncthreads();
        for (int jj = k >> 2; jj >= 1; jj >>= 1) {
            for (int t = tid; t < (N >> 1); t += NT) {
                const int i = (t / jj) * 2 * jj + (t % jj);
                if (i < n) cmpswap(a, i, i + jj, n);
            }
            __syncthreads();
        }
    }
}

constexpr int SORT_WAVE_MAX = 1024;   // entries one wave sorts in LDS (8 KiB)
constexpr int SORT_WG_MAX = 8192;     // entries one workgroup sorts in LDS (64 KiB)

// one wave per tile, segments of <= SORT_WAVE_MAX entries
__global__ void __launch_bounds__(64) k_sort_tiles_wave(int T, const int32_t *__restrict__ tile_offsets,
                                                        uint64_t *__restrict__ keys, int32_t *__restrict__ point_list,
                                                        const gsvc_raster_counters *__restrict__ counters)
{
    __shared__ uint64_t s[SORT_WAVE_MAX];
    if (counters->overflow) return;
    const int t = blockIdx.x, tid = threadIdx.x;
    const int beg = tile_offsets[t], n = tile_offsets[t + 1] - beg;
    if (n <= 0 || n > SORT_WAVE_MAX) return;
    for (int i = tid; i < n; i += 64) s[i] = keys[beg + i];
    __syncthreads();
    bitonic_sort<64>(s, n, tid);
    for (int i = tid; i < n; i += 64) point_list[beg + i] = (int32_t)(uint32_t)s[i];
}

// one 256-lane workgroup per tile, segments longer than SORT_WAVE_MAX: LDS up to SORT_WG_MAX, in place in
// global memory beyond that (correct for any length; such tiles are pathological)
__global__ void __launch_bounds__(256) k_sort_tiles_wg(int T, const int32_t *__restrict__ tile_offsets,
                                                       uint64_t *__restrict__ keys, int32_t *__restrict__ point_list,
                                                       const gsvc_raster_counters *__restrict__ counters)
{
    extern __shared__ uint64_t sbig[];
    if (counters->overflow || counters->max_tile_len <= SORT_WAVE_MAX) return;
    const int t = blockIdx.x, tid = threadIdx.x;
    const int beg = tile_offsets[t], n = tile_offsets[t + 1] - beg;
    if (n <= SORT_WAVE_MAX) return;
    if (n <= SORT_WG_MAX) {
        for (int i = tid; i < n; i += 256) sbig[i] = keys[beg + i];
        __syncthreads();
        bitonic_sort<256>(sbig, n, tid);
        for (int i = tid; i < n; i += 256) point_list[beg + i] = (int32_t)(uint32_t)sbig[i];
    } else {
        uint64_t *a = keys + beg;
        __threadfence_block();
        bitonic_sort<256>(a, n, tid);  // __syncthreads() orders the workgroup's own global accesses
        for (int i = tid; i < n; i += 256) point_list[beg + i] = (int32_t)(uint32_t)a[i];
    }
}

// ---------------------------------------------------------------------------------------------- K5
__global__ void __launch_bounds__(256) k_blend(RasterParams st, const int32_t *__restrict__ tile_offsets,
                                               const int32_t *__restrict__ point_list,
                                               const GeomRec *__restrict__ geom, float *__restrict__ image,
                                               float *__restrict__ final_T, int32_t *__restrict__ n_contrib,
                                               const gsvc_raster_counters *__restrict__ counters)
{
    __shared__ float4 s0[256];  // u v A B
    __shared__ float4 s1[256];  // C opacity r g
    __shared__ float s2[256];   // b
    if (counters->overflow) return;
    const int tid = threadIdx.x;
    const int lx = tid & 15, ly = tid >> 4;
    const int tile = blockIdx.y * st.gx + blockIdx.x;
    const int px = blockIdx.x * TILE + lx, py = blockIdx.y * TILE + ly;
    const bool inside = px < st.W && py < st.H;
    const int beg = tile_offsets[tile], end = tile_offsets[tile + 1];
    const float fx = (float)px, fy = (float)py;

    float T = 1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f;
    int last = 0, contributor = 0;
    bool done = !inside;
    for (int base = beg; base < end; base += 256) {
        if (__syncthreads_count(done) == 256) break;
        const int k = base + tid;
        if (k < end) {
            const int id = point_list[k];
            const float4 *src = reinterpret_cast<const float4 *>(geom + id);
            s0[tid] = src[0];
            s1[tid] = src[1];
            s2[tid] = src[2].x;
        }
        __syncthreads();
        const int m = min(256, end - base);
        for (int j = 0; j < m && !done; j++) {
            contributor++;
            const float4 a = s0[j];
            const float4 b = s1[j];
            const float dx = a.x - fx, dy = a.y - fy;
            const float power = -0.5f * (a.z * dx * dx + b.x * dy * dy) - a.w * dx * dy;
            if (power > 0.0f) continue;
            const float alpha = fminf(ALPHA_MAX, b.y * __expf(power));
            if (alpha < ALPHA_MIN) continue;
            const float test_T = T * (1.0f - alpha);
            if (test_T < T_MIN) { done = true; continue; }
            const float w = alpha * T;
            C0 += b.z * w; C1 += b.w * w; C2 += s2[j] * w;
            T = test_T;
            last = contributor;
        }
    }
    if (inside) {
        const int HW = st.H * st.W, pix = py * st.W + px;
        final_T[pix] = T;
        n_contrib[pix] = last;
        image[pix] = C0 + T * st.bg0;
        image[HW + pix] = C1 + T * st.bg1;
        image[2 * HW + pix] = C2 + T * st.bg2;
    }
}

static int check_settings(const gsvc_raster_settings *s, int64_t P)
{
    GSVC_REQUIRE(s != nullptr, "raster: settings is NULL");
    GSVC_REQUIRE(s->image_height > 0 && s->image_width > 0, "raster: image size must be positive (got %d x %d)",
                 s->image_height, s->image_width);
    GSVC_REQUIRE(P >= 0 && P < (int64_t)1 << 31, "raster: P out of range (%lld)", (long long)P);
    GSVC_REQUIRE((s->image_width + TILE - 1) / TILE < 65536 && (s->image_height + TILE - 1) / TILE < 65536,
                 "raster: image too large for 16-bit tile coordinates");
    return GSVC_OK;
}

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
    { ProfScope _prof("k_preprocess", s); hipLaunchKernelGGL(k_preprocess<true>, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, s, p, (int)P, means3D,
                       (const float *)nullptr, (const float *)nullptr, scales, rotations, radii, (GeomRec *)nullptr,
                       (int32_t *)nullptr, (gsvc_raster_counters *)nullptr); }
    return check_launch("visible_filter");
}

extern "C" int gsvc_raster_forward(const gsvc_raster_settings *settings, int64_t P, int64_t max_instances,
                                   const float *means3D, const float *colors, const float *opacities,
                                   const float *scales, const float *rotations, float *image, int32_t *radii,
                                   void *geom, void *binning, void *image_state, void *stream)
{
    if (int rc = check_settings(settings, P)) return rc;
    GSVC_REQUIRE(max_instances >= 0 && max_instances < (int64_t)1 << 31, "raster_forward: max_instances out of range");
    GSVC_REQUIRE(image && geom && binning && image_state, "raster_forward: NULL output/state pointer");
    GSVC_REQUIRE(P == 0 || (means3D && colors && opacities && scales && rotations && radii),
                 "raster_forward: NULL input pointer");
    const RasterParams p = make_params(*settings);
    const RasterLayout L = raster_layout(*settings, P, max_instances);
    hipStream_t s = (hipStream_t)stream;
    char *bin = (char *)binning;
    auto *counters = (gsvc_raster_counters *)(bin + L.off_counters);
    auto *tile_offsets = (int32_t *)(bin + L.off_tile_offsets);
    auto *tile_fill = (int32_t *)(bin + L.off_tile_fill);
    auto *keys = (uint64_t *)(bin + L.off_keys);
    auto *point_list = (int32_t *)(bin + L.off_point_list);
    auto *final_T = (float *)((char *)image_state + L.off_final_T);
    auto *n_contrib = (int32_t *)((char *)image_state + L.off_n_contrib);

    // counters + tile_offsets + tile_fill are contiguous at the head of the blob
    if (hipMemsetAsync(bin, 0, L.off_keys, s) != hipSuccess) {
        set_error("raster_forward: hipMemsetAsync failed");
        return GSVC_E_LAUNCH;
    }
    if (P > 0) {
        { ProfScope _prof("k_preprocess", s); hipLaunchKernelGGL(k_preprocess<false>, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, s, p, (int)P, means3D,
                           colors, opacities, scales, rotations, radii, (GeomRec *)geom, tile_fill, counters); }
    }
    { ProfScope _prof("k_scan_tiles", s); hipLaunchKernelGGL(k_scan_tiles, dim3(1), dim3(1024), 0, s, L.tiles, tile_fill, tile_offsets, counters,
                       (long long)max_instances); }
    if (P > 0) {
        { ProfScope _prof("k_scatter", s); hipLaunchKernelGGL(k_scatter, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, s, (int)P, L.gx,
                           (const GeomRec *)geom, tile_offsets, tile_fill, keys, counters); }
        { ProfScope _prof("k_sort_tiles_wave", s); hipLaunchKernelGGL(k_sort_tiles_wave, dim3(L.tiles), dim3(64), 0, s, L.tiles, tile_offsets, keys, point_list,
                           counters); }
        { ProfScope _prof("k_sort_tiles_wg", s); hipLaunchKernelGGL(k_sort_tiles_wg, dim3(L.tiles), dim3(256), SORT_WG_MAX * sizeof(uint64_t), s, L.tiles,
                           tile_offsets, keys, point_list, counters); }
    }
    { ProfScope _prof("k_blend", s); hipLaunchKernelGGL(k_blend, dim3(L.gx, L.gy), dim3(256), 0, s, p, tile_offsets, point_list,
                       (const GeomRec *)geom, image, final_T, n_contrib, counters); }
    return check_launch("raster_forward");
}

// gsvc_amd/csrc/grid.hip — multi-resolution hash-grid encoder (forward, dy_dx, table and input backward), gfx950.
//
// Replaces `_gridencoder.grid_encode_forward / grid_encode_backward` (reference
// submodules/gridencoder.zip!gridencoder/src/gridencoder.cu: kernel_grid :100-660, kernel_grid_backward
// :664-853, kernel_input_backward :856-882; call sites reference utils/encodings.py:529-553,582-610).
// Semantics kept: position = x*(res-2)+0.5, corners on the border cell are dropped and the remaining
// weights renormalised (forward/table-backward), dy_dx is the un-renormalised finite difference, dense
// index while the running stride fits the level's table else the prime-XOR hash, out-of-range inputs
// give zeros.  binary_vxl / per-point min_level_id are never passed by GSVC and are not part of this ABI.
//
// Mapping to the hardware: one lane per (point, level); a corner's C features are one 4*C-byte row, read
// with 16-byte loads; the [L,N,C] output row of a lane is contiguous so a wave writes 64*4*C contiguous
// bytes.  The table backward accumulates table slices in LDS (k_grid_bwd_lds below).
#include "common.h"
#include <algorithm>
#include <atomic>

namespace gsvc {

// Where a launch reads its points and writes / reads the per-level features.  Default (the reference's layout,
// gridencoder.cu): inputs [N, D] dense, features [L, N, C].  A caller that wants the concatenated [N, sum of L C] matrix of
// several grids (Mix3d2dEncoding: scene/gaussian_model.py:81-147) points every grid at its column block of that matrix and at
// its columns of the shared [N, 3] positions — no input slices, no permute / cat copies on either pass.
struct GridIO {
    long long feat_level_stride, feat_point_stride;     // floats: feature (level l, point b, channel c) at l * ls + b * ps + c
    int in_stride, in_col[3];                           // x[d] of point b at inputs[b * in_stride + in_col[d]]
};


template <uint32_t D>
__device__ __forceinline__ uint32_t fast_hash(const uint32_t (&p)[D])
{
    constexpr uint32_t primes[3] = {1u, 2654435761u, 805459861u};
    uint32_t h = 0;
#pragma unroll
    for (uint32_t d = 0; d < D; d++) h ^= p[d] * primes[d];
    return h;
}

template <uint32_t D>
__device__ __forceinline__ uint32_t grid_row(const uint32_t (&p)[D], uint32_t hashmap_size, uint32_t resolution)
{
    uint32_t stride = 1, index = 0;
#pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        if (stride <= hashmap_size) {
            index += p[d] * stride;
            stride *= resolution;
        }
    }
    if (stride > hashmap_size) index = fast_hash<D>(p);
    // index % hashmap_size without the ~30-instruction division where it is not needed: a dense index is already below the
    // table size, a hashed level's table has 2^log2_hashmap_size rows
    if (index >= hashmap_size) index = (hashmap_size & (hashmap_size - 1)) == 0 ? index & (hashmap_size - 1) : index % hashmap_size;
    return index;
}

template <uint32_t D>
struct Cell {
    float frac[D];
    uint32_t cell[D];
    float w[1 << D];
    uint32_t row[1 << D];
    uint32_t valid;  // bit per corner
    float wn_re;
};

template <uint32_t D>
__device__ __forceinline__ void locate(const float (&x)[D], uint32_t resolution, uint32_t hashmap_size, Cell<D> &c)
{
    // the cell index and the fractional position must round exactly as in the reference (float product, then
    // + 0.5): an FMA here moves points across cell boundaries / changes weights by ~1e-5
#pragma clang fp contract(off)
#pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        const float pos = x[d] * (float)(resolution - 2) + 0.5f;
        c.cell[d] = (uint32_t)floorf(pos);
        c.frac[d] = pos - (float)c.cell[d];
    }
    float wn = 0.f;
    c.valid = 0;
#pragma unroll
    for (uint32_t idx = 0; idx < (1u << D); idx++) {
        float w = 1.f;
        uint32_t pl[D];
        bool border = false;
#pragma unroll
        for (uint32_t d = 0; d < D; d++) {
            if ((idx & (1u << d)) == 0) {
                w *= 1.f - c.frac[d];
                pl[d] = c.cell[d];
            } else {
                w *= c.frac[d];
                pl[d] = min(c.cell[d] + 1, resolution - 1);
            }
            border |= (pl[d] == 0) | (pl[d] == resolution - 1);
        }
        c.w[idx] = w;
        c.row[idx] = 0;
        if (!border) {
            c.row[idx] = grid_row<D>(pl, hashmap_size, resolution);
            c.valid |= 1u << idx;
            wn += w;
        }
    }
    if (wn == 0.f) wn = 1e-9f;
    c.wn_re = 1.0f / wn;
}

template <uint32_t C>
__device__ __forceinline__ void load_row(const float *__restrict__ p, float (&v)[C])
{
    if constexpr (C >= 4) {
#pragma unroll
        for (uint32_t k = 0; k < C / 4; k++) {
            const float4 t = reinterpret_cast<const float4 *>(p)[k];
            v[4 * k] = t.x; v[4 * k + 1] = t.y; v[4 * k + 2] = t.z; v[4 * k + 3] = t.w;
        }
    } else if constexpr (C == 2) {
        const float2 t = *reinterpret_cast<const float2 *>(p);
        v[0] = t.x; v[1] = t.y;
    } else {
        v[0] = p[0];
    }
}

template <uint32_t C>
__device__ __forceinline__ void store_row(float *__restrict__ p, const float (&v)[C])
{
    if constexpr (C >= 4) {
#pragma unroll
        for (uint32_t k = 0; k < C / 4; k++)
            reinterpret_cast<float4 *>(p)[k] = make_float4(v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]);
    } else if constexpr (C == 2) {
        *reinterpret_cast<float2 *>(p) = make_float2(v[0], v[1]);
    } else {
        p[0] = v[0];
    }
}

// One row of a BIT-PACKED binarised table: byte `row` holds the signs of the row's 8 features (bit f set: +1, clear: -1) — the
// table as the bitstream carries it (reference utils/encodings.py:375-392 STE_binary over whole tables; 1/32 of the float form)
__device__ __forceinline__ void load_row_bits(const uint8_t *__restrict__ p, uint32_t row, float (&v)[8])
{
    const uint32_t b = p[row];
#pragma unroll
    for (uint32_t f = 0; f < 8; f++) v[f] = (b >> f) & 1u ? 1.0f : -1.0f;
}

template <uint32_t D, uint32_t C, bool PACKED = false>
__device__ __forceinline__ void grid_fwd_point(const float *__restrict__ inputs, const float *__restrict__ grid,
                                               const int32_t *__restrict__ offsets, const int32_t *__restrict__ resolutions,
                                               float *__restrict__ outputs, uint32_t L, float *__restrict__ dy_dx, const GridIO &io,
                                               uint32_t b, uint32_t level)
{
    const uint8_t *bits = reinterpret_cast<const uint8_t *>(grid) + (uint32_t)offsets[level];      // PACKED: one byte per row
    grid += (size_t)(uint32_t)offsets[level] * C;
    const uint32_t hashmap_size = (uint32_t)(offsets[level + 1] - offsets[level]);
    const uint32_t resolution = (uint32_t)resolutions[level];
    float x[D];
    bool oob = false;
#pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        x[d] = inputs[(size_t)b * io.in_stride + io.in_col[d]];
        oob |= (x[d] < 0.f) | (x[d] > 1.f);
    }
    float *out = outputs + (size_t)level * io.feat_level_stride + (size_t)b * io.feat_point_stride;
    float *dd = dy_dx ? dy_dx + (size_t)b * D * L * C + (size_t)level * D * C : nullptr;
    float res[C];
#pragma unroll
    for (uint32_t ch = 0; ch < C; ch++) res[ch] = 0.f;
    if (oob) {
        store_row<C>(out, res);
        if (dd) {
#pragma unroll
            for (uint32_t d = 0; d < D; d++) store_row<C>(dd + d * C, res);
        }
        return;
    }
    Cell<D> c;
    locate<D>(x, resolution, hashmap_size, c);
    // the 2^D corner rows are read once and kept: the interpolation and all D finite differences use the same rows
    // (a corner is dropped by both exactly when one of its coordinates lies on the border)
    float g[1 << D][C];
#pragma unroll
    for (uint32_t idx = 0; idx < (1u << D); idx++) {
        if (c.valid & (1u << idx)) {
            if constexpr (PACKED) load_row_bits(bits, c.row[idx], g[idx]);
            else load_row<C>(grid + (size_t)c.row[idx] * C, g[idx]);
        } else {
#pragma unroll
            for (uint32_t ch = 0; ch < C; ch++) g[idx][ch] = 0.f;
        }
    }
#pragma unroll
    for (uint32_t idx = 0; idx < (1u << D); idx++) {
        const float w = c.w[idx] * c.wn_re;
#pragma unroll
        for (uint32_t ch = 0; ch < C; ch++) res[ch] += (c.valid & (1u << idx)) ? w * g[idx][ch] : 0.f;
    }
    store_row<C>(out, res);
    if (!dd) return;
#pragma unroll
    for (uint32_t gd = 0; gd < D; gd++) {
        float rg[C];
#pragma unroll
        for (uint32_t ch = 0; ch < C; ch++) rg[ch] = 0.f;
#pragma unroll
        for (uint32_t idx = 0; idx < (1u << (D - 1)); idx++) {
            float w = (float)(resolution - 2);
            uint32_t lo = 0;      // corner index with bit gd clear
#pragma unroll
            for (uint32_t nd = 0; nd + 1 < D; nd++) {
                const uint32_t d = (nd >= gd) ? (nd + 1) : nd;
                if ((idx & (1u << nd)) == 0) {
                    w *= 1.f - c.frac[d];
                } else {
                    w *= c.frac[d];
                    lo |= 1u << d;
                }
            }
            const uint32_t hi = lo | (1u << gd);
#pragma unroll
            for (uint32_t ch = 0; ch < C; ch++) rg[ch] += w * (g[hi][ch] - g[lo][ch]);
        }
        store_row<C>(dd + gd * C, rg);
    }
}

template <uint32_t D, uint32_t C, bool PACKED = false>
__global__ void __launch_bounds__(256) k_grid_fwd(const float *__restrict__ inputs, const float *__restrict__ grid,
                                                  const int32_t *__restrict__ offsets,
                                                  const int32_t *__restrict__ resolutions, float *__restrict__ outputs,
                                                  uint32_t N, uint32_t L, float *__restrict__ dy_dx, GridIO io)
{
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= N) return;
    grid_fwd_point<D, C, PACKED>(inputs, grid, offsets, resolutions, outputs, L, dy_dx, io, b, blockIdx.y);
}

// Several grids over the same points in ONE launch (Mix3d2dEncoding: one 3-D and three 2-D grids, reference
// scene/gaussian_model.py:81-147): blockIdx.y runs over the grids' levels one grid after the other.
constexpr int GRID_MANY = 4;
struct GridManyJob {
    const float *table;             // forward: embeddings; backward: unused
    const float *grad;              // backward: the gradient's column block
    const int32_t *offsets, *resolutions;
    float *out;                     // forward: outputs' column block; backward: the table's gradient
    uint32_t D, L, first, chunks, chunk;      // first: the grid's first level in blockIdx.y; chunks x chunk points (backward)
    GridIO io;
};
struct GridManyJobs {
    int n;
    GridManyJob j[GRID_MANY];
};
__device__ __forceinline__ int grid_many_pick(const GridManyJobs &t, uint32_t y)
{
    int k = 0;
#pragma unroll
    for (int q = 1; q < GRID_MANY; q++)
        if (q < t.n && y >= t.j[q].first) k = q;
    return k;
}

template <uint32_t C>
__global__ void __launch_bounds__(256) k_grid_fwd_many(const float *__restrict__ inputs, GridManyJobs t, uint32_t N)
{
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= N) return;
    const int k = grid_many_pick(t, blockIdx.y);
    const GridManyJob &g = t.j[k];
    const uint32_t level = blockIdx.y - g.first;
    if (g.D == 3) grid_fwd_point<3, C>(inputs, g.table, g.offsets, g.resolutions, g.out, g.L, nullptr, g.io, b, level);
    else grid_fwd_point<2, C>(inputs, g.table, g.offsets, g.resolutions, g.out, g.L, nullptr, g.io, b, level);
}

// Table backward with the level's table privatised in LDS.  GSVC's tables are small (2^13 rows per 3-D level, 2^15
// per 2-D level in cfg_20240919) and every visible anchor adds into 2^D rows of every level: with global atomics
// that is N*L*2^D scattered memory-side requests (~17 G requests/s on MI355X: 1.0 ms for 181k points x 12 levels;
// measured with C consecutive lanes per point so that a corner is one 4*C-byte request).
// Here a workgroup owns (level, table slice, chunk of points): it accumulates the chunk's contributions to its slice
// with LDS atomics and then adds the slice to the gradient table with contiguous atomics (zeros skipped).  A level
// that needs more than gridDim.z slices falls back to global atomics in its z == 0 workgroups, so one launch covers
// every table size without the host knowing the (device-resident) offsets.
//
// The LDS accumulators are 64-bit FIXED-POINT integers, not floats.  Measured on gfx950 (tools/micro/lds_atomics.hip, random
// rows, 1024-thread workgroups): ds_add_f32 retires 0.33 lanes per cycle and CU, ds_add_u64 5.9, ds_add_u32 10.9 (rows
// C + 1 words apart; with a stride of C = 8 words a wave instruction falls on 8 banks and the integer forms drop to ~3).
// With float accumulators the atomics were the kernel's time (115 k points: 3-D 12 levels 515 us, 2-D 4 levels 125 us; now
// 128 / 79).  Scale: per LEVEL, a power of two 2^e with (largest |grad| of the level) x (most contributions a word can
// receive: chunk points x 2^D corners) x 2^e <= 2^62, so a word cannot overflow; a contribution is rounded to 2^-47 of the
// level's largest gradient or better (chunk <= 32 k points), i.e. entries down to 2^-23 of that maximum keep float accuracy
// and smaller ones lose it gradually (a float sum rounds to 2^-24 of its running value), and the sum of a slice no longer
// depends on the order of the adds.  The maxima come from k_grid_absmax (block maxima into a slot of a module-level ring; +inf
// stands for "not finite": that level takes the float path so that NaN / inf propagate as they did).
constexpr uint32_t BWD_SLICE_BYTES = 147456;       // 144 KiB of LDS: 2048 rows of 8 + 1 eight-byte words
constexpr uint32_t BWD_THREADS = 1024;
constexpr uint32_t BWD_MAX_SLICES = 16;
constexpr uint32_t ABSMAX_BLOCKS = 1024, ABSMAX_RING = 32;     // up to ABSMAX_BLOCKS / L (<= 64) blocks per level
__device__ float g_grid_absmax[ABSMAX_RING][ABSMAX_BLOCKS];

// largest |grad| of each level: gridDim = (blocks per level, L), block maxima to g_grid_absmax[slot][level * bpl + block]
template <uint32_t C>
__device__ __forceinline__ void grid_absmax_body(const float *__restrict__ grad, uint32_t N, const GridIO &io, uint32_t slot, uint32_t level,
                                                 uint32_t slot_level)
{
    __shared__ float red[4];
    float m = 0.f;
    bool bad = false;
    const float *gl = grad + (size_t)level * io.feat_level_stride;
    for (uint32_t b = blockIdx.x * 256 + threadIdx.x; b < N; b += gridDim.x * 256) {
        float g[C];
        load_row<C>(gl + (size_t)b * io.feat_point_stride, g);
#pragma unroll
        for (uint32_t ch = 0; ch < C; ch++) {
            const float a = fabsf(g[ch]);
            bad |= !(a <= 3.402823466e38f);
            m = fmaxf(m, a);
        }
    }
    if (bad) m = __builtin_inff();
#pragma unroll
    for (int k = 32; k >= 1; k >>= 1) m = fmaxf(m, __shfl_xor(m, k, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) g_grid_absmax[slot][slot_level * gridDim.x + blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

template <uint32_t C>
__global__ void __launch_bounds__(256) k_grid_absmax(const float *__restrict__ grad, uint32_t N, GridIO io, uint32_t slot)
{
    grid_absmax_body<C>(grad, N, io, slot, blockIdx.y, blockIdx.y);
}

template <uint32_t C>
__global__ void __launch_bounds__(256) k_grid_absmax_many(GridManyJobs t, uint32_t N, uint32_t slot)
{
    const GridManyJob &g = t.j[grid_many_pick(t, blockIdx.y)];
    grid_absmax_body<C>(g.grad, N, g.io, slot, blockIdx.y - g.first, blockIdx.y);
}

// x = value * 2^e, |x| < 2^51 -> nearest 64-bit integer: adding 1.5 * 2^52 in double leaves the integer in the low mantissa
// bits (two's complement), so the difference of the bit patterns is the result — 4 instructions (a float-only split into
// two 31-bit halves took 13, and with 2^D * C values per point the conversion was the loop's largest cost)
__device__ __forceinline__ long long fx_from_float(float x)
{
    const double magic = 6755399441055744.0;
    return __double_as_longlong((double)x + magic) - __double_as_longlong(magic);
}

// A located point in the form the table backward keeps in registers: the fractional position and renormalisation (the
// corner weights are re-multiplied from them, in locate()'s order), the 2^D row indices (16 bits each) and the valid mask.
template <uint32_t D>
struct BwdPoint {
    float frac[D], wn_re;
    unsigned long long rlo, rhi;       // 16-bit row indices of corners 0-3 / 4-7 (scalars: an array here ends up in scratch)
    uint32_t valid;                    // bit per corner; 0 for a point outside [0, 1]^D or past the end
};

// locate() for the table backward: the same float operations in the same order, results straight into the packed form
template <uint32_t D>
__device__ __forceinline__ void locate_packed(const float (&x)[D], uint32_t resolution, uint32_t hashmap_size, BwdPoint<D> &o)
{
#pragma clang fp contract(off)
    uint32_t cell[D];
    bool oob = false;
#pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        oob |= (x[d] < 0.f) | (x[d] > 1.f);
        const float pos = x[d] * (float)(resolution - 2) + 0.5f;
        cell[d] = (uint32_t)floorf(pos);
        o.frac[d] = pos - (float)cell[d];
    }
    float wn = 0.f;
    uint32_t valid = 0;
    o.rlo = o.rhi = 0;
#pragma unroll
    for (uint32_t idx = 0; idx < (1u << D); idx++) {
        float w = 1.f;
        uint32_t pl[D];
        bool border = false;
#pragma unroll
        for (uint32_t d = 0; d < D; d++) {
            if ((idx & (1u << d)) == 0) {
                w *= 1.f - o.frac[d];
                pl[d] = cell[d];
            } else {
                w *= o.frac[d];
                pl[d] = min(cell[d] + 1, resolution - 1);
            }
            border |= (pl[d] == 0) | (pl[d] == resolution - 1);
        }
        if (!border) {
            const unsigned long long rw = (unsigned long long)grid_row<D>(pl, hashmap_size, resolution) << (16 * (idx & 3));
            if (idx < 4) o.rlo |= rw; else o.rhi |= rw;
            valid |= 1u << idx;
            wn += w;
        }
    }
    if (wn == 0.f) wn = 1e-9f;
    o.wn_re = 1.0f / wn;
    o.valid = oob ? 0u : valid;
}

// corners of one located point that fall into the slice [row_lo, row_lo + rows): scaled gradient row -> fixed point -> LDS.
// A loop over the SET bits of `mine`: a wave runs as many rounds as its busiest lane has corners in the slice (a level
// of 4 slices: ~5 of 8), and an LDS atomic costs the same ~11 cycles whether 4 or 64 of its lanes are active.
template <uint32_t D, uint32_t C, uint32_t CP>
__device__ __forceinline__ void bwd_accumulate(const BwdPoint<D> &pt, uint32_t mine, const float (&g)[C], unsigned long long *acc,
                                               uint32_t row_lo)
{
    while (mine) {
        const uint32_t idx = (uint32_t)__ffs((int)mine) - 1u;
        mine &= mine - 1u;
        const unsigned long long word = (D == 3 && idx >= 4) ? pt.rhi : pt.rlo;
        const uint32_t r = ((uint32_t)(word >> (16 * (idx & 3))) & 0xffffu) - row_lo;
        float w = 1.f;               // the corner's weight, multiplied in the order locate() uses
#pragma unroll
        for (uint32_t d = 0; d < D; d++) w *= (idx & (1u << d)) ? pt.frac[d] : 1.f - pt.frac[d];
        w *= pt.wn_re;
        unsigned long long *dst = acc + r * CP;
#pragma unroll
        for (uint32_t ch = 0; ch < C; ch++) atomicAdd(dst + ch, (unsigned long long)fx_from_float(w * g[ch]));
    }
}

// A workgroup owns (chunk of points, level, table slice).  The chunks are LARGE (~N * L * slices / 256 points: one round of
// workgroups on the chip): every workgroup ends by adding its slice to the gradient table with float atomics, and the chip
// retires only ~0.13 T of those per second however contiguous they are (measured: 23 M in 170 us) — with 2.7 k-point chunks
// that flush, not the accumulation, was the kernel's time.  A workgroup walks its chunk 1024 points at a time; the loop is
// software-pipelined over two points per thread (A / B): the positions of the next point and the gradient row of the
// current one are in flight while the previous point's corners are added.
// level: the level inside its grid; slot_level: its row of the absmax slot (= level, or the launch's running level when several
// grids share a launch); chunk_idx / slice: the workgroup's chunk of points and table slice
template <uint32_t D, uint32_t C>
__device__ __forceinline__ void grid_bwd_lds_body(const float *__restrict__ grad, const float *__restrict__ inputs,
                                                  const int32_t *__restrict__ offsets, const int32_t *__restrict__ resolutions,
                                                  float *__restrict__ grad_grid, uint32_t N, uint32_t chunk, const GridIO &io,
                                                  uint32_t slot, uint32_t absmax_bpl, uint32_t chunk_idx, uint32_t level,
                                                  uint32_t slot_level, long long *acc, float *s_max)
{
    constexpr uint32_t CP = C > 1 ? C + 1 : 1;
    constexpr uint32_t SLICE_ROWS = BWD_SLICE_BYTES / (8 * CP);
    const uint32_t slice = blockIdx.z, tid = threadIdx.x;
    const uint32_t hashmap_size = (uint32_t)(offsets[level + 1] - offsets[level]);
    const uint32_t resolution = (uint32_t)resolutions[level];
    const uint32_t slices = (hashmap_size + SLICE_ROWS - 1) / SLICE_ROWS;
    const uint32_t p0 = chunk_idx * chunk, p1 = min(N, p0 + chunk);
    // largest |grad| of this level (block maxima of k_grid_absmax)
    float gmax = tid < absmax_bpl ? g_grid_absmax[slot][slot_level * absmax_bpl + tid] : 0.f;
#pragma unroll
    for (int k = 32; k >= 1; k >>= 1) gmax = fmaxf(gmax, __shfl_xor(gmax, k, 64));
    if ((tid & 63) == 0) s_max[tid >> 6] = gmax;
    __syncthreads();
    gmax = 0.f;
#pragma unroll
    for (uint32_t w = 0; w < BWD_THREADS / 64; w++) gmax = fmaxf(gmax, s_max[w]);
    // (a level whose largest |grad| is below 2^-60 keeps the float path: its scale would leave float's exponent range)
    const bool fallback = absmax_bpl == 0 || slices > gridDim.z || hashmap_size > 65536u || !(gmax <= 3.402823466e38f) ||
                          (gmax != 0.f && gmax < 8.673617379884035e-19f);
    if (fallback ? slice != 0 : slice >= slices) return;         // block-uniform
    grad_grid += (size_t)(uint32_t)offsets[level] * C;
    if (fallback) {
        // float atomics straight into the table; C consecutive lanes per point: one contiguous 4*C-byte segment per corner
        for (uint32_t t = p0 * C + tid; t < p1 * C; t += BWD_THREADS) {
            const uint32_t b = t / C, ch = t - b * C;
            float x[D];
            bool oob = false;
#pragma unroll
            for (uint32_t d = 0; d < D; d++) {
                x[d] = inputs[(size_t)b * io.in_stride + io.in_col[d]];
                oob |= (x[d] < 0.f) | (x[d] > 1.f);
            }
            if (oob) continue;
            const float g = grad[(size_t)level * io.feat_level_stride + (size_t)b * io.feat_point_stride + ch];
            Cell<D> c;
            locate<D>(x, resolution, hashmap_size, c);
#pragma unroll
            for (uint32_t idx = 0; idx < (1u << D); idx++)
                if (c.valid & (1u << idx)) atomicAdd(grad_grid + (size_t)c.row[idx] * C + ch, c.w[idx] * c.wn_re * g);
        }
        return;
    }
    if (gmax == 0.f || p0 >= p1) return;                          // every contribution of the launch is zero / empty chunk
    // 2^e: gmax < 2^eg, contributions per word <= chunk * 2^D <= 2^ec (and one contribution below 2^51: fx_from_float)
    const int eg = ilogbf(gmax) + 1, ec = max(32 - __clz((int)max(chunk, 2u) - 1) + (int)D, 11);
    const int e = 62 - eg - ec;                                   // -93 .. 111
    const float scale = ldexpf(1.f, e), inv_scale = ldexpf(1.f, -e);
    const uint32_t row_lo = slice * SLICE_ROWS, rows = min(SLICE_ROWS, hashmap_size - row_lo);
    for (uint32_t i = tid; i < rows * CP; i += BWD_THREADS) acc[i] = 0;
    __syncthreads();
    unsigned long long *uacc = reinterpret_cast<unsigned long long *>(acc);
    const float *glev = grad + (size_t)level * io.feat_level_stride;

    auto load_x = [&](uint32_t b, float (&x)[D]) {
#pragma unroll
        for (uint32_t d = 0; d < D; d++) x[d] = b < p1 ? inputs[(size_t)b * io.in_stride + io.in_col[d]] : -1.f;
    };
    // locate point b; which of its corners are this slice's; request its gradient row if any is
    auto stage1 = [&](uint32_t b, const float (&x)[D], BwdPoint<D> &pt, uint32_t &mine, float (&g)[C]) {
        locate_packed<D>(x, resolution, hashmap_size, pt);
        mine = 0;
#pragma unroll
        for (uint32_t idx = 0; idx < (1u << D); idx++) {
            const uint32_t r = ((uint32_t)((idx < 4 ? pt.rlo : pt.rhi) >> (16 * (idx & 3))) & 0xffffu) - row_lo;
            if ((pt.valid & (1u << idx)) && r < rows) mine |= 1u << idx;
        }
        if (mine) load_row<C>(glev + (size_t)b * io.feat_point_stride, g);
    };
    auto stage2 = [&](const BwdPoint<D> &pt, uint32_t mine, float (&g)[C]) {
        if (!mine) return;
#pragma unroll
        for (uint32_t ch = 0; ch < C; ch++) g[ch] *= scale;      // a power of two: exact
        bwd_accumulate<D, C, CP>(pt, mine, g, uacc, row_lo);
    };
    BwdPoint<D> ptA, ptB;
    float gA[C], gB[C], xA[D], xB[D];
    uint32_t mineA = 0, mineB = 0;
    uint32_t b = p0 + tid;
    load_x(b, xA);
    load_x(b + BWD_THREADS, xB);
    stage1(b, xA, ptA, mineA, gA);
    // invariant at the top: A = point b (located, row requested), xB = positions of point b + 1024 (requested)
    for (; b < p1; b += 2 * BWD_THREADS) {
        load_x(b + 2 * BWD_THREADS, xA);
        stage1(b + BWD_THREADS, xB, ptB, mineB, gB);
        stage2(ptA, mineA, gA);
        load_x(b + 3 * BWD_THREADS, xB);
        stage1(b + 2 * BWD_THREADS, xA, ptA, mineA, gA);
        stage2(ptB, mineB, gB);
    }
    __syncthreads();
    float *out = grad_grid + (size_t)row_lo * C;
    for (uint32_t i = tid; i < rows * C; i += BWD_THREADS) {
        const long long v = acc[(i / C) * CP + i % C];
        if (v != 0) atomicAdd(out + i, (float)v * inv_scale);
    }
}

template <uint32_t D, uint32_t C>
__global__ void __launch_bounds__(BWD_THREADS) k_grid_bwd_lds(const float *__restrict__ grad, const float *__restrict__ inputs,
                                                              const int32_t *__restrict__ offsets,
                                                              const int32_t *__restrict__ resolutions,
                                                              float *__restrict__ grad_grid, uint32_t N, uint32_t chunk, GridIO io,
                                                              uint32_t slot, uint32_t absmax_bpl)
{
    extern __shared__ long long acc[];
    __shared__ float s_max[BWD_THREADS / 64];
    grid_bwd_lds_body<D, C>(grad, inputs, offsets, resolutions, grad_grid, N, chunk, io, slot, absmax_bpl, blockIdx.x, blockIdx.y,
                            blockIdx.y, acc, s_max);
}

template <uint32_t C>
__global__ void __launch_bounds__(BWD_THREADS) k_grid_bwd_lds_many(const float *__restrict__ inputs, GridManyJobs t, uint32_t N,
                                                                   uint32_t slot, uint32_t absmax_bpl)
{
    extern __shared__ long long acc[];
    __shared__ float s_max[BWD_THREADS / 64];
    const GridManyJob &g = t.j[grid_many_pick(t, blockIdx.y)];
    if (blockIdx.x >= g.chunks) return;                  // (workgroup-uniform) the grids differ in their number of chunks
    const uint32_t level = blockIdx.y - g.first;
    if (g.D == 3)
        grid_bwd_lds_body<3, C>(g.grad, inputs, g.offsets, g.resolutions, g.out, N, g.chunk, g.io, slot, absmax_bpl, blockIdx.x, level,
                                blockIdx.y, acc, s_max);
    else
        grid_bwd_lds_body<2, C>(g.grad, inputs, g.offsets, g.resolutions, g.out, N, g.chunk, g.io, slot, absmax_bpl, blockIdx.x, level,
                                blockIdx.y, acc, s_max);
}

// grad_inputs[b,d] = sum_l sum_c grad[l,b,c] * dy_dx[b,l,d,c]
template <uint32_t D, uint32_t C>
__global__ void __launch_bounds__(256) k_grid_input_bwd(const float *__restrict__ grad, const float *__restrict__ dy_dx,
                                                        float *__restrict__ grad_inputs, uint32_t N, uint32_t L, GridIO io)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= N * D) return;
    const uint32_t b = t / D, d = t - b * D;
    const float *dd = dy_dx + (size_t)b * L * D * C;
    float r = 0.f;
    for (uint32_t l = 0; l < L; l++) {
        float g[C], y[C];
        load_row<C>(grad + (size_t)l * io.feat_level_stride + (size_t)b * io.feat_point_stride, g);
        load_row<C>(dd + (size_t)l * D * C + d * C, y);
#pragma unroll
        for (uint32_t ch = 0; ch < C; ch++) r += g[ch] * y[ch];
    }
    grad_inputs[t] = r;
}

static std::atomic<uint32_t> g_absmax_ring{0};      // slot of g_grid_absmax a backward launch uses (single-grid and multi-grid alike)

template <uint32_t D, uint32_t C>
static void launch_fwd(const float *inputs, const float *emb, const int32_t *off, const int32_t *res, float *out,
                       uint32_t N, uint32_t L, float *dy_dx, hipStream_t s, const GridIO &io)
{
    { ProfScope _prof("k_grid_fwd", s); hipLaunchKernelGGL((k_grid_fwd<D, C>), dim3((N + 255) / 256, L), dim3(256), 0, s, inputs, emb, off, res, out, N, L,
                       dy_dx, io); }
}

template <uint32_t D, uint32_t C>
static void launch_bwd(const float *grad, const float *inputs, const int32_t *off, const int32_t *res, float *gemb,
                       uint32_t N, uint32_t L, const float *dy_dx, float *ginp, hipStream_t s, const GridIO &io)
{
    {
        // one round of workgroups on the chip.  The table sizes are device-resident: the slice count per level is taken as
        // what cfg_20240919's tables need (2^13 rows per 3-D level, 2^15 per 2-D level, 8 features); smaller tables leave
        // workgroups that return at once
        static bool attr_set = false;
        if (!attr_set) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_grid_bwd_lds<D, C>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)BWD_SLICE_BYTES);
            attr_set = true;
        }
        static const uint32_t want_wgs = getenv("GSVC_GRID_BWD_WGS") ? (uint32_t)atoi(getenv("GSVC_GRID_BWD_WGS")) : 256;
        const uint32_t slices_guess = D == 3 ? 4 : (D == 2 ? 16 : 1);
        uint32_t chunks = std::max(1u, want_wgs / (L * slices_guess));      // rounded down: a second round of workgroups doubles the time
        chunks = std::min(chunks, (N + 2047) / 2048);              // at least 2048 points per workgroup
        if (chunks == 0 || deterministic()) chunks = 1;           // deterministic: one workgroup per (level, slice) — every table word
                                                                   // receives ONE float add of an exact fixed-point sum
        const uint32_t chunk = (N + chunks - 1) / chunks;
        const uint32_t slot = g_absmax_ring.fetch_add(1) % ABSMAX_RING;
        const uint32_t bpl = std::max(1u, std::min(std::min(ABSMAX_BLOCKS / L, 64u), (N + 1023) / 1024));
        ProfScope _prof("k_grid_bwd", s);
        if (L <= ABSMAX_BLOCKS)
            hipLaunchKernelGGL((k_grid_absmax<C>), dim3(bpl, L), dim3(256), 0, s, grad, N, io, slot);
        hipLaunchKernelGGL((k_grid_bwd_lds<D, C>), dim3(chunks, L, BWD_MAX_SLICES), dim3(BWD_THREADS), BWD_SLICE_BYTES, s, grad,
                           inputs, off, res, gemb, N, chunk, io, slot, L <= ABSMAX_BLOCKS ? bpl : 0u);
    }
    if (dy_dx && ginp)
        { ProfScope _prof("k_grid_input_bwd", s); hipLaunchKernelGGL((k_grid_input_bwd<D, C>), dim3((N * D + 255) / 256), dim3(256), 0, s, grad, dy_dx, ginp, N, L, io); }
}

#define GSVC_DISPATCH_C(D_, CALL)                                                        \
    switch (C) {                                                                         \
        case 1: CALL(D_, 1); break;                                                      \
        case 2: CALL(D_, 2); break;                                                      \
        case 4: CALL(D_, 4); break;                                                      \
        case 8: CALL(D_, 8); break;                                                      \
        case 16: CALL(D_, 16); break;                                                    \
        case 32: CALL(D_, 32); break;                                                    \
        default:                                                                         \
            set_error("GridEncoding: n_fearures must be 1, 2, 4, 8, 16 or 32.");         \
            return GSVC_E_UNSUPPORTED;                                                   \
    }

#define GSVC_DISPATCH_DC(CALL)                                             \
    switch (D) {                                                           \
        case 1: GSVC_DISPATCH_C(1, CALL) break;                            \
        case 2: GSVC_DISPATCH_C(2, CALL) break;                            \
        case 3: GSVC_DISPATCH_C(3, CALL) break;                            \
        default:                                                           \
            set_error("GridEncoding: num_dim must be 1, 2, 3.");           \
            return GSVC_E_UNSUPPORTED;                                     \
    }

// bits[row] = sign bits of x[row][0..8) (x >= 0 -> 1: the STE_binary rule, 0 -> +1)
__global__ void __launch_bounds__(256) k_pack_sign_bits(const float *__restrict__ x, long long rows, uint8_t *__restrict__ bits)
{
    const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    const float4 a = reinterpret_cast<const float4 *>(x)[2 * r], b = reinterpret_cast<const float4 *>(x)[2 * r + 1];
    bits[r] = (uint8_t)((a.x >= 0.f) | ((a.y >= 0.f) << 1) | ((a.z >= 0.f) << 2) | ((a.w >= 0.f) << 3) | ((b.x >= 0.f) << 4) |
                        ((b.y >= 0.f) << 5) | ((b.z >= 0.f) << 6) | ((b.w >= 0.f) << 7));
}

}  // namespace gsvc

using namespace gsvc;

static GridIO grid_io_default(uint32_t N, uint32_t D, uint32_t C)
{
    GridIO io;
    io.feat_level_stride = (long long)N * C; io.feat_point_stride = C;
    io.in_stride = (int)D; io.in_col[0] = 0; io.in_col[1] = 1; io.in_col[2] = 2;
    return io;
}

static int grid_io_from(const gsvc_grid_io *d, uint32_t D, uint32_t C, GridIO &io, const char *what)
{
    GSVC_REQUIRE(d && d->feat_level_stride >= 0 && d->feat_point_stride >= (int64_t)C && d->in_stride >= 1, "%s: bad layout", what);
    // vector row accesses: a point's C features must start on a 4 * min(C, 4)-byte boundary
    const int64_t al = C >= 4 ? 4 : (int64_t)C;
    GSVC_REQUIRE(d->feat_level_stride % al == 0 && d->feat_point_stride % al == 0, "%s: feature strides must be multiples of %d floats",
                 what, (int)al);
    io.feat_level_stride = d->feat_level_stride; io.feat_point_stride = d->feat_point_stride; io.in_stride = d->in_stride;
    for (uint32_t k = 0; k < 3; k++) {
        io.in_col[k] = k < D ? d->in_col[k] : 0;
        GSVC_REQUIRE(io.in_col[k] >= 0 && io.in_col[k] < d->in_stride, "%s: input column outside the row", what);
    }
    return GSVC_OK;
}

static int grid_forward_impl(const float *inputs, const float *embeddings, const int32_t *offsets, const int32_t *resolutions,
                             float *outputs, uint32_t N, uint32_t D, uint32_t C, uint32_t L, float *dy_dx, void *stream, const GridIO &io)
{
    hipStream_t s = (hipStream_t)stream;
    if (N == 0 || L == 0) {
        if (!(D >= 1 && D <= 3)) { set_error("GridEncoding: num_dim must be 1, 2, 3."); return GSVC_E_UNSUPPORTED; }
        return GSVC_OK;
    }
    GSVC_REQUIRE(inputs && embeddings && offsets && resolutions && outputs, "grid_forward: NULL pointer");
#define FWD_CALL(D_, C_) launch_fwd<D_, C_>(inputs, embeddings, offsets, resolutions, outputs, N, L, dy_dx, s, io)
    GSVC_DISPATCH_DC(FWD_CALL)
#undef FWD_CALL
    return check_launch("grid_forward");
}

static int grid_backward_impl(const float *grad, const float *inputs, const int32_t *offsets, const int32_t *resolutions,
                              float *grad_embeddings, uint32_t N, uint32_t D, uint32_t C, uint32_t L, const float *dy_dx,
                              float *grad_inputs, void *stream, const GridIO &io)
{
    hipStream_t s = (hipStream_t)stream;
    if (N == 0 || L == 0) return GSVC_OK;
    GSVC_REQUIRE(grad && inputs && offsets && resolutions && grad_embeddings, "grid_backward: NULL pointer");
#define BWD_CALL(D_, C_) launch_bwd<D_, C_>(grad, inputs, offsets, resolutions, grad_embeddings, N, L, dy_dx, grad_inputs, s, io)
    GSVC_DISPATCH_DC(BWD_CALL)
#undef BWD_CALL
    return check_launch("grid_backward");
}

extern "C" int gsvc_grid_forward(const float *inputs, const float *embeddings, const int32_t *offsets,
                                 const int32_t *resolutions, float *outputs, uint32_t N, uint32_t D, uint32_t C,
                                 uint32_t L, float *dy_dx, void *stream)
{
    return grid_forward_impl(inputs, embeddings, offsets, resolutions, outputs, N, D, C, L, dy_dx, stream, grid_io_default(N, D, C));
}

extern "C" int gsvc_grid_backward(const float *grad, const float *inputs, const float *embeddings,
                                  const int32_t *offsets, const int32_t *resolutions, float *grad_embeddings, uint32_t N,
                                  uint32_t D, uint32_t C, uint32_t L, const float *dy_dx, float *grad_inputs,
                                  void *stream)
{
    (void)embeddings;
    return grid_backward_impl(grad, inputs, offsets, resolutions, grad_embeddings, N, D, C, L, dy_dx, grad_inputs, stream,
                              grid_io_default(N, D, C));
}

extern "C" int gsvc_grid_forward_ex(const float *inputs, const float *embeddings, const int32_t *offsets, const int32_t *resolutions,
                                    float *outputs, uint32_t N, uint32_t D, uint32_t C, uint32_t L, const gsvc_grid_io *layout,
                                    void *stream)
{
    GridIO io;
    if (int rc = grid_io_from(layout, D, C, io, "grid_forward_ex")) return rc;
    return grid_forward_impl(inputs, embeddings, offsets, resolutions, outputs, N, D, C, L, nullptr, stream, io);
}

extern "C" int gsvc_grid_backward_ex(const float *grad, const float *inputs, const int32_t *offsets, const int32_t *resolutions,
                                     float *grad_embeddings, uint32_t N, uint32_t D, uint32_t C, uint32_t L, const gsvc_grid_io *layout,
                                     void *stream)
{
    GridIO io;
    if (int rc = grid_io_from(layout, D, C, io, "grid_backward_ex")) return rc;
    return grid_backward_impl(grad, inputs, offsets, resolutions, grad_embeddings, N, D, C, L, nullptr, nullptr, stream, io);
}

// ---- several grids over the same points, one launch each way
static int grid_many_fill(const char *what, const gsvc_grid_many_job *jobs, int32_t n, uint32_t C, bool backward, GridManyJobs &t, uint32_t &levels)
{
    GSVC_REQUIRE(jobs && n >= 1 && n <= GRID_MANY, "%s: 1..%d grids", what, GRID_MANY);
    t.n = n;
    levels = 0;
    for (int k = 0; k < n; k++) {
        const gsvc_grid_many_job &j = jobs[k];
        GSVC_REQUIRE((j.D == 2 || j.D == 3) && j.L >= 1, "%s: grids of 2 or 3 dimensions with at least one level", what);
        GSVC_REQUIRE(j.offsets && j.resolutions && j.features && (backward ? j.grad_embeddings != nullptr : j.embeddings != nullptr), "%s: NULL pointer", what);
        GridManyJob &g = t.j[k];
        if (int rc = grid_io_from(&j.layout, j.D, C, g.io, what)) return rc;
        g.table = j.embeddings; g.grad = j.features; g.out = backward ? j.grad_embeddings : j.features;
        g.offsets = j.offsets; g.resolutions = j.resolutions; g.D = j.D; g.L = j.L; g.first = levels; g.chunks = g.chunk = 0;
        levels += j.L;
    }
    for (int k = n; k < GRID_MANY; k++) t.j[k] = t.j[0];
    return GSVC_OK;
}

extern "C" int gsvc_grid_forward_many(const float *inputs, const gsvc_grid_many_job *jobs, int32_t n_jobs, uint32_t N, uint32_t C, void *stream)
{
    GridManyJobs t;
    uint32_t levels = 0;
    if (int rc = grid_many_fill("grid_forward_many", jobs, n_jobs, C, false, t, levels)) return rc;
    if (N == 0) return GSVC_OK;
    GSVC_REQUIRE(inputs, "grid_forward_many: NULL pointer");
    hipStream_t s = (hipStream_t)stream;
    ProfScope _prof("k_grid_fwd_many", s);
    const dim3 grid((N + 255) / 256, levels);
    switch (C) {
        case 2: hipLaunchKernelGGL((k_grid_fwd_many<2>), grid, dim3(256), 0, s, inputs, t, N); break;
        case 4: hipLaunchKernelGGL((k_grid_fwd_many<4>), grid, dim3(256), 0, s, inputs, t, N); break;
        case 8: hipLaunchKernelGGL((k_grid_fwd_many<8>), grid, dim3(256), 0, s, inputs, t, N); break;
        default: set_error("grid_forward_many: 2, 4 or 8 features per level"); return GSVC_E_UNSUPPORTED;
    }
    return check_launch("grid_forward_many");
}

template <uint32_t C>
static void launch_bwd_many(const float *inputs, GridManyJobs &t, uint32_t levels, uint32_t N, hipStream_t s)
{
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_grid_bwd_lds_many<C>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)BWD_SLICE_BYTES);
        attr_set = true;
    }
    // every grid gets the chunks it would get in a launch of its own (one round of workgroups on the chip per grid: launch_bwd)
    static const uint32_t want_wgs = getenv("GSVC_GRID_BWD_WGS") ? (uint32_t)atoi(getenv("GSVC_GRID_BWD_WGS")) : 256;
    uint32_t max_chunks = 1;
    for (int k = 0; k < t.n; k++) {
        GridManyJob &g = t.j[k];
        const uint32_t slices_guess = g.D == 3 ? 4 : 16;
        uint32_t chunks = std::max(1u, want_wgs / (g.L * slices_guess));
        chunks = std::max(1u, std::min(chunks, (N + 2047) / 2048));
        if (deterministic()) chunks = 1;                           // (as launch_bwd)
        g.chunks = chunks;
        g.chunk = (N + chunks - 1) / chunks;
        max_chunks = std::max(max_chunks, chunks);
    }
    const uint32_t slot = g_absmax_ring.fetch_add(1) % ABSMAX_RING;
    const uint32_t bpl = std::max(1u, std::min(std::min(ABSMAX_BLOCKS / levels, 64u), (N + 1023) / 1024));
    ProfScope _prof("k_grid_bwd_many", s);
    hipLaunchKernelGGL((k_grid_absmax_many<C>), dim3(bpl, levels), dim3(256), 0, s, t, N, slot);
    hipLaunchKernelGGL((k_grid_bwd_lds_many<C>), dim3(max_chunks, levels, BWD_MAX_SLICES), dim3(BWD_THREADS), BWD_SLICE_BYTES, s, inputs, t, N,
                       slot, bpl);
}

extern "C" int gsvc_grid_backward_many(const float *inputs, const gsvc_grid_many_job *jobs, int32_t n_jobs, uint32_t N, uint32_t C, void *stream)
{
    GridManyJobs t;
    uint32_t levels = 0;
    if (int rc = grid_many_fill("grid_backward_many", jobs, n_jobs, C, true, t, levels)) return rc;
    if (N == 0) return GSVC_OK;
    GSVC_REQUIRE(inputs && levels <= ABSMAX_BLOCKS, "grid_backward_many: NULL pointer / too many levels");
    hipStream_t s = (hipStream_t)stream;
    switch (C) {
        case 2: launch_bwd_many<2>(inputs, t, levels, N, s); break;
        case 4: launch_bwd_many<4>(inputs, t, levels, N, s); break;
        case 8: launch_bwd_many<8>(inputs, t, levels, N, s); break;
        default: set_error("grid_backward_many: 2, 4 or 8 features per level"); return GSVC_E_UNSUPPORTED;
    }
    return check_launch("grid_backward_many");
}

extern "C" int gsvc_pack_sign_bits(const float *x, int64_t rows, uint8_t *bits, void *stream)
{
    GSVC_REQUIRE(rows >= 0, "pack_sign_bits: bad size");
    if (rows == 0) return GSVC_OK;
    GSVC_REQUIRE(x && bits && (reinterpret_cast<uintptr_t>(x) & 15) == 0, "pack_sign_bits: NULL or unaligned pointer");
    hipLaunchKernelGGL(k_pack_sign_bits, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, (long long)rows, bits);
    return check_launch("pack_sign_bits");
}

extern "C" int gsvc_grid_forward_packed(const float *inputs, const uint8_t *table_bits, const int32_t *offsets, const int32_t *resolutions,
                                        float *outputs, uint32_t N, uint32_t D, uint32_t L, const gsvc_grid_io *layout, void *stream)
{
    const uint32_t C = 8;
    GridIO io = grid_io_default(N, D, C);
    if (layout)
        if (int rc = grid_io_from(layout, D, C, io, "grid_forward_packed")) return rc;
    if (D != 2 && D != 3) {
        set_error("grid_forward_packed: num_dim must be 2 or 3 (8 features per row)");
        return GSVC_E_UNSUPPORTED;
    }
    if (N == 0 || L == 0) return GSVC_OK;
    GSVC_REQUIRE(inputs && table_bits && offsets && resolutions && outputs, "grid_forward_packed: NULL pointer");
    hipStream_t s = (hipStream_t)stream;
    ProfScope _prof("k_grid_fwd", s);
    const float *tb = reinterpret_cast<const float *>(table_bits);
    if (D == 3)
        hipLaunchKernelGGL((k_grid_fwd<3, 8, true>), dim3((N + 255) / 256, L), dim3(256), 0, s, inputs, tb, offsets, resolutions, outputs, N, L,
                           (float *)nullptr, io);
    else
        hipLaunchKernelGGL((k_grid_fwd<2, 8, true>), dim3((N + 255) / 256, L), dim3(256), 0, s, inputs, tb, offsets, resolutions, outputs, N, L,
                           (float *)nullptr, io);
    return check_launch("grid_forward_packed");
}

// gsvc_amd/csrc/anchor.hip — GPU side of the anchor-geometry decoder (gsvc_amd/anchor_codec.py; stands in for the G-PCC tmc3 step of
// reference utils/encodings.py:780-826 decode_anchor): the per-level interleaved rANS streams of the occupancy octree are decoded by
// one workgroup per level (the levels' entropy streams are independent; only the node expansion is level after level), then every
// level expands its nodes' occupancy masks into the next level's nodes.
#include "common.h"

namespace gsvc {

constexpr int AC_PROB_BITS = 12;
constexpr uint32_t AC_RANS_L = 1u << 16;
constexpr int AC_THREADS = 1024;
constexpr int AC_MAX_LPT = 16;          // lanes per thread: 16 384 lanes at most
constexpr int AC_MAX_LEVELS = 16;

struct AnchorLevels {
    const uint32_t *states[AC_MAX_LEVELS];      // [lanes] initial decoder states
    const uint16_t *words[AC_MAX_LEVELS];       // [n_words] renormalisation words in consumption order
    const uint16_t *freq[AC_MAX_LEVELS];        // [256] frequencies, sum 2^12
    uint8_t *out[AC_MAX_LEVELS];                // [n] symbols
    long long n[AC_MAX_LEVELS], n_words[AC_MAX_LEVELS];
    int lanes[AC_MAX_LEVELS];
};

// Symbol i of a level belongs to lane i % lanes; rows of `lanes` symbols are decoded first to last, and within a row the lanes that
// renormalise take their 16-bit words from ONE shared stream in lane order — the position of a lane's word is the number of
// renormalising lanes before it: thread t owns the contiguous lanes [t LPT, (t + 1) LPT), counts its own, one workgroup scan per row.
__global__ void __launch_bounds__(AC_THREADS) k_anchor_rans_decode(AnchorLevels lv, int *__restrict__ error)
{
    __shared__ uint8_t slot2sym[1 << AC_PROB_BITS];
    __shared__ uint16_t s_freq[256], s_cum[257];
    __shared__ int wave_sum[AC_THREADS / 64];
    const int level = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const uint32_t *__restrict__ states = lv.states[level];
    const uint16_t *__restrict__ words = lv.words[level];
    uint8_t *__restrict__ out = lv.out[level];
    const long long n = lv.n[level], n_words = lv.n_words[level];
    const int lanes = lv.lanes[level];
    if (n == 0) return;
    if (t < 256) s_freq[t] = lv.freq[level][t];
    __syncthreads();
    if (t == 0) {
        int c = 0;
        for (int s = 0; s < 256; s++) { s_cum[s] = (uint16_t)c; c += s_freq[s]; }
        s_cum[256] = (uint16_t)c;
        if (c != (1 << AC_PROB_BITS)) atomicOr(error, 1);
    }
    __syncthreads();
    if (s_cum[256] != (1 << AC_PROB_BITS)) return;
    for (int slot = t; slot < (1 << AC_PROB_BITS); slot += AC_THREADS) {
        int lo = 0, hi = 255;                    // the symbol whose [cum, cum + freq) holds the slot
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (s_cum[mid] <= slot) lo = mid; else hi = mid - 1;
        }
        slot2sym[slot] = (uint8_t)lo;
    }
    __syncthreads();
    const int lpt = (lanes + AC_THREADS - 1) / AC_THREADS;
    const int l0 = t * lpt;
    uint32_t x[AC_MAX_LPT];
#pragma unroll
    for (int q = 0; q < AC_MAX_LPT; q++) x[q] = (q < lpt && l0 + q < lanes) ? states[l0 + q] : AC_RANS_L;
    const long long rows = (n + lanes - 1) / lanes;
    long long ptr = 0;
    for (long long r = 0; r < rows; r++) {
        const long long lo = r * lanes;
        const int k = (int)((n - lo) < lanes ? (n - lo) : lanes);
        int c = 0;
        unsigned need = 0;
#pragma unroll
        for (int q = 0; q < AC_MAX_LPT; q++) {
            if (q < lpt && l0 + q < k) {
                const uint32_t slot = x[q] & ((1u << AC_PROB_BITS) - 1);
                const int s = slot2sym[slot];
                out[lo + l0 + q] = (uint8_t)s;
                x[q] = (uint32_t)s_freq[s] * (x[q] >> AC_PROB_BITS) + slot - s_cum[s];
                if (x[q] < AC_RANS_L) { need |= 1u << q; c++; }
            }
        }
        // exclusive scan of c over the workgroup (thread order = lane order)
        int inc = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(inc, d, 64);
            if (lane >= d) inc += o;
        }
        if (lane == 63) wave_sum[wave] = inc;
        __syncthreads();
        int base = 0, total = 0;
#pragma unroll
        for (int w = 0; w < AC_THREADS / 64; w++) {
            const int v = wave_sum[w];
            base += w < wave ? v : 0;
            total += v;
        }
        __syncthreads();
        long long at = ptr + base + inc - c;
        if (ptr + total > n_words) {          // truncated stream: every thread sees it
            if (t == 0) atomicOr(error, 2);
            return;
        }
#pragma unroll
        for (int q = 0; q < AC_MAX_LPT; q++)
            if (need & (1u << q)) x[q] = (x[q] << 16) | words[at++];
        ptr += total;
    }
    // the coder must have returned to its initial state and used every word
    bool bad = ptr != n_words;
#pragma unroll
    for (int q = 0; q < AC_MAX_LPT; q++) bad |= x[q] != AC_RANS_L;
    if (bad) atomicOr(error, 4);
}

// next-level nodes: child j of node i is the j-th set bit of its occupancy mask; first[i] = children of the nodes before i
__global__ void __launch_bounds__(256) k_octree_expand(const long long *__restrict__ nodes, const uint8_t *__restrict__ occ,
                                                       const long long *__restrict__ first_incl, long long n, long long capacity,
                                                       long long *__restrict__ out, int *__restrict__ error)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const unsigned m = occ[i];
    const long long key = nodes ? nodes[i] << 3 : 0;
    long long at = first_incl[i] - __popc(m);
    if (m == 0 || first_incl[i] > capacity) {
        atomicOr(error, 8);
        return;
    }
#pragma unroll
    for (int c = 0; c < 8; c++)
        if (m & (1u << c)) out[at++] = key | c;
}

__global__ void __launch_bounds__(256) k_popcount_u8(const uint8_t *__restrict__ occ, long long n, long long *__restrict__ cnt)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) cnt[i] = __popc((unsigned)occ[i]);
}

__device__ __forceinline__ unsigned long long ac_compact(unsigned long long v)
{
    v &= 0x1249249249249249ull;
    v = (v | (v >> 2)) & 0x10C30C30C30C30C3ull;
    v = (v | (v >> 4)) & 0x100F00F00F00F00Full;
    v = (v | (v >> 8)) & 0x001F0000FF0000FFull;
    v = (v | (v >> 16)) & 0x001F00000000FFFFull;
    v = (v | (v >> 32)) & 0xFFFFull;
    return v;
}

// Morton key (x at bit 3 b + 2, y at 3 b + 1, z at 3 b) -> (x, y, z)
__global__ void __launch_bounds__(256) k_demorton(const long long *__restrict__ key, long long n, long long *__restrict__ xyz)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const unsigned long long k = (unsigned long long)key[i];
    xyz[3 * i] = (long long)ac_compact(k >> 2);
    xyz[3 * i + 1] = (long long)ac_compact(k >> 1);
    xyz[3 * i + 2] = (long long)ac_compact(k);
}

}  // namespace gsvc

using namespace gsvc;

extern "C" int gsvc_anchor_rans_decode(const gsvc_anchor_level *levels_host, int32_t n_levels, int32_t *error, void *stream)
{
    GSVC_REQUIRE(levels_host && error && n_levels >= 1 && n_levels <= AC_MAX_LEVELS, "anchor_rans_decode: 1..16 levels");
    AnchorLevels lv;
    for (int l = 0; l < AC_MAX_LEVELS; l++) {
        const gsvc_anchor_level &d = levels_host[l < n_levels ? l : 0];
        GSVC_REQUIRE(d.n >= 0 && d.n_words >= 0 && d.lanes >= 1 && d.lanes <= AC_THREADS * AC_MAX_LPT, "anchor_rans_decode: bad level %d", l);
        GSVC_REQUIRE(d.n == 0 || (d.states && d.freq && d.out && (d.n_words == 0 || d.words)), "anchor_rans_decode: NULL pointer in level %d", l);
        lv.states[l] = d.states; lv.words[l] = d.words; lv.freq[l] = d.freq; lv.out[l] = d.out;
        lv.n[l] = d.n; lv.n_words[l] = d.n_words; lv.lanes[l] = d.lanes;
    }
    hipStream_t s = (hipStream_t)stream;
    ProfScope _prof("k_anchor_rans_decode", s);
    hipLaunchKernelGGL(k_anchor_rans_decode, dim3(n_levels), dim3(AC_THREADS), 0, s, lv, error);
    return check_launch("anchor_rans_decode");
}

extern "C" int gsvc_octree_popcount(const uint8_t *occ, int64_t n, int64_t *counts, void *stream)
{
    GSVC_REQUIRE(n >= 0, "octree_popcount: bad size");
    if (n == 0) return GSVC_OK;
    GSVC_REQUIRE(occ && counts, "octree_popcount: NULL pointer");
    hipLaunchKernelGGL(k_popcount_u8, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, occ, (long long)n, (long long *)counts);
    return check_launch("octree_popcount");
}

extern "C" int gsvc_octree_expand(const int64_t *nodes, const uint8_t *occ, const int64_t *counts_inclusive_scan, int64_t n,
                                  int64_t capacity, int64_t *nodes_out, int32_t *error, void *stream)
{
    GSVC_REQUIRE(n >= 0 && capacity >= 0, "octree_expand: bad size");
    if (n == 0) return GSVC_OK;
    GSVC_REQUIRE(occ && counts_inclusive_scan && nodes_out && error, "octree_expand: NULL pointer");
    hipLaunchKernelGGL(k_octree_expand, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const long long *)nodes, occ,
                       (const long long *)counts_inclusive_scan, (long long)n, (long long)capacity, (long long *)nodes_out, error);
    return check_launch("octree_expand");
}

extern "C" int gsvc_morton_decode(const int64_t *keys, int64_t n, int64_t *xyz, void *stream)
{
    GSVC_REQUIRE(n >= 0, "morton_decode: bad size");
    if (n == 0) return GSVC_OK;
    GSVC_REQUIRE(keys && xyz, "morton_decode: NULL pointer");
    hipLaunchKernelGGL(k_demorton, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const long long *)keys, (long long)n,
                       (long long *)xyz);
    return check_launch("morton_decode");
}

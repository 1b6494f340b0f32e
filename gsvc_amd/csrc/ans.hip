// gsvc_amd/csrc/ans.hip — entropy CODING of quantised anchor attributes with the learned Gaussian model, gfx950.
//
// Replaces the external package the reference imports as gsvc_cuda_ans.ANSCoder (reference README.md:51; used through
// utils/encodings.py:102-245 `encoder_gaussian` / `decoder_gaussian`: symbols in [min, max], one Normal(mu, sigma) per
// symbol with mu = mean / Q, sigma = scale / Q).  Its source is not in the reference tree, so the bitstream is our own;
// what is kept is the interface (integer symbols + per-symbol mu, sigma in, bytes out, and back) and the model:
// P(s) = Phi((s + 1/2 - mu) / sigma) - Phi((s - 1/2 - mu) / sigma), tails folded into min and max.
//
// Coder: range-ANS, 32-bit state in [2^23, 2^31), byte renormalisation, 20-bit probabilities.  The cumulative
// frequency of symbol s is  C(s) = floor(Phi((s - 1/2 - mu) / sigma) * (2^20 - R)) + (s - min),  R = max - min + 1, C(min) = 0,
// C(max + 1) = 2^20: every symbol keeps a frequency >= 1 whatever the model says, and encoder and decoder evaluate the
// same double-precision expression (no table of R entries per symbol).  The symbols are cut into segments of `seg_len`;
// one lane codes one segment sequentially (rANS is serial), segments are independent streams: [4-byte final state]
// [renormalisation bytes in the order the decoder reads them].  K1 codes into a fixed-stride scratch (from the end of
// each slot backwards), K2 packs the segments back to back.  The decoder finds s by bisection on C.
// This is an offline path (the reference runs it on the CPU): sized for correctness and a bounded run time, not tuned.
#include "common.h"

namespace gsvc {

constexpr int ANS_SCALE_BITS = 20;
constexpr uint32_t ANS_M = 1u << ANS_SCALE_BITS;
constexpr uint32_t ANS_L = 1u << 23;
constexpr int ANS_SLOT_BYTES_PER_SYMBOL = 3;     // 20 bits per symbol at most, +8 bytes per segment for the state

__host__ __device__ inline int64_t ans_slot_bytes(int32_t seg_len) { return (int64_t)seg_len * ANS_SLOT_BYTES_PER_SYMBOL + 8; }

// C(s) for s in [min, max + 1]
__device__ __forceinline__ uint32_t ans_cdf(int s, double mu, double sigma, int smin, int smax)
{
#pragma clang fp contract(off)
    if (s <= smin) return 0u;
    if (s > smax) return ANS_M;
    const uint32_t R = (uint32_t)(smax - smin + 1);
    const double z = ((double)s - 0.5 - mu) / sigma;
    double p = 0.5 * erfc(-z * 0.70710678118654752440);
    p = p < 0.0 ? 0.0 : (p > 1.0 ? 1.0 : p);
    if (!(p == p)) p = 0.5;                        // NaN model (sigma = 0 with s - 1/2 == mu): any fixed value works
    uint32_t c = (uint32_t)(p * (double)(ANS_M - R));
    if (c > ANS_M - R) c = ANS_M - R;
    return c + (uint32_t)(s - smin);
}

__global__ void __launch_bounds__(64) k_ans_encode(const int32_t *__restrict__ sym, const float *__restrict__ mu,
                                                   const float *__restrict__ sigma, int64_t n, int smin, int smax, int seg_len,
                                                   int64_t n_seg, uint8_t *__restrict__ scratch, uint32_t *__restrict__ seg_bytes,
                                                   int32_t *__restrict__ error_flag)
{
    const int64_t seg = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (seg >= n_seg) return;
    const int64_t i0 = seg * seg_len;
    const int64_t i1 = i0 + seg_len < n ? i0 + seg_len : n;
    const int64_t slot = ans_slot_bytes(seg_len);
    uint8_t *end = scratch + (seg + 1) * slot;
    uint8_t *p = end;
    uint32_t x = ANS_L;
    for (int64_t i = i1 - 1; i >= i0; i--) {
        const int s = sym[i];
        if (s < smin || s > smax) { atomicExch(error_flag, 1); continue; }
        const double m = (double)mu[i], sg = (double)sigma[i];
        const uint32_t start = ans_cdf(s, m, sg, smin, smax), freq = ans_cdf(s + 1, m, sg, smin, smax) - start;
        if (freq == 0u || freq > ANS_M) { atomicExch(error_flag, 2); continue; }
        const uint64_t x_max = ((uint64_t)(ANS_L >> ANS_SCALE_BITS) << 8) * freq;
        for (int r = 0; r < 4 && (uint64_t)x >= x_max; r++) {     // at most 3 bytes leave per symbol
            *--p = (uint8_t)(x & 0xffu);
            x >>= 8;
        }
        x = ((x / freq) << ANS_SCALE_BITS) + (x % freq) + start;
    }
    p -= 4;
    p[0] = (uint8_t)(x >> 24); p[1] = (uint8_t)(x >> 16); p[2] = (uint8_t)(x >> 8); p[3] = (uint8_t)x;
    seg_bytes[seg] = (uint32_t)(end - p);
}

// exclusive scan of the segment sizes (one workgroup; the codec's segment counts are in the thousands)
__global__ void __launch_bounds__(1024) k_ans_scan(const uint32_t *__restrict__ seg_bytes, int64_t n_seg, uint64_t *__restrict__ seg_offsets)
{
    __shared__ uint64_t part[1024];
    const int tid = threadIdx.x;
    const int64_t per = (n_seg + 1023) / 1024, a = (int64_t)tid * per, b = a + per < n_seg ? a + per : n_seg;
    uint64_t s = 0;
    for (int64_t i = a; i < b; i++) s += seg_bytes[i];
    part[tid] = s;
    __syncthreads();
    if (tid == 0) {
        uint64_t run = 0;
        for (int t = 0; t < 1024; t++) { const uint64_t v = part[t]; part[t] = run; run += v; }
        seg_offsets[n_seg] = run;
    }
    __syncthreads();
    uint64_t run = part[tid];
    for (int64_t i = a; i < b; i++) { seg_offsets[i] = run; run += seg_bytes[i]; }
}

__global__ void __launch_bounds__(256) k_ans_pack(const uint8_t *__restrict__ scratch, const uint32_t *__restrict__ seg_bytes,
                                                  const uint64_t *__restrict__ seg_offsets, int seg_len, int64_t n_seg,
                                                  uint8_t *__restrict__ out)
{
    const int64_t seg = blockIdx.x;
    if (seg >= n_seg) return;
    const int64_t slot = ans_slot_bytes(seg_len);
    const uint32_t nb = seg_bytes[seg];
    const uint8_t *src = scratch + (seg + 1) * slot - nb;
    uint8_t *dst = out + seg_offsets[seg];
    for (uint32_t i = threadIdx.x; i < nb; i += 256) dst[i] = src[i];
}

__global__ void __launch_bounds__(64) k_ans_decode(const uint8_t *__restrict__ bytes, const uint64_t *__restrict__ seg_offsets,
                                                   const float *__restrict__ mu, const float *__restrict__ sigma, int64_t n,
                                                   int smin, int smax, int seg_len, int64_t n_seg, int32_t *__restrict__ sym,
                                                   int32_t *__restrict__ error_flag)
{
    const int64_t seg = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (seg >= n_seg) return;
    const int64_t i0 = seg * seg_len;
    const int64_t i1 = i0 + seg_len < n ? i0 + seg_len : n;
    const uint8_t *p = bytes + seg_offsets[seg], *pend = bytes + seg_offsets[seg + 1];
    if (pend - p < 4) { atomicExch(error_flag, 3); return; }
    uint32_t x = ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | (uint32_t)p[3];
    p += 4;
    for (int64_t i = i0; i < i1; i++) {
        const double m = (double)mu[i], sg = (double)sigma[i];
        const uint32_t slot = x & (ANS_M - 1u);
        // largest s in [min, max] with C(s) <= slot: bisection, at most 32 steps
        int lo = smin, hi = smax;
        for (int it = 0; it < 32 && lo < hi; it++) {
            const int mid = lo + (int)(((int64_t)hi - lo + 1) >> 1);
            if (ans_cdf(mid, m, sg, smin, smax) <= slot) lo = mid; else hi = mid - 1;
        }
        const uint32_t start = ans_cdf(lo, m, sg, smin, smax), freq = ans_cdf(lo + 1, m, sg, smin, smax) - start;
        sym[i] = lo;
        if (freq == 0u || slot < start || slot - start >= freq) { atomicExch(error_flag, 4); return; }
        x = freq * (x >> ANS_SCALE_BITS) + slot - start;
        for (int r = 0; r < 4 && x < ANS_L; r++) {
            if (p >= pend) { atomicExch(error_flag, 5); return; }
            x = (x << 8) | (uint32_t)(*p++);
        }
    }
}

}  // namespace gsvc

using namespace gsvc;

extern "C" int64_t gsvc_ans_segments(int64_t n, int32_t seg_len) { return seg_len > 0 && n > 0 ? (n + seg_len - 1) / seg_len : 0; }

extern "C" int64_t gsvc_ans_scratch_bytes(int64_t n, int32_t seg_len)
{
    return gsvc_ans_segments(n, seg_len) * ans_slot_bytes(seg_len);
}

extern "C" int gsvc_ans_encode(const int32_t *symbols, const float *mu, const float *sigma, int64_t n, int32_t min_symbol,
                               int32_t max_symbol, int32_t seg_len, void *scratch, uint32_t *seg_bytes, uint64_t *seg_offsets,
                               uint8_t *out, int32_t *error_flag, void *stream)
{
    GSVC_REQUIRE(n >= 0 && seg_len > 0 && seg_len <= (1 << 20), "ans_encode: bad sizes");
    GSVC_REQUIRE(max_symbol >= min_symbol && (int64_t)max_symbol - min_symbol + 1 < (int64_t)(ANS_M / 2),
                 "ans_encode: symbol range [%d, %d] does not fit 20-bit frequencies", min_symbol, max_symbol);
    if (n == 0) return GSVC_OK;
    GSVC_REQUIRE(symbols && mu && sigma && scratch && seg_bytes && seg_offsets && out && error_flag, "ans_encode: NULL pointer");
    hipStream_t s = (hipStream_t)stream;
    const int64_t n_seg = gsvc_ans_segments(n, seg_len);
    {
        ProfScope _p("k_ans_encode", s);
        hipLaunchKernelGGL(k_ans_encode, dim3((unsigned)((n_seg + 63) / 64)), dim3(64), 0, s, symbols, mu, sigma, n, min_symbol,
                           max_symbol, seg_len, n_seg, (uint8_t *)scratch, seg_bytes, error_flag);
    }
    hipLaunchKernelGGL(k_ans_scan, dim3(1), dim3(1024), 0, s, seg_bytes, n_seg, seg_offsets);
    hipLaunchKernelGGL(k_ans_pack, dim3((unsigned)n_seg), dim3(256), 0, s, (const uint8_t *)scratch, seg_bytes, seg_offsets, seg_len,
                       n_seg, out);
    return check_launch("ans_encode");
}

extern "C" int gsvc_ans_decode(const uint8_t *bytes, const uint64_t *seg_offsets, const float *mu, const float *sigma, int64_t n,
                               int32_t min_symbol, int32_t max_symbol, int32_t seg_len, int32_t *symbols, int32_t *error_flag,
                               void *stream)
{
    GSVC_REQUIRE(n >= 0 && seg_len > 0 && seg_len <= (1 << 20), "ans_decode: bad sizes");
    GSVC_REQUIRE(max_symbol >= min_symbol && (int64_t)max_symbol - min_symbol + 1 < (int64_t)(ANS_M / 2),
                 "ans_decode: symbol range [%d, %d] does not fit 20-bit frequencies", min_symbol, max_symbol);
    if (n == 0) return GSVC_OK;
    GSVC_REQUIRE(bytes && seg_offsets && mu && sigma && symbols && error_flag, "ans_decode: NULL pointer");
    hipStream_t s = (hipStream_t)stream;
    const int64_t n_seg = gsvc_ans_segments(n, seg_len);
    ProfScope _p("k_ans_decode", s);
    hipLaunchKernelGGL(k_ans_decode, dim3((unsigned)((n_seg + 63) / 64)), dim3(64), 0, s, bytes, seg_offsets, mu, sigma, n, min_symbol,
                       max_symbol, seg_len, n_seg, symbols, error_flag);
    return check_launch("ans_decode");
}

// gsvc_amd/csrc/ans.hip — entropy CODING of quantised anchor attributes with the learned Gaussian model, gfx950.
//
// Replaces the external package the reference imports as gsvc_cuda_ans.ANSCoder (reference README.md:51; used through
// utils/encodings.py:102-245 `encoder_gaussian` / `decoder_gaussian`: symbols in [min, max], one Normal(mu, sigma) per
// symbol with mu = mean / Q, sigma = scale / Q).  Its source is not in the reference tree, so the bitstream is our own;
// what is kept is the interface (integer symbols + per-symbol mu, sigma in, bytes out, and back) and the model:
// P(s) = Phi((s + 1/2 - mu) / sigma) - Phi((s - 1/2 - mu) / sigma), tails folded into min and max.
//
// Coder: range-ANS, 32-bit state in [2^23, 2^31), byte renormalisation, 20-bit probabilities.  The cumulative
// frequency of symbol s is  C(s) = floor(Phi((s - 1/2 - mu) / sigma) * (2^20 - F R)) + F (s - min),  R = max - min + 1, F = ans_floor(R)
// (16 for alphabets up to 512 symbols), C(min) = 0, C(max + 1) = 2^20: every symbol keeps a frequency >= F whatever the model says, and encoder and decoder evaluate the
// same exactly specified expression (Phi from a fixed-point table, see ans_cdf; no table of R entries per symbol).  The symbols are cut into segments of `seg_len`;
// one lane codes one segment sequentially (rANS is serial), segments are independent streams: [4-byte final state]
// [renormalisation bytes in the order the decoder reads them].  K1 codes into a fixed-stride scratch (from the end of
// each slot backwards), K2 packs the segments back to back.  The decoder finds s from a float inverse of the model as a
// first guess, then an exact bracket + bisection on C (ans_decode_symbol).
#include "common.h"

#include <cstdlib>

namespace gsvc {

constexpr int ANS_SCALE_BITS = 20;
constexpr uint32_t ANS_M = 1u << ANS_SCALE_BITS;
constexpr uint32_t ANS_L = 1u << 23;
constexpr int ANS_SLOT_BYTES_PER_SYMBOL = 3;     // 20 bits per symbol at most, +8 bytes per segment for the state

__host__ __device__ inline int64_t ans_slot_bytes(int32_t seg_len) { return (int64_t)seg_len * ANS_SLOT_BYTES_PER_SYMBOL + 8; }

// Phi(z) as a 32-bit fixed-point number from a table of 4097 samples over z in [-8, 8] (step 1/256) and linear
// interpolation in integer arithmetic: every operation between the model's floats and the frequency is an exactly specified
// one (IEEE +, -, *, /, float -> integer conversion, integer multiply / shift), so encoder and decoder agree bit for bit on any
// hardware and any math-library version — a stream no longer depends on how a particular erfc() rounds — and an evaluation
// costs ~20 instructions instead of a double-precision erfc (~600 cycles; the decoder needs 2-4 of them per symbol).  The
// interpolation error (< 5e-7) is below the 2^-20 frequency resolution.  The table sits in LDS (16 KiB per workgroup).
constexpr int PHI_STEPS = 4096;
constexpr double PHI_ZMAX = 8.0;
__device__ uint32_t g_phi_table[PHI_STEPS + 1];

__device__ __forceinline__ void load_phi_table(uint32_t *lds, int tid, int nthreads)
{
    for (int i = tid; i <= PHI_STEPS; i += nthreads) lds[i] = g_phi_table[i];
    __syncthreads();
}

// Frequency floor of every symbol, in units of 2^-20: 16 (= 2^-16, the likelihood floor the rate model is trained and priced with:
// reference utils/entropy_models.py Low_bound, `likelihood >= 2^-16`) while the floors of the whole alphabet take at most 1/128 of the
// probability mass, down to 1 for very wide alphabets.  With a floor of 2^-20 (GSA3) a symbol in the model's far tail cost 20 bits
// where the estimate charges 16: 1-3 % of a fitted model's attribute bytes, 7 % after a long fit (round 5, profiles/r05).
__host__ __device__ inline uint32_t ans_floor(uint32_t R)
{
    const uint32_t f = (ANS_M >> 7) / (R ? R : 1u);
    return f > 16u ? 16u : (f < 1u ? 1u : f);
}

// C(s) for s in [min, max + 1]; inv_sigma = 1.0 / sigma, computed once per symbol by encoder and decoder alike
__device__ __forceinline__ uint32_t ans_cdf(const uint32_t *__restrict__ phi, int s, double mu, double inv_sigma, int smin, int smax)
{
#pragma clang fp contract(off)
    if (s <= smin) return 0u;
    if (s > smax) return ANS_M;
    const uint32_t R = (uint32_t)(smax - smin + 1);
    const double z = ((double)s - 0.5 - mu) * inv_sigma;
    const double t = (z + PHI_ZMAX) * ((double)PHI_STEPS / (2.0 * PHI_ZMAX));
    uint32_t p32;
    if (!(t > 0.0)) p32 = (z != z) ? 0x80000000u : 0u;            // NaN model (sigma = 0 with s - 1/2 == mu): any fixed value
    else if (t >= (double)PHI_STEPS) p32 = 0xffffffffu;
    else {
        const int i = (int)t;
        const uint32_t f = (uint32_t)((t - (double)i) * 65536.0);  // 16-bit fraction
        const uint32_t a = phi[i], b = phi[i + 1];
        p32 = a + (uint32_t)(((uint64_t)(b - a) * f) >> 16);
    }
    const uint32_t F = ans_floor(R);
    const uint32_t c = (uint32_t)(((uint64_t)p32 * (uint64_t)(ANS_M - F * R)) >> 32);
    return c + F * (uint32_t)(s - smin);
}

__global__ void __launch_bounds__(64) k_ans_encode(const int32_t *__restrict__ sym, const float *__restrict__ mu,
                                                   const float *__restrict__ sigma, int64_t n, int smin, int smax, int seg_len,
                                                   int64_t n_seg, uint8_t *__restrict__ scratch, uint32_t *__restrict__ seg_bytes,
                                                   int32_t *__restrict__ error_flag)
{
    __shared__ uint32_t phi[PHI_STEPS + 1];
    load_phi_table(phi, threadIdx.x, 64);
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
        const double m = (double)mu[i], sg = 1.0 / (double)sigma[i];
        const uint32_t start = ans_cdf(phi, s, m, sg, smin, smax), freq = ans_cdf(phi, s + 1, m, sg, smin, smax) - start;
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

struct ByteReader {
    const uint4 *chunk;       // next aligned 16-byte chunk
    uint4 buf;
    int pos;                  // next unread byte of buf
    __device__ __forceinline__ void open(const uint8_t *p)
    {
        const uintptr_t a = reinterpret_cast<uintptr_t>(p);
        chunk = reinterpret_cast<const uint4 *>(a & ~(uintptr_t)15);
        pos = (int)(a & 15);
        buf = *chunk++;
    }
    __device__ __forceinline__ uint32_t next()
    {
        if (pos == 16) { buf = *chunk++; pos = 0; }
        const uint32_t w = pos < 8 ? (pos < 4 ? buf.x : buf.y) : (pos < 12 ? buf.z : buf.w);
        const uint32_t b = (w >> (8 * (pos & 3))) & 0xffu;
        pos++;
        return b;
    }
};

// Slow path of the decoder: the symbol is further than one step from the model's mode s0.  Known on entry: either
// slot >= C(s0 + 2) (`above`, c_edge = C(s0 + 2)) or slot < C(s0 - 1) (c_edge = C(s0 - 1)).  Finds the largest s with
// C(s) <= slot by doubling steps away from that edge, then bisection; for a broad model (sigma > 4 symbols) the first probe is a
// float inverse of the model, s ~ mu + 1/2 + sigma * Phi^-1(slot / 2^20), which usually lands within a few symbols.  The guess
// only orders the search: the result is exact.
__device__ __forceinline__ int ans_decode_symbol(const uint32_t *__restrict__ phi, uint32_t slot, double m, double inv_sg, float sgf,
                                                 int smin, int smax, int s0, bool above, uint32_t c_edge, uint32_t &start,
                                                 uint32_t &freq)
{
    int lo, hi;
    uint32_t c_lo, c_hi1;      // C(lo), C(hi + 1) of the current bracket: C(lo) <= slot < C(hi + 1)
    int guess = 0;
    const bool use_guess = sgf > 4.0f;
    if (use_guess) {
        const float qf = ((float)slot + 0.5f) * (1.0f / (float)ANS_M);
        float g = (float)m + 0.5f + sgf * normcdfinvf(qf);
        g = g == g ? fminf(fmaxf(g, (float)smin), (float)smax) : (float)smin;
        guess = (int)floorf(g);
    }
    if (above) {                           // answer >= s0 + 2: gallop upwards
        lo = s0 + 2; c_lo = c_edge;
        if (use_guess && guess > lo) {
            const uint32_t c = ans_cdf(phi, guess, m, inv_sg, smin, smax);
            if (c <= slot) { lo = guess; c_lo = c; }
        }
        int64_t step = 1;
        for (;;) {
            const int64_t t = (int64_t)lo + step;
            const int probe = t > (int64_t)smax + 1 ? smax + 1 : (int)t;
            const uint32_t c = ans_cdf(phi, probe, m, inv_sg, smin, smax);       // C(max + 1) = 2^20 > slot ends the loop
            if (c > slot) { hi = probe - 1; c_hi1 = c; break; }
            lo = probe; c_lo = c; step <<= 1;
        }
    } else {                               // answer <= s0 - 2: gallop downwards (C(min) = 0 <= slot ends the loop)
        hi = s0 - 2; c_hi1 = c_edge;
        if (use_guess && guess < hi) {
            const uint32_t c = ans_cdf(phi, guess + 1, m, inv_sg, smin, smax);
            if (c > slot) { hi = guess; c_hi1 = c; }
        }
        int64_t step = 1;
        for (;;) {
            const int64_t t = (int64_t)hi + 1 - step;
            const int probe = t < (int64_t)smin ? smin : (int)t;
            const uint32_t c = ans_cdf(phi, probe, m, inv_sg, smin, smax);
            if (c <= slot) { lo = probe; c_lo = c; break; }
            hi = probe - 1; c_hi1 = c; step <<= 1;
        }
    }
    for (int it = 0; it < 32 && lo < hi; it++) {
        const int mid = lo + (int)(((int64_t)hi - lo + 1) >> 1);
        const uint32_t c = ans_cdf(phi, mid, m, inv_sg, smin, smax);
        if (c <= slot) { lo = mid; c_lo = c; } else { hi = mid - 1; c_hi1 = c; }
    }
    start = c_lo;
    freq = c_hi1 - c_lo;
    return lo;
}

__device__ __forceinline__ int ans_mode(float m, int smin, int smax)
{
    float g = floorf(m + 0.5f);
    g = g == g ? fminf(fmaxf(g, (float)smin), (float)smax) : (float)smin;
    return (int)g;
}

// Per symbol, from the model alone (no coder state): the cumulative frequencies around the model's mode s0 = round(mu),
// (C(s0 - 1), C(s0), C(s0 + 1), C(s0 + 2)), written together with (mu, sigma) as one 32-byte record in the order the decode
// kernel's lanes walk: record of symbol t of segment g at [(g / 64) * seg_len + t] * 64 + g % 64, so that the 64 lanes of a
// decode wave — one segment each, 16 KB apart in the model arrays — read ONE contiguous 2-KB piece per step.  Embarrassingly
// parallel (one lane per symbol over the whole chip) while the decode kernel has one lane per SEGMENT, and a wave of it pays for
// its slowest lane at every step: with the three symbols around the mode decided by compares, the search (hundreds of cycles)
// runs only for symbols further out — rare in a fitted model, where a wave would otherwise hit it at most steps through one lane
// or another.
__global__ void __launch_bounds__(256) k_ans_model(const float *__restrict__ mu, const float *__restrict__ sigma, int64_t n, int smin,
                                                   int smax, int seg_len, uint4 *__restrict__ records)
{
    __shared__ uint32_t phi[PHI_STEPS + 1];
    load_phi_table(phi, threadIdx.x, 256);
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float m = mu[i], sg = sigma[i];
    const double inv = 1.0 / (double)sg, md = (double)m;
    const int s0 = ans_mode(m, smin, smax);
    const int64_t g = i / seg_len, t = i - g * seg_len;
    uint4 *dst = records + (((g >> 6) * seg_len + t) * 64 + (g & 63)) * 2;
    dst[0] = make_uint4(ans_cdf(phi, s0 - 1, md, inv, smin, smax), ans_cdf(phi, s0, md, inv, smin, smax),
                        ans_cdf(phi, s0 + 1, md, inv, smin, smax), ans_cdf(phi, s0 + 2, md, inv, smin, smax));
    dst[1] = make_uint4(__float_as_uint(m), __float_as_uint(sg), 0u, 0u);
}

constexpr int ANS_AHEAD = 16;     // records a decode lane keeps in flight (the stream is coalesced, its latency is not short)

// One lane decodes one segment.  Everything a lane reads from memory is fetched ahead of its use: the byte stream through a
// 16-byte register buffer (a renormalisation byte used to be a dependent global load of ~1 us each with the few waves a
// decode launch has), the model records ANS_AHEAD symbols ahead (each record register is reloaded as soon as it is consumed).
__device__ __forceinline__ void ans_decode_body(const uint32_t *__restrict__ phi, const uint8_t *__restrict__ bytes,
                                                const uint64_t *__restrict__ seg_offsets, int64_t n, int smin, int smax, int seg_len,
                                                int64_t n_seg, int32_t *__restrict__ sym, int32_t *__restrict__ error_flag,
                                                const uint4 *__restrict__ records, int64_t block)
{
    const int64_t seg = block * 64 + threadIdx.x;
    if (seg >= n_seg) return;
    const int64_t i0 = seg * seg_len;
    const int len = (int)((i0 + seg_len < n ? i0 + seg_len : n) - i0);
    const uint64_t b0 = seg_offsets[seg], b1 = seg_offsets[seg + 1];
    if (b1 < b0 + 4) { atomicExch(error_flag, 3); return; }
    int64_t left = (int64_t)(b1 - b0) - 4;          // renormalisation bytes this segment still holds
    ByteReader rd;
    rd.open(bytes + b0);
    uint32_t x = rd.next() << 24;
    x |= rd.next() << 16;
    x |= rd.next() << 8;
    x |= rd.next();
    const uint4 *rec = records + (block * seg_len * 64 + threadIdx.x) * 2;      // + 128 per symbol
    const uint4 zero = make_uint4(0u, 0u, 0u, 0u);
    uint4 ra[ANS_AHEAD], rb[ANS_AHEAD];
#pragma unroll
    for (int k = 0; k < ANS_AHEAD; k++) {
        ra[k] = k < len ? rec[(int64_t)k * 128] : zero;
        rb[k] = k < len ? rec[(int64_t)k * 128 + 1] : zero;
    }
    for (int t0 = 0; t0 < len; t0 += ANS_AHEAD) {
        bool ok = true;
#pragma unroll
        for (int k = 0; k < ANS_AHEAD; k++) {
            if (ok && t0 + k < len) {
                const uint4 c = ra[k];
                const float m = __uint_as_float(rb[k].x), sg = __uint_as_float(rb[k].y);
                if (t0 + ANS_AHEAD + k < len) {          // this register pair is free again: the record ANS_AHEAD steps ahead
                    ra[k] = rec[(int64_t)(t0 + ANS_AHEAD + k) * 128];
                    rb[k] = rec[(int64_t)(t0 + ANS_AHEAD + k) * 128 + 1];
                }
                const int s0 = ans_mode(m, smin, smax);
                const uint32_t slot = x & (ANS_M - 1u);
                uint32_t start, freq;
                int out;
                if (slot >= c.y && slot < c.z) { out = s0; start = c.y; freq = c.z - c.y; }
                else if (slot >= c.z && slot < c.w) { out = s0 + 1; start = c.z; freq = c.w - c.z; }
                else if (slot >= c.x && slot < c.y) { out = s0 - 1; start = c.x; freq = c.y - c.x; }
                else {
                    const bool above = slot >= c.w;
                    out = ans_decode_symbol(phi, slot, (double)m, 1.0 / (double)sg, sg, smin, smax, s0, above, above ? c.w : c.x, start, freq);
                }
                sym[i0 + t0 + k] = out;
                if (freq == 0u || slot < start || slot - start >= freq) {
                    atomicExch(error_flag, 4);
                    ok = false;
                } else {
                    x = freq * (x >> ANS_SCALE_BITS) + slot - start;
                    for (int r = 0; r < 4 && x < ANS_L; r++) {
                        if (--left < 0) { atomicExch(error_flag, 5); ok = false; break; }
                        x = (x << 8) | rd.next();
                    }
                }
            }
        }
        if (!ok) return;
    }
}

__global__ void __launch_bounds__(64) k_ans_decode(const uint8_t *__restrict__ bytes, const uint64_t *__restrict__ seg_offsets,
                                                   int64_t n, int smin, int smax, int seg_len, int64_t n_seg,
                                                   int32_t *__restrict__ sym, int32_t *__restrict__ error_flag,
                                                   const uint4 *__restrict__ records)
{
    __shared__ uint32_t phi[PHI_STEPS + 1];
    load_phi_table(phi, threadIdx.x, 64);
    ans_decode_body(phi, bytes, seg_offsets, n, smin, smax, seg_len, n_seg, sym, error_flag, records, blockIdx.x);
}

// Several streams in ONE launch (a coded model is dozens of streams, each a handful of waves: launched one by one they run
// one after another — streams of different HIP queues share hardware queues with whatever else the caller has queued — and
// every launch lasts as long as its slowest lane's 4 096 serial symbols, ~1.8 ms; together they fill the chip for that long once).
constexpr int ANS_MAX_JOBS = 16;
struct AnsJob {
    const uint8_t *bytes;
    const uint64_t *seg_offsets;
    const float *mu, *sigma;
    int32_t *sym, *err;
    uint4 *records;
    long long n, n_seg;
    int smin, smax, seg_len, first_block, first_model_block, pad;
};
struct AnsJobs {
    AnsJob j[ANS_MAX_JOBS];
    int n;
};

__global__ void __launch_bounds__(64) k_ans_decode_many(AnsJobs t)
{
    __shared__ uint32_t phi[PHI_STEPS + 1];
    load_phi_table(phi, threadIdx.x, 64);
    int k = 0;
    while (k + 1 < t.n && (int)blockIdx.x >= t.j[k + 1].first_block) k++;
    const AnsJob &q = t.j[k];
    ans_decode_body(phi, q.bytes, q.seg_offsets, q.n, q.smin, q.smax, q.seg_len, q.n_seg, q.sym, q.err, q.records,
                    (int64_t)blockIdx.x - q.first_block);
}

__global__ void __launch_bounds__(256) k_ans_model_many(AnsJobs t)
{
    __shared__ uint32_t phi[PHI_STEPS + 1];
    load_phi_table(phi, threadIdx.x, 256);
    int k = 0;
    while (k + 1 < t.n && (int)blockIdx.x >= t.j[k + 1].first_model_block) k++;
    const AnsJob &q = t.j[k];
    const int64_t i = ((int64_t)blockIdx.x - q.first_model_block) * 256 + threadIdx.x;
    if (i >= q.n) return;
    const float m = q.mu[i], sg = q.sigma[i];
    const double inv = 1.0 / (double)sg, md = (double)m;
    const int s0 = ans_mode(m, q.smin, q.smax);
    const int64_t g = i / q.seg_len, tt = i - g * q.seg_len;
    uint4 *dst = q.records + (((g >> 6) * q.seg_len + tt) * 64 + (g & 63)) * 2;
    dst[0] = make_uint4(ans_cdf(phi, s0 - 1, md, inv, q.smin, q.smax), ans_cdf(phi, s0, md, inv, q.smin, q.smax),
                        ans_cdf(phi, s0 + 1, md, inv, q.smin, q.smax), ans_cdf(phi, s0 + 2, md, inv, q.smin, q.smax));
    dst[1] = make_uint4(__float_as_uint(m), __float_as_uint(sg), 0u, 0u);
}

}  // namespace gsvc

using namespace gsvc;

#include <cmath>
#include <mutex>

// The table is filled once per process from the host's double-precision erfc, rounded to 32-bit fixed point.  Two platforms
// whose erfc differ in the last place could round one of the 4097 entries differently, so a stream carries the table's
// checksum (gsvc_ans_table_checksum) and the decoder compares it with its own before decoding.
static uint32_t g_phi_host[PHI_STEPS + 1];
static uint32_t g_phi_crc = 0;
static int ensure_phi_table()
{
    static std::once_flag once;
    static int rc = GSVC_OK;
    std::call_once(once, [] {
        uint32_t h = 2166136261u;
        for (int i = 0; i <= PHI_STEPS; i++) {
            const double z = -PHI_ZMAX + (double)i * (2.0 * PHI_ZMAX / (double)PHI_STEPS);
            double p = 0.5 * std::erfc(-z * 0.70710678118654752440) * 4294967296.0;
            p = std::floor(p + 0.5);
            g_phi_host[i] = p >= 4294967295.0 ? 0xffffffffu : (p <= 0.0 ? 0u : (uint32_t)p);
            if (i > 0 && g_phi_host[i] < g_phi_host[i - 1]) g_phi_host[i] = g_phi_host[i - 1];     // monotone by construction
            for (int b = 0; b < 4; b++) h = (h ^ ((g_phi_host[i] >> (8 * b)) & 0xffu)) * 16777619u;   // FNV-1a
        }
        g_phi_crc = h;
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_phi_table), g_phi_host, sizeof(g_phi_host)) != hipSuccess) {
            set_error("ans: uploading the Phi table failed");
            rc = GSVC_E_LAUNCH;
        }
    });
    return rc;
}

extern "C" uint32_t gsvc_ans_table_checksum(void)
{
    ensure_phi_table();
    return g_phi_crc;
}

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
    if (int rc = ensure_phi_table()) return rc;
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

extern "C" int64_t gsvc_ans_decode_scratch_bytes(int64_t n, int32_t seg_len)
{
    // one 32-byte record per symbol, laid out per group of 64 segments (the last group is padded to 64 segments)
    const int64_t n_seg = gsvc_ans_segments(n, seg_len);
    return ((n_seg + 63) / 64) * 64 * (int64_t)(seg_len > 0 ? seg_len : 1) * 2 * (int64_t)sizeof(uint4) + 16;
}

extern "C" int gsvc_ans_decode(const uint8_t *bytes, const uint64_t *seg_offsets, const float *mu, const float *sigma, int64_t n,
                               int32_t min_symbol, int32_t max_symbol, int32_t seg_len, int32_t *symbols, int32_t *error_flag,
                               void *scratch, void *stream)
{
    GSVC_REQUIRE(n >= 0 && seg_len > 0 && seg_len <= (1 << 20), "ans_decode: bad sizes");
    GSVC_REQUIRE(max_symbol >= min_symbol && (int64_t)max_symbol - min_symbol + 1 < (int64_t)(ANS_M / 2),
                 "ans_decode: symbol range [%d, %d] does not fit 20-bit frequencies", min_symbol, max_symbol);
    if (n == 0) return GSVC_OK;
    GSVC_REQUIRE(bytes && seg_offsets && mu && sigma && symbols && error_flag && scratch, "ans_decode: NULL pointer");
    GSVC_REQUIRE((reinterpret_cast<uintptr_t>(scratch) & 15) == 0, "ans_decode: scratch must be 16-byte aligned");
    if (int rc = ensure_phi_table()) return rc;
    hipStream_t s = (hipStream_t)stream;
    const int64_t n_seg = gsvc_ans_segments(n, seg_len);
    {
        ProfScope _p("k_ans_model", s);
        hipLaunchKernelGGL(k_ans_model, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, mu, sigma, n, min_symbol, max_symbol, seg_len,
                           (uint4 *)scratch);
    }
    ProfScope _p("k_ans_decode", s);
    hipLaunchKernelGGL(k_ans_decode, dim3((unsigned)((n_seg + 63) / 64)), dim3(64), 0, s, bytes, seg_offsets, n, min_symbol, max_symbol,
                       seg_len, n_seg, symbols, error_flag, (const uint4 *)scratch);
    return check_launch("ans_decode");
}

extern "C" int gsvc_ans_decode_many(const gsvc_ans_decode_job *jobs, int32_t n_jobs, void *stream)
{
    GSVC_REQUIRE(jobs && n_jobs >= 0, "ans_decode_many: bad arguments");
    if (int rc = ensure_phi_table()) return rc;
    hipStream_t s = (hipStream_t)stream;
    for (int j0 = 0; j0 < n_jobs;) {
        AnsJobs t;
        t.n = 0;
        long long blocks = 0, mblocks = 0;
        int j = j0;
        // a batch is the next ANS_MAX_JOBS NON-EMPTY jobs: the next batch starts behind the last index this one consumed
        // (empty jobs are skipped without taking a slot; advancing by ANS_MAX_JOBS decoded the surplus jobs a second time)
        for (; j < n_jobs && t.n < ANS_MAX_JOBS; j++) {
            const gsvc_ans_decode_job &d = jobs[j];
            GSVC_REQUIRE(d.n >= 0 && d.seg_len > 0 && d.seg_len <= (1 << 20), "ans_decode_many: bad sizes (job %d)", j);
            GSVC_REQUIRE(d.max_symbol >= d.min_symbol && (int64_t)d.max_symbol - d.min_symbol + 1 < (int64_t)(ANS_M / 2),
                         "ans_decode_many: symbol range [%d, %d] does not fit 20-bit frequencies", d.min_symbol, d.max_symbol);
            if (d.n == 0) continue;
            GSVC_REQUIRE(d.bytes && d.seg_offsets && d.mu && d.sigma && d.symbols && d.error_flag && d.scratch,
                         "ans_decode_many: NULL pointer (job %d)", j);
            GSVC_REQUIRE((reinterpret_cast<uintptr_t>(d.scratch) & 15) == 0, "ans_decode_many: scratch must be 16-byte aligned");
            AnsJob &q = t.j[t.n++];
            q.bytes = d.bytes; q.seg_offsets = d.seg_offsets; q.mu = d.mu; q.sigma = d.sigma; q.sym = d.symbols; q.err = d.error_flag;
            q.records = (uint4 *)d.scratch; q.n = d.n; q.n_seg = gsvc_ans_segments(d.n, d.seg_len);
            q.smin = d.min_symbol; q.smax = d.max_symbol; q.seg_len = d.seg_len; q.pad = 0;
            q.first_block = (int)blocks; q.first_model_block = (int)mblocks;
            blocks += (q.n_seg + 63) / 64;
            mblocks += (q.n + 255) / 256;
            GSVC_REQUIRE(blocks < (1ll << 31) && mblocks < (1ll << 31), "ans_decode_many: too many symbols for one launch");
        }
        j0 = j;
        if (t.n == 0) continue;
        {
            ProfScope _p("k_ans_model", s);
            hipLaunchKernelGGL(k_ans_model_many, dim3((unsigned)mblocks), dim3(256), 0, s, t);
        }
        ProfScope _p("k_ans_decode", s);
        hipLaunchKernelGGL(k_ans_decode_many, dim3((unsigned)blocks), dim3(64), 0, s, t);
    }
    return check_launch("ans_decode_many");
}

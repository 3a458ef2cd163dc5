// Exact squared Euclidean distance transform (R2) and the operators that are
// thresholds of it: disk(r) dilation (A6) and fill_particle_area (A8).
//
// Separable, integers only.  The input is read ONCE and reduced to one bit per
// pixel (32-row column words); vertical distances come from bit scans of those
// words plus a per-column carry across words, are staged per 8-row block in LDS
// as uint16, and the horizontal pass is an expanding search that stops as soon
// as k*k >= best (exact: the candidate at offset k is >= k*k).  HBM traffic is
// the input read plus the int32 (or uint8) result write.
//
// The THRESHOLD operators with a reach of at most 31 pixels (every call of the reference: disk(2), disk(20), distance < 2) never
// form a distance: reach_bits_kernel dilates the bit words themselves (see there); the row-block pass with its prefix scans
// (edt_reach_kernel) serves larger radii.
#include <type_traits>

#include "common.h"

namespace pcseg {

constexpr int EDT_CH = 32;          // rows per bit word
#ifndef PCSEG_EDT_REACH_ROWS
#define PCSEG_EDT_REACH_ROWS 8
#endif
#ifndef PCSEG_EDT_REACH_BITS
#define PCSEG_EDT_REACH_BITS 1  // 0 (A/B builds): every threshold epilogue takes the row-block pass (edt_reach_kernel)
#endif
#ifndef PCSEG_EDT_REACH_OCC
#define PCSEG_EDT_REACH_OCC 1
#endif
constexpr int EDT_RB = PCSEG_EDT_REACH_ROWS;  // rows per block of the threshold pass (a divisor of the 32-row word)
static_assert(EDT_CH % EDT_RB == 0, "a block's rows lie in one bit word");
constexpr unsigned G_INF = 0xFFFFu;  // "no zero pixel in this column"
constexpr int EDT_STAGE_TRIPS = 4;   // column words a thread fetches as one batch when it stages a row block

// ---- foreground predicates (distance is measured TO the nearest non-foreground pixel)
struct FgNzU8 {
    static constexpr bool kBytes = true;  // one input byte per pixel, no side output: eligible for 4-column loads
    const uint8_t *p;
    typedef uint8_t raw_t;
    __device__ __forceinline__ bool byte(unsigned v) const { return v != 0; }
    __device__ __forceinline__ raw_t raw(int b, int64_t pix, int64_t n) const { return p[b * n + pix]; }
    __device__ __forceinline__ bool from_raw(raw_t v, int, int64_t, int64_t) const { return v != 0; }
    __device__ __forceinline__ bool operator()(int b, int64_t pix, int64_t n) const { return p[b * n + pix] != 0; }
};
struct FgLtF32 {
    static constexpr bool kBytes = false;
    const float *p;
    float thr;
    uint8_t *mask_out;
    int64_t frame_stride;  // elements between frames of p (a plane inside a (B,C,H,W) stack)
    typedef float raw_t;
    // load and test are separate so that a thread can issue all the loads of its word before the first side store
    // (a byte store may alias anything: the compiler will not move a later load above it)
    __device__ __forceinline__ raw_t raw(int b, int64_t pix, int64_t) const { return p[b * frame_stride + pix]; }
    __device__ __forceinline__ bool from_raw(raw_t v, int b, int64_t pix, int64_t n) const
    {
        const bool m = v < thr;
        if (mask_out) mask_out[b * n + pix] = m;
        return m;
    }
    __device__ __forceinline__ bool operator()(int b, int64_t pix, int64_t n) const { return from_raw(raw(b, pix, n), b, pix, n); }
};
struct FgNotInSetU8 {
    static constexpr bool kBytes = true;
    const uint8_t *p;
    unsigned long long bits;
    typedef uint8_t raw_t;
    __device__ __forceinline__ bool byte(unsigned v) const { return !(v < 64 && ((bits >> v) & 1ull)); }
    __device__ __forceinline__ raw_t raw(int b, int64_t pix, int64_t n) const { return p[b * n + pix]; }
    __device__ __forceinline__ bool from_raw(raw_t v, int, int64_t, int64_t) const { return byte(v); }
    __device__ __forceinline__ bool operator()(int b, int64_t pix, int64_t n) const { return byte(p[b * n + pix]); }
};

// one thread per (word, column): fg bits of 32 rows
// (launch bounds: 94 scalar registers = seven workgroups per CU without the second argument, 78 = eight with it)
// any_bg (may be null): any_bg[b] = 1 if the frame has a zero pixel at all -- one plain store per wave that saw one
template <typename Fg>
__global__ void __launch_bounds__(256, 8) edt_bits_kernel(Fg fg, unsigned *__restrict__ bits, int *__restrict__ any_bg, int H, int W, int nch)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    const int ch = blockIdx.y, b = blockIdx.z;
    if (c >= W) return;
    const int64_t n = (int64_t)H * W;
    const int r0 = ch * EDT_CH;
    unsigned word = 0;
    if (r0 + EDT_CH <= H) {  // full word: no row test around the loads, all 32 of them in flight together
        typename Fg::raw_t raw[EDT_CH];
#pragma unroll
        for (int j = 0; j < EDT_CH; ++j) raw[j] = fg.raw(b, rowoff((r0 + j), W) + c, n);
#pragma unroll
        for (int j = 0; j < EDT_CH; ++j)
            if (fg.from_raw(raw[j], b, rowoff((r0 + j), W) + c, n)) word |= 1u << j;
    } else {
#pragma unroll 8
        for (int j = 0; j < EDT_CH; ++j) {
            int r = r0 + j;
            if (r < H && fg(b, rowoff(r, W) + c, n)) word |= 1u << j;
        }
    }
    bits[((int64_t)b * nch + ch) * W + c] = word;
    if (any_bg) {
        const int rows = min(EDT_CH, H - r0);
        const bool zero = (~word & (rows >= 32 ? 0xFFFFFFFFu : ((1u << rows) - 1u))) != 0;
        if (__any(zero) && __ffsll((long long)__ballot(true)) - 1 == lane_id()) any_bg[b] = 1;
    }
}

// byte inputs with W % 4 == 0: a lane takes four adjacent columns (one 4-byte load per row, one 16-byte store)
template <typename Fg>
__global__ void __launch_bounds__(256) edt_bits4_kernel(Fg fg, unsigned *__restrict__ bits, int *__restrict__ any_bg, int H, int W, int nch)
{
    const int c = (blockIdx.x * 256 + threadIdx.x) * 4;
    const int ch = blockIdx.y, b = blockIdx.z;
    if (c >= W) return;
    const uint8_t *src = fg.p + (int64_t)b * H * W + c;
    const int r0 = ch * EDT_CH;
    unsigned w0 = 0, w1 = 0, w2 = 0, w3 = 0;
    // all 32 rows' loads in flight together (a row test around each load made them 32 dependent round trips): rows past the
    // frame's end re-read its last row and are masked out of the words
    unsigned v[EDT_CH];
#pragma unroll
    for (int j = 0; j < EDT_CH; ++j) v[j] = *reinterpret_cast<const unsigned *>(src + rowoff(min(r0 + j, H - 1), W));
#pragma unroll
    for (int j = 0; j < EDT_CH; ++j) {
        if (fg.byte(v[j] & 255u)) w0 |= 1u << j;
        if (fg.byte((v[j] >> 8) & 255u)) w1 |= 1u << j;
        if (fg.byte((v[j] >> 16) & 255u)) w2 |= 1u << j;
        if (fg.byte(v[j] >> 24)) w3 |= 1u << j;
    }
    const int rows = min(EDT_CH, H - r0);
    const unsigned valid = rows >= 32 ? 0xFFFFFFFFu : ((1u << rows) - 1u);
    w0 &= valid; w1 &= valid; w2 &= valid; w3 &= valid;
    *reinterpret_cast<uint4 *>(bits + ((int64_t)b * nch + ch) * W + c) = make_uint4(w0, w1, w2, w3);
    if (any_bg) {
        const bool zero = ((~w0 | ~w1 | ~w2 | ~w3) & valid) != 0;
        if (__any(zero) && __ffsll((long long)__ballot(true)) - 1 == lane_id()) any_bg[b] = 1;
    }
}

// per column: distance from the first row of each word to the nearest zero above it (up),
// and from the last row of each word to the nearest zero below it (dn); G_INF if none.
// any_bg[b] = 1 if the frame has a zero pixel at all: one plain store per block (a flag that every thread of the bit
// pass touches would make all of them queue on one cache line).
__global__ void __launch_bounds__(256) edt_carry_kernel(const unsigned *__restrict__ bits, uint16_t *__restrict__ up,
                                                         uint16_t *__restrict__ dn, int *__restrict__ any_bg, int H, int W, int nch)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    const int64_t base = (int64_t)b * nch * W + c;
    bool has_zero = false;
    unsigned d = G_INF;  // distance from row (r0 - 1) ... tracked as "distance of first row of the word to the zero"
    // (the words of a column are loaded eight at a time -- the carry depends on the previous word, the LOADS do not; one
    // load per step of the carry was a memory round trip per word with a quarter of the chip's wave slots in use)
    constexpr int CB = 8;
    for (int ch0 = 0; c < W && ch0 < nch; ch0 += CB) {
        unsigned wv[CB];
#pragma unroll
        for (int k = 0; k < CB; ++k) wv[k] = bits[base + rowoff(min(ch0 + k, nch - 1), W)];
#pragma unroll
        for (int k = 0; k < CB; ++k) {
            const int ch = ch0 + k;
            if (ch >= nch) break;
            up[base + rowoff(ch, W)] = (uint16_t)d;
            int rows = min(EDT_CH, H - ch * EDT_CH);
            unsigned valid = rows == 32 ? 0xFFFFFFFFu : ((1u << rows) - 1u);
            unsigned zero = ~wv[k] & valid;
            if (zero) { d = rows - (31 - __clz(zero)); has_zero = true; }  // from first row of next word to the last zero of this word
            else d = d == G_INF ? G_INF : d + rows;
        }
    }
    if (__syncthreads_or(has_zero) && threadIdx.x == 0) any_bg[b] = 1;
    d = G_INF;
    for (int ch0 = nch - 1; c < W && ch0 >= 0; ch0 -= CB) {
        unsigned wv[CB];
#pragma unroll
        for (int k = 0; k < CB; ++k) wv[k] = bits[base + rowoff(max(ch0 - k, 0), W)];
#pragma unroll
        for (int k = 0; k < CB; ++k) {
            const int ch = ch0 - k;
            if (ch < 0) break;
            dn[base + rowoff(ch, W)] = (uint16_t)d;
            int rows = min(EDT_CH, H - ch * EDT_CH);
            unsigned valid = rows == 32 ? 0xFFFFFFFFu : ((1u << rows) - 1u);
            unsigned zero = ~wv[k] & valid;
            if (zero) d = (__ffs(zero) - 1) + 1;  // from last row of previous word to the first zero of this word
            else d = d == G_INF ? G_INF : d + rows;
        }
    }
}

// vertical distance of row j (0..31) of a word
__device__ __forceinline__ unsigned vdist(unsigned word, unsigned valid, int j, unsigned up, unsigned dn, int rows)
{
    if (!((word >> j) & 1u)) return 0;
    unsigned zero = ~word & valid;
    unsigned below_or_at = zero & (j == 31 ? 0xFFFFFFFFu : ((2u << j) - 1u));  // zeros in rows <= j
    unsigned above = zero & ~(j == 31 ? 0xFFFFFFFFu : ((2u << j) - 1u));        // zeros in rows > j
    unsigned d1 = below_or_at ? (unsigned)(j - (31 - __clz(below_or_at))) : (up == G_INF ? G_INF : up + j);
    unsigned d2 = above ? (unsigned)((__ffs(above) - 1) - j) : (dn == G_INF ? G_INF : dn + (rows - 1 - j));
    return min(min(d1, d2), G_INF);
}

// ---- epilogues of the horizontal pass
struct EpiD2 {
    static constexpr bool kThreshold = false;
    int *d2;
    int cap;  // < 0: exact
    __device__ __forceinline__ int kmax(int W) const { return cap < 0 ? W : (int)sqrtf((float)cap) + 1; }
    __device__ __forceinline__ void store(int64_t i, long long v, bool any_bg, int r, int c, unsigned long long &cnt) const
    {
        if (!any_bg) v = (long long)(r + 1) * (r + 1) + (long long)c * c;  // scipy: virtual zero pixel at (-1, 0)
        if (cap >= 0 && v > cap) v = (long long)cap + 1;
        d2[i] = (int)v;
    }
};
// threshold epilogues only need "is there a zero pixel within sqrt(R2)": they run on edt_reach_kernel
// The threshold epilogues work on a row block staged in LDS: `input(i)` (if kInput) is fetched for the whole block as
// one batch of 4-byte loads, `decide` turns a staged byte and the verdict "a zero pixel is within reach" into the output
// byte in LDS, and the block writes its rows back with 4-byte stores -- no global load sits between two steps of a scan.
struct EpiDilate {
    static constexpr bool kThreshold = true;
    static constexpr bool kInput = false;
    uint8_t *out;
    int r2;
    __host__ __device__ __forceinline__ int R2() const { return r2; }
    __device__ __forceinline__ const uint8_t *input() const { return nullptr; }
    __device__ __forceinline__ uint8_t decide(uint8_t, bool within, bool any_bg, int, int, unsigned long long &) const
    {
        return (any_bg && within) ? 1 : 0;  // empty set dilates to the empty set
    }
    // four pixels at once (a frame WITH a zero pixel): w4 = the verdicts, 0 / 1 per byte
    __device__ __forceinline__ unsigned decide4(unsigned, unsigned w4, unsigned long long &) const { return w4; }
};
struct EpiFillParticle {
    static constexpr bool kThreshold = true;
    static constexpr bool kInput = true;
    const uint8_t *ds;
    uint8_t *out;
    int cell_label, overlap_label, r2, thr2;  // overlap if d2 <= r2 (dilation) or d2 < thr2 (distance)
    __host__ __device__ __forceinline__ int R2() const { return max(r2, thr2 - 1); }
    __device__ __forceinline__ const uint8_t *input() const { return ds; }
    __device__ __forceinline__ uint8_t decide(uint8_t z, bool within, bool any_bg, int r, int c, unsigned long long &cnt) const
    {
        if (z == cell_label) {
            bool ov;
            if (any_bg) ov = within;
            else ov = ((long long)(r + 1) * (r + 1) + (long long)c * c) < thr2;  // only the EDT term sees the virtual pixel
            if (ov) { z = (uint8_t)overlap_label; ++cnt; }
        }
        return z;
    }
    // four pixels at once (a frame WITH a zero pixel): z4 = the input bytes, w4 = the verdicts, 0 / 1 per byte
    __device__ __forceinline__ unsigned decide4(unsigned z4, unsigned w4, unsigned long long &cnt) const
    {
        const unsigned x = z4 ^ (0x01010101u * (unsigned)cell_label);
        unsigned nz = (x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu;  // bit 7 of a byte: its low seven bits are not all zero (no carry leaves a byte)
        nz = ~(nz | x | 0x7F7F7F7Fu);                   // 0x80 exactly where the byte of x is zero, i.e. the pixel is a cell pixel
        const unsigned hit = (nz >> 7) & w4;
        cnt += __popc(hit);
        return z4 ^ (hit * (unsigned)(cell_label ^ overlap_label));  // a hit byte holds cell_label: it becomes overlap_label
    }
};

// Horizontal pass for the distance itself.  The block stages the SQUARED vertical distances of its rows in LDS as
// uint32 (EDT_D2_INF where the column has no zero pixel; one such guard cell either side of a row, so the search needs
// no bounds tests, only clamped indices); a candidate then costs an add and a min.  The search expands four offsets per
// trip (the 8 LDS reads are issued together) and stops as soon as k^2 >= best, which is exact: offsets past the exit
// point are still true candidates (g^2 + k^2 of a real pixel), and no closer one is left.
constexpr unsigned EDT_D2_INF = 0x40000000u;  // > any g^2 (g <= 32767); g^2 + k^2 stays below 2^32

// guard cells either side of a staged row: the search reads five neighbouring cells from a CLAMPED base address without any
// test (see the kernel).  (Round 3 had one guard cell and a branch per trip that clamped ten indices one by one near a row end,
// and a division per pixel pair: 296 us a launch against 224, profiles/r04/ab_logs/r4k_*.)
constexpr int EDT_GUARD = 4;

template <typename Epi, int RB>
__global__ void __launch_bounds__(256) edt_row_kernel(const unsigned *__restrict__ bits, const uint16_t *__restrict__ up,
                                                       const uint16_t *__restrict__ dn, const int *__restrict__ any_bg,
                                                       Epi epi, unsigned long long *__restrict__ count, int H, int W, int nch)
{
    extern __shared__ __attribute__((aligned(16))) unsigned g2[];  // [RB][W + 2 * EDT_GUARD]
    const int P = W + 2 * EDT_GUARD;
    const int b = blockIdx.y;
    const int r0 = blockIdx.x * RB;
    const int ch = r0 / EDT_CH, j0 = r0 % EDT_CH;
    const int rows_in_word = min(EDT_CH, H - ch * EDT_CH);
    const unsigned valid = rows_in_word == 32 ? 0xFFFFFFFFu : ((1u << rows_in_word) - 1u);
    const int nrows = min(RB, H - r0);
    const int64_t wbase = ((int64_t)b * nch + ch) * W;
    // the column words of all trips are fetched as one batch (up to 4 trips = W <= 1024), then turned into distances
    for (int cbase = 0; cbase < W; cbase += 256 * EDT_STAGE_TRIPS) {
        unsigned wordv[EDT_STAGE_TRIPS], uv[EDT_STAGE_TRIPS], dv[EDT_STAGE_TRIPS];
#pragma unroll
        for (int t = 0; t < EDT_STAGE_TRIPS; ++t) {
            const int c = min(cbase + (int)threadIdx.x + 256 * t, W - 1);
            wordv[t] = bits[wbase + c];
            uv[t] = up[wbase + c];
            dv[t] = dn[wbase + c];
        }
#pragma unroll
        for (int t = 0; t < EDT_STAGE_TRIPS; ++t) {
            const int c = cbase + (int)threadIdx.x + 256 * t;
            if (c < W) {
#pragma unroll
                for (int j = 0; j < RB; ++j)
                    if (j < nrows) {
                        const unsigned v = vdist(wordv[t], valid, j0 + j, uv[t], dv[t], rows_in_word);
                        g2[j * P + c + EDT_GUARD] = v == G_INF ? EDT_D2_INF : v * v;
                    }
            }
        }
    }
    if (threadIdx.x < 2 * EDT_GUARD * RB) {
        const int j = threadIdx.x / (2 * EDT_GUARD), q = threadIdx.x % (2 * EDT_GUARD);
        g2[j * P + (q < EDT_GUARD ? q : W + q)] = EDT_D2_INF;
    }
    __syncthreads();
    const bool anybg = any_bg[b] != 0;
    const int kmax = epi.kmax(W);
    const int64_t fbase = (int64_t)b * H * W;
    unsigned long long cnt = 0;
    // a thread takes two neighbouring pixels: their candidate columns overlap (ten LDS reads per trip serve sixteen
    // candidates) and the two min chains are independent.  A wave walks whole rows (row = wave, wave + 4, ..; lane = pair):
    // no division per pair
    constexpr int WPR = RB >= 4 ? 1 : 4 / RB, RSTEP = 4 / WPR;  // waves per row; rows the block's four waves cover at a time
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int j = wave / WPR; j < nrows; j += RSTEP)
    for (int c = 2 * (lane + 64 * (wave % WPR)); c < W; c += 128 * WPR) {
        const int64_t gi = fbase + rowoff((r0 + j), W) + c;
        const unsigned *gr = g2 + j * P + EDT_GUARD;  // gr[-EDT_GUARD .. -1] and gr[W .. W + EDT_GUARD - 1] are the guard cells
        const bool two = c + 1 < W;
        unsigned best0 = gr[c], best1 = two ? gr[c + 1] : 0u;
        // Offsets k .. k + 3 of both pixels: columns c-k-3 .. c-k+1 on the left, c+k .. c+k+4 on the right -- two base
        // addresses plus constant offsets.  The bases are CLAMPED to the row (0 and W - 1) instead of testing for the row's ends:
        // a clamped read either lands in the guard cells (infinite) or on the end column, which is then NEARER to the pixel
        // than the offset its candidate is priced with -- an over-estimate of a candidate that was examined at its true
        // offset in an earlier trip, so the minimum is unchanged.  Past klim both pixels' offsets are outside the row on
        // both sides (or beyond the cap): the loop's one compare covers that bound and "k^2 >= best".
        const int klim = min(kmax, max(c + 1, W - 1 - c));
        const unsigned lim2 = (unsigned)(klim + 1) * (unsigned)(klim + 1);
        for (int k = 1; (unsigned)(k * k) < min(max(best0, best1), lim2); k += 4) {
            const unsigned *pl = gr + max(c - k + 1, 0), *pr = gr + min(c + k, W - 1);
            unsigned gl[5], gq[5];
#pragma unroll
            for (int t = 0; t < 5; ++t) {
                gl[t] = pl[-t];
                gq[t] = pr[t];
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const unsigned kk = (unsigned)((k + t) * (k + t));
                best0 = min(best0, min(gl[t + 1], gq[t]) + kk);      // pixel c:     c-(k+t), c+(k+t)
                best1 = min(best1, min(gl[t], gq[t + 1]) + kk);      // pixel c + 1: c+1-(k+t), c+1+(k+t)
            }
        }
        epi.store(gi, best0 >= EDT_D2_INF ? (1ll << 40) : (long long)best0, anybg, r0 + j, c, cnt);
        if (two) epi.store(gi + 1, best1 >= EDT_D2_INF ? (1ll << 40) : (long long)best1, anybg, r0 + j, c + 1, cnt);
    }
    if (count) {
        for (int off = 32; off; off >>= 1) cnt += __shfl_xor(cnt, off);
        if (lane_id() == 0 && cnt) atomicAdd(&count[b], cnt);
    }
}

// Horizontal pass of the threshold epilogues.  "Some zero pixel within distance sqrt(R2)" means: some column c' of the
// row with vertical distance g(c') and g^2 + (c - c')^2 <= R2, i.e. c lies in [c' - w, c' + w] with
// w = floor(sqrt(R2 - g^2)).  The union of those intervals is one prefix maximum of (c' + w) from the left and one
// suffix minimum of (c' - w) from the right: O(1) per pixel instead of a search over up to 2 sqrt(R2) + 1 columns.
// A wave owns two of the block's 8 rows and walks them in 64-column chunks (lane = column: conflict-free LDS,
// coalesced stores); the left-pass verdict rides in bit 15 of the staged distance.
__device__ __forceinline__ int reach_halfwidth(unsigned g, int R2)
{
    const int t = R2 - (int)(g * g);  // caller guarantees g * g <= R2
    int w = (int)__fsqrt_rn((float)t);
    if (w * w > t) --w;
    else if ((w + 1) * (w + 1) <= t) ++w;
    return w;
}

template <typename Epi>
__global__ void __launch_bounds__(256, PCSEG_EDT_REACH_OCC) edt_reach_kernel(const unsigned *__restrict__ bits, const uint16_t *__restrict__ up,
                                                         const uint16_t *__restrict__ dn, const int *__restrict__ any_bg,
                                                         Epi epi, unsigned long long *__restrict__ count, int H, int W, int nch)
{
    extern __shared__ __attribute__((aligned(16))) uint16_t g[];  // [EDT_RB][W] distances, then [EDT_RB][W4] in / out bytes
    const int W4 = (W + 3) & ~3;
    uint8_t *zb = reinterpret_cast<uint8_t *>(g + EDT_RB * W4);
    const int b = blockIdx.y;
    const int r0 = blockIdx.x * EDT_RB;
    const int ch = r0 / EDT_CH, j0 = r0 % EDT_CH;
    const int rows_in_word = min(EDT_CH, H - ch * EDT_CH);
    const unsigned valid = rows_in_word == 32 ? 0xFFFFFFFFu : ((1u << rows_in_word) - 1u);
    const int nrows = min(EDT_RB, H - r0);
    const int64_t wbase = ((int64_t)b * nch + ch) * W;
    const int64_t fbase = (int64_t)b * H * W;
    const int R2 = epi.R2();
    const unsigned gmax = (unsigned)sqrtf((float)R2) + 1;  // larger distances can never be within reach
    // half width of the interval a zero pixel at vertical distance g reaches, for every g that can reach at all: a table
    // in LDS (radius <= 180) instead of a square root and two fix-up compares per column and pass
    __shared__ int reach_w[192];
    for (unsigned gq = threadIdx.x; gq < 192; gq += 256) reach_w[gq] = (gq <= gmax && (int)(gq * gq) <= R2) ? reach_halfwidth(gq, R2) : -1;
    const bool wide = (W & 3) == 0 && ((uintptr_t)epi.out & 3) == 0 && (!Epi::kInput || ((uintptr_t)epi.input() & 3) == 0);
    // input bytes of the block (one 4-byte load per thread and row at W = 1024, all in flight together)
    if (Epi::kInput) {
        const uint8_t *src = epi.input() + fbase + rowoff(r0, W);
        if (wide) {
            for (int i = threadIdx.x; i < nrows * (W / 4); i += 256) {
                const int j = i / (W / 4), q = i % (W / 4);
                *reinterpret_cast<unsigned *>(zb + j * W4 + 4 * q) = *reinterpret_cast<const unsigned *>(src + rowoff(j, W) + 4 * q);
            }
        } else {
            for (int i = threadIdx.x; i < nrows * W; i += 256) zb[(i / W) * W4 + i % W] = src[(int64_t)(i / W) * W + i % W];
        }
    }
    // rows whose every column is further than that from a zero pixel vertically (most of a frame when the zero set is
    // one compact object, like the particle of fill_particle_area): nothing is within reach, no scan is needed
    bool near = false;
    for (int cbase = 0; cbase < W; cbase += 256 * EDT_STAGE_TRIPS) {  // batched like the staging of edt_row_kernel
        unsigned wordv[EDT_STAGE_TRIPS], uv[EDT_STAGE_TRIPS], dv[EDT_STAGE_TRIPS];
#pragma unroll
        for (int t = 0; t < EDT_STAGE_TRIPS; ++t) {
            const int c = min(cbase + (int)threadIdx.x + 256 * t, W - 1);
            wordv[t] = bits[wbase + c];
            uv[t] = up[wbase + c];
            dv[t] = dn[wbase + c];
        }
#pragma unroll
        for (int t = 0; t < EDT_STAGE_TRIPS; ++t) {
            const int c = cbase + (int)threadIdx.x + 256 * t;
            if (c < W) {
#pragma unroll
                for (int j = 0; j < EDT_RB; ++j)
                    if (j < nrows) {
#if defined(PCSEG_EXP_REACH) && (PCSEG_EXP_REACH & 2)  // (... without the bit scans of its staging)
                        const unsigned v = ((wordv[t] >> (j0 + j)) & 1u) ? min(uv[t] + dv[t], 0x7FFFu) : 0u;
#else
                        const unsigned v = min(vdist(wordv[t], valid, j0 + j, uv[t], dv[t], rows_in_word), 0x7FFFu);
#endif
                        g[j * W4 + c] = (uint16_t)v;  // bit 15 stays free
                        near = near || v <= gmax;
                    }
            }
        }
    }
#if defined(PCSEG_EXP_REACH) && (PCSEG_EXP_REACH & 1)  // (ablation builds, profiles/r04/time_ops.py: the pass without its scans)
    const bool any_near = __syncthreads_or(near) && false;
#else
    const bool any_near = __syncthreads_or(near);
#endif
    const bool anybg = any_bg[b] != 0;
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const int nchunks = (W + WAVE - 1) / WAVE;
    unsigned long long cnt = 0;
    if (!any_near && anybg) {
        for (int idx = threadIdx.x; idx < nrows * W; idx += 256) {
            const int j = idx / W, c = idx % W;
            zb[j * W4 + c] = epi.decide(zb[j * W4 + c], false, true, r0 + j, c, cnt);
        }
    } else {
        for (int j = wave; j < nrows; j += 4) {
            uint16_t *gr = g + j * W4;
            int carry = -1;
            for (int k = 0; k < nchunks; ++k) {  // left to right: furthest column reached by the intervals that start at or before c
                const int c = k * WAVE + lane;
                const unsigned gv = c < W ? gr[c] : 0x7FFFu;
                const int wl = reach_w[min(gv, 191u)];
                int x = wl >= 0 ? c + wl : -1;
                x = max(wave_prefix_max(x), carry);
                carry = wave_last_lane(x);
                if (c < W && x >= c) gr[c] = (uint16_t)(gv | 0x8000u);
            }
            carry = 0x7FFFFFFF;
            for (int k = nchunks - 1; k >= 0; --k) {  // right to left, then the verdict
                // lanes take the chunk's columns in REVERSE order: the suffix minimum over columns is a prefix minimum
                // over lanes (the DPP scan only runs towards higher lanes)
                const int c = k * WAVE + (WAVE - 1 - lane);
                const unsigned raw = c < W ? gr[c] : 0x7FFFu;
                const unsigned gv = raw & 0x7FFFu;
                const int wr = reach_w[min(gv, 191u)];
                int x = wr >= 0 ? c - wr : 0x7FFFFFFF;
                x = min(wave_prefix_min(x), carry);
                carry = wave_last_lane(x);
                if (c < W) zb[j * W4 + c] = epi.decide(zb[j * W4 + c], (raw & 0x8000u) != 0 || x <= c, anybg, r0 + j, c, cnt);
            }
        }
    }
    __syncthreads();
    uint8_t *dst = epi.out + fbase + rowoff(r0, W);
    if (wide) {
        for (int i = threadIdx.x; i < nrows * (W / 4); i += 256) {
            const int j = i / (W / 4), q = i % (W / 4);
            *reinterpret_cast<unsigned *>(dst + rowoff(j, W) + 4 * q) = *reinterpret_cast<const unsigned *>(zb + j * W4 + 4 * q);
        }
    } else {
        for (int i = threadIdx.x; i < nrows * W; i += 256) dst[(int64_t)(i / W) * W + i % W] = zb[(i / W) * W4 + i % W];
    }
    if (count) {  // per-block partial (plain store): block_counts[b][blockIdx.x], summed by edt_count_kernel
        __shared__ unsigned long long wsum[4];
        for (int off = 32; off; off >>= 1) cnt += __shfl_xor(cnt, off);
        if (lane == 0) wsum[wave] = cnt;
        __syncthreads();
        if (threadIdx.x == 0) count[(int64_t)b * gridDim.x + blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    }
}

// count[b] += sum of the frame's block partials (one wave per frame)
__global__ void __launch_bounds__(64) edt_count_kernel(const unsigned long long *__restrict__ block_counts, int nblocks,
                                                        unsigned long long *__restrict__ count)
{
    const int b = blockIdx.x;
    unsigned long long s = 0;
    for (int i = threadIdx.x; i < nblocks; i += 64) s += block_counts[(int64_t)b * nblocks + i];
    for (int off = 32; off; off >>= 1) s += __shfl_xor(s, off);
    if (threadIdx.x == 0) count[b] += s;
}

// ---- threshold epilogues with a SMALL reach (floor(sqrt(R2)) <= 31: disk(2), fill_particle_area's radius 20), on the BIT
// words.  "Some zero pixel within sqrt(R2) of (r, c)" is the union over the column offsets dx of "column c + dx has a zero
// pixel within h(|dx|) = floor(sqrt(R2 - dx^2)) rows of r" -- and the second statement for all 32 rows of a bit word at once is
// the word of zero bits dilated vertically by h(|dx|) rows: three word operations per row of reach, whatever the image
// holds.  A block takes 448 columns of one word row: (1) every column (and 32 columns either side) dilates its word one
// row at a time and leaves the dilations the offsets ask for in LDS, (2) every column ORs the 2 D + 1 entries its
// neighbours left for it, (3) the verdict bits meet the input bytes four columns at a time.  No distances, no staged rows,
// no scans: the row-block pass this replaces (edt_reach_kernel) spent 78 of its 148 us a launch in prefix scans and 20 in
// bit scans (profiles/r04/ab_logs/r4n_*), 0.97 of its SIMD-cycles issuing vector instructions.
constexpr int RBIT_TC = 448, RBIT_HALO = 32, RBIT_THREADS = RBIT_TC + 2 * RBIT_HALO, RBIT_MAXD = 31;
struct ReachTab {
    int D, nslots;
    int tc;  // columns a block finishes: W split evenly over ceil(W / 448) blocks, a multiple of 4
    // hd[d] = rows a zero pixel reaches at column offset d (falls with d).  The kernel needs no table, only two masks (scalar
    // bit tests instead of a dependent load per loop trip): bit h of store_mask = some offset asks for the dilation by h
    // rows (they go to LDS slots in rising order of h), bit d of change_mask = hd[d] differs from hd[d - 1] (one slot down)
    unsigned store_mask, change_mask;
};

template <typename Epi>
__global__ void __launch_bounds__(RBIT_THREADS) reach_bits_kernel(const unsigned *__restrict__ bits, const int *__restrict__ any_bg,
                                                                  Epi epi, unsigned long long *__restrict__ count, ReachTab tab,
                                                                  int H, int W, int nch)
{
    extern __shared__ __attribute__((aligned(16))) unsigned rb_lds[];  // [nslots][RBIT_THREADS] dilated words, then [RBIT_THREADS] verdicts
    unsigned *acc_lds = rb_lds + tab.nslots * RBIT_THREADS;
    const int b = blockIdx.z, ch = blockIdx.y, c0 = blockIdx.x * tab.tc;
    const int t = threadIdx.x, c = c0 - RBIT_HALO + t;
    // (1) the column's zero bits in the word rows above, here and below (nothing outside the frame is a zero pixel)
    unsigned w0 = 0, w1 = 0, w2 = 0;
    {
        const int cc = min(max(c, 0), W - 1);
        const bool col_in = c >= 0 && c < W;
        const int64_t base = (int64_t)b * nch * W + cc;
        const unsigned r0w = bits[base + rowoff(max(ch - 1, 0), W)], r1w = bits[base + rowoff(ch, W)],
                       r2w = bits[base + rowoff(min(ch + 1, nch - 1), W)];
        const int rows_last = H - (nch - 1) * EDT_CH;  // rows of the frame's last word
        const unsigned v_last = rows_last >= 32 ? 0xFFFFFFFFu : ((1u << rows_last) - 1u);
        if (col_in) {
            w1 = ~r1w & (ch == nch - 1 ? v_last : 0xFFFFFFFFu);
            if (ch > 0) w0 = ~r0w;
            if (ch + 1 < nch) w2 = ~r2w & (ch + 1 == nch - 1 ? v_last : 0xFFFFFFFFu);
        }
    }
    {
        int slot = 0;
        for (int h = 0; h <= tab.D; ++h) {
            if ((tab.store_mask >> h) & 1u) rb_lds[(slot++) * RBIT_THREADS + t] = w1;
            // one more row either way (the outer words lose a row of context per step: 31 steps never reach the middle word)
            const unsigned n1 = w1 | __builtin_amdgcn_alignbit(w1, w0, 31) | __builtin_amdgcn_alignbit(w2, w1, 1);
            const unsigned n0 = w0 | (w0 << 1) | __builtin_amdgcn_alignbit(w1, w0, 1);
            const unsigned n2 = w2 | __builtin_amdgcn_alignbit(w2, w1, 31) | (w2 >> 1);
            w0 = n0; w1 = n1; w2 = n2;
        }
    }
    __syncthreads();
    // (2) the union over the column offsets
    if (t >= RBIT_HALO && t < RBIT_HALO + tab.tc) {
        int slot = tab.nslots - 1;  // offset 0 asks for the largest dilation
        unsigned acc = rb_lds[slot * RBIT_THREADS + t];
        for (int d = 1; d <= tab.D; ++d) {
            slot -= (int)((tab.change_mask >> d) & 1u);
            const unsigned *row = rb_lds + slot * RBIT_THREADS + t;
            acc |= row[-d] | row[d];
        }
        acc_lds[t] = acc;
    }
    __syncthreads();
    // (3) verdict bits -> output bytes
    const bool anybg = any_bg[b] != 0;
    const int64_t fbase = (int64_t)b * H * W;
    unsigned long long cnt = 0;
    const bool wide = anybg && (W & 3) == 0 && ((uintptr_t)epi.out & 3) == 0 && (!Epi::kInput || ((uintptr_t)epi.input() & 3) == 0);
    if (wide) {
        // thread = (group of 8 rows, quad of columns): the quad's four verdict words once, then eight 4-byte loads in one batch
        const int QUADS = tab.tc / 4;
        const int g = t / QUADS, q = t % QUADS, cq = c0 + 4 * q;
        if (g < 4 && cq < W) {
            const uint4 a = *reinterpret_cast<const uint4 *>(acc_lds + RBIT_HALO + 4 * q);
            // the eight rows' bits of the four columns, one byte per column
            const unsigned m = ((a.x >> (8 * g)) & 255u) | (((a.y >> (8 * g)) & 255u) << 8) | (((a.z >> (8 * g)) & 255u) << 16) |
                               (((a.w >> (8 * g)) & 255u) << 24);
            const int rr0 = ch * EDT_CH + 8 * g;
            unsigned z4[8];
            if (Epi::kInput) {
#pragma unroll
                for (int jj = 0; jj < 8; ++jj)
                    z4[jj] = *reinterpret_cast<const unsigned *>(epi.input() + fbase + rowoff(min(rr0 + jj, H - 1), W) + cq);
            }
#pragma unroll
            for (int jj = 0; jj < 8; ++jj)
                if (rr0 + jj < H) {
                    const unsigned w4 = (m >> jj) & 0x01010101u;
                    *reinterpret_cast<unsigned *>(epi.out + fbase + rowoff(rr0 + jj, W) + cq) = epi.decide4(Epi::kInput ? z4[jj] : 0u, w4, cnt);
                }
        }
    } else if (t < tab.tc && c0 + t < W) {
        // ragged widths, unaligned images, and frames without any zero pixel (the verdict is then the epilogue's own
        // business per pixel: scipy's virtual zero pixel): a column per thread, byte by byte
        const unsigned a = acc_lds[RBIT_HALO + t];
        const int cc = c0 + t;
        for (int j = 0; j < EDT_CH; ++j) {
            const int r = ch * EDT_CH + j;
            if (r >= H) break;
            const int64_t i = fbase + rowoff(r, W) + cc;
            epi.out[i] = epi.decide(Epi::kInput ? epi.input()[i] : (uint8_t)0, ((a >> j) & 1u) != 0, anybg, r, cc, cnt);
        }
    }
    if (count) {
        // ONE atomic per block that counted anything: a frame's blocks all add to the same word, and same-address atomics
        // queue (one per wave -- 43 000 a launch on 64 words -- made this kernel 195 us instead of 60)
        __shared__ unsigned long long wsum[RBIT_THREADS / WAVE];
        for (int off = 32; off; off >>= 1) cnt += __shfl_xor(cnt, off);
        if (lane_id() == 0) wsum[t / WAVE] = cnt;
        __syncthreads();
        if (t == 0) {
            unsigned long long s = 0;
            for (int w = 0; w < (int)(blockDim.x / WAVE); ++w) s += wsum[w];
            if (s) atomicAdd(&count[b], s);
        }
    }
}

struct EdtWs {
    unsigned *bits;
    uint16_t *up, *dn;
    int *any_bg;
    unsigned long long *block_counts;  // [B][ceil(H / EDT_RB)] partial counts of the threshold epilogues
    int nch;
};

static size_t edt_ws_bytes(int B, int H, int W)
{
    int nch = (H + EDT_CH - 1) / EDT_CH;
    size_t words = (size_t)B * nch * W;
    return align_up(words * 4) + 2 * align_up(words * 2) + align_up(sizeof(int) * B) +
           align_up(sizeof(unsigned long long) * B * ((H + EDT_RB - 1) / EDT_RB));
}

static EdtWs edt_carve(Carver &cv, int B, int H, int W)
{
    EdtWs ws;
    ws.nch = (H + EDT_CH - 1) / EDT_CH;
    size_t words = (size_t)B * ws.nch * W;
    ws.bits = cv.take<unsigned>(words);
    ws.up = cv.take<uint16_t>(words);
    ws.dn = cv.take<uint16_t>(words);
    ws.any_bg = cv.take<int>(B);
    ws.block_counts = cv.take<unsigned long long>((size_t)B * ((H + EDT_RB - 1) / EDT_RB));
    return ws;
}

template <typename Fg, typename Epi>
static int edt_run(Fg fg, Epi epi, unsigned long long *count, int B, int H, int W, void *workspace, size_t workspace_bytes,
                   hipStream_t s, const char *who)
{
    Carver cv(workspace, workspace_bytes);
    EdtWs ws = edt_carve(cv, B, H, W);
    if (!cv.ok()) {
        set_error("%s: workspace too small (%zu < %zu)", who, workspace_bytes, cv.off);
        return PCSEG_ERR_WORKSPACE;
    }
    size_t lds = (size_t)EDT_RB * ((W + 3) & ~3) * (sizeof(uint16_t) + 1);  // distances + the block's in / out bytes
    if (Epi::kThreshold && lds > 160 * 1024) {
        set_error("%s: W = %d too wide for the LDS row stage", who, W);
        return PCSEG_ERR_ARG;
    }
    PCSEG_CHECK_HIP(hipMemsetAsync(ws.any_bg, 0, sizeof(int) * B, s));
    dim3 g1((W + 255) / 256, ws.nch, B);
    bool wide = false;
    if constexpr (Fg::kBytes) wide = (W & 3) == 0 && ((uintptr_t)fg.p & 3) == 0 && ((uintptr_t)ws.bits & 15) == 0;
    // a threshold epilogue with a reach of at most 31 pixels runs on the bit words alone (reach_bits_kernel): no carries
    bool bit_path = false;
    ReachTab tab{};
    if constexpr (Epi::kThreshold) {
        const int R2 = epi.R2();
        int D = (int)sqrt((double)R2);
        while (D * D > R2) --D;
        while ((D + 1) * (D + 1) <= R2) ++D;
        if (PCSEG_EDT_REACH_BITS && D <= RBIT_MAXD) {
            bit_path = true;
            tab.D = D;
            tab.nslots = 0;
            tab.store_mask = tab.change_mask = 0;
            int prev = -1;
            for (int d = 0; d <= D; ++d) {
                int h = (int)sqrt((double)(R2 - d * d));
                while (h * h > R2 - d * d) --h;
                while ((h + 1) * (h + 1) <= R2 - d * d) ++h;
                if (h != prev) {
                    ++tab.nslots;
                    tab.store_mask |= 1u << h;
                    if (d > 0) tab.change_mask |= 1u << d;
                }
                prev = h;
            }
        }
    }
    int *bits_any = bit_path ? ws.any_bg : nullptr;  // (the carry pass sets the flag on the other paths)
    if (wide) {
        if constexpr (Fg::kBytes)
            PCSEG_LAUNCH((edt_bits4_kernel<Fg>), dim3((W / 4 + 255) / 256, ws.nch, B), dim3(256), 0, s, fg, ws.bits, bits_any, H, W, ws.nch);
    } else {
        PCSEG_LAUNCH((edt_bits_kernel<Fg>), g1, dim3(256), 0, s, fg, ws.bits, bits_any, H, W, ws.nch);
    }
    PCSEG_CHECK_LAUNCH();
    if constexpr (Epi::kThreshold) {
        if (bit_path) {
            const size_t bytes = (size_t)(tab.nslots + 1) * RBIT_THREADS * sizeof(unsigned);
            if (bytes > 64 * 1024)
                PCSEG_CHECK_HIP(hipFuncSetAttribute((const void *)reach_bits_kernel<Epi>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
            const int nblk = (W + RBIT_TC - 1) / RBIT_TC;
            tab.tc = std::min(RBIT_TC, (((W + nblk - 1) / nblk) + 3) & ~3);
            const int threads = (tab.tc + 2 * RBIT_HALO + WAVE - 1) / WAVE * WAVE;  // whole waves; the rest of the 512 would idle
            PCSEG_LAUNCH((reach_bits_kernel<Epi>), dim3((W + tab.tc - 1) / tab.tc, ws.nch, B), dim3(threads), bytes, s,
                         (const unsigned *)ws.bits, (const int *)ws.any_bg, epi, count, tab, H, W, ws.nch);
            PCSEG_CHECK_LAUNCH();
            return PCSEG_OK;
        }
    }
    PCSEG_LAUNCH(edt_carry_kernel, dim3((W + 255) / 256, B), dim3(256), 0, s, ws.bits, ws.up, ws.dn, ws.any_bg, H, W, ws.nch);
    PCSEG_CHECK_LAUNCH();
    dim3 g2((H + EDT_RB - 1) / EDT_RB, B);
    if constexpr (Epi::kThreshold) {
        if (lds > 64 * 1024)
            PCSEG_CHECK_HIP(hipFuncSetAttribute((const void *)edt_reach_kernel<Epi>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        PCSEG_LAUNCH((edt_reach_kernel<Epi>), g2, dim3(256), lds, s, ws.bits, ws.up, ws.dn, ws.any_bg, epi,
                     count ? ws.block_counts : nullptr, H, W, ws.nch);
        if (count) {
            PCSEG_CHECK_LAUNCH();
            PCSEG_LAUNCH(edt_count_kernel, dim3(B), dim3(64), 0, s, (const unsigned long long *)ws.block_counts, (int)g2.x, count);
        }
    } else {
        // rows per block: as many as fit the LDS (uint32 per column and row, 2 * EDT_GUARD guard cells per row)
        auto launch_rows = [&](auto rb_tag) -> int {
            constexpr int RB = decltype(rb_tag)::value;
            const size_t bytes = (size_t)RB * (W + 2 * EDT_GUARD) * sizeof(unsigned);
            if (bytes > 64 * 1024)
                PCSEG_CHECK_HIP(hipFuncSetAttribute((const void *)edt_row_kernel<Epi, RB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
            PCSEG_LAUNCH((edt_row_kernel<Epi, RB>), dim3((H + RB - 1) / RB, B), dim3(256), bytes, s, ws.bits, ws.up, ws.dn, ws.any_bg, epi,
                         count, H, W, ws.nch);
            return PCSEG_OK;
        };
        const size_t per_row = (size_t)(W + 2 * EDT_GUARD) * sizeof(unsigned);
        int rc;
#ifndef PCSEG_EDT_ROWS
#define PCSEG_EDT_ROWS 4
#endif
        // rows per block: the search is a chain of dependent LDS reads, so occupancy beats amortising the staging --
        // 4 rows (17 KB at W = 1024, 8+ blocks per CU) measured 20 % faster than 8, 2 rows were tried too
        if (PCSEG_EDT_ROWS == 8 && 8 * per_row <= 64 * 1024) rc = launch_rows(std::integral_constant<int, 8>());
        else if (PCSEG_EDT_ROWS == 2 && 2 * per_row <= 160 * 1024) rc = launch_rows(std::integral_constant<int, 2>());
        else if (4 * per_row <= 160 * 1024) rc = launch_rows(std::integral_constant<int, 4>());
        else rc = launch_rows(std::integral_constant<int, 1>());
        if (rc) return rc;
    }
    PCSEG_CHECK_LAUNCH();
    return PCSEG_OK;
}

}  // namespace pcseg

using namespace pcseg;

extern "C" {

size_t pcseg_edt_workspace_bytes(int B, int H, int W)
{
    if (!check_shape(B, H, W)) return 0;
    return edt_ws_bytes(B, H, W);
}

int pcseg_edt_sq_u8(const uint8_t *mask, int32_t *d2, int B, int H, int W, int cap, void *workspace, size_t workspace_bytes,
                    pcseg_stream_t stream)
{
    PCSEG_REQUIRE(mask && d2 && workspace && check_shape(B, H, W), "bad arguments");
    return edt_run(FgNzU8{mask}, EpiD2{d2, cap}, nullptr, B, H, W, workspace, workspace_bytes, (hipStream_t)stream, "edt_sq_u8");
}

int pcseg_edt_sq_lt_f32(const float *img, int64_t frame_stride, float threshold, int32_t *d2, uint8_t *mask_out, int B,
                        int H, int W, void *workspace, size_t workspace_bytes, pcseg_stream_t stream)
{
    PCSEG_REQUIRE(img && d2 && workspace && check_shape(B, H, W), "bad arguments");
    PCSEG_REQUIRE(frame_stride == 0 || frame_stride >= (int64_t)H * W, "frame_stride smaller than a frame");
    if (frame_stride == 0) frame_stride = (int64_t)H * W;
    return edt_run(FgLtF32{img, threshold, mask_out, frame_stride}, EpiD2{d2, -1}, nullptr, B, H, W, workspace,
                   workspace_bytes, (hipStream_t)stream, "edt_sq_lt_f32");
}

int pcseg_dilate_disk_u8(const uint8_t *in, uint64_t value_bits, int radius, uint8_t *out, int B, int H, int W,
                         void *workspace, size_t workspace_bytes, pcseg_stream_t stream)
{
    PCSEG_REQUIRE(in && out && in != out && workspace && radius >= 0 && radius <= 180 && check_shape(B, H, W), "bad arguments");
    return edt_run(FgNotInSetU8{in, value_bits}, EpiDilate{out, radius * radius}, nullptr, B, H, W, workspace, workspace_bytes,
                   (hipStream_t)stream, "dilate_disk_u8");
}

int pcseg_fill_particle(const uint8_t *ds, uint8_t *out, int particle_label, int cell_label, int overlap_label,
                        int dilation_radius, int dist_threshold, int64_t *overlap_area, int B, int H, int W, void *workspace,
                        size_t workspace_bytes, pcseg_stream_t stream)
{
    PCSEG_REQUIRE(ds && out && ds != out && workspace && check_shape(B, H, W), "bad arguments");
    PCSEG_REQUIRE(particle_label >= 0 && particle_label < 64 && cell_label >= 0 && cell_label < 256 && overlap_label >= 0 &&
                      overlap_label < 256 && dilation_radius >= 0 && dilation_radius <= 180 && dist_threshold >= 0 &&
                      dist_threshold <= 180,
                  "labels / radii out of range");
    EpiFillParticle epi{ds, out, cell_label, overlap_label, dilation_radius * dilation_radius, dist_threshold * dist_threshold};
    return edt_run(FgNotInSetU8{ds, 1ull << particle_label}, epi, (unsigned long long *)overlap_area, B, H, W, workspace,
                   workspace_bytes, (hipStream_t)stream, "fill_particle");
}

}  // extern "C"

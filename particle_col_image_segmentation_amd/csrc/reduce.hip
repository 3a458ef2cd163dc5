// Per-label reductions: regionprops fields (A3), per-ROI isotope sums (M1),
// region classification / cluster cell counts (A3/A4 tail) and the proximity
// merge grouping (A6 tail).
//
// Main kernel (W % 4 == 0): a lane owns 4 adjacent columns and walks down 32 rows, accumulating each vertical run
// of equal labels in registers (int4 / float4 loads, no cross-lane traffic) and committing it when the label
// changes.  All table columns are integers, so the result does not depend on the order of the atomics; a 256-slot
// direct-mapped LDS table absorbs the hot labels (background, particle) and is flushed once per block.  The generic
// fallback walks rows: one wave per 64-pixel row segment, label runs from a ballot, closed-form run sums.
#include "common.h"

#ifndef PCSEG_SUMS_NT_LOADS
// A/B (profiles/r04/ab_logs/r5a_*): the fused sums pass 378 us with non-temporal plane loads against 389-394 with plain ones; the
// front end, whose tiles share halo rows through L2, lost a quarter with them (526 us against 426) and keeps plain loads
#define PCSEG_SUMS_NT_LOADS 1
#endif
#ifndef PCSEG_LABELS_NT_LOADS
#define PCSEG_LABELS_NT_LOADS 0  // A/B: the label images of the region passes (each read once per pass) as non-temporal loads too
#endif
namespace pcseg {
__device__ __forceinline__ int4 ld_labels4(const int *p)
{
#if PCSEG_LABELS_NT_LOADS
    typedef int i4v __attribute__((ext_vector_type(4)));
    const i4v t = __builtin_nontemporal_load(reinterpret_cast<const i4v *>(p));
    return make_int4(t.x, t.y, t.z, t.w);
#else
    return *reinterpret_cast<const int4 *>(p);
#endif
}
}  // namespace pcseg
namespace pcseg {

constexpr int RED_SLOTS = 256;
constexpr int RED_MAXC = 8;
constexpr int RED_ROWS = 16;  // rows per block

__device__ __forceinline__ void atomic_min_i64(long long *p, long long v) { atomicMin(p, v); }
__device__ __forceinline__ void atomic_max_i64(long long *p, long long v) { atomicMax(p, v); }

// (eight lanes per table row, one per column: a store instruction then covers eight whole 64-byte rows instead of one word
// of 64 different rows)
__global__ void __launch_bounds__(256) region_init_kernel(long long *__restrict__ stats, double *__restrict__ sums,
                                                           const int *__restrict__ counts, int cap, int C, int H, int W)
{
    const int b = blockIdx.y;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int l = idx >> 3, f = idx & 7;
    int nl = counts ? min(counts[b], cap) : cap;
    if (l >= nl) return;
    stats[((int64_t)b * cap + l) * 8 + f] = f == 3 ? (long long)H : (f == 4 ? (long long)W : (f == 7 ? 0x7FFFFFFFFFFFFFFFLL : 0LL));
    if (sums && f < C) sums[((int64_t)b * cap + l) * C + f] = 0.0;
}

template <bool HAS_PLANES>
__global__ void __launch_bounds__(256) region_reduce_kernel(const int *__restrict__ labels, const float *__restrict__ planes,
                                                             const uint8_t *__restrict__ cls, unsigned long long sel, int C, int H,
                                                             int W, int cap, long long *__restrict__ stats,
                                                             double *__restrict__ sums, int *__restrict__ overflow)
{
    __shared__ int tags[RED_SLOTS];
    __shared__ long long lstat[RED_SLOTS][8];
    __shared__ double lsum[HAS_PLANES ? RED_SLOTS : 1][RED_MAXC];
    const int b = blockIdx.y;
    const int64_t n = (int64_t)H * W;
    const int *lab = labels + (int64_t)b * n;
    const float *pl = HAS_PLANES ? planes + (int64_t)b * C * n : nullptr;
    long long *gst = stats + (int64_t)b * cap * 8;
    double *gsum = HAS_PLANES ? sums + (int64_t)b * cap * C : nullptr;
    for (int i = threadIdx.x; i < RED_SLOTS; i += 256) {
        tags[i] = 0;
        lstat[i][0] = 0; lstat[i][1] = 0; lstat[i][2] = 0; lstat[i][3] = H; lstat[i][4] = W; lstat[i][5] = 0; lstat[i][6] = 0;
        lstat[i][7] = 0x7FFFFFFFFFFFFFFFLL;
        if (HAS_PLANES)
            for (int k = 0; k < RED_MAXC; ++k) lsum[i][k] = 0.0;
    }
    __syncthreads();
    const int lane = lane_id(), wid = threadIdx.x >> 6;
    const int row0 = blockIdx.x * RED_ROWS;
    const int segs = (W + 63) / 64;
    // wave w handles (row, segment) pairs round-robin
    for (int item = wid; item < RED_ROWS * segs; item += 4) {
        const int r = row0 + item / segs;
        if (r >= H) break;
        const int c = (item % segs) * 64 + lane;
        const bool inb = c < W;
        const int l = inb ? lab[rowoff(r, W) + c] : 0;
        float v[RED_MAXC];
        if (HAS_PLANES) {
            // sel != 0: plane sums only where the class map holds one of the selected values
            bool want = inb && l > 0;
            if (want && sel) {
                const unsigned cv = cls[(int64_t)b * n + rowoff(r, W) + c];
                want = cv < 64 && ((sel >> cv) & 1ull);
            }
#pragma unroll
            for (int k = 0; k < RED_MAXC; ++k) v[k] = (k < C && want) ? pl[(int64_t)k * n + rowoff(r, W) + c] : 0.f;
        }
        const int lprev = __shfl_up(l, 1);
        const bool head = (lane == 0) || (l != lprev);
        const unsigned long long heads = __ballot(head);
        // run end: next head after this lane (exclusive), 64 if none
        unsigned long long after = lane == 63 ? 0ull : (heads >> (lane + 1));
        const int len = after ? (__ffsll((long long)after)) : (64 - lane);
        // segmented (per run) float64 sums of the channel values: suffix sums inside the run
        double acc[RED_MAXC];
        if (HAS_PLANES) {
#pragma unroll
            for (int k = 0; k < RED_MAXC; ++k) acc[k] = (double)v[k];
            // distance to the end of the run for this lane
            unsigned long long after_me = lane == 63 ? 0ull : (heads >> (lane + 1));
            int remain = after_me ? (__ffsll((long long)after_me) - 1) : (63 - lane);  // lanes after me in my run
            for (int off = 1; off < 64; off <<= 1) {
#pragma unroll
                for (int k = 0; k < RED_MAXC; ++k) {
                    double t = __shfl_down(acc[k], off);
                    if (k < C && off <= remain) acc[k] += t;
                }
            }
        }
        if (head && l > 0) {
            if (l > cap) {
                if (overflow) overflow[b] = 1;
            } else {
                const long long L = len, c0 = c, c1 = c + len - 1;
                const long long s_area = L, s_r = (long long)r * L, s_c = (c0 + c1) * L / 2;
                const long long first = (long long)r * W + c0;
                const int slot = l & (RED_SLOTS - 1);
                int tag = atomicCAS(&tags[slot], 0, l);
                if (tag == 0 || tag == l) {
                    atomicAdd((unsigned long long *)&lstat[slot][0], (unsigned long long)s_area);
                    atomicAdd((unsigned long long *)&lstat[slot][1], (unsigned long long)s_r);
                    atomicAdd((unsigned long long *)&lstat[slot][2], (unsigned long long)s_c);
                    atomic_min_i64(&lstat[slot][3], (long long)r);
                    atomic_min_i64(&lstat[slot][4], c0);
                    atomic_max_i64(&lstat[slot][5], (long long)r + 1);
                    atomic_max_i64(&lstat[slot][6], c1 + 1);
                    atomic_min_i64(&lstat[slot][7], first);
                    if (HAS_PLANES)
#pragma unroll
                        for (int k = 0; k < RED_MAXC; ++k)
                            if (k < C) atomicAdd(&lsum[slot][k], acc[k]);
                } else {
                    long long *t = gst + (int64_t)(l - 1) * 8;
                    atomicAdd((unsigned long long *)&t[0], (unsigned long long)s_area);
                    atomicAdd((unsigned long long *)&t[1], (unsigned long long)s_r);
                    atomicAdd((unsigned long long *)&t[2], (unsigned long long)s_c);
                    atomic_min_i64(&t[3], (long long)r);
                    atomic_min_i64(&t[4], c0);
                    atomic_max_i64(&t[5], (long long)r + 1);
                    atomic_max_i64(&t[6], c1 + 1);
                    atomic_min_i64(&t[7], first);
                    if (HAS_PLANES)
#pragma unroll
                        for (int k = 0; k < RED_MAXC; ++k)
                            if (k < C && acc[k] != 0.0) atomicAdd(&gsum[(int64_t)(l - 1) * C + k], acc[k]);
                }
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < RED_SLOTS; i += 256) {
        const int l = tags[i];
        if (l == 0) continue;
        long long *t = gst + (int64_t)(l - 1) * 8;
        atomicAdd((unsigned long long *)&t[0], (unsigned long long)lstat[i][0]);
        atomicAdd((unsigned long long *)&t[1], (unsigned long long)lstat[i][1]);
        atomicAdd((unsigned long long *)&t[2], (unsigned long long)lstat[i][2]);
        atomic_min_i64(&t[3], lstat[i][3]);
        atomic_min_i64(&t[4], lstat[i][4]);
        atomic_max_i64(&t[5], lstat[i][5]);
        atomic_max_i64(&t[6], lstat[i][6]);
        atomic_min_i64(&t[7], lstat[i][7]);
        if (HAS_PLANES)
            for (int k = 0; k < C; ++k) atomicAdd(&gsum[(int64_t)(l - 1) * C + k], lsum[i][k]);
    }
}

// shared commit step of the column-run reduce kernels: LDS slot if the label owns (or can claim) it, else global atomics.
// A block covers COL_ROWS rows x 1024 columns (32768 pixels at most) of a frame no larger than 32768 x 32768, so every
// block-local partial -- area, row and column sums, the first raster index -- fits 32 bits: the LDS table and the lane
// accumulators are 32-bit (full-rate ds atomics, half the LDS traffic of 64-bit ones), widened at the flush.
struct RegionSlots {
    int *tags;
    int (*lstat)[8];
    double (*lsum)[RED_MAXC];
};

__device__ __forceinline__ void region_slots_clear(int *t, int H, int W)
{
    t[0] = 0; t[1] = 0; t[2] = 0; t[3] = H; t[4] = W; t[5] = 0; t[6] = 0; t[7] = 0x7FFFFFFF;
}

__device__ __forceinline__ void region_slots_flush(const int *s, long long *t)
{
    atomicAdd((unsigned long long *)&t[0], (unsigned long long)(unsigned)s[0]);
    atomicAdd((unsigned long long *)&t[1], (unsigned long long)(unsigned)s[1]);
    atomicAdd((unsigned long long *)&t[2], (unsigned long long)(unsigned)s[2]);
    atomic_min_i64(&t[3], (long long)s[3]);
    atomic_min_i64(&t[4], (long long)s[4]);
    atomic_max_i64(&t[5], (long long)s[5]);
    atomic_max_i64(&t[6], (long long)s[6]);
    atomic_min_i64(&t[7], (long long)s[7]);
}

template <int NC>
__device__ __forceinline__ void region_commit(const RegionSlots &ls, long long *gst, double *gsum, int *overflow, int b, int cap,
                                              int C, int l, int s_area, int s_r, int s_c, int rmin, int rmax1, int c0, int c1,
                                              int first, const double *acc)
{
    if (l > cap) {
        if (overflow) overflow[b] = 1;
        return;
    }
    // (direct-mapped on purpose.  Linear probing over eight slots keeps more labels in the block's LDS table, and measured
    // SLOWER where it matters: the float64 plane sums of a colliding label then queue at an LDS float64 atomic instead of
    // going to the memory-side one -- the fused sums pass 540 us against 407; the integer pass did not move, 197 against 202)
    const int slot = l & (RED_SLOTS - 1);
    const int tag = atomicCAS(&ls.tags[slot], 0, l);
    if (tag == 0 || tag == l) {
        int *t = ls.lstat[slot];
        atomicAdd((unsigned *)&t[0], (unsigned)s_area);
        atomicAdd((unsigned *)&t[1], (unsigned)s_r);
        atomicAdd((unsigned *)&t[2], (unsigned)s_c);
        atomicMin(&t[3], rmin);
        atomicMin(&t[4], c0);
        atomicMax(&t[5], rmax1);
        atomicMax(&t[6], c1 + 1);
        atomicMin(&t[7], first);
#pragma unroll
        for (int k = 0; k < NC; ++k)
            if (k < C && acc[k] != 0.0) atomicAdd(&ls.lsum[slot][k], acc[k]);  // (regions outside the class selection sum to 0)
    } else {
        long long *t = gst + (int64_t)(l - 1) * 8;
        atomicAdd((unsigned long long *)&t[0], (unsigned long long)(unsigned)s_area);
        atomicAdd((unsigned long long *)&t[1], (unsigned long long)(unsigned)s_r);
        atomicAdd((unsigned long long *)&t[2], (unsigned long long)(unsigned)s_c);
        atomic_min_i64(&t[3], (long long)rmin);
        atomic_min_i64(&t[4], (long long)c0);
        atomic_max_i64(&t[5], (long long)rmax1);
        atomic_max_i64(&t[6], (long long)c1 + 1);
        atomic_min_i64(&t[7], (long long)first);
#pragma unroll
        for (int k = 0; k < NC; ++k)
            if (k < C && acc[k] != 0.0) atomicAdd(&gsum[(int64_t)(l - 1) * C + k], acc[k]);
    }
}

// The row walks below fetch row r + 1 before they process row r.  The compiler's wait-count pass cannot count loads across
// the loop's back edge: left alone it puts `s_waitcnt vmcnt(0)` at the first USE of row r -- after the loads of row r + 1
// went out -- and every step then waits a full memory latency (the plane-free pass ran at 1.3 TB/s for that reason).
// "Using" row r's registers in an empty asm ahead of the fetch moves that wait to the top of the step, where only row
// r's loads are outstanding.
__device__ __forceinline__ void landed(const int4 &q) { asm volatile("" ::"v"(q.x), "v"(q.y), "v"(q.z), "v"(q.w) : "memory"); }
__device__ __forceinline__ void landed(const float4 &q) { asm volatile("" ::"v"(q.x), "v"(q.y), "v"(q.z), "v"(q.w) : "memory"); }
__device__ __forceinline__ void landed(unsigned q) { asm volatile("" ::"v"(q) : "memory"); }

// Column-run variant (W % 4 == 0): a lane owns 4 adjacent columns and walks DOWN COL_ROWS rows; it accumulates the
// vertical run of equal labels in registers (area, row sum, plane sums in float64) and commits when the label
// changes.  No cross-lane traffic at all; loads are int4 / float4 and coalesced along the row.
#ifndef PCSEG_COL_ROWS
#define PCSEG_COL_ROWS 32
#endif
constexpr int COL_ROWS = PCSEG_COL_ROWS;

#ifndef PCSEG_RED_WAVES
#define PCSEG_RED_WAVES 2
#endif
// (NC == 0 is kept for completeness; the pipeline's plane-free pass is region_stats_col_kernel below)
template <int NC>
__global__ void __launch_bounds__(256, NC > 0 ? PCSEG_RED_WAVES : 4) region_reduce_col_kernel(const int *__restrict__ labels, const float *__restrict__ planes,
                                                                 const uint8_t *__restrict__ cls, unsigned long long sel, int C,
                                                                 int H, int W, int cap, long long *__restrict__ stats,
                                                                 double *__restrict__ sums, int *__restrict__ overflow)
{
    __shared__ int tags[RED_SLOTS];
    __shared__ int lstat[RED_SLOTS][8];
    __shared__ double lsum[NC > 0 ? RED_SLOTS : 1][RED_MAXC];
    const TileIndex ti = xcd_tile_index();  // (a frame's blocks on one XCD: their atomics on the frame's tables meet in one L2)
    const int b = ti.z;
    const int64_t n = (int64_t)H * W;
    const int *lab = labels + (int64_t)b * n;
    const float *pl = NC > 0 ? planes + (int64_t)b * C * n : nullptr;
    long long *gst = stats + (int64_t)b * cap * 8;
    double *gsum = NC > 0 ? sums + (int64_t)b * cap * C : nullptr;
    for (int i = threadIdx.x; i < RED_SLOTS; i += 256) {
        tags[i] = 0;
        region_slots_clear(lstat[i], H, W);
        if (NC > 0)
            for (int k = 0; k < RED_MAXC; ++k) lsum[i][k] = 0.0;
    }
    __syncthreads();
    const RegionSlots ls{tags, lstat, lsum};
    const int c = (ti.x * 256 + threadIdx.x) * 4;
    const int r0 = ti.y * COL_ROWS, r1 = min(H, r0 + COL_ROWS);
    if (c < W) {
        int cur[4] = {0, 0, 0, 0}, start[4] = {0, 0, 0, 0};
        int area[4] = {0, 0, 0, 0}, srow[4] = {0, 0, 0, 0};
        double acc[4][NC > 0 ? NC : 1];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int k = 0; k < (NC > 0 ? NC : 1); ++k) acc[j][k] = 0.0;
        // the next row's six 16-byte loads are issued before this row is processed (the run logic below is a chain of
        // branches the compiler does not move loads across)
        int4 l4n = make_int4(0, 0, 0, 0);
        float4 vn[NC > 0 ? NC : 1];
        // sel != 0: plane sums only where the class map holds one of the selected values -- the planes of a 4-pixel
        // group without such a pixel are not even read (for class-map components only cell regions need sums, and they
        // are a small part of a frame)
        unsigned wantn = 0xF;
        int64_t at = rowoff(r0, W) + c;  // (fetch() walks the rows in order)
        auto fetch = [&](int r) {
            l4n = make_int4(0, 0, 0, 0);
            if (r < r1) {
                l4n = *reinterpret_cast<const int4 *>(lab + at);
                if (NC > 0) {
                    if (sel) {
                        const unsigned cw = *reinterpret_cast<const unsigned *>(cls + (int64_t)b * n + at);
                        wantn = 0;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const unsigned cv = (cw >> (8 * j)) & 255u;
                            if (cv < 64 && ((sel >> cv) & 1ull)) wantn |= 1u << j;
                        }
                    }
#pragma unroll
                    for (int k = 0; k < NC; ++k)
                        vn[k] = (k < C && wantn) ? *reinterpret_cast<const float4 *>(pl + (int64_t)k * n + at)
                                                 : make_float4(0.f, 0.f, 0.f, 0.f);
                }
                at += W;
            }
        };
        // one row of the walk (r == r1: the flush after the last row, with an all-zero label quad)
        auto step = [&](const int r, const int4 l4, const unsigned want, const float4 *v) {
            const int ll[4] = {l4.x, l4.y, l4.z, l4.w};
            if (r == r1) {
                // end of the block: the four columns of a lane usually sit in the same region; folding them first
                // quarters the number of same-slot LDS atomics the whole block fires at once
                int scol[4], rmin[4], rmax1[4], cmin[4], cmax[4], first[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    scol[j] = __mul24(c + j, area[j]);
                    rmin[j] = start[j];
                    rmax1[j] = start[j] + area[j];
                    cmin[j] = cmax[j] = c + j;
                    first[j] = __mul24(start[j], W) + c + j;
                }
#pragma unroll
                for (int j = 1; j < 4; ++j)
#pragma unroll
                    for (int i = 0; i < j; ++i)
                        if (cur[j] > 0 && cur[j] == cur[i]) {
                            area[i] += area[j]; srow[i] += srow[j]; scol[i] += scol[j];
                            rmin[i] = min(rmin[i], rmin[j]); rmax1[i] = max(rmax1[i], rmax1[j]);
                            cmin[i] = min(cmin[i], cmin[j]); cmax[i] = max(cmax[i], cmax[j]); first[i] = min(first[i], first[j]);
#pragma unroll
                            for (int k = 0; k < (NC > 0 ? NC : 1); ++k) acc[i][k] += acc[j][k];
                            cur[j] = 0;
                        }
                // ... and when a whole WAVE sits in one region (background, the particle: most waves of a class map) the 64
                // lanes are reduced with shuffles and one lane commits: otherwise they all fire their eight atomics at the
                // same LDS slot, which serialises them
                const int l0 = __builtin_amdgcn_readfirstlane(cur[0]);
                // (only a fully active wave: a shuffle from a lane beyond the frame's width would read nothing defined)
                if (__ballot(true) == ~0ull && __all(cur[0] == l0 && l0 > 0 && cur[1] == 0 && cur[2] == 0 && cur[3] == 0)) {
                    int v_area = area[0], v_srow = srow[0], v_scol = scol[0], v_rmin = rmin[0], v_rmax = rmax1[0], v_cmin = cmin[0],
                        v_cmax = cmax[0], v_first = first[0];
                    for (int off = 32; off; off >>= 1) {
                        v_area += __shfl_xor(v_area, off);
                        v_srow += __shfl_xor(v_srow, off);
                        v_scol += __shfl_xor(v_scol, off);
                        v_rmin = min(v_rmin, __shfl_xor(v_rmin, off));
                        v_rmax = max(v_rmax, __shfl_xor(v_rmax, off));
                        v_cmin = min(v_cmin, __shfl_xor(v_cmin, off));
                        v_cmax = max(v_cmax, __shfl_xor(v_cmax, off));
                        v_first = min(v_first, __shfl_xor(v_first, off));
#pragma unroll
                        for (int k = 0; k < (NC > 0 ? NC : 1); ++k) acc[0][k] += __shfl_xor(acc[0][k], off);
                    }
                    if (lane_id() == 0)
                        region_commit<NC>(ls, gst, gsum, overflow, b, cap, C, l0, v_area, v_srow, v_scol, v_rmin, v_rmax, v_cmin, v_cmax,
                                          v_first, acc[0]);
                    return;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (cur[j] > 0)
                        region_commit<NC>(ls, gst, gsum, overflow, b, cap, C, cur[j], area[j], srow[j], scol[j], rmin[j], rmax1[j],
                                          cmin[j], cmax[j], first[j], acc[j]);
                return;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (ll[j] != cur[j]) {
                    if (cur[j] > 0)
                        region_commit<NC>(ls, gst, gsum, overflow, b, cap, C, cur[j], area[j], srow[j], __mul24(c + j, area[j]),
                                          start[j], start[j] + area[j], c + j, c + j, __mul24(start[j], W) + c + j, acc[j]);
                    cur[j] = ll[j];
                    start[j] = r;
                    area[j] = 0;
                    srow[j] = 0;
#pragma unroll
                    for (int k = 0; k < (NC > 0 ? NC : 1); ++k) acc[j][k] = 0.0;
                }
                if (ll[j] > 0) {
                    area[j] += 1;
                    srow[j] += r;
                    if (NC > 0 && ((want >> j) & 1u)) {
#pragma unroll
                        for (int k = 0; k < NC; ++k) {
                            const float4 f = v[k];
                            acc[j][k] += (double)(j == 0 ? f.x : (j == 1 ? f.y : (j == 2 ? f.z : f.w)));
                        }
                    }
                }
            }
                };
        fetch(r0);
        for (int r = r0; r <= r1; ++r) {
            const int4 l4 = l4n;
            const unsigned want = wantn;
            float4 v[NC > 0 ? NC : 1];
#pragma unroll
            for (int k = 0; k < (NC > 0 ? NC : 1); ++k) v[k] = vn[k];
            landed(l4);
            if (NC > 0) {
#pragma unroll
                for (int k = 0; k < NC; ++k) landed(v[k]);
            }
            fetch(r + 1);
            step(r, l4, want, v);
        }
    }
    __syncthreads();
    // flush, eight lanes per slot (see region_stats_col_kernel): one per column of the integer row, one per plane sum
    for (int base = 0; base < RED_SLOTS; base += 32) {
        const int i = base + (int)(threadIdx.x >> 3), f = threadIdx.x & 7;
        const int l = tags[i];
        if (l == 0) continue;
        long long *t = gst + (int64_t)(l - 1) * 8 + f;
        const int v = lstat[i][f];
        if (f < 3) atomicAdd((unsigned long long *)t, (unsigned long long)(unsigned)v);
        else if (f == 5 || f == 6) atomic_max_i64(t, (long long)v);
        else atomic_min_i64(t, (long long)v);
        if (NC > 0 && f < C) atomicAdd(&gsum[(int64_t)(l - 1) * C + f], lsum[i][f]);
    }
}

// ---- the plane-free pass (area, centroid sums, bounding box, first pixel): 4 bytes per pixel.
// Same column walk, but a vertical run is just (label, first row, end row) -- its area, row sum and column sum follow --
// and what bounds the pass is not the walk (64 us with the commits taken out: 4.2 TB/s) but the LDS atomics of the
// commits, eight per run and column, most of them aimed at the slot the neighbouring lanes aim at too (72 us), and the
// block's flush to the frame's table (55 us).  So a finished run is PARKED in two registers, and at the end of the block
// the wave adds up the runs of ADJACENT LANES THAT CARRY THE SAME LABEL with a segmented shuffle reduction -- a region a
// few dozen pixels wide is eight lanes -- and only the first lane of each segment goes to the LDS table.
// (measured on the way: a four- and an eight-row load ring on the old form, 180 and 197 us against 179 -- not the loads;
// parking alone, every lane still committing for itself at the end: 200 us against 185 -- the atomics, not the branch.)
#ifndef PCSEG_STATS_ROWS
#define PCSEG_STATS_ROWS 32
#endif
constexpr int STATS_ROWS = PCSEG_STATS_ROWS;  // rows per block of the plane-free pass (block partials must fit 32 bits: <= 64)
static_assert(STATS_ROWS <= 64, "block-local sums are 32-bit");

struct RunSum {
    int label;  // 0 = none
    int area, srow, scol, rmin, rmax1, cmin, cmax, first;
};

__device__ __forceinline__ RunSum run_sum(int label, int start, int end, int col, int W)
{
    const int area = end - start;
    return RunSum{label, area, __mul24(area, start) + ((area * (area - 1)) >> 1), __mul24(col, area), start, end, col, col,
                  __mul24(start, W) + col};
}

__device__ __forceinline__ void run_merge(RunSum &a, const RunSum &o)
{
    a.area += o.area; a.srow += o.srow; a.scol += o.scol;
    a.rmin = min(a.rmin, o.rmin); a.rmax1 = max(a.rmax1, o.rmax1);
    a.cmin = min(a.cmin, o.cmin); a.cmax = max(a.cmax, o.cmax); a.first = min(a.first, o.first);
}

__device__ __forceinline__ void run_commit(const RegionSlots &ls, long long *gst, int *overflow, int b, int cap, const RunSum &a)
{
    region_commit<0>(ls, gst, nullptr, overflow, b, cap, 0, a.label, a.area, a.srow, a.scol, a.rmin, a.rmax1, a.cmin, a.cmax, a.first,
                     nullptr);
}

// all 64 lanes call this (label 0 = nothing to add): lanes next to each other with the same label are summed into the
// first of them, which commits
__device__ __forceinline__ void wave_commit(const RegionSlots &ls, long long *gst, int *overflow, int b, int cap, RunSum a)
{
    const int lane = lane_id();
    const int left = __shfl_up(a.label, 1);
    const bool head = lane == 0 || a.label != left;
    const unsigned long long heads = __ballot(head);
    const unsigned long long after = lane == 63 ? 0ull : heads >> (lane + 1);
    const int remain = after ? __ffsll((long long)after) - 1 : 63 - lane;  // lanes after this one in its segment
    for (int off = 1; off < 64; off <<= 1) {
        RunSum o;
        o.area = __shfl_down(a.area, off); o.srow = __shfl_down(a.srow, off); o.scol = __shfl_down(a.scol, off);
        o.rmin = __shfl_down(a.rmin, off); o.rmax1 = __shfl_down(a.rmax1, off); o.cmin = __shfl_down(a.cmin, off);
        o.cmax = __shfl_down(a.cmax, off); o.first = __shfl_down(a.first, off);
        if (off <= remain) run_merge(a, o);
    }
    if (head && a.label > 0) run_commit(ls, gst, overflow, b, cap, a);
}

#ifndef PCSEG_STATS_OCC
#define PCSEG_STATS_OCC 4
#endif
__global__ void __launch_bounds__(256, PCSEG_STATS_OCC) region_stats_col_kernel(const int *__restrict__ labels, int H, int W, int cap,
                                                                   long long *__restrict__ stats, int *__restrict__ overflow)
{
    __shared__ int tags[RED_SLOTS];
    __shared__ int lstat[RED_SLOTS][8];
    // every block of a frame adds to the same few lines of the frame's table (the background's, the particle's): with a
    // frame's blocks on ONE XCD those atomics meet in one L2 instead of bouncing the line between eight
    const TileIndex ti = xcd_tile_index();
    const int b = ti.z;
    const int64_t n = (int64_t)H * W;
    const int *lab = labels + (int64_t)b * n;
    long long *gst = stats + (int64_t)b * cap * 8;
    for (int i = threadIdx.x; i < RED_SLOTS; i += 256) {
        tags[i] = 0;
        region_slots_clear(lstat[i], H, W);
    }
    __syncthreads();
    const RegionSlots ls{tags, lstat, nullptr};
    const int c = (ti.x * 256 + threadIdx.x) * 4;
    const int r0 = ti.y * STATS_ROWS, r1 = min(H, r0 + STATS_ROWS);
    // (lanes beyond the frame's width walk zeros: the reductions at the end want all 64 lanes)
    const bool inside = c < W;
    int cur[4] = {0, 0, 0, 0}, start[4] = {0, 0, 0, 0};
    int parked_label[4] = {0, 0, 0, 0}, parked_rows[4] = {0, 0, 0, 0};  // first row | end row << 16 (rows < 2^15)
    const int *at = lab + rowoff(r0, W) + (inside ? c : 0);
    int4 l4n = inside ? ld_labels4(at) : make_int4(0, 0, 0, 0);
    for (int r = r0; r < r1; ++r) {
        const int4 l4 = l4n;
        landed(l4);
        at += W;
        if (inside && r + 1 < r1) l4n = ld_labels4(at);
        const int ll[4] = {l4.x, l4.y, l4.z, l4.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (ll[j] != cur[j]) {
                if (cur[j] > 0) {
                    // (a column seldom ends two runs inside one block)
                    if (parked_label[j])
                        run_commit(ls, gst, overflow, b, cap,
                                   run_sum(parked_label[j], parked_rows[j] & 0xFFFF, parked_rows[j] >> 16, c + j, W));
                    parked_label[j] = cur[j];
                    parked_rows[j] = start[j] | (r << 16);
                }
                cur[j] = ll[j];
                start[j] = r;
            }
        }
    }
    // end of the block: the open runs and the parked ones, each folded over the lane's four columns first (they usually sit
    // in the same region), then over the lanes
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        RunSum q[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            q[j] = pass == 0 ? run_sum(cur[j], start[j], r1, c + j, W)
                             : run_sum(parked_label[j], parked_rows[j] & 0xFFFF, parked_rows[j] >> 16, c + j, W);
#pragma unroll
        for (int j = 1; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < j; ++i)
                if (q[j].label > 0 && q[j].label == q[i].label) {
                    run_merge(q[i], q[j]);
                    q[j].label = 0;
                }
        // a lane whose first column carries no label hands another column's run to the lane reduction instead
#pragma unroll
        for (int j = 1; j < 4; ++j)
            if (q[0].label <= 0 && q[j].label > 0) {
                q[0] = q[j];
                q[j].label = 0;
            }
        if (q[0].label < 0) q[0].label = 0;
        wave_commit(ls, gst, overflow, b, cap, q[0]);
#pragma unroll
        for (int j = 1; j < 4; ++j)
            if (q[j].label > 0) run_commit(ls, gst, overflow, b, cap, q[j]);
    }
    __syncthreads();
    // flush: EIGHT LANES PER SLOT, one per column of the table row, so that one atomic instruction carries up to eight
    // neighbouring 8-byte words of a row's 64-byte line (the three adds, the three mins, the two maxes) instead of 64 lanes
    // aiming at 64 different lines eight times over
    for (int base = 0; base < RED_SLOTS; base += 32) {
        const int i = base + (int)(threadIdx.x >> 3), f = threadIdx.x & 7;
        const int l = tags[i];
        if (l == 0) continue;
        long long *t = gst + (int64_t)(l - 1) * 8 + f;
        const int v = lstat[i][f];
        if (f < 3) atomicAdd((unsigned long long *)t, (unsigned long long)(unsigned)v);
        else if (f == 5 || f == 6) atomic_max_i64(t, (long long)v);
        else atomic_min_i64(t, (long long)v);
    }
}

// ---- plane sums of TWO label images in one pass over the planes (M1 for the class-map components and for the refined
// ROIs: both tables report isotope sums, both label images are final by the end of a batch's chains, and the planes are
// by far the largest thing either reduction reads).  Same column-run scheme as region_reduce_col_kernel -- a lane owns 4
// adjacent columns, walks down COL_ROWS rows and keeps each vertical run's float64 sums in registers -- but only the
// sums: the integer columns (area, centroid sums, bounding box, first pixel) come from the plane-free kernel, which
// reads 4 bytes per pixel.  Image A is restricted to the classes in `sel` (bit v = class value v; 0 = every pixel).
struct SumSlots {
    int *tags;
    double (*lsum)[RED_MAXC];
};

template <int NC>
__device__ __forceinline__ void sums_commit(const SumSlots &ls, double *gsum, int cap, int C, int l, const double *acc)
{
    if (l <= 0 || l > cap) return;
    bool any = false;
#pragma unroll
    for (int k = 0; k < NC; ++k) any = any || acc[k] != 0.0;
    if (!any) return;  // (regions outside the class selection, runs of zero-valued planes)
    const int slot = l & (RED_SLOTS - 1);
    const int tag = atomicCAS(&ls.tags[slot], 0, l);
    if (tag == 0 || tag == l) {
#pragma unroll
        for (int k = 0; k < NC; ++k)
            if (k < C && acc[k] != 0.0) atomicAdd(&ls.lsum[slot][k], acc[k]);
    } else {
#pragma unroll
        for (int k = 0; k < NC; ++k)
            if (k < C && acc[k] != 0.0) atomicAdd(&gsum[(int64_t)(l - 1) * C + k], acc[k]);
    }
}

// STATS_B: image B's integer columns (area, centroid sums, bounding box, first pixel) are accumulated in the same walk
// (its runs are tracked anyway): the refined ROIs' table then needs no pass of its own.
// (five planes without image B's integer columns fit 168 registers: three blocks per CU instead of two, 409 us a launch
// against 497)
// EXACT: C == NC, the plane loads carry no test of C (each test is a scalar branch, and the compiler drains the load queue
// at some of them).
template <int NC, bool STATS_B, bool EXACT>
__global__ void __launch_bounds__(256, (NC <= 5 && !STATS_B) ? 3 : 2) region_sums2_col_kernel(const int *__restrict__ labels_a, const uint8_t *__restrict__ cls,
                                                                  unsigned long long sel, const int *__restrict__ labels_b,
                                                                  const float *__restrict__ planes, int C, int H, int W, int cap_a,
                                                                  int cap_b, double *__restrict__ sums_a, double *__restrict__ sums_b,
                                                                  long long *__restrict__ stats_b, int *__restrict__ overflow_b)
{
    __shared__ int tags_a[RED_SLOTS], tags_b[RED_SLOTS];
    __shared__ double lsum_a[RED_SLOTS][RED_MAXC], lsum_b[RED_SLOTS][RED_MAXC];
    __shared__ int lstat_b[STATS_B ? RED_SLOTS : 1][8];
    const TileIndex ti = xcd_tile_index();  // (a frame's blocks on one XCD: their atomics on the frame's tables meet in one L2)
    const int b = ti.z;
    const int64_t n = (int64_t)H * W;
    const int *la = labels_a + (int64_t)b * n, *lb = labels_b + (int64_t)b * n;
    const float *pl = planes + (int64_t)b * C * n;
    double *ga = sums_a + (int64_t)b * cap_a * C, *gb = sums_b + (int64_t)b * cap_b * C;
    long long *gst_b = STATS_B ? stats_b + (int64_t)b * cap_b * 8 : nullptr;
    for (int i = threadIdx.x; i < RED_SLOTS; i += 256) {
        tags_a[i] = 0;
        tags_b[i] = 0;
        for (int k = 0; k < RED_MAXC; ++k) { lsum_a[i][k] = 0.0; lsum_b[i][k] = 0.0; }
        if (STATS_B) region_slots_clear(lstat_b[i], H, W);
    }
    __syncthreads();
    const SumSlots sa{tags_a, lsum_a}, sb{tags_b, lsum_b};
    const RegionSlots sbb{tags_b, lstat_b, lsum_b};
    const int c = (ti.x * 256 + threadIdx.x) * 4;
    const int r0 = ti.y * COL_ROWS, r1 = min(H, r0 + COL_ROWS);
    if (c < W) {
        int cur_a[4] = {0, 0, 0, 0}, cur_b[4] = {0, 0, 0, 0};
        int start_b[4] = {0, 0, 0, 0}, area_b[4] = {0, 0, 0, 0};
        int srow_b[4] = {0, 0, 0, 0};
        double acc_a[4][NC], acc_b[4][NC];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int k = 0; k < NC; ++k) { acc_a[j][k] = 0.0; acc_b[j][k] = 0.0; }
        // the next row's eight 16-byte loads are issued before this row is processed
        int4 a4n = make_int4(0, 0, 0, 0), b4n = make_int4(0, 0, 0, 0);
        float4 vn[NC];
        // (the class bytes are decoded when the row is processed, not when it is fetched: a decode here would wait for the
        // label loads of the row ahead and undo the look-ahead)
        // without a selection the load reads bytes of image A instead (never used): a conditional load would make the
        // compiler wait for the previous one before it can keep its value
        unsigned cwn = 0;
        const uint8_t *cbytes = sel ? cls + (int64_t)b * n : reinterpret_cast<const uint8_t *>(la);
        int64_t at = rowoff(r0, W) + c;  // (fetch() walks the rows in order)
        auto fetch = [&](int r) {
            a4n = make_int4(0, 0, 0, 0);
            b4n = make_int4(0, 0, 0, 0);
            if (r < r1) {
                a4n = ld_labels4(la + at);
                b4n = ld_labels4(lb + at);
                cwn = *reinterpret_cast<const unsigned *>(cbytes + at);
#pragma unroll
                for (int k = 0; k < NC; ++k)
#if PCSEG_SUMS_NT_LOADS  // the planes are read once by this pass and not again: non-temporal loads (no reuse to keep in the caches)
                    {
                        typedef float f4v __attribute__((ext_vector_type(4)));
                        f4v t = {0.f, 0.f, 0.f, 0.f};
                        if (EXACT || k < C) t = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(pl + (int64_t)k * n + at));
                        vn[k] = make_float4(t.x, t.y, t.z, t.w);
                    }
#else
                    vn[k] = (EXACT || k < C) ? *reinterpret_cast<const float4 *>(pl + (int64_t)k * n + at) : make_float4(0.f, 0.f, 0.f, 0.f);
#endif
                at += W;
            }
        };
        fetch(r0);
        for (int r = r0; r <= r1; ++r) {
            const int4 a4 = a4n, b4 = b4n;
            const unsigned cw = cwn;
            float4 v[NC];
#pragma unroll
            for (int k = 0; k < NC; ++k) v[k] = vn[k];
            landed(a4);
            landed(b4);
            landed(cw);
#pragma unroll
            for (int k = 0; k < NC; ++k) landed(v[k]);
            fetch(r + 1);
            unsigned want = 0xF;
            if (sel) {
                want = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned cv = (cw >> (8 * j)) & 255u;
                    if (cv < 64 && ((sel >> cv) & 1ull)) want |= 1u << j;
                }
            }
            const int aa[4] = {a4.x, a4.y, a4.z, a4.w}, bb[4] = {b4.x, b4.y, b4.z, b4.w};  // (row r1: zeros -> every run ends)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (aa[j] != cur_a[j]) {
                    sums_commit<NC>(sa, ga, cap_a, C, cur_a[j], acc_a[j]);
                    cur_a[j] = aa[j];
#pragma unroll
                    for (int k = 0; k < NC; ++k) acc_a[j][k] = 0.0;
                }
                if (bb[j] != cur_b[j]) {
                    if (STATS_B) {
                        if (cur_b[j] > 0)
                            region_commit<NC>(sbb, gst_b, gb, overflow_b, b, cap_b, C, cur_b[j], area_b[j], srow_b[j],
                                              __mul24(c + j, area_b[j]), start_b[j], start_b[j] + area_b[j], c + j, c + j,
                                              __mul24(start_b[j], W) + c + j, acc_b[j]);
                        start_b[j] = r;
                        area_b[j] = 0;
                        srow_b[j] = 0;
                    } else {
                        sums_commit<NC>(sb, gb, cap_b, C, cur_b[j], acc_b[j]);
                    }
                    cur_b[j] = bb[j];
#pragma unroll
                    for (int k = 0; k < NC; ++k) acc_b[j][k] = 0.0;
                }
                if (r < r1) {
                    const bool in_a = aa[j] > 0 && ((want >> j) & 1u), in_b = bb[j] > 0;
                    if (STATS_B && in_b) {
                        area_b[j] += 1;
                        srow_b[j] += r;
                    }
#pragma unroll
                    for (int k = 0; k < NC; ++k) {
                        const float4 f = v[k];
                        const double x = (double)(j == 0 ? f.x : (j == 1 ? f.y : (j == 2 ? f.z : f.w)));
                        if (in_a) acc_a[j][k] += x;
                        if (in_b) acc_b[j][k] += x;
                    }
                }
            }
        }
    }
    __syncthreads();
    // flush: eight lanes per slot, one per plane, so that one atomic instruction carries a row's neighbouring sums (merged per
    // 64-byte line by the hardware) instead of 64 lanes aiming at 64 different rows once per plane
    static_assert(RED_MAXC == 8, "eight lanes per slot");
    for (int base = 0; base < RED_SLOTS; base += 32) {
        const int i = base + (int)(threadIdx.x >> 3), k = threadIdx.x & 7;
        const int l1 = tags_a[i], l2 = tags_b[i];
        if (l1 && k < C && lsum_a[i][k] != 0.0) atomicAdd(&ga[(int64_t)(l1 - 1) * C + k], lsum_a[i][k]);
        if (l2 && k < C && lsum_b[i][k] != 0.0) atomicAdd(&gb[(int64_t)(l2 - 1) * C + k], lsum_b[i][k]);
        if (STATS_B && l2 && k == 0) region_slots_flush(lstat_b[i], gst_b + (int64_t)(l2 - 1) * 8);
    }
}

__global__ void __launch_bounds__(256) region_class_kernel(const long long *__restrict__ stats, const uint8_t *__restrict__ cls,
                                                            const int *__restrict__ counts, uint8_t *__restrict__ cls_out,
                                                            int cap, int64_t n)
{
    const int b = blockIdx.y;
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    int nl = counts ? min(counts[b], cap) : cap;
    if (l >= nl) return;
    long long first = stats[((int64_t)b * cap + l) * 8 + 7];
    cls_out[(int64_t)b * cap + l] = (first >= 0 && first < n) ? cls[(int64_t)b * n + first] : 0;
}

// ---- A6 tail: grouping by dilated label at the truncated centroid ------------
__device__ __forceinline__ int block_exclusive_scan256(int v, int *total, int *wsum)
{
    int lane = lane_id(), wid = threadIdx.x >> 6;
    int inc = v;
    for (int off = 1; off < 64; off <<= 1) {
        int t = __shfl_up(inc, off);
        if (lane >= off) inc += t;
    }
    if (lane == 63) wsum[wid] = inc;
    __syncthreads();
    int base = 0, tot = 0;
    for (int w = 0; w < 4; ++w) {
        if (w < wid) base += wsum[w];
        tot += wsum[w];
    }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

// keys (tiff_analysis.py:844-848): dilated label at (int(cy), int(cx)); exact with integer floor division.  One thread
// per list entry over the whole batch: the look-up walks a union-find chain in global memory, which is pure latency --
// it wants many blocks in flight, not the one block per frame of the grouping kernel below.
__global__ void __launch_bounds__(256) merge_keys_kernel(const int *__restrict__ dl, const long long *__restrict__ stats,
                                                          const int *__restrict__ region_list, const int *__restrict__ n_list,
                                                          int *__restrict__ key_ws, int H, int W, int cap, int list_cap,
                                                          int keys_are_roots, const unsigned *__restrict__ run_bits)
{
    const int b = blockIdx.y;
    const int R = min(n_list[b], list_cap);
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= R) return;
    const int64_t n = (int64_t)H * W;
    const int *dlab = dl + (int64_t)b * n;
    const long long *st = stats + (int64_t)b * cap * 8;
    int key_k = 0;
    const int r = region_list[(int64_t)b * list_cap + k];
    if (r >= 0 && r < cap) {
        const long long a = st[(int64_t)r * 8 + 0];
        if (a > 0) {
            const long long y = st[(int64_t)r * 8 + 1] / a, x = st[(int64_t)r * 8 + 2] / a;
            if (y >= 0 && y < H && x >= 0 && x < W) {
                if (run_bits) {
                    // components by vertical runs (pcseg_dilate_ccl_runs_u8): the node of a set pixel is the top
                    // pixel of its run inside the 32-row word; only those entries of the parent array exist
                    const int nch = (H + 31) / 32;
                    const unsigned word = run_bits[((int64_t)b * nch + (int)(y >> 5)) * W + x];
                    const int j = (int)(y & 31);
                    if ((word >> j) & 1u) {
                        const unsigned below = ~word & ((1u << j) - 1u);
                        const int start = below ? 32 - __clz(below) : 0;
                        key_k = walk_root(dlab, (int)((y - j + start) * W + x)) + 1;  // fenced (common.h, walk_ok)
                    }
                } else {
                    key_k = dlab[y * W + x];
                    if (keys_are_roots) {  // parent image of a union-find (-1 = background): walk to the root
                        if (key_k >= 0) key_k = (unsigned)key_k < (unsigned)(H * W) ? walk_root(dlab, key_k) : -1;  // fenced
                        key_k += 1;
                    }
                }
            }
        }
    }
    key_ws[(int64_t)b * list_cap + k] = key_k;
}

// grouping of one frame's list by key (one block per frame; keys from merge_keys_kernel)
__global__ void __launch_bounds__(256) merge_groups_kernel(const int *__restrict__ n_list, int *__restrict__ group_of,
                                                            int *__restrict__ n_groups, int *__restrict__ key_ws,
                                                            int *__restrict__ first_ws, int *__restrict__ gid_ws, int list_cap)
{
    __shared__ int wsum[4];
    const int b = blockIdx.x;
    const int R = min(n_list[b], list_cap);
    int *key = key_ws + (int64_t)b * list_cap;
    int *first = first_ws + (int64_t)b * (list_cap + 1);
    int *gid = gid_ws + (int64_t)b * list_cap;
    int *gof = group_of + (int64_t)b * list_cap;
    // dilated labels are arbitrary in 1..K (K may exceed the list length): remap through the list itself --
    // first[] is indexed by the LIST POSITION of the first entry seen with that key, found by a tiny hash on key
    // (open addressing over list_cap + 1 slots, keys stored in gid as scratch)
    for (int k = threadIdx.x; k <= R; k += 256) first[k] = 0x7FFFFFFF;
    for (int k = threadIdx.x; k < R; k += 256) gid[k] = 0;
    __syncthreads();
    // slot table: gid[slot] holds the key owning the slot (0 = free), first[slot] the smallest list position
    for (int k = threadIdx.x; k < R; k += 256) {
        int kk = key[k];
        if (kk <= 0) continue;
        unsigned slot = ((unsigned)kk * 2654435761u) % (unsigned)R;
        for (;;) {
            int owner = atomicCAS(&gid[slot], 0, kk);
            if (owner == 0 || owner == kk) break;
            slot = slot + 1 == (unsigned)R ? 0 : slot + 1;
        }
        atomicMin(&first[slot], k);
        key[k] = -(int)slot - 1;  // remember the slot (negative marks "resolved")
    }
    __syncthreads();
    int carry = 0;
    // group ids in order of the first member: leaders are list positions k with first[slot(k)] == k
    for (int base = 0; base < R; base += 256) {
        int k = base + threadIdx.x;
        int leader = 0;
        if (k < R && key[k] < 0) leader = first[-key[k] - 1] == k;
        int total;
        int ex = block_exclusive_scan256(leader, &total, wsum);
        if (leader) gof[k] = carry + ex + 1;
        carry += total;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < R; k += 256) {
        if (key[k] >= 0) gof[k] = 0;
        else {
            int f = first[-key[k] - 1];
            if (f != k) gof[k] = gof[f];  // leaders were written before the barrier; members only read leaders
        }
    }
    if (threadIdx.x == 0) n_groups[b] = carry;
}

// ---- A6 tail in ONE launch per merge (tiff_analysis.py:843-878): centroid keys, grouping in order of the first member
// and the member sums of every group.  One 1024-thread block per frame: a frame lists a few hundred regions, so every
// entry has a thread of its own and the union-find chain walks of the key look-ups (pure latency) all run at once; the
// hash table of the grouping lives in LDS (the three separate kernels kept it in global memory, where every probe of
// a block's dependent chain was a memory round trip: 117 us per launch against a handful).  Lists longer than MG_LDS
// entries fall back to the caller's global scratch, same code.  region_list / n_list are addressed in place inside the
// (B, n_slots, cap) / (B, n_slots) arrays of pcseg_classify_regions (no per-slot copies).
constexpr int MG_THREADS = 1024, MG_LDS = 4096;

__device__ __forceinline__ int block_exclusive_scan1024(int v, int *total, int *wsum)
{
    const int lane = lane_id(), wid = threadIdx.x >> 6;
    int inc = v;
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(inc, off);
        if (lane >= off) inc += t;
    }
    if (lane == 63) wsum[wid] = inc;
    __syncthreads();
    int base = 0, tot = 0;
    for (int w = 0; w < MG_THREADS / 64; ++w) {
        if (w < wid) base += wsum[w];
        tot += wsum[w];
    }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

// (blockIdx.y = mask: the bit planes / run parents / outputs of mask m sit m * B frames further, its list is slot_of_mask[m])
struct MergeSlots {
    int slot[4];
};
__global__ void __launch_bounds__(MG_THREADS) merge_fused_kernel(const unsigned *__restrict__ run_bits_all, const int *__restrict__ run_parent_all,
                                                                 const long long *__restrict__ stats, const int *__restrict__ region_list,
                                                                 const int *__restrict__ n_list, MergeSlots slot_of_mask, int n_slots,
                                                                 int *__restrict__ group_of_all, int *__restrict__ n_groups_all,
                                                                 long long *__restrict__ gstats_all, int *__restrict__ key_ws_all,
                                                                 int *__restrict__ first_ws_all, int *__restrict__ gid_ws_all, int H, int W, int cap,
                                                                 int list_cap)
{
    __shared__ int s_key[MG_LDS], s_first[MG_LDS + 1], s_gid[MG_LDS];
    __shared__ int wsum[MG_THREADS / 64];
    const int b = blockIdx.x;
    const int mask = blockIdx.y, B = gridDim.x, slot = slot_of_mask.slot[mask];
    const int64_t mf = (int64_t)mask * B;  // frames in front of this mask's arrays
    const unsigned *run_bits = run_bits_all + mf * ((H + 31) / 32) * W;
    const int *run_parent = run_parent_all + mf * H * W;
    int *group_of = group_of_all + mf * list_cap, *n_groups = n_groups_all + mf;
    long long *gstats = gstats_all + mf * list_cap * 8;
    int *key_ws = key_ws_all + mf * list_cap, *first_ws = first_ws_all + mf * (list_cap + 1), *gid_ws = gid_ws_all + mf * list_cap;
    const int R = min(n_list[b * n_slots + slot], list_cap);
    const int *lst = region_list + ((int64_t)b * n_slots + slot) * cap;
    const bool in_lds = R <= MG_LDS;  // block-uniform
    int *key = in_lds ? s_key : key_ws + (int64_t)b * list_cap;
    int *first = in_lds ? s_first : first_ws + (int64_t)b * (list_cap + 1);
    int *gid = in_lds ? s_gid : gid_ws + (int64_t)b * list_cap;
    int *gof = group_of + (int64_t)b * list_cap;
    long long *gs = gstats + (int64_t)b * list_cap * 8;
    const int64_t n = (int64_t)H * W;
    const int *par = run_parent + (int64_t)b * n;
    const long long *st = stats + (int64_t)b * cap * 8;
    const int nch = (H + 31) / 32;
    // (1) keys: the run component under the truncated centroid (0: centroid on a clear pixel -> dropped, :848)
    for (int k = threadIdx.x; k < R; k += MG_THREADS) {
        int key_k = 0;
        const int r = lst[k];
        if (r >= 0 && r < cap) {
            const long long a = st[(int64_t)r * 8 + 0];
            if (a > 0) {
                const long long y = st[(int64_t)r * 8 + 1] / a, x = st[(int64_t)r * 8 + 2] / a;
                if (y >= 0 && y < H && x >= 0 && x < W) {
                    const unsigned word = run_bits[((int64_t)b * nch + (int)(y >> 5)) * W + x];
                    const int j = (int)(y & 31);
                    if ((word >> j) & 1u) {
                        const unsigned below = ~word & ((1u << j) - 1u);
                        const int start = below ? 32 - __clz(below) : 0;
                        key_k = walk_root(par, (int)((y - j + start) * W + x)) + 1;  // fenced (common.h, walk_ok)
                    }
                }
            }
        }
        key[k] = key_k;
        gid[k] = 0;
    }
    for (int k = threadIdx.x; k <= R; k += MG_THREADS) first[k] = 0x7FFFFFFF;
    __syncthreads();
    // (2) slot table (open addressing over R slots): gid[slot] = key owning the slot, first[slot] = smallest list position
    for (int k = threadIdx.x; k < R; k += MG_THREADS) {
        const int kk = key[k];
        if (kk <= 0) continue;
        unsigned slot_i = ((unsigned)kk * 2654435761u) % (unsigned)R;
        for (;;) {
            const int owner = atomicCAS(&gid[slot_i], 0, kk);
            if (owner == 0 || owner == kk) break;
            slot_i = slot_i + 1 == (unsigned)R ? 0 : slot_i + 1;
        }
        atomicMin(&first[slot_i], k);
        key[k] = -(int)slot_i - 1;  // remember the slot (negative marks "resolved")
    }
    __syncthreads();
    // (3) group ids in order of the first member: leaders are the list positions k with first[slot(k)] == k
    int carry = 0;
    for (int base = 0; base < R; base += MG_THREADS) {
        const int k = base + threadIdx.x;
        int leader = 0;
        if (k < R && key[k] < 0) leader = first[-key[k] - 1] == k;
        int total;
        const int ex = block_exclusive_scan1024(leader, &total, wsum);
        if (leader) {
            const int g = carry + ex + 1;
            gof[k] = g;
            gid[-key[k] - 1] = g;  // the slot's key is not needed any more: it now holds the group id for the members
            long long *t = gs + (int64_t)(g - 1) * 8;
            t[0] = 0; t[1] = 0; t[2] = 0; t[3] = H; t[4] = W; t[5] = 0; t[6] = 0; t[7] = 0;
        }
        carry += total;
    }
    __syncthreads();  // (group ids and the zeroed rows of this block's groups: written above, used below by the same block)
    // (4) members take their leader's id; every listed region adds itself to its group's row (tiff_analysis.py:855-872)
    // (eight lanes per member, one per column of the row: the reads are one 64-byte line per member and the atomics of an
    // instruction that fall into one line travel together -- see region_stats_col_kernel's flush)
    for (int idx = threadIdx.x; idx < R * 8; idx += MG_THREADS) {
        const int k = idx >> 3, f = idx & 7;
        int g = 0;
        if (key[k] < 0) g = gid[-key[k] - 1];
        if (f == 0) gof[k] = g;
        if (g <= 0) continue;
        const long long v = st[(int64_t)lst[k] * 8 + f];
        long long *t = gs + (int64_t)(g - 1) * 8 + f;
        if (f < 3) atomicAdd((unsigned long long *)t, (unsigned long long)v);
        else if (f < 5) atomicMin(t, v);
        else if (f < 7) atomicMax(t, v);
        else atomicAdd((unsigned long long *)t, 1ull);
    }
    if (threadIdx.x == 0) n_groups[b] = carry;
}

// ---- A3 tail + A4: the reference's per-region loop (tiff_analysis.py:754-781) and the region lists that
// get_cell_clusters_from_distances builds (:794-796, 811, 820) for one frame per block.
constexpr int CLS_T = 4;  // cell-type slots (the reference has 3: CELL_TYPES)

struct ClassTables {
    uint8_t slot[256];      // class value -> cell-type slot, 255 = not a cell type
    uint8_t particle[256];  // class value is "Particle"
    int min_cell[CLS_T];
    int min_cluster[CLS_T];
    int n_slots;
};

// numpy's float64 floor_divide (npy_divmod): fmod-based, exact for the rounded divisor
__device__ __forceinline__ double npy_floor_divide(double a, double b)
{
    double mod = fmod(a, b);
    if (!(b == b) || !(mod == mod)) return mod;  // NaN
    double div = (a - mod) / b;
    if (mod != 0.0 && ((b < 0) != (mod < 0))) div -= 1.0;
    double fl;
    if (div != 0.0) {
        fl = floor(div);
        if (div - fl > 0.5) fl += 1.0;
    } else fl = 0.0;
    return fl;
}

// one block of CLS_THREADS per frame: the passes over the frame's few thousand regions are loops of dependent loads, so
// the block is as wide as a block gets (256 threads: 63 us a launch, 1024: 21)
constexpr int CLS_THREADS = 1024;
__global__ void __launch_bounds__(CLS_THREADS) classify_regions_kernel(const long long *__restrict__ stats, const uint8_t *__restrict__ cls_out,
                                                                const int *__restrict__ counts, ClassTables tab,
                                                                uint8_t *__restrict__ kind, uint8_t *__restrict__ slot_of,
                                                                int *__restrict__ cells, long long *__restrict__ particle_area,
                                                                long long *__restrict__ type_stats /*(B,CLS_T,4)*/,
                                                                int *__restrict__ region_list /*(B,CLS_T+1,cap)*/,
                                                                int *__restrict__ n_list /*(B,CLS_T+1)*/, int *__restrict__ nan_flag,
                                                                int cap)
{
    __shared__ unsigned long long s_particle;
    __shared__ int s_first[CLS_T], s_ncell[CLS_T], s_nclu[CLS_T], s_base[CLS_T];
    __shared__ unsigned long long s_sumcell[CLS_T];
    __shared__ int s_nan;
    const int b = blockIdx.x;
    const int R = min(counts[b], cap);
    const long long *st = stats + (int64_t)b * cap * 8;
    const uint8_t *co = cls_out + (int64_t)b * cap;
    uint8_t *kd = kind + (int64_t)b * cap, *so = slot_of + (int64_t)b * cap;
    int *cl = cells + (int64_t)b * cap;
    if (threadIdx.x < CLS_T) {
        s_first[threadIdx.x] = 0x7FFFFFFF; s_ncell[threadIdx.x] = 0; s_nclu[threadIdx.x] = 0; s_sumcell[threadIdx.x] = 0;
    }
    if (threadIdx.x == 0) { s_particle = 0; s_nan = 0; }
    __syncthreads();
    for (int r = threadIdx.x; r < R; r += CLS_THREADS) {
        int c = co[r];
        long long area = st[(int64_t)r * 8];
        int slot = tab.slot[c];
        int k = 0;
        if (tab.particle[c]) atomicAdd(&s_particle, (unsigned long long)area);
        if (slot != 255) {
            atomicMin(&s_first[slot], r);
            if (area >= tab.min_cell[slot] && area < tab.min_cluster[slot]) {
                k = 1;
                atomicAdd(&s_ncell[slot], 1);
                atomicAdd(&s_sumcell[slot], (unsigned long long)area);
            } else if (area >= tab.min_cluster[slot]) {
                k = 2;
                atomicAdd(&s_nclu[slot], 1);
            }
        }
        kd[r] = (uint8_t)k;
        so[r] = (uint8_t)slot;
        cl[r] = k == 1 ? 1 : 0;
    }
    __syncthreads();
    // cluster.cells = int(area // mean(cell areas))   (:776-781)
    for (int r = threadIdx.x; r < R; r += CLS_THREADS) {
        if (kd[r] != 2) continue;
        int slot = so[r];
        if (s_ncell[slot] == 0) { cl[r] = -1; s_nan = 1; continue; }  // the reference raises ValueError here
        double avg = (double)s_sumcell[slot] / (double)s_ncell[slot];
        cl[r] = (int)npy_floor_divide((double)st[(int64_t)r * 8], avg);
    }
    // combined-list bases: types in the order of their first region (dict insertion order of the reference)
    if (threadIdx.x < CLS_T) {
        int me = threadIdx.x, base = 0;
        for (int t = 0; t < CLS_T; ++t)
            if (t != me && s_first[t] < s_first[me]) base += s_ncell[t] + s_nclu[t];
        s_base[me] = base;
    }
    __syncthreads();
    int *lists = region_list + (int64_t)b * (CLS_T + 1) * cap;
    // list positions: ONE pass over the regions for all (slot, kind) pairs.  A region belongs to at most one of the
    // 2 * CLS_T lists, so its position is the number of earlier regions of the same list: per wave a ballot per list and a
    // population count, across the waves one exchange through LDS per CLS_THREADS regions (it used to be a block scan with
    // two barriers per list and per 256 regions)
    __shared__ int s_wcount[CLS_THREADS / 64][2 * CLS_T];
    __shared__ int s_carry[2 * CLS_T];
    if (threadIdx.x < 2 * CLS_T) s_carry[threadIdx.x] = 0;
    __syncthreads();
    const int lane = lane_id(), wid = threadIdx.x >> 6;
    const unsigned long long below = (1ull << lane) - 1ull;
    for (int base = 0; base < R; base += CLS_THREADS) {
        const int r = base + threadIdx.x;
        const int mine = (r < R && kd[r] != 0) ? (int)so[r] * 2 + (kd[r] - 1) : -1;  // list of this region (kind 1 or 2 has a slot)
        int before = 0;
#pragma unroll
        for (int f = 0; f < 2 * CLS_T; ++f) {
            const unsigned long long m = __ballot(mine == f);
            if (mine == f) before = __popcll(m & below);
            if (lane == 0) s_wcount[wid][f] = __popcll(m);
        }
        __syncthreads();
        if (mine >= 0) {
            int pos = s_carry[mine] + before;
            for (int w = 0; w < wid; ++w) pos += s_wcount[w][mine];
            const int slot = mine >> 1;
            if (mine & 1) pos += s_ncell[slot];  // a type's clusters follow its cells
            lists[(int64_t)slot * cap + pos] = r;
            lists[(int64_t)CLS_T * cap + s_base[slot] + pos] = r;
        }
        __syncthreads();
        if (threadIdx.x < 2 * CLS_T)
            for (int w = 0; w < CLS_THREADS / 64; ++w) s_carry[threadIdx.x] += s_wcount[w][threadIdx.x];
        __syncthreads();
    }
    if (threadIdx.x < CLS_T) {
        int t = threadIdx.x;
        n_list[b * (CLS_T + 1) + t] = s_ncell[t] + s_nclu[t];
        long long *ts = type_stats + ((int64_t)b * CLS_T + t) * 4;
        ts[0] = s_ncell[t]; ts[1] = s_nclu[t]; ts[2] = (long long)s_sumcell[t]; ts[3] = s_first[t];
    }
    if (threadIdx.x == 0) {
        int tot = 0;
        for (int t = 0; t < CLS_T; ++t) tot += s_ncell[t] + s_nclu[t];
        n_list[b * (CLS_T + 1) + CLS_T] = tot;
        particle_area[b] = (long long)s_particle;
        nan_flag[b] = s_nan;
    }
}

// group table of pcseg_merge_groups: (B, list_cap, 8) = area, sum_row, sum_col, bbox(4), members
__global__ void __launch_bounds__(256) group_reduce_kernel(const long long *__restrict__ stats, const int *__restrict__ region_list,
                                                            const int *__restrict__ n_list, const int *__restrict__ group_of,
                                                            const int *__restrict__ n_groups, long long *__restrict__ gstats,
                                                            int cap, int list_cap, int H, int W)
{
    const int b = blockIdx.x;
    const int R = min(n_list[b], list_cap), G = n_groups[b];
    long long *gs = gstats + (int64_t)b * list_cap * 8;
    for (int g = threadIdx.x; g < G; g += 256) {
        long long *t = gs + (int64_t)g * 8;
        t[0] = 0; t[1] = 0; t[2] = 0; t[3] = H; t[4] = W; t[5] = 0; t[6] = 0; t[7] = 0;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < R; k += 256) {
        int g = group_of[(int64_t)b * list_cap + k];
        if (g <= 0) continue;
        const long long *s = stats + ((int64_t)b * cap + region_list[(int64_t)b * list_cap + k]) * 8;
        long long *t = gs + (int64_t)(g - 1) * 8;
        atomicAdd((unsigned long long *)&t[0], (unsigned long long)s[0]);
        atomicAdd((unsigned long long *)&t[1], (unsigned long long)s[1]);
        atomicAdd((unsigned long long *)&t[2], (unsigned long long)s[2]);
        atomicMin(&t[3], s[3]);
        atomicMin(&t[4], s[4]);
        atomicMax(&t[5], s[5]);
        atomicMax(&t[6], s[6]);
        atomicAdd((unsigned long long *)&t[7], 1ull);
    }
}

// ---- C14: nearest neighbour between two point sets (pdist2 + min, .m:260-263, 301-305)
__global__ void __launch_bounds__(256) nearest_kernel(const double *__restrict__ a, int na, const double *__restrict__ b, int nb,
                                                       double *__restrict__ out)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= na) return;
    const double ax = a[2 * i], ay = a[2 * i + 1];
    double best = 1.0 / 0.0;
    for (int j = 0; j < nb; ++j) {
        double dx = ax - b[2 * j], dy = ay - b[2 * j + 1];
        double d2 = dx * dx + dy * dy;
        best = d2 < best ? d2 : best;
    }
    out[i] = sqrt(best);
}

}  // namespace pcseg

using namespace pcseg;

extern "C" {

int pcseg_region_reduce(const int32_t *labels, const uint8_t *cls, const float *planes, int C, int B, int H, int W, int cap,
                        int64_t *stats, uint8_t *cls_out, double *sums, int32_t *overflow, pcseg_stream_t stream);

/* counts-aware variant used by the host layer: rows >= counts[b] are left untouched */
int pcseg_region_reduce_n(const int32_t *labels, const int32_t *counts, const uint8_t *cls, const float *planes, int C,
                          int B, int H, int W, int cap, int64_t *stats, uint8_t *cls_out, double *sums, int32_t *overflow,
                          pcseg_stream_t stream)
{
    return pcseg_region_reduce_sel(labels, counts, cls, 0, planes, C, B, H, W, cap, stats, cls_out, sums, overflow, stream);
}

int pcseg_region_reduce_sel(const int32_t *labels, const int32_t *counts, const uint8_t *cls, uint64_t sum_class_bits,
                            const float *planes, int C, int B, int H, int W, int cap, int64_t *stats, uint8_t *cls_out,
                            double *sums, int32_t *overflow, pcseg_stream_t stream)
{
    PCSEG_REQUIRE(labels && stats && cap >= 1 && check_shape(B, H, W), "bad arguments");
    PCSEG_REQUIRE(sum_class_bits == 0 || (cls && planes), "a class selection needs the class map and planes");
    const unsigned long long sel = sum_class_bits;
    PCSEG_REQUIRE((!planes && !sums) || (sums && C >= 1 && C <= RED_MAXC), "planes/sums/C mismatch (C <= 8)");
    PCSEG_REQUIRE(!cls || cls_out, "cls needs cls_out");
    hipStream_t s = (hipStream_t)stream;
    if (overflow) PCSEG_CHECK_HIP(hipMemsetAsync(overflow, 0, sizeof(int32_t) * B, s));
    dim3 gi((cap * 8 + 255) / 256, B);  // eight lanes per row
    PCSEG_LAUNCH(region_init_kernel, gi, dim3(256), 0, s, (long long *)stats, sums, counts, cap, C, H, W);
    PCSEG_CHECK_LAUNCH();
    dim3 grid((H + RED_ROWS - 1) / RED_ROWS, B);
    const bool vec = (W % 4) == 0 && ((uintptr_t)labels % 16) == 0 && (!planes || ((uintptr_t)planes % 16) == 0) &&
                     (!sel || ((uintptr_t)cls % 4) == 0);
    const dim3 cgrid((W / 4 + 255) / 256, (H + COL_ROWS - 1) / COL_ROWS, B);
    if (vec && planes && C <= 5)
        PCSEG_LAUNCH(region_reduce_col_kernel<5>, cgrid, dim3(256), 0, s, labels, planes, cls, sel, C, H, W, cap, (long long *)stats, sums,
                     overflow);
    else if (vec && planes)
        PCSEG_LAUNCH(region_reduce_col_kernel<8>, cgrid, dim3(256), 0, s, labels, planes, cls, sel, C, H, W, cap, (long long *)stats, sums,
                     overflow);
    else if (vec)
        PCSEG_LAUNCH(region_stats_col_kernel, dim3(cgrid.x, (H + STATS_ROWS - 1) / STATS_ROWS, B), dim3(256), 0, s, labels, H, W, cap,
                     (long long *)stats, overflow);
    else if (planes)
        PCSEG_LAUNCH(region_reduce_kernel<true>, grid, dim3(256), 0, s, labels, planes, cls, sel, C, H, W, cap, (long long *)stats,
                           sums, overflow);
    else
        PCSEG_LAUNCH(region_reduce_kernel<false>, grid, dim3(256), 0, s, labels, planes, cls, sel, C, H, W, cap,
                           (long long *)stats, sums, overflow);
    PCSEG_CHECK_LAUNCH();
    if (cls) {
        PCSEG_LAUNCH(region_class_kernel, gi, dim3(256), 0, s, (const long long *)stats, cls, counts, cls_out, cap,
                           (int64_t)H * W);
        PCSEG_CHECK_LAUNCH();
    }
    return PCSEG_OK;
}

int pcseg_region_reduce(const int32_t *labels, const uint8_t *cls, const float *planes, int C, int B, int H, int W, int cap,
                        int64_t *stats, uint8_t *cls_out, double *sums, int32_t *overflow, pcseg_stream_t stream)
{
    return pcseg_region_reduce_n(labels, nullptr, cls, planes, C, B, H, W, cap, stats, cls_out, sums, overflow, stream);
}

int pcseg_region_init(const int32_t *counts, int cap, int C, int B, int H, int W, int64_t *stats, double *sums, int32_t *overflow,
                      pcseg_stream_t stream)
{
    PCSEG_REQUIRE(stats && cap >= 1 && C >= 0 && C <= RED_MAXC && (C == 0 || sums) && check_shape(B, H, W), "bad arguments (C <= 8)");
    hipStream_t s = (hipStream_t)stream;
    if (overflow) PCSEG_CHECK_HIP(hipMemsetAsync(overflow, 0, sizeof(int32_t) * B, s));
    static_assert(RED_MAXC <= 8, "region_init_kernel has eight lanes per row");
    PCSEG_LAUNCH(region_init_kernel, dim3((cap * 8 + 255) / 256, B), dim3(256), 0, s, (long long *)stats, C ? sums : nullptr, counts, cap, C, H, W);
    PCSEG_CHECK_LAUNCH();
    return PCSEG_OK;
}

int pcseg_region_sums2(const int32_t *labels_a, const uint8_t *cls, uint64_t sum_class_bits, int cap_a, double *sums_a,
                       const int32_t *labels_b, int cap_b, double *sums_b, int64_t *stats_b, int32_t *overflow_b,
                       const float *planes, int C, int B, int H, int W, pcseg_stream_t stream)
{
    PCSEG_REQUIRE(labels_a && labels_b && sums_a && sums_b && planes && cap_a >= 1 && cap_b >= 1 && C >= 1 && C <= RED_MAXC &&
                      check_shape(B, H, W),
                  "bad arguments (C <= 8)");
    PCSEG_REQUIRE(sum_class_bits == 0 || cls, "a class selection needs the class map");
    PCSEG_REQUIRE((W % 4) == 0 && (((uintptr_t)labels_a | (uintptr_t)labels_b | (uintptr_t)planes) % 16) == 0 &&
                      (!sum_class_bits || ((uintptr_t)cls % 4) == 0),
                  "W must be a multiple of 4 and the images 16-byte aligned (use pcseg_region_reduce_sel per image otherwise)");
    hipStream_t s = (hipStream_t)stream;
    const dim3 cgrid((W / 4 + 255) / 256, (H + COL_ROWS - 1) / COL_ROWS, B);
    const unsigned long long sel = sum_class_bits;
#define PCSEG_SUMS2(NCV, ST)                                                                                                        \
    if (C == NCV)                                                                                                                   \
        PCSEG_LAUNCH((region_sums2_col_kernel<NCV, ST, true>), cgrid, dim3(256), 0, s, labels_a, cls, sel, labels_b, planes, C, H, W, \
                     cap_a, cap_b, sums_a, sums_b, (long long *)stats_b, overflow_b);                                              \
    else                                                                                                                            \
        PCSEG_LAUNCH((region_sums2_col_kernel<NCV, ST, false>), cgrid, dim3(256), 0, s, labels_a, cls, sel, labels_b, planes, C, H, W, cap_a, cap_b, \
                 sums_a, sums_b, (long long *)stats_b, overflow_b)
    if (C <= 5 && stats_b) { PCSEG_SUMS2(5, true); }
    else if (C <= 5) { PCSEG_SUMS2(5, false); }
    else if (stats_b) { PCSEG_SUMS2(8, true); }
    else { PCSEG_SUMS2(8, false); }
#undef PCSEG_SUMS2
    PCSEG_CHECK_LAUNCH();
    return PCSEG_OK;
}

size_t pcseg_merge_groups_workspace_bytes(int B, int list_cap)
{
    if (B < 1 || list_cap < 1) return 0;
    return align_up(sizeof(int) * (size_t)B * list_cap) * 2 + align_up(sizeof(int) * (size_t)B * (list_cap + 1));
}

int pcseg_merge_groups(const int32_t *dilated_labels, int keys_are_roots, const int64_t *stats, const int32_t *region_list,
                       const int32_t *n_list, int32_t *group_of, int32_t *n_groups, int B, int H, int W, int cap, int list_cap,
                       void *workspace, size_t workspace_bytes, pcseg_stream_t stream)
{
    PCSEG_REQUIRE(dilated_labels && stats && region_list && n_list && group_of && n_groups && workspace && cap >= 1 &&
                      list_cap >= 1 && check_shape(B, H, W),
                  "bad arguments");
    Carver cv(workspace, workspace_bytes);
    int *key = cv.take<int>((size_t)B * list_cap);
    int *gid = cv.take<int>((size_t)B * list_cap);
    int *first = cv.take<int>((size_t)B * (list_cap + 1));
    if (!cv.ok()) {
        set_error("merge_groups: workspace too small (%zu < %zu)", workspace_bytes, cv.off);
        return PCSEG_ERR_WORKSPACE;
    }
    PCSEG_LAUNCH(merge_keys_kernel, dim3((list_cap + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, dilated_labels,
                 (const long long *)stats, region_list, n_list, key, H, W, cap, list_cap, keys_are_roots, (const unsigned *)nullptr);
    PCSEG_CHECK_LAUNCH();
    PCSEG_LAUNCH(merge_groups_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, n_list, group_of, n_groups, key, first, gid, list_cap);
    PCSEG_CHECK_LAUNCH();
    return PCSEG_OK;
}

int pcseg_merge_groups_runs(const uint32_t *dilated_bits, const int32_t *run_parent, const int64_t *stats,
                            const int32_t *region_list, const int32_t *n_list, int32_t *group_of, int32_t *n_groups, int B, int H,
                            int W, int cap, int list_cap, void *workspace, size_t workspace_bytes, pcseg_stream_t stream)
{
    PCSEG_REQUIRE(dilated_bits && run_parent && stats && region_list && n_list && group_of && n_groups && workspace && cap >= 1 &&
                      list_cap >= 1 && check_shape(B, H, W),
                  "bad arguments");
    Carver cv(workspace, workspace_bytes);
    int *key = cv.take<int>((size_t)B * list_cap);
    int *gid = cv.take<int>((size_t)B * list_cap);
    int *first = cv.take<int>((size_t)B * (list_cap + 1));
    if (!cv.ok()) {
        set_error("merge_groups_runs: workspace too small (%zu < %zu)", workspace_bytes, cv.off);
        return PCSEG_ERR_WORKSPACE;
    }
    PCSEG_LAUNCH(merge_keys_kernel, dim3((list_cap + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, run_parent,
                 (const long long *)stats, region_list, n_list, key, H, W, cap, list_cap, 1, (const unsigned *)dilated_bits);
    PCSEG_CHECK_LAUNCH();
    PCSEG_LAUNCH(merge_groups_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, n_list, group_of, n_groups, key, first, gid, list_cap);
    PCSEG_CHECK_LAUNCH();
    return PCSEG_OK;
}

int pcseg_merge_groups_fused_multi(const uint32_t *dilated_bits, const int32_t *run_parent, const int64_t *stats,
                                   const int32_t *region_lists, const int32_t *n_lists, const int32_t *slots, int n_masks, int n_slots,
                                   int32_t *group_of, int32_t *n_groups, int64_t *group_stats, int B, int H, int W, int cap,
                                   void *workspace, size_t workspace_bytes, pcseg_stream_t stream)
{
    PCSEG_REQUIRE(dilated_bits && run_parent && stats && region_lists && n_lists && slots && group_of && n_groups && group_stats &&
                      workspace && cap >= 1 && n_slots >= 1 && n_masks >= 1 && n_masks <= 4 && check_shape(B, H, W),
                  "bad arguments (1..4 masks)");
    MergeSlots ms;
    for (int m = 0; m < 4; ++m) {
        ms.slot[m] = m < n_masks ? slots[m] : 0;
        PCSEG_REQUIRE(ms.slot[m] >= 0 && ms.slot[m] < n_slots, "list slot out of range");
    }
    const int list_cap = cap;
    const size_t BM = (size_t)B * n_masks;
    Carver cv(workspace, workspace_bytes);
    int *key = cv.take<int>(BM * list_cap);
    int *gid = cv.take<int>(BM * list_cap);
    int *first = cv.take<int>(BM * (list_cap + 1));
    if (!cv.ok()) {
        set_error("merge_groups_fused: workspace too small (%zu < %zu)", workspace_bytes, cv.off);
        return PCSEG_ERR_WORKSPACE;
    }
    PCSEG_LAUNCH(merge_fused_kernel, dim3(B, n_masks), dim3(MG_THREADS), 0, (hipStream_t)stream, (const unsigned *)dilated_bits,
                 (const int *)run_parent, (const long long *)stats, (const int *)region_lists, (const int *)n_lists, ms, n_slots,
                 (int *)group_of, (int *)n_groups, (long long *)group_stats, key, first, gid, H, W, cap, list_cap);
    PCSEG_CHECK_LAUNCH();
    return PCSEG_OK;
}

int pcseg_merge_groups_fused(const uint32_t *dilated_bits, const int32_t *run_parent, const int64_t *stats,
                             const int32_t *region_lists, const int32_t *n_lists, int slot, int n_slots, int32_t *group_of,
                             int32_t *n_groups, int64_t *group_stats, int B, int H, int W, int cap, void *workspace,
                             size_t workspace_bytes, pcseg_stream_t stream)
{
    const int32_t slots[1] = {slot};
    return pcseg_merge_groups_fused_multi(dilated_bits, run_parent, stats, region_lists, n_lists, slots, 1, n_slots, group_of, n_groups,
                                          group_stats, B, H, W, cap, workspace, workspace_bytes, stream);
}

int pcseg_classify_regions(const int64_t *stats, const uint8_t *cls_out, const int32_t *counts, const uint8_t *class_slot,
                           const uint8_t *class_particle, const int32_t *min_cell, const int32_t *min_cluster, int n_slots,
                           uint8_t *kind, uint8_t *slot_of, int32_t *cells, int64_t *particle_area, int64_t *type_stats,
                           int32_t *region_list, int32_t *n_list, int32_t *nan_flag, int B, int cap, pcseg_stream_t stream)
{
    PCSEG_REQUIRE(stats && cls_out && counts && class_slot && class_particle && min_cell && min_cluster && kind && slot_of &&
                      cells && particle_area && type_stats && region_list && n_list && nan_flag && B >= 1 && cap >= 1 &&
                      n_slots >= 0 && n_slots <= CLS_T,
                  "bad arguments (class tables are HOST arrays: slot[256], particle[256], min_cell[n], min_cluster[n])");
    ClassTables tab;
    memcpy(tab.slot, class_slot, 256);
    memcpy(tab.particle, class_particle, 256);
    for (int t = 0; t < CLS_T; ++t) {
        tab.min_cell[t] = t < n_slots ? min_cell[t] : 0x7FFFFFFF;
        tab.min_cluster[t] = t < n_slots ? min_cluster[t] : 0x7FFFFFFF;
    }
    tab.n_slots = n_slots;
    PCSEG_LAUNCH(classify_regions_kernel, dim3(B), dim3(CLS_THREADS), 0, (hipStream_t)stream, (const long long *)stats, cls_out,
                       counts, tab, kind, slot_of, cells, (long long *)particle_area, (long long *)type_stats, region_list, n_list,
                       nan_flag, cap);
    PCSEG_CHECK_LAUNCH();
    return PCSEG_OK;
}

int pcseg_group_reduce(const int64_t *stats, const int32_t *region_list, const int32_t *n_list, const int32_t *group_of,
                       const int32_t *n_groups, int64_t *group_stats, int B, int H, int W, int cap, int list_cap,
                       pcseg_stream_t stream)
{
    PCSEG_REQUIRE(stats && region_list && n_list && group_of && n_groups && group_stats && B >= 1 && cap >= 1 && list_cap >= 1,
                  "bad arguments");
    PCSEG_LAUNCH(group_reduce_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, (const long long *)stats, region_list,
                       n_list, group_of, n_groups, (long long *)group_stats, cap, list_cap, H, W);
    PCSEG_CHECK_LAUNCH();
    return PCSEG_OK;
}

int pcseg_nearest_dist_f64(const double *a, int na, const double *b, int nb, double *out_a, pcseg_stream_t stream)
{
    PCSEG_REQUIRE(a && b && out_a && na >= 1 && nb >= 1, "bad arguments");
    PCSEG_LAUNCH(nearest_kernel, dim3((na + 255) / 256), dim3(256), 0, (hipStream_t)stream, a, na, b, nb, out_a);
    PCSEG_CHECK_LAUNCH();
    return PCSEG_OK;
}

}  // extern "C"

// Region-table building blocks shared by the reduction kernels (reduce.hip) and the label pass that accumulates the
// table while it writes the final labels (ccl.hip, ccl_relabel_stats_kernel): the block's direct-mapped LDS table, the
// commit of a vertical run, the segmented lane reduction in front of it, the eight-lanes-per-row flush.
#pragma once
#include "common.h"

namespace pcseg {

constexpr int RED_SLOTS = 256;
constexpr int RED_MAXC = 8;

__device__ __forceinline__ void atomic_min_i64(long long *p, long long v) { atomicMin(p, v); }
__device__ __forceinline__ void atomic_max_i64(long long *p, long long v) { atomicMax(p, v); }

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


// flush of a block's LDS table: EIGHT LANES PER SLOT, one per column of the table row, so that one atomic instruction carries
// up to eight neighbouring 8-byte words of a row's 64-byte line (the three adds, the three mins, the two maxes) instead of 64
// lanes aiming at 64 different lines eight times over.  All 256 threads of the block call it, after a barrier.
__device__ __forceinline__ void region_slots_flush8(const int *tags, const int (*lstat)[8], long long *gst)
{
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

// host-side launchers of reduce.hip's table kernels for other translation units
int region_init_launch(const int *counts, int cap, int C, int B, int H, int W, long long *stats, double *sums, hipStream_t s);
int region_class_launch(const long long *stats, const uint8_t *cls, const int *counts, uint8_t *cls_out, int cap, int B, int H, int W,
                        hipStream_t s);

}  // namespace pcseg

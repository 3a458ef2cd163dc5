// Per-label reductions: regionprops fields (A3), per-ROI isotope sums (M1),
// region classification / cluster cell counts (A3/A4 tail) and the proximity
// merge grouping (A6 tail).
//
// The reduction walks rows: one wave covers 64 consecutive pixels of a row,
// finds label runs with a ballot, and the head lane of each run adds closed-form
// run sums (length, row*length, column arithmetic series, bbox) -- integers only,
// so the result is independent of the order of the atomics.  A 256-slot
// direct-mapped LDS table absorbs the hot labels (background, particle) of the
// block's rows and is flushed once per block.
#include "common.h"

namespace pcseg {

constexpr int RED_SLOTS = 256;
constexpr int RED_MAXC = 8;
constexpr int RED_ROWS = 16;  // rows per block

__device__ __forceinline__ void atomic_min_i64(long long *p, long long v) { atomicMin(p, v); }
__device__ __forceinline__ void atomic_max_i64(long long *p, long long v) { atomicMax(p, v); }

__global__ void __launch_bounds__(256) region_init_kernel(long long *__restrict__ stats, double *__restrict__ sums,
                                                           const int *__restrict__ counts, int cap, int C, int H, int W)
{
    const int b = blockIdx.y;
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    int nl = counts ? min(counts[b], cap) : cap;
    if (l >= nl) return;
    long long *t = stats + ((int64_t)b * cap + l) * 8;
    t[0] = 0; t[1] = 0; t[2] = 0; t[3] = H; t[4] = W; t[5] = 0; t[6] = 0; t[7] = 0x7FFFFFFFFFFFFFFFLL;
    if (sums)
        for (int k = 0; k < C; ++k) sums[((int64_t)b * cap + l) * C + k] = 0.0;
}

template <bool HAS_PLANES>
__global__ void __launch_bounds__(256) region_reduce_kernel(const int *__restrict__ labels, const float *__restrict__ planes,
                                                             int C, int H, int W, int cap, long long *__restrict__ stats,
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
        const int l = inb ? lab[(int64_t)r * W + c] : 0;
        float v[RED_MAXC];
        if (HAS_PLANES) {
#pragma unroll
            for (int k = 0; k < RED_MAXC; ++k) v[k] = (k < C && inb && l > 0) ? pl[(int64_t)k * n + (int64_t)r * W + c] : 0.f;
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
                            if (k < C) atomicAdd(&gsum[(int64_t)(l - 1) * C + k], acc[k]);
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

__global__ void __launch_bounds__(256) merge_groups_kernel(const int *__restrict__ dl, const long long *__restrict__ stats,
                                                            const uint8_t *__restrict__ select, const int *__restrict__ n_regions,
                                                            int *__restrict__ group_of, int *__restrict__ n_groups,
                                                            int *__restrict__ key_ws, int *__restrict__ first_ws,
                                                            int *__restrict__ gid_ws, int H, int W, int cap)
{
    __shared__ int wsum[4];
    const int b = blockIdx.x;
    const int R = min(n_regions[b], cap);
    const int64_t n = (int64_t)H * W;
    const int *dlab = dl + (int64_t)b * n;
    const long long *st = stats + (int64_t)b * cap * 8;
    const uint8_t *sel = select + (int64_t)b * cap;
    int *key = key_ws + (int64_t)b * cap;
    int *first = first_ws + (int64_t)b * (cap + 1);
    int *gid = gid_ws + (int64_t)b * cap;
    int *gof = group_of + (int64_t)b * cap;
    // keys (tiff_analysis.py:844-848): dilated label at (int(cy), int(cx)); exact with integer floor division
    for (int r = threadIdx.x; r < R; r += 256) {
        int k = 0;
        if (sel[r]) {
            long long a = st[(int64_t)r * 8 + 0];
            if (a > 0) {
                long long y = st[(int64_t)r * 8 + 1] / a, x = st[(int64_t)r * 8 + 2] / a;
                if (y >= 0 && y < H && x >= 0 && x < W) k = dlab[y * W + x];
            }
        }
        if (k > cap) k = 0;  // cannot happen: a dilated image has no more components than regions
        key[r] = k;
        if (k > 0) first[k] = 0x7FFFFFFF;
    }
    __syncthreads();
    for (int r = threadIdx.x; r < R; r += 256)
        if (key[r] > 0) atomicMin(&first[key[r]], r);
    __syncthreads();
    int carry = 0;
    for (int base = 0; base < R; base += 256) {
        int r = base + threadIdx.x;
        int leader = (r < R && key[r] > 0 && first[key[r]] == r) ? 1 : 0;
        int total;
        int ex = block_exclusive_scan256(leader, &total, wsum);
        if (leader) gid[r] = carry + ex + 1;
        carry += total;
    }
    __syncthreads();
    for (int r = threadIdx.x; r < R; r += 256) gof[r] = key[r] > 0 ? gid[first[key[r]]] : 0;
    if (threadIdx.x == 0) n_groups[b] = carry;
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
    PCSEG_REQUIRE(labels && stats && cap >= 1 && check_shape(B, H, W), "bad arguments");
    PCSEG_REQUIRE((!planes && !sums) || (planes && sums && C >= 1 && C <= RED_MAXC), "planes/sums/C mismatch (C <= 8)");
    PCSEG_REQUIRE(!cls || cls_out, "cls needs cls_out");
    hipStream_t s = (hipStream_t)stream;
    if (overflow) PCSEG_CHECK_HIP(hipMemsetAsync(overflow, 0, sizeof(int32_t) * B, s));
    dim3 gi((cap + 255) / 256, B);
    hipLaunchKernelGGL(region_init_kernel, gi, dim3(256), 0, s, (long long *)stats, sums, counts, cap, C, H, W);
    PCSEG_CHECK_LAUNCH();
    dim3 grid((H + RED_ROWS - 1) / RED_ROWS, B);
    if (planes)
        hipLaunchKernelGGL(region_reduce_kernel<true>, grid, dim3(256), 0, s, labels, planes, C, H, W, cap, (long long *)stats,
                           sums, overflow);
    else
        hipLaunchKernelGGL(region_reduce_kernel<false>, grid, dim3(256), 0, s, labels, planes, C, H, W, cap,
                           (long long *)stats, sums, overflow);
    PCSEG_CHECK_LAUNCH();
    if (cls) {
        hipLaunchKernelGGL(region_class_kernel, gi, dim3(256), 0, s, (const long long *)stats, cls, counts, cls_out, cap,
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

size_t pcseg_merge_groups_workspace_bytes(int B, int cap)
{
    if (B < 1 || cap < 1) return 0;
    return align_up(sizeof(int) * (size_t)B * cap) * 2 + align_up(sizeof(int) * (size_t)B * (cap + 1));
}

int pcseg_merge_groups(const int32_t *dilated_labels, const int64_t *stats, const uint8_t *select, const int32_t *n_regions,
                       int32_t *group_of, int32_t *n_groups, int B, int H, int W, int cap, void *workspace,
                       size_t workspace_bytes, pcseg_stream_t stream)
{
    PCSEG_REQUIRE(dilated_labels && stats && select && n_regions && group_of && n_groups && workspace && cap >= 1 &&
                      check_shape(B, H, W),
                  "bad arguments");
    Carver cv(workspace, workspace_bytes);
    int *key = cv.take<int>((size_t)B * cap);
    int *gid = cv.take<int>((size_t)B * cap);
    int *first = cv.take<int>((size_t)B * (cap + 1));
    if (!cv.ok()) {
        set_error("merge_groups: workspace too small (%zu < %zu)", workspace_bytes, cv.off);
        return PCSEG_ERR_WORKSPACE;
    }
    hipLaunchKernelGGL(merge_groups_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, dilated_labels,
                       (const long long *)stats, select, n_regions, group_of, n_groups, key, first, gid, H, W, cap);
    PCSEG_CHECK_LAUNCH();
    return PCSEG_OK;
}

}  // extern "C"

// Per-ROI / per-cell / per-group table assembly on the device (SURVEY.md 8b "ROI-mask / per-ROI-table output", 8e:
// the table is what the ranks all-gather).  The fixed-capacity per-frame tables the reductions leave behind
// ((B, cap, ...) with a row count per frame) are compacted into dense float64 row tables:
//   rois    one row per refined ROI with a pixel          [frame, label, area, centroid_row, centroid_col, S_0..S_C-1, ratios]
//   cells   one row per cell / cluster region             [frame, label, class, kind, area, centroid (2), bbox (4), cells,
//                                                          group, group_combined, S_0..S_C-1, ratios]
//   groups  one row per merged group (tiff_analysis.py:855-872)  [frame, slot, group, area, centroid (2), bbox (4), members]
//   frames  one int64 record per frame: the integer ingredients of get_cell_counts_and_densities (:1018-1038); the two
//           round(x, 5) of that function are Python's decimal rounding and stay a B-row host epilogue
// Row order is the reference's: frames in batch order, labels ascending.  Three launches: per-frame row counts,
// one-block scan over the frames, per-frame writers (block scan over the frame's rows gives every row its slot).
#include "common.h"

namespace pcseg {

constexpr int TB_SLOTS = 5;  // CLS_T cell-type slots + the "combined" list

struct ClassSlots {
    uint8_t slot[256];  // class value -> cell-type slot, 255 = not a cell type
};

__device__ __forceinline__ int tb_block_scan(int v, int *total, int *wsum)
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
    for (int w = 0; w < 4; ++w) {
        if (w < wid) base += wsum[w];
        tot += wsum[w];
    }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

// counts[b] = {roi rows, cell rows, group rows}
__global__ void __launch_bounds__(256) table_count_kernel(pcseg_table_inputs in, long long *__restrict__ counts)
{
    __shared__ int wsum[4];
    const int b = blockIdx.x;
    int n_roi = 0, n_cell = 0;
    const int m = min(in.n_markers[b], in.cap), n = min(in.counts[b], in.cap);
    for (int r = threadIdx.x; r < m; r += 256) n_roi += in.ws_stats[((int64_t)b * in.cap + r) * 8] > 0;
    for (int r = threadIdx.x; r < n; r += 256) n_cell += in.kind[(int64_t)b * in.cap + r] > 0;
    int tot_roi, tot_cell;
    tb_block_scan(n_roi, &tot_roi, wsum);
    tb_block_scan(n_cell, &tot_cell, wsum);
    if (threadIdx.x == 0) {
        long long g = 0;
        for (int s = 0; s < TB_SLOTS; ++s)
            if (in.n_groups[s]) g += in.n_groups[s][b];
        counts[b * 3 + 0] = tot_roi;
        counts[b * 3 + 1] = tot_cell;
        counts[b * 3 + 2] = g;
    }
}

// offsets[b] = exclusive prefix over the frames; totals[0..2] = rows of rois / cells / groups, totals[3..5] = number of
// frames whose region table overflowed (class components, refined ROIs) and of frames on which the reference raises
// int(NaN): the caller's checks ride on the one read-back that sizes the tables
__global__ void __launch_bounds__(64) table_scan_kernel(pcseg_table_inputs in, const long long *__restrict__ counts,
                                                         long long *__restrict__ offsets, long long *__restrict__ totals, int B)
{
    const int t = threadIdx.x;
    if (t < 3) {
        long long acc = 0;
        for (int b = 0; b < B; ++b) {
            offsets[b * 3 + t] = acc;
            acc += counts[b * 3 + t];
        }
        totals[t] = acc;
    } else if (t < 6) {
        const int32_t *flag = t == 3 ? in.overflow : (t == 4 ? in.ws_overflow : in.nan_flag);
        long long acc = 0;
        for (int b = 0; flag && b < B; ++b) acc += flag[b] != 0;
        totals[t] = acc;
    }
}

__device__ __forceinline__ void tb_ratios(const pcseg_table_inputs &in, const double *s, double *row)
{
    for (int k = 0; k < in.n_ratios; ++k) {
        double d = 0.0;
        bool ok = in.ratio_num[k] < in.C;
        for (int j = 0; j < 4; ++j) {
            const int p = in.ratio_den[k][j];
            if (p < 0) continue;
            if (p >= in.C) { ok = false; continue; }
            d = __dadd_rn(d, s[p]);
        }
        row[k] = ok ? __ddiv_rn(s[in.ratio_num[k]], d) : __longlong_as_double(0x7FF8000000000000LL);
    }
}

__global__ void __launch_bounds__(256) table_write_kernel(pcseg_table_inputs in, const long long *__restrict__ offsets,
                                                           double *__restrict__ rois, double *__restrict__ cells,
                                                           double *__restrict__ groups, long long *__restrict__ frames,
                                                           int *__restrict__ own_ws, int *__restrict__ comb_ws)
{
    __shared__ int wsum[4];
    __shared__ long long s_clu_cells[4], s_clu_area[4];
    const int b = blockIdx.x;
    const int cap = in.cap, C = in.C, nr = in.n_ratios;
    const double fid = (double)in.frame_ids[b];
    const int m = min(in.n_markers[b], cap), n = min(in.counts[b], cap);
    // ---- rois
    {
        const int ncol = 5 + C + nr;
        double *out = rois + offsets[b * 3 + 0] * ncol;
        int carry = 0;
        for (int base = 0; base < m; base += 256) {
            const int r = base + threadIdx.x;
            const int64_t *st = in.ws_stats + ((int64_t)b * cap + (r < m ? r : 0)) * 8;
            const int valid = r < m && st[0] > 0;
            int total;
            const int pos = carry + tb_block_scan(valid, &total, wsum);
            if (valid) {
                double *row = out + (int64_t)pos * ncol;
                const double a = (double)st[0];
                row[0] = fid; row[1] = (double)(r + 1); row[2] = a;
                row[3] = __ddiv_rn((double)st[1], a); row[4] = __ddiv_rn((double)st[2], a);
                const double *s = in.ws_sums + ((int64_t)b * cap + r) * C;
                for (int k = 0; k < C; ++k) row[5 + k] = s[k];
                tb_ratios(in, s, row + 5 + C);
            }
            carry += total;
        }
    }
    // ---- group membership of the regions (tiff_analysis.py:867-872: "regions" of a merged group), own type and combined
    int *own = own_ws + (int64_t)b * cap, *comb = comb_ws + (int64_t)b * cap;
    for (int r = threadIdx.x; r < n; r += 256) { own[r] = 0; comb[r] = 0; }
    __syncthreads();
    for (int s = 0; s < TB_SLOTS; ++s) {
        if (!in.group_of[s]) continue;
        const int k_n = min(in.n_list[b * TB_SLOTS + s], cap);
        const int *lst = in.region_list + ((int64_t)b * TB_SLOTS + s) * cap;
        const int *gof = in.group_of[s] + (int64_t)b * cap;
        int *tgt = s == TB_SLOTS - 1 ? comb : own;
        for (int k = threadIdx.x; k < k_n; k += 256) {
            const int r = lst[k];
            if (r >= 0 && r < n) tgt[r] = gof[k];
        }
    }
    __syncthreads();
    // ---- cells
    {
        const int ncol = 14 + C + nr;
        double *out = cells + offsets[b * 3 + 1] * ncol;
        int carry = 0;
        for (int base = 0; base < n; base += 256) {
            const int r = base + threadIdx.x;
            const int valid = r < n && in.kind[(int64_t)b * cap + r] > 0;
            int total;
            const int pos = carry + tb_block_scan(valid, &total, wsum);
            if (valid) {
                const int64_t *st = in.stats + ((int64_t)b * cap + r) * 8;
                double *row = out + (int64_t)pos * ncol;
                const double a = (double)st[0];
                row[0] = fid; row[1] = (double)(r + 1);
                row[2] = (double)in.cls_out[(int64_t)b * cap + r]; row[3] = (double)in.kind[(int64_t)b * cap + r];
                row[4] = a; row[5] = __ddiv_rn((double)st[1], a); row[6] = __ddiv_rn((double)st[2], a);
                row[7] = (double)st[3]; row[8] = (double)st[4]; row[9] = (double)st[5]; row[10] = (double)st[6];
                row[11] = (double)in.cells[(int64_t)b * cap + r]; row[12] = (double)own[r]; row[13] = (double)comb[r];
                const double *s = in.cc_sums + ((int64_t)b * cap + r) * C;
                for (int k = 0; k < C; ++k) row[14 + k] = s[k];
                tb_ratios(in, s, row + 14 + C);
            }
            carry += total;
        }
    }
    // ---- groups: slots in the order the reference's dict holds them (types, then "combined")
    {
        double *out = groups + offsets[b * 3 + 2] * 11;
        int done = 0;
        for (int s = 0; s < TB_SLOTS; ++s) {
            if (!in.n_groups[s]) continue;
            const int G = in.n_groups[s][b];
            const int64_t *gst = in.group_stats[s] + (int64_t)b * cap * 8;
            for (int g = threadIdx.x; g < G; g += 256) {
                const int64_t *t = gst + (int64_t)g * 8;
                double *row = out + (int64_t)(done + g) * 11;
                const double a = (double)t[0];
                row[0] = fid; row[1] = (double)s; row[2] = (double)(g + 1); row[3] = a;
                row[4] = __ddiv_rn((double)t[1], a); row[5] = __ddiv_rn((double)t[2], a);
                row[6] = (double)t[3]; row[7] = (double)t[4]; row[8] = (double)t[5]; row[9] = (double)t[6]; row[10] = (double)t[7];
            }
            done += G;
        }
    }
    // ---- frames: [n_labels, n_rois, particle_area, particle_area + overlap, tie_flag, then per slot
    //              present, count = n_cells + sum(cluster.cells), area_px = sum(cell areas) + sum(cluster areas)]
    if (threadIdx.x < 4) { s_clu_cells[threadIdx.x] = 0; s_clu_area[threadIdx.x] = 0; }
    __syncthreads();
    for (int r = threadIdx.x; r < n; r += 256) {
        if (in.kind[(int64_t)b * cap + r] != 2) continue;
        const int s = in.slot_of[(int64_t)b * cap + r];
        if (s >= 4) continue;
        atomicAdd((unsigned long long *)&s_clu_cells[s], (unsigned long long)(long long)in.cells[(int64_t)b * cap + r]);
        atomicAdd((unsigned long long *)&s_clu_area[s], (unsigned long long)in.stats[((int64_t)b * cap + r) * 8]);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        long long *f = frames + (int64_t)b * (5 + 3 * 4);
        f[0] = n; f[1] = m; f[2] = in.particle_area[b]; f[3] = in.particle_area[b] + in.overlap_area[b]; f[4] = in.tie_flags[b];
        for (int s = 0; s < 4; ++s) {
            const int64_t *ts = in.type_stats + ((int64_t)b * 4 + s) * 4;  // n_cells, n_clusters, sum of cell areas, first region
            f[5 + 3 * s + 0] = ts[3] != 0x7FFFFFFF;
            f[5 + 3 * s + 1] = ts[0] + s_clu_cells[s];
            f[5 + 3 * s + 2] = ts[2] + s_clu_area[s];
        }
    }
}

// C14 for a whole batch (.m:260-268): for every cell / cluster row of type slot 0 or 1, the distance to the nearest row of
// the OTHER of the two types in the same frame, over the dense `cells` table that table_write_kernel has just written
// (rows of a frame are contiguous: offsets / counts of pcseg_table_layout).  Positions as MATLAB reports them:
// (x, y) = (centroid_col + 1, centroid_row + 1).  pdist2 + min: sqrt(min over the other set of dx^2 + dy^2), each product
// and the sum rounded on their own (no fused multiply-add), then / (512 / raster).  One block per frame; a frame lists a
// few hundred rows, every thread scans the frame's rows of the other type.  NaN marks a row that has no entry in the
// distance table (another type, or a frame in which one of the two types is absent).
__global__ void __launch_bounds__(256) cell_distance_kernel(const double *__restrict__ cells, int ncol, const long long *__restrict__ counts,
                                                             const long long *__restrict__ offsets, ClassSlots slots, double scale,
                                                             double *__restrict__ dist)
{
    __shared__ int s_n[2];
    const int b = blockIdx.x;
    const long long row0 = offsets[b * 3 + 1];
    const int n = (int)counts[b * 3 + 1];
    const double *rows = cells + row0 * ncol;
    if (threadIdx.x < 2) s_n[threadIdx.x] = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 256) {
        const int sl = slots.slot[(int)rows[(int64_t)i * ncol + 2] & 255];
        if (sl < 2) atomicAdd(&s_n[sl], 1);
    }
    __syncthreads();
    const bool both = s_n[0] > 0 && s_n[1] > 0;
    const double nan = __longlong_as_double(0x7FF8000000000000LL);
    for (int i = threadIdx.x; i < n; i += 256) {
        const double *ri = rows + (int64_t)i * ncol;
        const int sl = slots.slot[(int)ri[2] & 255];
        double out = nan;
        if (both && sl < 2) {
            const double x = ri[6] + 1.0, y = ri[5] + 1.0;
            double best = __longlong_as_double(0x7FF0000000000000LL);
            for (int j = 0; j < n; ++j) {
                const double *rj = rows + (int64_t)j * ncol;
                if (slots.slot[(int)rj[2] & 255] != 1 - sl) continue;
                const double dx = x - (rj[6] + 1.0), dy = y - (rj[5] + 1.0);
                const double d2 = __dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy));
                best = d2 < best ? d2 : best;
            }
            out = __ddiv_rn(__dsqrt_rn(best), scale);
        }
        dist[row0 + i] = out;
    }
}

}  // namespace pcseg

using namespace pcseg;

extern "C" {

size_t pcseg_table_workspace_bytes(int B, int cap)
{
    if (B < 1 || cap < 1) return 0;
    return 2 * align_up(sizeof(long long) * 3 * (size_t)B) + 2 * align_up(sizeof(int) * (size_t)B * cap);
}

static int table_check(const pcseg_table_inputs *in)
{
    return in && in->B >= 1 && in->cap >= 1 && in->C >= 0 && in->C <= 8 && in->n_ratios >= 0 && in->n_ratios <= 8 && in->frame_ids &&
           in->counts && in->stats && in->cls_out && in->cc_sums && in->kind && in->slot_of && in->cells && in->particle_area &&
           in->overlap_area && in->type_stats && in->tie_flags && in->region_list && in->n_list && in->n_markers && in->ws_stats &&
           in->ws_sums;
}

int pcseg_table_layout(const pcseg_table_inputs *in, int64_t *totals, void *workspace, size_t workspace_bytes, pcseg_stream_t stream)
{
    PCSEG_REQUIRE(table_check(in) && totals && workspace, "bad arguments");
    Carver cv(workspace, workspace_bytes);
    long long *counts = cv.take<long long>(3 * (size_t)in->B);
    long long *offsets = cv.take<long long>(3 * (size_t)in->B);
    if (!cv.ok()) {
        set_error("table_layout: workspace too small (%zu < %zu)", workspace_bytes, cv.off);
        return PCSEG_ERR_WORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    PCSEG_LAUNCH(table_count_kernel, dim3(in->B), dim3(256), 0, s, *in, counts);
    PCSEG_CHECK_LAUNCH();
    PCSEG_LAUNCH(table_scan_kernel, dim3(1), dim3(64), 0, s, *in, (const long long *)counts, offsets, (long long *)totals, in->B);
    PCSEG_CHECK_LAUNCH();
    return PCSEG_OK;
}

int pcseg_table_write(const pcseg_table_inputs *in, double *rois, double *cells, double *groups, int64_t *frames,
                      void *workspace, size_t workspace_bytes, pcseg_stream_t stream)
{
    PCSEG_REQUIRE(table_check(in) && rois && cells && groups && frames && workspace, "bad arguments");
    Carver cv(workspace, workspace_bytes);
    cv.take<long long>(3 * (size_t)in->B);
    long long *offsets = cv.take<long long>(3 * (size_t)in->B);
    int *own = cv.take<int>((size_t)in->B * in->cap);
    int *comb = cv.take<int>((size_t)in->B * in->cap);
    if (!cv.ok()) {
        set_error("table_write: workspace too small (%zu < %zu)", workspace_bytes, cv.off);
        return PCSEG_ERR_WORKSPACE;
    }
    PCSEG_LAUNCH(table_write_kernel, dim3(in->B), dim3(256), 0, (hipStream_t)stream, *in, (const long long *)offsets, rois, cells,
                 groups, (long long *)frames, own, comb);
    PCSEG_CHECK_LAUNCH();
    return PCSEG_OK;
}

int pcseg_cell_distances(const double *cells, int64_t n_rows, int ncol, const uint8_t *class_slot, double raster, double size,
                         double *dist, int B, const void *workspace, size_t workspace_bytes, pcseg_stream_t stream)
{
    PCSEG_REQUIRE(cells && class_slot && dist && workspace && B >= 1 && ncol >= 14 && n_rows >= 0 && raster > 0.0 && size > 0.0,
                  "bad arguments");
    Carver cv(const_cast<void *>(workspace), workspace_bytes);
    const long long *counts = cv.take<long long>(3 * (size_t)B);
    const long long *offsets = cv.take<long long>(3 * (size_t)B);
    if (!cv.ok()) {
        set_error("cell_distances: workspace too small (%zu < %zu)", workspace_bytes, cv.off);
        return PCSEG_ERR_WORKSPACE;
    }
    ClassSlots slots;
    memcpy(slots.slot, class_slot, 256);
    PCSEG_LAUNCH(cell_distance_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, cells, ncol, counts, offsets, slots, size / raster,
                 dist);
    PCSEG_CHECK_LAUNCH();
    return PCSEG_OK;
}

}  // extern "C"

/*
 * pcseg.h -- C ABI of libpcseg.so, the MI355X (gfx950) implementation of the
 * per-frame segmentation hot path of ssilverman16/particle_col_image_segmentation.
 *
 * The reference has NO FFI of its own (it is pure Python calling scipy /
 * scikit-image); each entry point below replaces the library call or loop the
 * reference makes at the cited file:line.  The Python host layer
 * (particle_col_image_segmentation_amd/) binds these with ctypes; the stub a
 * reference maintainer would add is shown in INTEGRATION.md.
 *
 * Conventions
 *   - every image argument is a DEVICE pointer to a contiguous batch-major,
 *     row-major array (B, H, W) or (B, C, H, W); buffers are caller-owned
 *     (torch tensors' data_ptr()); the library allocates nothing persistent;
 *   - scratch comes from a caller-provided workspace sized by the matching
 *     *_workspace_bytes(B, H, W) query (256-byte aligned device memory);
 *   - work is enqueued on `stream` (a hipStream_t) and every compute entry point
 *     returns without waiting for it: no hidden host synchronisation, fixed
 *     points included (they finish in device-side tail kernels);
 *   - return value: PCSEG_OK or a negative pcseg_status; the message is in
 *     pcseg_last_error() (thread local);
 *   - there is no CPU fallback anywhere: without a HIP device every compute
 *     entry point returns PCSEG_ERR_HIP.
 */
#ifndef PCSEG_H
#define PCSEG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *pcseg_stream_t; /* hipStream_t */

enum pcseg_status {
    PCSEG_OK = 0,
    PCSEG_ERR_ARG = -1,       /* bad shape / null pointer / unsupported size */
    PCSEG_ERR_HIP = -2,       /* a HIP call or launch failed */
    PCSEG_ERR_WORKSPACE = -3, /* workspace too small */
    PCSEG_ERR_CAPACITY = -4   /* more labels than the caller's table capacity */
};

/* layout of one row of the region table (int64 each) -- skimage regionprops
 * fields the reference consumes: area (tiff_analysis.py:769-781), centroid sums
 * (:406,844,1054), bbox half-open (:860-863), raster-first pixel (:1041-1044) */
enum pcseg_region_col {
    PCSEG_R_AREA = 0, PCSEG_R_SUM_ROW = 1, PCSEG_R_SUM_COL = 2,
    PCSEG_R_MIN_ROW = 3, PCSEG_R_MIN_COL = 4, PCSEG_R_MAX_ROW1 = 5, PCSEG_R_MAX_COL1 = 6,
    PCSEG_R_FIRST = 7, PCSEG_R_NCOLS = 8
};

int pcseg_version(void);
const char *pcseg_last_error(void);
int pcseg_device_count(void);

/* ---- measurement aid (bench.py's roofline leg): when enabled, every kernel launch is bracketed by hipEvents
 * recorded on the launch stream.  pcseg_timing_report waits for them and writes one line per kernel
 * "name<TAB>launches<TAB>total_ms"; it returns the number of bytes written (needed size when buf is NULL)
 * and clears the records.  Off by default. */
void pcseg_timing_enable(int on);
int pcseg_timing_report(char *buf, size_t buf_bytes);

/* ---- ingest: class map = argmax over the C planes + 1 (what ilastik's
 * "Simple Segmentation" export holds; read at tiff_analysis.py:118-121, 639-642) */
int pcseg_argmax_planes_f32(const float *stack, uint8_t *cls, int B, int C, int H, int W, pcseg_stream_t stream);

/* ---- A1: scipy.ndimage.median_filter(ds_arr, size=5), mode='reflect'
 * (tiff_analysis.py:122, 643) */
int pcseg_median5_u8(const uint8_t *in, uint8_t *out, int B, int H, int W, pcseg_stream_t stream);

/* ---- ingest + A1 + A2 in one call: class map (argmax + 1) -> median_filter(size=5) -> label(z_slice)
 * (tiff_analysis.py:639-643, 743).  Same results as pcseg_argmax_planes_f32, pcseg_median5_u8, pcseg_ccl8_equal_u8 in
 * sequence, but for C <= 5 planes the three tile passes are ONE kernel: the raw class map is never written, the medians
 * are labelled while they sit in LDS.  denoised: uint8 (B,H,W) = the median-filtered class map (what the reference
 * calls ds_arr / z_slice); labels / counts as pcseg_ccl8_equal_u8. */
size_t pcseg_classmap_label_workspace_bytes(int B, int H, int W);
int pcseg_classmap_label_f32(const float *stack, int C, uint8_t *denoised, int32_t *labels, int32_t *counts,
                             int B, int H, int W, void *workspace, size_t workspace_bytes, pcseg_stream_t stream);

/* ---- A2: skimage.measure.label (tiff_analysis.py:743, 260, 829;
 * refine_boundaries.py:64) and scipy.ndimage.label (inside binary_fill_holes).
 * labels: int32 (B,H,W), 0 = background, 1..N in raster order of each
 * component's first pixel; counts[b] = N of frame b (device int32[B]). */
size_t pcseg_ccl_workspace_bytes(int B, int H, int W);
int pcseg_ccl8_equal_u8(const uint8_t *in, int32_t *labels, int32_t *counts, int B, int H, int W,
                        void *workspace, size_t workspace_bytes, pcseg_stream_t stream);
int pcseg_ccl8_bool(const uint8_t *in, int32_t *labels, int32_t *counts, int B, int H, int W,
                    void *workspace, size_t workspace_bytes, pcseg_stream_t stream);
int pcseg_ccl4_bool(const uint8_t *in, int32_t *labels, int32_t *counts, int B, int H, int W,
                    void *workspace, size_t workspace_bytes, pcseg_stream_t stream);
/* renumber a root image (value = linear index of the component's first pixel
 * + 1, 0 = background) into 1..N raster order; `labels` must not alias `roots`. */
int pcseg_compact_labels(const int32_t *roots, int32_t *labels, int32_t *counts, int B, int H, int W,
                         void *workspace, size_t workspace_bytes, pcseg_stream_t stream);

/* ---- A3 + M1: regionprops fields and per-ROI isotope sums
 * (tiff_analysis.py:746-773, 1041-1044; .m:122-135, 186-199).
 * stats: int64 (B, cap, 8) rows as pcseg_region_col; cls_out: uint8 (B, cap) =
 * class-map value at the raster-first pixel (NULL if cls is NULL); sums:
 * float64 (B, cap, C) (NULL if planes is NULL).  Labels above cap are dropped
 * and overflow[b] (device int32[B], may be NULL) is set. */
int pcseg_region_reduce(const int32_t *labels, const uint8_t *cls, const float *planes, int C,
                        int B, int H, int W, int cap, int64_t *stats, uint8_t *cls_out, double *sums,
                        int32_t *overflow, pcseg_stream_t stream);
/* same, but only rows l < counts[b] (device int32[B], e.g. from pcseg_ccl*) are
 * initialised and filled; rows beyond are left untouched. */
int pcseg_region_reduce_n(const int32_t *labels, const int32_t *counts, const uint8_t *cls, const float *planes,
                          int C, int B, int H, int W, int cap, int64_t *stats, uint8_t *cls_out, double *sums,
                          int32_t *overflow, pcseg_stream_t stream);
/* same, with the plane sums restricted to the pixels whose class-map value is in sum_class_bits (bit v = value v,
 * v < 64; 0 = every pixel): the regions of the other classes keep sums of 0 and their planes are not read.  The
 * reference only ever sums isotopes over cell ROIs (tiff_analysis.py:1041-1044). */
int pcseg_region_reduce_sel(const int32_t *labels, const int32_t *counts, const uint8_t *cls, uint64_t sum_class_bits,
                            const float *planes, int C, int B, int H, int W, int cap, int64_t *stats,
                            uint8_t *cls_out, double *sums, int32_t *overflow, pcseg_stream_t stream);

/* (pcseg_region_reduce_sel with planes == NULL but sums != NULL and C >= 1: the integer columns only, the first
 * counts[b] rows of sums are ZEROED -- for pcseg_region_sums2.)
 *
 * Table initialisation alone: the first min(counts[b], cap) rows of stats (B, cap, 8) get the neutral element of the
 * reduction (0 sums, empty bounding box), those of sums (B, cap, C) (C may be 0) are zeroed, overflow[b] (may be NULL)
 * cleared; counts == NULL: every row. */
int pcseg_region_init(const int32_t *counts, int cap, int C, int B, int H, int W, int64_t *stats, double *sums,
                      int32_t *overflow, pcseg_stream_t stream);

/* Plane sums of TWO label images over the same planes in ONE pass (.m:122-135 for the class-map components and for the
 * refined ROIs; the planes are the largest thing a reduction reads): sums_a (B, cap_a, C) += per-label sums of labels_a
 * restricted to the pixels whose class-map value is in sum_class_bits (0 = every pixel), sums_b (B, cap_b, C) += per-label
 * sums of labels_b.  stats_b != NULL: image B's integer columns (pcseg_region_col) are accumulated in the same walk into
 * stats_b (B, cap_b, 8) / overflow_b.  Every table must have been initialised (pcseg_region_init, or
 * pcseg_region_reduce_sel for image A); without stats_b labels above the capacity are skipped silently.  W % 4 == 0,
 * 16-byte aligned images. */
int pcseg_region_sums2(const int32_t *labels_a, const uint8_t *cls, uint64_t sum_class_bits, int cap_a, double *sums_a,
                       const int32_t *labels_b, int cap_b, double *sums_b, int64_t *stats_b, int32_t *overflow_b,
                       const float *planes, int C, int B, int H, int W, pcseg_stream_t stream);

/* ---- R1: binary_mask = boundary_map < threshold (refine_boundaries.py:44-45) */
int pcseg_threshold_lt_f32(const float *img, float threshold, uint8_t *mask, int B, int H, int W,
                           pcseg_stream_t stream);

/* ---- R2: scipy.ndimage.distance_transform_edt as exact integer squared
 * distance to the nearest zero pixel (refine_boundaries.py:60,
 * tiff_analysis.py:996); the float64 distance is sqrt((double)d2).  cap < 0:
 * exact everywhere; cap >= 0: values above cap are reported as cap + 1.  A
 * frame without any zero pixel follows scipy: virtual zero pixel at (-1, 0).
 * The threshold members of the EDT family (dilation, particle fill) stage 8 rows of uint16 distances plus the rows' bytes in
 * LDS: W <= 6800. */
size_t pcseg_edt_workspace_bytes(int B, int H, int W);
int pcseg_edt_sq_u8(const uint8_t *mask, int32_t *d2, int B, int H, int W, int cap,
                    void *workspace, size_t workspace_bytes, pcseg_stream_t stream);
/* fused R1 + R2: mask = img < threshold (also written to mask_out if not NULL).
 * frame_stride = float32 elements between consecutive frames of img (0 = H*W),
 * so that a plane of a (B,C,H,W) stack is read in place (refine_boundaries.py:34). */
int pcseg_edt_sq_lt_f32(const float *img, int64_t frame_stride, float threshold, int32_t *d2, uint8_t *mask_out,
                        int B, int H, int W, void *workspace, size_t workspace_bytes, pcseg_stream_t stream);

/* ---- A6: skimage.morphology.binary_dilation(binary, disk(radius))
 * (tiff_analysis.py:827-828, 990) with binary = ((value_bits >> in) & 1),
 * i.e. `z_slice == v` for one bit and the OR of several classes for several
 * (tiff_analysis.py:812, 816-818); out is 0/1. */
int pcseg_dilate_disk_u8(const uint8_t *in, uint64_t value_bits, int radius, uint8_t *out,
                         int B, int H, int W, void *workspace, size_t workspace_bytes, pcseg_stream_t stream);

/* ---- A6 fused: components of binary_dilation(((value_bits >> in) & 1), disk(radius)) as a union-find parent
 * image (int32 (B,H,W): linear index of a pixel of the same component, -1 = background), computed on a 1-bit image
 * (the dilation is shifts / ORs of 32-row column words).  Feeds pcseg_merge_groups(keys_are_roots = 1). */
size_t pcseg_dilate_ccl_workspace_bytes(int B, int H, int W);
int pcseg_dilate_ccl_roots_u8(const uint8_t *in, uint64_t value_bits, int radius, int32_t *roots,
                              int B, int H, int W, void *workspace, size_t workspace_bytes, pcseg_stream_t stream);
/* the same components WITHOUT a label image, for callers that only look the components up at a few pixels (the merge
 * step reads them at the region centroids, tiff_analysis.py:844-847): dilated_bits = the dilated mask as 32-row column
 * words, uint32 (B, ceil(H/32), W), bit j of word (ch, c) = pixel (32 ch + j, c); run_parent = union-find over the
 * vertical runs of set bits, int32 (B, H, W) of which ONLY the entries at the top pixel of each run (within its word)
 * are written and meaningful.  Feeds pcseg_merge_groups_runs. */
size_t pcseg_dilate_ccl_runs_workspace_bytes(int B, int H, int W);
int pcseg_dilate_ccl_runs_u8(const uint8_t *in, uint64_t value_bits, int radius, uint32_t *dilated_bits,
                             int32_t *run_parent, int B, int H, int W, void *workspace, size_t workspace_bytes,
                             pcseg_stream_t stream);
/* the same for n_masks (<= 4) masks of one class map at once -- get_cell_clusters_from_distances dilates and labels one
 * mask per cell type plus the union of all types (tiff_analysis.py:806-822): the map is read once, every pass behind the
 * bit planes is ONE launch over n_masks * B frames.  value_bits: HOST array [n_masks]; dilated_bits (n_masks, B,
 * ceil(H/32), W); run_parent (n_masks, B, H, W).  W % 4 == 0.  Workspace: pcseg_dilate_ccl_runs_workspace_bytes(B * n_masks, H, W). */
int pcseg_dilate_ccl_runs_multi_u8(const uint8_t *in, const uint64_t *value_bits, int n_masks, int radius,
                                   uint32_t *dilated_bits, int32_t *run_parent, int B, int H, int W,
                                   void *workspace, size_t workspace_bytes, pcseg_stream_t stream);

/* ---- A8: fill_particle_area (tiff_analysis.py:982-1015) in one pass pair:
 * out = ds with overlap pixels set to overlap_label, where overlap =
 * (ds == cell_label) & (EDT(ds != particle) < dist_threshold | dilate(ds ==
 * particle, disk(dilation_radius))); overlap_area[b] += count (device int64[B]). */
int pcseg_fill_particle(const uint8_t *ds, uint8_t *out, int particle_label, int cell_label,
                        int overlap_label, int dilation_radius, int dist_threshold, int64_t *overlap_area,
                        int B, int H, int W, void *workspace, size_t workspace_bytes, pcseg_stream_t stream);

/* ---- A7: scipy.ndimage.binary_fill_holes (tiff_analysis.py:880) */
size_t pcseg_fill_holes_workspace_bytes(int B, int H, int W);
int pcseg_fill_holes(const uint8_t *mask, uint8_t *out, int B, int H, int W,
                     void *workspace, size_t workspace_bytes, pcseg_stream_t stream);

/* ---- R3 + R4: skimage.morphology.local_maxima(distance) and
 * measure.label(local_max) (refine_boundaries.py:63-64) on an int32 image
 * (d2 is order-isomorphic to the float64 distance).  is_max: uint8 (NULL to
 * skip); markers: int32 1..K raster order (NULL to skip); counts: int32[B]. */
size_t pcseg_local_maxima_workspace_bytes(int B, int H, int W);
int pcseg_local_maxima_i32(const int32_t *img, uint8_t *is_max, int32_t *markers, int32_t *counts,
                           int B, int H, int W, void *workspace, size_t workspace_bytes, pcseg_stream_t stream);

/* ---- W1: skimage.segmentation.watershed(image, markers, mask=mask),
 * connectivity 1, no compactness, no watershed line (refine_boundaries.py:73).
 * Asynchronous like everything else: the two fixed points (minimax levels, second-level keys) run a fixed number
 * of grid rounds and finish in one-block-per-frame tail kernels, the frames that need the second level / the exact
 * flood are listed and counted on the device.
 * mode 0: parallel flood + proof check, frames that fail the check
 * are re-run by the exact sequential priority flood; mode 1: exact sequential
 * flood for every frame; mode 2: parallel flood only (tie_flags tells which
 * frames are NOT proven exact); add 4 to also run the explicit per-pixel proof
 * check (implied by the component test, kept for verification).  tie_flags:
 * device int32[B] (may be NULL).
 * frame_stride: as for pcseg_edt_sq_lt_f32 (0 = H*W). */
size_t pcseg_watershed_workspace_bytes(int B, int H, int W);
/* measurement aid: out[0] = 64x64 tiles the minimax relaxation actually processed (marked tiles over all rounds;
 * kept in a counter on the current device and read with a blocking copy, i.e. after everything queued so far),
 * out[1] = relaxation grid launches (including the rounds that find nothing marked), out[2] = watershed
 * calls since the last reset (process-wide).  reset != 0 synchronises the device. */
void pcseg_watershed_counters(int64_t *out, int reset);
int pcseg_watershed4_f32(const float *img, int64_t frame_stride, const int32_t *markers, const uint8_t *mask,
                         int32_t *out, int32_t *tie_flags, int B, int H, int W, int mode,
                         void *workspace, size_t workspace_bytes, pcseg_stream_t stream);

/* ---- A6 tail: get_merged_regions grouping (tiff_analysis.py:843-878).
 * region_list[b][k] (k < n_list[b]) is the reference's og_cell_regions as
 * 0-based row indices into the (B, cap, 8) stats table, in list order.  key =
 * dilated_labels at the truncated centroid; list entries sharing a non-zero
 * key form one group, groups numbered 1.. in the order of their first member;
 * group_of[b][k] = group id or 0 (centroid on a zero pixel -> dropped, :848).
 * region_list / group_of: int32 (B, list_cap); n_list / n_groups: int32[B].
 * keys_are_roots != 0: `dilated_labels` is the parent image of pcseg_dilate_ccl_roots_u8 (no numbering pass needed,
 * groups only need "same component"). */
size_t pcseg_merge_groups_workspace_bytes(int B, int list_cap);
int pcseg_merge_groups(const int32_t *dilated_labels, int keys_are_roots, const int64_t *stats,
                       const int32_t *region_list, const int32_t *n_list, int32_t *group_of, int32_t *n_groups,
                       int B, int H, int W, int cap, int list_cap, void *workspace, size_t workspace_bytes,
                       pcseg_stream_t stream);
/* the same grouping on the run-based components of pcseg_dilate_ccl_runs_u8 (workspace: pcseg_merge_groups_workspace_bytes) */
int pcseg_merge_groups_runs(const uint32_t *dilated_bits, const int32_t *run_parent, const int64_t *stats,
                            const int32_t *region_list, const int32_t *n_list, int32_t *group_of, int32_t *n_groups,
                            int B, int H, int W, int cap, int list_cap, void *workspace, size_t workspace_bytes,
                            pcseg_stream_t stream);

/* get_merged_regions' grouping AND the member sums of its groups (tiff_analysis.py:843-878) in ONE launch, on the
 * run-based components of pcseg_dilate_ccl_runs_u8: what pcseg_merge_groups_runs followed by pcseg_group_reduce
 * computes, with the list of type slot `slot` read in place from the (B, n_slots, cap) region lists and (B, n_slots)
 * list lengths pcseg_classify_regions writes (list capacity = cap).  group_of int32 (B, cap): every entry below the
 * list length is written; n_groups int32[B]; group_stats int64 (B, cap, 8): rows below n_groups[b] are written.
 * Workspace: pcseg_merge_groups_workspace_bytes(B, cap) (only touched by frames that list more than 4096 regions). */
int pcseg_merge_groups_fused(const uint32_t *dilated_bits, const int32_t *run_parent, const int64_t *stats,
                             const int32_t *region_lists, const int32_t *n_lists, int slot, int n_slots,
                             int32_t *group_of, int32_t *n_groups, int64_t *group_stats, int B, int H, int W, int cap,
                             void *workspace, size_t workspace_bytes, pcseg_stream_t stream);

/* pcseg_merge_groups_fused for n_masks (<= 4) masks in ONE launch: dilated_bits (n_masks, B, ceil(H/32), W) and run_parent
 * (n_masks, B, H, W) as pcseg_dilate_ccl_runs_multi_u8 leaves them, mask m grouped over the list of type slot slots[m]
 * (HOST array); group_of (n_masks, B, cap), n_groups (n_masks, B), group_stats (n_masks, B, cap, 8).
 * Workspace: pcseg_merge_groups_workspace_bytes(B * n_masks, cap). */
int pcseg_merge_groups_fused_multi(const uint32_t *dilated_bits, const int32_t *run_parent, const int64_t *stats,
                                   const int32_t *region_lists, const int32_t *n_lists, const int32_t *slots, int n_masks,
                                   int n_slots, int32_t *group_of, int32_t *n_groups, int64_t *group_stats, int B, int H,
                                   int W, int cap, void *workspace, size_t workspace_bytes, pcseg_stream_t stream);

/* member sums of the groups: group_stats int64 (B, list_cap, 8) = area, sum_row,
 * sum_col, min_row, min_col, max_row+1, max_col+1, members (tiff_analysis.py:855-872) */
int pcseg_group_reduce(const int64_t *stats, const int32_t *region_list, const int32_t *n_list,
                       const int32_t *group_of, const int32_t *n_groups, int64_t *group_stats,
                       int B, int H, int W, int cap, int list_cap, pcseg_stream_t stream);

/* ---- A3 tail + A4: the per-region loop of get_cell_positions_and_areas
 * (tiff_analysis.py:754-781) and the region lists get_cell_clusters_from_distances
 * builds (:794-796).  Class tables are HOST arrays: class_slot[256] maps a class
 * value to a cell-type slot (255 = none), class_particle[256] flags "Particle",
 * min_cell / min_cluster [n_slots] are MIN_CELL_AREA / MIN_CLUSTER_AREA (:54-60).
 * Outputs (device): kind (B,cap) 0 none / 1 cell / 2 cluster; slot_of (B,cap);
 * cells (B,cap) = 1 for cells, int(area // mean cell area) for clusters, -1 when
 * the reference would raise (clusters but no single cell; nan_flag[b] = 1);
 * particle_area int64[B]; type_stats int64 (B,4,4) = n_cells, n_clusters,
 * sum of cell areas, first region index; region_list int32 (B,5,cap): per slot
 * cells then clusters in label order, row 4 = the "combined" list (types in the
 * order of their first region); n_list int32 (B,5). */
int pcseg_classify_regions(const int64_t *stats, const uint8_t *cls_out, const int32_t *counts,
                           const uint8_t *class_slot, const uint8_t *class_particle,
                           const int32_t *min_cell, const int32_t *min_cluster, int n_slots,
                           uint8_t *kind, uint8_t *slot_of, int32_t *cells, int64_t *particle_area,
                           int64_t *type_stats, int32_t *region_list, int32_t *n_list, int32_t *nan_flag,
                           int B, int cap, pcseg_stream_t stream);

/* ---- C14: nearest distance from every point of a (na, 2) float64 set to a (nb, 2) set = min(pdist2(a, b), [], 2)
 * (.m:260-263 between the two ROI classes, .m:301-305 to the aggregate boundary); out_a: float64[na]. */
int pcseg_nearest_dist_f64(const double *a, int na, const double *b, int nb, double *out_a, pcseg_stream_t stream);

/* ---- C6: combine_cell_positions_and_clusters (tiff_analysis.py:252-287):
 * out = dapi with every 8-connected component of (dapi == 1) whose overlap
 * with (other == 1) exceeds `threshold` of its area set to 2. */
size_t pcseg_overlap_workspace_bytes(int B, int H, int W);
int pcseg_remove_overlapping(const uint8_t *dapi, const uint8_t *other, double threshold, uint8_t *out,
                             int B, int H, int W, void *workspace, size_t workspace_bytes,
                             pcseg_stream_t stream);

/* ---- per-ROI table output (SURVEY.md 8b output formats / 8e: what the ranks all-gather): the fixed-capacity
 * per-frame tables of the calls above, compacted on the device into dense float64 row tables.
 *   rois   (n, 5 + C + n_ratios)   frame, label, area, centroid_row, centroid_col, S_0.., ratios (.m:136-139 form:
 *                                  S[num] / sum of S[den...]); one row per refined ROI that owns a pixel
 *   cells  (n, 14 + C + n_ratios)  frame, label, class, kind (1 cell, 2 cluster), area, centroid (2), bbox (4), cells,
 *                                  group, group_combined, S_0.., ratios; one row per cell / cluster region
 *   groups (n, 11)                 frame, slot (4 = combined), group, area, centroid (2), bbox (4), members
 *   frames int64 (B, 17)           n_labels, n_rois, particle_area, particle_area + overlap, tie_flag, then per
 *                                  cell-type slot: present, count, area in pixels (tiff_analysis.py:1018-1038
 *                                  before its two round(x, 5), which the host applies)
 * pcseg_table_layout counts and scans (totals: device int64[6] = rows of rois, cells, groups, then the number of frames
 * with overflow / ws_overflow / nan_flag set, so that one read-back serves the caller's checks too); the caller reads the
 * totals, allocates, and pcseg_table_write fills the tables.  Both asynchronous on `stream`; every pointer of the
 * struct is a device pointer, group_of / n_groups / group_stats entries may be NULL (slot absent / merged = False). */
typedef struct pcseg_table_inputs {
    int32_t B, cap, C, n_ratios;
    const int64_t *frame_ids;                                   /* (B) id written into column 0 */
    const int32_t *counts; const int64_t *stats; const uint8_t *cls_out; const double *cc_sums;  /* class-map components */
    const uint8_t *kind; const uint8_t *slot_of; const int32_t *cells;                            /* pcseg_classify_regions */
    const int64_t *particle_area; const int64_t *overlap_area; const int64_t *type_stats; const int32_t *tie_flags;
    const int32_t *region_list; const int32_t *n_list;          /* (B, 5, cap), (B, 5) */
    const int32_t *group_of[5]; const int32_t *n_groups[5]; const int64_t *group_stats[5];       /* (B, cap), (B), (B, cap, 8) */
    const int32_t *n_markers; const int64_t *ws_stats; const double *ws_sums;                     /* refined ROIs */
    int32_t ratio_num[8]; int32_t ratio_den[8][4];              /* plane indices, -1 = unused */
    const int32_t *overflow; const int32_t *ws_overflow; const int32_t *nan_flag;  /* per-frame flags (B), may be NULL */
} pcseg_table_inputs;
size_t pcseg_table_workspace_bytes(int B, int cap);
int pcseg_table_layout(const pcseg_table_inputs *in, int64_t *totals, void *workspace, size_t workspace_bytes,
                       pcseg_stream_t stream);
int pcseg_table_write(const pcseg_table_inputs *in, double *rois, double *cells, double *groups, int64_t *frames,
                      void *workspace, size_t workspace_bytes, pcseg_stream_t stream);

/* ---- C14 for a batch (HCN_nanosims_rois_activity_distance_5iso_YG.m:260-268): dist[i] = distance from row i of the dense
 * `cells` table pcseg_table_write has just filled to the nearest row of the OTHER of the two cell types (slots 0 and 1 of
 * class_slot, a HOST array class value -> slot) in the same frame, / (size / raster) (the script: size 512, raster 19);
 * positions (centroid_col + 1, centroid_row + 1) as MATLAB's regionprops reports them; NaN for rows without an entry
 * (another type, or a frame in which one of the two types is absent).  `workspace` is the one pcseg_table_layout /
 * pcseg_table_write used (it holds the frames' row offsets).  PARITY UNPINNED (no MATLAB here). */
int pcseg_cell_distances(const double *cells, int64_t n_rows, int ncol, const uint8_t *class_slot, double raster, double size,
                         double *dist, int B, const void *workspace, size_t workspace_bytes, pcseg_stream_t stream);

/* ---- X1 (north_star extension; refine_boundaries.py:22 imports skimage.filters and never calls it): the library
 * SURVEY.md 8a names is the oracle -- skimage.filters.threshold_otsu(float32 image, nbins=256), pinned by
 * tests/golden/extensions.npz.  pcseg_otsu_f32: threshold[b] (device float64 (B,), the value is the float32 bin
 * centre the library returns; a constant frame returns its value), plus the histogram it was taken from: hist device
 * int64 (B,256) over each frame's own [min, max] with numpy.histogram's float32 binning, lohi device float32 (B,2).
 * Entirely on the device, asynchronous on `stream`.  pcseg_otsu_hist_f32: the histogram alone. */
int pcseg_otsu_f32(const float *img, double *threshold, int64_t *hist, float *lohi, int B, int H, int W,
                   pcseg_stream_t stream);
int pcseg_otsu_hist_f32(const float *img, int64_t *hist, float *lohi, int B, int H, int W,
                        pcseg_stream_t stream);

/* ---- X2 (north_star extension, no reference call site): 3x3 square binary
 * erosion (erode != 0, outside = True) or dilation (outside = False). */
int pcseg_morph3x3(const uint8_t *mask, uint8_t *out, int erode, int B, int H, int W, pcseg_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* PCSEG_H */

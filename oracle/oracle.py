"""CPU oracle: numpy + liboracle.so restatement of the reference's hot path.

TEST INFRASTRUCTURE ONLY (see ``pcseg_oracle.c``).  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg import
this module; the product package never does.

Every function keeps the name, arguments and return shape of the reference
function it restates (``tiff_analysis.py`` / ``refine_boundaries.py`` /
``HCN_nanosims_rois_activity_distance_5iso_YG.m`` of
ssilverman16/particle_col_image_segmentation) and cites its lines.  Pinned
against ``tests/golden`` (made by running the real reference).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

# --- constants restated from tiff_analysis.py:47-82 -------------------------
CELL_TYPES = ["3D05", "6B07", "C3M10"]
MIN_CELL_AREA = {"3D05": 20, "6B07": 20, "C3M10": 20}
MIN_CLUSTER_AREA = {"3D05": 200, "6B07": 200, "C3M10": 370}
DENOISE_SIZE = 5
DILATION_RADIUS = 20
DISTANCE_THRESHOLD = 2
CELL_CLUSTER_DISTANCE_THRESHOLD = 5
DAPI_RFP_OVERLAP_THRESHOLD = 0.1
PX_TO_UM_CONV = 9.95


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "pcseg_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"],
                              stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
        _LIB.orc_label_i32.restype = ctypes.c_int
        _LIB.orc_otsu_f32.restype = ctypes.c_double
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


# ---------------------------------------------------------------- primitives
def median_filter(ds_arr, size=DENOISE_SIZE):
    """scipy.ndimage.median_filter(ds_arr, size=5) -- tiff_analysis.py:122, 643."""
    assert size == 5
    a = _c(ds_arr, np.uint8)
    out = np.empty_like(a)
    lib().orc_median5_u8(_p(a), _p(out), a.shape[0], a.shape[1])
    return out


def label(image, connectivity=None, return_num=False):
    """skimage.measure.label -- tiff_analysis.py:743, 260, 829; refine_boundaries.py:64.

    bool input: non-zero components; integer input: equal-valued components;
    connectivity None -> 2 (8-neighbours) for 2-D."""
    image = np.asarray(image)
    conn8 = 1 if connectivity in (None, 2) else 0
    eq = 0 if image.dtype == bool else 1
    a = _c(image, np.int32)
    out = np.empty(a.shape, np.int32)
    n = lib().orc_label_i32(_p(a), _p(out), a.shape[0], a.shape[1], conn8, eq)
    return (out, n) if return_num else out


def region_table(label_im, n=None):
    label_im = _c(label_im, np.int32)
    if n is None:
        n = int(label_im.max()) if label_im.size else 0
    tab = np.zeros((n, 8), np.int64)
    lib().orc_region_table(_p(label_im), label_im.shape[0], label_im.shape[1], n, _p(tab))
    return tab


def channel_sums(label_im, planes, n=None):
    """S_k = sum(raw_k .* roimask) -- .m:122-135."""
    label_im = _c(label_im, np.int32)
    planes = _c(planes, np.float32)
    if n is None:
        n = int(label_im.max()) if label_im.size else 0
    C = planes.shape[0]
    sums = np.zeros((n, C), np.float64)
    lib().orc_channel_sums(_p(label_im), _p(planes), C, label_im.shape[0], label_im.shape[1], n, _p(sums))
    return sums


def binary_dilation_disk(mask, radius):
    """skimage.morphology.binary_dilation(mask, disk(r)) -- tiff_analysis.py:827-828, 990."""
    m = _c(mask, np.uint8)
    out = np.empty_like(m)
    lib().orc_dilate_disk(_p(m), _p(out), m.shape[0], m.shape[1], int(radius))
    return out.astype(bool)


def edt_sq(mask, brute=False):
    m = _c(mask, np.uint8)
    out = np.empty(m.shape, np.int32)
    (lib().orc_edt_sq_brute if brute else lib().orc_edt_sq)(_p(m), _p(out), m.shape[0], m.shape[1])
    return out


def distance_transform_edt(mask):
    """scipy.ndimage.distance_transform_edt -- refine_boundaries.py:60, tiff_analysis.py:996."""
    return np.sqrt(edt_sq(mask).astype(np.float64))


def binary_fill_holes(mask):
    """scipy.ndimage.binary_fill_holes -- tiff_analysis.py:880."""
    m = _c(mask, np.uint8)
    out = np.empty_like(m)
    lib().orc_fill_holes(_p(m), _p(out), m.shape[0], m.shape[1])
    return out.astype(bool)


def local_maxima(image):
    """skimage.morphology.local_maxima -- refine_boundaries.py:63."""
    a = _c(image, np.float64)
    out = np.empty(a.shape, np.uint8)
    lib().orc_local_maxima_f64(_p(a), _p(out), a.shape[0], a.shape[1])
    return out.astype(bool)


def watershed(image, markers, mask):
    """skimage.segmentation.watershed(image, markers, mask=mask) -- refine_boundaries.py:73."""
    img = _c(image, np.float64)
    mk = _c(markers, np.int32)
    mm = _c(mask, np.uint8)
    out = np.empty(mk.shape, np.int32)
    lib().orc_watershed(_p(img), _p(mk), _p(mm), _p(out), mk.shape[0], mk.shape[1])
    return out


def threshold_otsu(image):
    """north_star extension (no reference call site): PARITY UNPINNED BY THE REFERENCE."""
    a = _c(image, np.float32).ravel()
    hist = np.zeros(256, np.int64)
    thr = lib().orc_otsu_f32(_p(a), ctypes.c_size_t(a.size), _p(hist))
    return float(thr), hist


def morph3x3(mask, erode):
    """north_star extension (no reference call site): PARITY UNPINNED BY THE REFERENCE."""
    m = _c(mask, np.uint8)
    out = np.empty_like(m)
    lib().orc_morph3x3(_p(m), _p(out), m.shape[0], m.shape[1], int(bool(erode)))
    return out.astype(bool)


# ------------------------------------------------------------------- regions
class Region:
    """The attributes of skimage RegionProperties the reference touches
    (tiff_analysis.py:270, 275, 406, 769-781, 844, 855-863, 912, 1033, 1042)."""

    def __init__(self, label_id, row, label_im):
        self.label = int(label_id)
        self.area = int(row[0])
        self.centroid = (float(row[1]) / float(row[0]), float(row[2]) / float(row[0]))
        self.bbox = (int(row[3]), int(row[4]), int(row[5]), int(row[6]))
        self.first = (int(row[7]) // label_im.shape[1], int(row[7]) % label_im.shape[1])
        self._label_im = label_im

    @property
    def coords(self):
        return np.argwhere(self._label_im == self.label)

    def __getitem__(self, key):
        return getattr(self, key)


def regionprops(label_im):
    tab = region_table(label_im)
    return [Region(i + 1, tab[i], label_im) for i in range(tab.shape[0]) if tab[i, 0] > 0]


def get_type(region, data):
    """tiff_analysis.py:1041-1044."""
    return data[region.first[0], region.first[1]]


# ------------------------------------------------------- reference functions
def get_cell_positions_and_areas(z_slice, cell_types, merged=False):
    """tiff_analysis.py:742-789."""
    label_im = label(z_slice)
    regions = regionprops(label_im)
    cell_pos, cell_clusters, particle_area = {}, {}, 0
    for region in regions:
        region_type = int(get_type(region, z_slice))
        cell_type = cell_types[region_type]  # KeyError like the reference (:756)
        if cell_type not in CELL_TYPES:
            if cell_type == "Particle":
                particle_area += region.area
            continue
        if cell_type not in cell_pos:
            cell_pos[cell_type] = []
            cell_clusters[cell_type] = []
        if MIN_CELL_AREA[cell_type] <= region.area < MIN_CLUSTER_AREA[cell_type]:
            cell_pos[cell_type].append(region)
        if region.area >= MIN_CLUSTER_AREA[cell_type]:
            cell_clusters[cell_type].append(region)
    avg = {}
    for cell_type, arr in cell_pos.items():
        with np.errstate(all="ignore"):
            avg[cell_type] = np.float64(np.sum([c.area for c in arr], dtype=np.float64)) / np.float64(len(arr)) \
                if len(arr) else np.float64("nan")
    for cell_type, clusters in cell_clusters.items():
        for cluster in clusters:
            q = np.float64(cluster.area) // avg[cell_type]
            if np.isnan(q):
                raise ValueError("cannot convert float NaN to integer")  # :776-781
            cluster.cells = int(q)
    if merged:
        merged_clusters, _ = get_cell_clusters_from_distances(z_slice, cell_pos, cell_clusters, cell_types)
    else:
        merged_clusters = {}
    return cell_pos, cell_clusters, particle_area, merged_clusters


def get_cell_clusters_from_distances(z_slice, cell_pos, cell_clusters, cell_types):
    """tiff_analysis.py:791-824 (types in insertion order instead of set order)."""
    combined = {}
    for key in list(cell_pos) + [k for k in cell_clusters if k not in cell_pos]:
        combined[key] = cell_pos.get(key, []) + cell_clusters.get(key, [])
    merged_regions, merged_images, img_vals, combined_regions = {}, {}, [], []
    for cell_type, cell_regions in combined.items():
        val = 0
        for cell_val, t in cell_types.items():
            if t == cell_type:
                val = cell_val
                break
        img_vals.append(val)
        combined_regions.extend(cell_regions)
        merged_regions[cell_type], merged_images[cell_type] = get_merged_regions(z_slice == val, cell_regions)
    comb = np.zeros(z_slice.shape, bool)
    for v in img_vals:
        comb |= z_slice == v
    merged_regions["combined"], merged_images["combined"] = get_merged_regions(comb, combined_regions)
    return merged_regions, merged_images


def get_merged_regions(binary_image, og_cell_regions):
    """tiff_analysis.py:826-883, O(R) grouping with the reference's semantics."""
    dilated = binary_dilation_disk(binary_image, CELL_CLUSTER_DISTANCE_THRESHOLD // 2)
    dilated_labels = label(dilated)
    H, W = dilated_labels.shape
    keys = []
    for r in og_cell_regions:
        y, x = int(r.centroid[0]), int(r.centroid[1])
        keys.append(int(dilated_labels[y, x]) if (0 <= y < H and 0 <= x < W) else 0)
    groups, order = {}, []
    for r, k in zip(og_cell_regions, keys):
        if k <= 0:
            continue
        if k not in groups:
            groups[k] = []
            order.append(k)
        groups[k].append(r)
    merged_regions = []
    merged_image = np.zeros(binary_image.shape, bool)
    for k in order:
        touching = groups[k]
        merged_regions.append({
            "area": sum(r.area for r in touching),
            "centroid": np.average([r.centroid for r in touching], axis=0, weights=[r.area for r in touching]),
            "regions": touching,
            "bbox": (min(r.bbox[0] for r in touching), min(r.bbox[1] for r in touching),
                     max(r.bbox[2] for r in touching), max(r.bbox[3] for r in touching)),
        })
        merged_image |= dilated_labels == k
    return merged_regions, binary_fill_holes(merged_image)


def fill_particle_area(ds_arr, particle_label, cell_label, overlap_label):
    """tiff_analysis.py:982-1015."""
    particle_mask = ds_arr == particle_label
    cell_mask = ds_arr == cell_label
    dilated_particle = binary_dilation_disk(particle_mask, DILATION_RADIUS)
    dist = distance_transform_edt(~particle_mask)
    combined = (cell_mask & (dist < DISTANCE_THRESHOLD)) | (cell_mask & dilated_particle)
    updated = ds_arr.copy()
    updated[combined] = overlap_label
    return updated, int(np.sum(combined))


def recreate_particle_area(ds_arr, cell_types, particle_area):
    """tiff_analysis.py:931-950."""
    particle_label = None
    for key, value in cell_types.items():
        if value == "Particle":
            particle_label = key
    for cell_type_label, cell_type in cell_types.items():
        if cell_type not in CELL_TYPES:
            continue
        ds_arr, overlap = fill_particle_area(ds_arr, particle_label, cell_type_label, particle_label)
        particle_area += overlap
    return ds_arr, particle_area


def get_cell_counts_and_densities(cell_pos, cell_clusters, particle_area):
    """tiff_analysis.py:1018-1038."""
    cell_count, cell_density, cell_area_ratio = {}, {}, {}
    particle_area = particle_area / (PX_TO_UM_CONV ** 2)
    for cell_type, arr in cell_pos.items():
        if cell_type not in CELL_TYPES:
            continue
        cluster_cells = sum(c.cells for c in cell_clusters[cell_type])
        cell_count[cell_type] = len(arr) + cluster_cells
        cell_area = np.sum([c.area for c in arr])
        for c in cell_clusters[cell_type]:
            cell_area += c["area"]
        area = cell_area / (PX_TO_UM_CONV ** 2)
        cell_density[cell_type] = round(cell_count[cell_type] / particle_area, 5)
        cell_area_ratio[cell_type] = round(area / particle_area, 5)
    return cell_count, cell_density, cell_area_ratio


def combine_cell_positions_and_clusters(dapi_channel, other_channel):
    """tiff_analysis.py:252-287."""
    dapi_mask = dapi_channel == 1
    rfp_mask = other_channel == 1
    labeled = label(dapi_mask)
    n = int(labeled.max()) if labeled.size else 0
    area = np.bincount(labeled.ravel(), minlength=n + 1)
    over = np.bincount(labeled.ravel(), weights=rfp_mask.ravel().astype(np.float64), minlength=n + 1)
    remove = np.zeros(n + 1, bool)
    for l in range(1, n + 1):
        remove[l] = (over[l] / area[l]) > DAPI_RFP_OVERLAP_THRESHOLD
    out = dapi_channel.copy()
    out[remove[labeled]] = 2
    return out


def refine_boundaries(boundary_map, threshold=0.5):
    """refine_boundaries.py:44-73 as a function; returns every stage."""
    binary_mask = np.asarray(boundary_map) < threshold
    distance = distance_transform_edt(binary_mask)
    local_max = local_maxima(distance)
    markers = label(local_max)
    labels = watershed(boundary_map, markers, binary_mask)
    return {"binary_mask": binary_mask, "distance": distance, "local_max": local_max,
            "markers": markers, "labels": labels}


# ------------------------------------------------------------ .m restatement
ISOTOPES_7 = ("12C", "13C", "14N12C", "15N12C", "16O", "17O", "18O")
# (numerator, denominator members) -- .m:136-139
RATIOS_7 = (("C13act", 1, (1, 0)), ("N15act", 3, (2, 3)), ("O17act", 5, (6, 5, 4)), ("O18act", 6, (6, 5, 4)))
ISOTOPES_5 = ("12C", "13C", "14N12C", "15N12C", "32S")
RATIOS_5 = (("C13act", 1, (1, 0)), ("N15act", 3, (2, 3)))


def roi_activity_table(label_im, planes, roi_class=1, ratios=RATIOS_7):
    """.m:122-170: rows [class, i, S_1..S_C, ratios..., 100*ratios...]; centroid
    (x = col + 1, y = row + 1) as MATLAB regionprops reports it (.m:164-165).
    PARITY UNPINNED (no MATLAB / Octave here, no fixtures in the reference)."""
    sums = channel_sums(label_im, planes)
    tab = region_table(label_im, sums.shape[0])
    rows, xy = [], []
    for i in range(sums.shape[0]):
        if tab[i, 0] == 0:
            continue
        s = sums[i]
        acts = []
        for _, num, den in ratios:
            d = 0.0
            for k in den:
                d = d + s[k]
            with np.errstate(all="ignore"):
                acts.append(np.float64(s[num]) / np.float64(d))
        rows.append([roi_class, i + 1] + list(s) + acts + [a * 100 for a in acts])
        xy.append([tab[i, 2] / tab[i, 0] + 1.0, tab[i, 1] / tab[i, 0] + 1.0])
    return np.array(rows, np.float64).reshape(len(rows), -1), np.array(xy, np.float64).reshape(len(xy), 2)


def nearest_distances(a_positions, b_positions, raster=19.0, size=512.0):
    """.m:260-268: pdist2 + min both ways, scaled by /(512/raster)."""
    a = np.asarray(a_positions, np.float64)
    b = np.asarray(b_positions, np.float64)
    d = np.sqrt(((a[:, None, :] - b[None, :, :]) ** 2).sum(-1))
    return np.concatenate([d.min(axis=1), d.min(axis=0)]) / (size / raster)


# ---------------------------------------------------------------- full chain
def segment_frame(stack, cell_types, merged=True, threshold=0.5, boundary_plane=3):
    """One frame of the hot path: SURVEY.md section 8d 'full chain'."""
    stack = np.asarray(stack, np.float32)
    cm = (np.argmax(stack, axis=0) + 1).astype(np.uint8)
    den = median_filter(cm)
    label_im = label(den)
    cell_pos, cell_clusters, particle_area, merged_clusters = get_cell_positions_and_areas(den, cell_types, merged)
    counts = get_cell_counts_and_densities(cell_pos, cell_clusters, particle_area)
    rec, pa2 = recreate_particle_area(den, cell_types, particle_area)
    rf = refine_boundaries(stack[boundary_plane], threshold)
    roi_sums = channel_sums(rf["labels"], stack)
    cc_sums = channel_sums(label_im, stack)
    return {"denoised": den, "label_im": label_im, "cell_pos": cell_pos, "cell_clusters": cell_clusters,
            "particle_area": particle_area, "merged_clusters": merged_clusters, "counts": counts,
            "recreated": rec, "particle_area2": pa2, "refine": rf, "roi_sums": roi_sums, "cc_sums": cc_sums}

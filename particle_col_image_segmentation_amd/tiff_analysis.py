"""Drop-in for the hot-path functions of the reference's ``tiff_analysis.py``.

Same names, positional arguments, return shapes and error behaviour as
ssilverman16/particle_col_image_segmentation ``tiff_analysis.py`` (cited per
function); the pixel work runs in the HIP kernels of ``libpcseg.so`` through
``ops`` -- there is no CPU path.  Images go in and come back as numpy arrays
(torch CUDA tensors are accepted and then returned as tensors).

Not reproduced on purpose (out of scope, SURVEY.md section 2: C9-C11, C13):
matplotlib plotting and the ``main()`` folder walk.
"""
import csv
import os

import numpy as np
import torch

from . import ops

# --- constants: tiff_analysis.py:47-82 (they are the reference's whole flag system)
BASE_TYPE_MAP = {1: "3D05", 2: "6B07", 3: "C3M10", 4: "Particle", 5: "Background"}
CELL_TYPES = ["3D05", "6B07", "C3M10"]
CHANNELS = ["RFP", "DAPI", "GFP"]
CHANNEL_MAP = {"RFP": "3D05", "DAPI": "6B07", "GFP": "C3M10"}
STRAIN_MAP = {"3D05": "RFP", "6B07": "DAPI", "C3M10": "GFP"}
MIN_CELL_AREA = {"3D05": 20, "6B07": 20, "C3M10": 20}
MIN_CLUSTER_AREA = {"3D05": 200, "6B07": 200, "C3M10": 370}
DENOISE_SIZE = 5
DILATION_RADIUS = 20
DISTANCE_THRESHOLD = 2
CELL_CLUSTER_DISTANCE_THRESHOLD = 5
DAPI_RFP_OVERLAP_THRESHOLD = 0.1
PX_TO_UM_CONV = 9.95


def _device():
    if not torch.cuda.is_available():
        raise RuntimeError("particle_col_image_segmentation_amd needs a ROCm GPU: the HIP path has no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def _to_dev_u8(a):
    """(H, W) integer / bool image -> (1, H, W) uint8 CUDA tensor, plus 'caller passed a tensor'."""
    if isinstance(a, torch.Tensor):
        t = a
        was_tensor = True
    else:
        arr = np.asarray(a)
        if arr.dtype != np.uint8 and arr.dtype != bool and arr.size and (arr.min() < 0 or arr.max() > 255):
            raise ValueError("class maps must hold values 0..255")
        t = torch.from_numpy(np.ascontiguousarray(arr.astype(np.uint8, copy=False)))
        was_tensor = False
    if t.dim() != 2:
        raise ValueError("expected a 2-D image, got shape %s" % (tuple(t.shape),))
    return t.to(device=_device(), dtype=torch.uint8).contiguous()[None], was_tensor


def _back(t, was_tensor, dtype=None):
    if was_tensor:
        return t
    a = t.cpu().numpy()
    return a.astype(dtype) if dtype is not None else a


class _LabelImage:
    """Device label image of one frame with a lazily fetched host copy (for Region.coords)."""

    def __init__(self, dev_labels):
        self.dev = dev_labels
        self._host = None

    @property
    def host(self):
        if self._host is None:
            self._host = self.dev.cpu().numpy()
        return self._host


class Region:
    """Duck type of skimage RegionProperties restricted to what the reference touches:
    .label (:270), .area (:275, 769-781, 855, 1031, 1055), ["area"] (:1033), .centroid (:406, 844, 1054),
    .bbox (:860-863, 912), .coords[0] (:1042), dynamically added .cells (:781, 1029, 1063)."""

    __slots__ = ("label", "area", "centroid", "bbox", "first", "sum_row", "sum_col", "cells", "_im")

    def __init__(self, label_id, row, width, label_image=None):
        self.label = int(label_id)
        # numpy scalars on purpose: the reference's CSV writers call round() on them, and numpy's round
        # (x*10^n -> rint -> /10^n) differs from Python's float round on decimal ties (tiff_analysis.py:1057)
        self.area = np.int64(row[0])
        self.sum_row = int(row[1])
        self.sum_col = int(row[2])
        # np.mean of integer coordinates = float64(sum) / count (skimage/measure/_regionprops.py:296-297)
        self.centroid = (np.float64(row[1]) / np.float64(row[0]), np.float64(row[2]) / np.float64(row[0]))
        self.bbox = (int(row[3]), int(row[4]), int(row[5]), int(row[6]))
        self.first = (int(row[7]) // width, int(row[7]) % width)
        self._im = label_image

    @property
    def coords(self):
        if self._im is None:
            return np.array([self.first])
        return np.argwhere(self._im.host == self.label)

    def __getitem__(self, key):
        return getattr(self, key)


def median_filter(ds_arr, size=DENOISE_SIZE):
    """scipy.ndimage.median_filter(ds_arr, size=5) as called at tiff_analysis.py:122, 643."""
    if size != 5:
        raise ValueError("only size=5 (DENOISE_SIZE) is built")
    t, was = _to_dev_u8(ds_arr)
    return _back(ops.median5(t)[0], was)


def label(image):
    """skimage.measure.label(image) as called at tiff_analysis.py:743 (int) and :260, :829 (bool)."""
    is_bool = (image.dtype == torch.bool) if isinstance(image, torch.Tensor) else (np.asarray(image).dtype == bool)
    t, was = _to_dev_u8(image)
    lab, _ = (ops.label_bool8 if is_bool else ops.label_equal8)(t)
    return _back(lab[0], was, np.int64 if not is_bool else np.int32)


def _regions_of(z_dev):
    """label + regionprops of one frame on the device: (regions, class at first pixel, label holder, and the device
    tensors stats / cls_out / counts that the classification and merge kernels take)."""
    labels, counts = ops.label_equal8(z_dev)
    n = int(counts[0].item())
    stats, cls_out, _, _ = ops.region_reduce(labels, counts, cls=z_dev, cap=max(n, 1))
    holder = _LabelImage(labels[0])
    st = stats[0, :n].cpu().numpy()
    cl = cls_out[0, :n].cpu().numpy()
    width = z_dev.shape[2]
    regions = [Region(i + 1, st[i], width, holder) for i in range(n)]
    return regions, cl, holder, (stats, cls_out, counts)


def regionprops(label_im):
    """skimage.measure.regionprops(label_im) reduced to the fields the reference reads (tiff_analysis.py:746)."""
    if isinstance(label_im, torch.Tensor):
        lab = label_im.to(device=_device(), dtype=torch.int32).contiguous()[None]
    else:
        lab = torch.from_numpy(np.ascontiguousarray(np.asarray(label_im).astype(np.int32)))[None].to(_device())
    n = int(lab.max().item()) if lab.numel() else 0
    stats, _, _, _ = ops.region_reduce(lab, cap=max(n, 1))
    st = stats[0, :n].cpu().numpy()
    holder = _LabelImage(lab[0])
    return [Region(i + 1, st[i], lab.shape[2], holder) for i in range(n) if st[i][0] > 0]


def get_type(region, data):
    """tiff_analysis.py:1041-1044."""
    point = getattr(region, "first", None)
    if point is None:
        point = region.coords[0]
    return data[point[0], point[1]]


def get_cell_positions_and_areas(z_slice, cell_types, merged=False):
    """tiff_analysis.py:742-789.  Returns (cell_pos, cell_clusters, particle_area, merged_clusters).

    The per-region decisions (Particle area, cell / cluster by the area thresholds of the region's type, cluster cell
    count = int(area // mean single-cell area)) are ``pcseg_classify_regions`` -- the kernel the batched pipeline uses
    -- and the host only files the regions under their type names."""
    z_dev, _ = _to_dev_u8(z_slice)
    regions, classes, _, (stats, cls_out, counts) = _regions_of(z_dev)
    for value in np.unique(classes):
        cell_types[int(value)]  # KeyError for an unmapped class value, as at :756
    tables = ops.ClassTables(cell_types, CELL_TYPES, MIN_CELL_AREA, MIN_CLUSTER_AREA)
    verdict = ops.classify_regions(stats, cls_out, counts, tables)
    n = len(regions)
    kind = verdict["kind"][0, :n].cpu().numpy()
    slot = verdict["slot_of"][0, :n].cpu().numpy()
    n_cells = verdict["cells"][0, :n].cpu().numpy()
    first_region = verdict["type_stats"][0, :, 3].cpu().numpy()
    # the reference's dicts are keyed in the order the types first appear among the regions (:763-765)
    order = sorted((int(first_region[t]), t) for t in range(len(tables.slot_names)) if first_region[t] != 0x7FFFFFFF)
    cell_pos = {tables.slot_names[t]: [] for _, t in order}
    cell_clusters = {tables.slot_names[t]: [] for _, t in order}
    if int(verdict["nan_flag"][0].item()):
        # a type with clusters but without a single cell: np.average([]) is NaN and int(NaN) raises (:776-781)
        raise ValueError("cannot convert float NaN to integer")
    for i in np.nonzero(kind)[0]:
        name = tables.slot_names[slot[i]]
        if kind[i] == 1:
            cell_pos[name].append(regions[i])
        else:
            regions[i].cells = int(n_cells[i])
            cell_clusters[name].append(regions[i])
    particle_area = int(verdict["particle_area"][0].item())
    merged_clusters = {}
    if merged:
        merged_clusters, _ = _clusters_from_distances(z_dev, stats, cell_pos, cell_clusters, cell_types, False)
    return cell_pos, cell_clusters, particle_area, merged_clusters


def _group_regions(dl_dev, stats_dev, og_cell_regions):
    """device grouping (pcseg_merge_groups) + host assembly of the reference's merged-region dicts (:850-872)."""
    n = len(og_cell_regions)
    dev = dl_dev.device
    lst = torch.full((1, max(n, 1)), -1, dtype=torch.int32)
    if n:
        lst[0, :n] = torch.tensor([r.label - 1 for r in og_cell_regions], dtype=torch.int32)
    group_of, n_groups = ops.merge_groups(dl_dev, stats_dev, lst.to(dev), torch.tensor([n], dtype=torch.int32, device=dev))
    gof = group_of[0, :n].cpu().numpy()
    merged_regions = []
    for g in range(1, int(n_groups[0].item()) + 1):
        members = [og_cell_regions[k] for k in np.nonzero(gof == g)[0]]
        areas = [m.area for m in members]
        boxes = np.array([m.bbox for m in members])
        merged_regions.append({
            "area": sum(areas),
            "centroid": np.average([m.centroid for m in members], axis=0, weights=areas),
            "regions": members,
            "bbox": (boxes[:, 0].min(), boxes[:, 1].min(), boxes[:, 2].max(), boxes[:, 3].max()),
        })
    return merged_regions, gof


def _stats_from_regions(regions, dev):
    cap = max([r.label for r in regions], default=1)
    st = np.zeros((1, cap, 8), np.int64)
    for r in regions:
        st[0, r.label - 1, :3] = (r.area, r.sum_row, r.sum_col)
    return torch.from_numpy(st).to(dev)


def _merged_regions_dev(z_dev, value_bits, stats_dev, og_cell_regions, want_image):
    dil = ops.dilate_disk(z_dev, value_bits, CELL_CLUSTER_DISTANCE_THRESHOLD // 2)
    dl, dl_counts = ops.label_bool8(dil)
    merged_regions, _ = _group_regions(dl, stats_dev, og_cell_regions)
    merged_image = None
    if want_image:
        # union of the dilated components that hold a listed centroid, holes filled (:876-880)
        K = int(dl_counts[0].item())
        keep = torch.zeros((K + 1,), dtype=torch.uint8, device=dl.device)
        H, W = dl.shape[1:]
        ys = torch.tensor([min(max(int(r.centroid[0]), 0), H - 1) for r in og_cell_regions], dtype=torch.long, device=dl.device)
        xs = torch.tensor([min(max(int(r.centroid[1]), 0), W - 1) for r in og_cell_regions], dtype=torch.long, device=dl.device)
        if len(og_cell_regions):
            keep[dl[0, ys, xs].long()] = 1
        keep[0] = 0
        sel = keep[dl.long()]
        merged_image = ops.fill_holes(sel)[0].bool()
    return merged_regions, merged_image


def _clusters_from_distances(z_dev, stats_dev, cell_pos, cell_clusters, cell_types, want_images):
    """:791-824 on the device: one dilated-mask grouping per cell type over that type's cells + clusters, and one over
    all of them on the union of the masks ("combined").  The reference walks a set of the type names (hash-randomised
    order, :794); here the types come in the insertion order of the two dicts."""
    names = list(cell_pos) + [k for k in cell_clusters if k not in cell_pos]
    first_value = {}
    for value, name in cell_types.items():
        first_value.setdefault(name, value)  # the class value of a type is the FIRST key that maps to it (:806-810)
    merged_regions, merged_images = {}, {}
    everything, all_bits = [], 0
    for name in names:
        members = cell_pos.get(name, []) + cell_clusters.get(name, [])
        value = first_value.get(name, 0)
        all_bits |= 1 << value
        everything += members
        merged_regions[name], merged_images[name] = _merged_regions_dev(z_dev, 1 << value, stats_dev, members, want_images)
    merged_regions["combined"], merged_images["combined"] = _merged_regions_dev(z_dev, all_bits, stats_dev, everything, want_images)
    return merged_regions, merged_images


def get_cell_clusters_from_distances(z_slice, cell_pos, cell_clusters, cell_types):
    """tiff_analysis.py:791-824.  Returns (merged_regions, merged_images)."""
    z_dev, was = _to_dev_u8(z_slice)
    all_regions = [r for regs in list(cell_pos.values()) + list(cell_clusters.values()) for r in regs]
    stats_dev = _stats_from_regions(all_regions, z_dev.device)
    merged_regions, merged_images = _clusters_from_distances(z_dev, stats_dev, cell_pos, cell_clusters, cell_types, True)
    return merged_regions, {k: _back(v, was) for k, v in merged_images.items()}


def get_merged_regions(binary_image, og_cell_regions):
    """tiff_analysis.py:826-883.  Returns (merged_regions, merged_image)."""
    b_dev, was = _to_dev_u8(binary_image)
    b_dev = (b_dev != 0).to(torch.uint8)
    stats_dev = _stats_from_regions(og_cell_regions, b_dev.device)
    merged_regions, merged_image = _merged_regions_dev(b_dev, 1 << 1, stats_dev, og_cell_regions, True)
    return merged_regions, _back(merged_image, was)


def fill_particle_area(ds_arr, particle_label, cell_label, overlap_label):
    """tiff_analysis.py:982-1015.  Returns (updated_ds_arr, overlap_area); the input is not modified (:1010)."""
    ds_dev, was = _to_dev_u8(ds_arr)
    out, area = ops.fill_particle(ds_dev, particle_label, cell_label, overlap_label, DILATION_RADIUS, DISTANCE_THRESHOLD)
    res = _back(out[0], was)
    if not was:
        res = res.astype(np.asarray(ds_arr).dtype, copy=False)
    return res, int(area[0].item())


def recreate_particle_area(ds_arr, cell_types, particle_area):
    """tiff_analysis.py:931-950: one fill per cell class, each on the output of the previous one; the overlap areas
    are accumulated on the device and read once."""
    cell_labels = [label for label, name in cell_types.items() if name in CELL_TYPES]
    if not cell_labels:
        return ds_arr, particle_area
    particle_labels = [label for label, name in cell_types.items() if name == "Particle"]
    if not particle_labels:
        raise TypeError("no 'Particle' entry in cell_types")  # the reference fails inside fill_particle_area
    particle_label = particle_labels[-1]  # the last match wins in the reference's loop (:933-935)
    ds_dev, was = _to_dev_u8(ds_arr)
    gained = None
    for cell_label in cell_labels:
        ds_dev, gained = ops.fill_particle(ds_dev, particle_label, cell_label, particle_label, DILATION_RADIUS,
                                           DISTANCE_THRESHOLD, gained)
    res = _back(ds_dev[0], was)
    if not was:
        res = res.astype(np.asarray(ds_arr).dtype, copy=False)
    return res, particle_area + int(gained[0].item())


def get_cell_counts_and_densities(cell_pos, cell_clusters, particle_area):
    """tiff_analysis.py:1018-1038: count = cells + cells inside clusters; density and area ratio per um^2 of particle,
    both rounded to 5 decimals (Python's round on a Python float, as there)."""
    um2 = PX_TO_UM_CONV ** 2
    particle_um2 = particle_area / um2
    counts, densities, ratios = {}, {}, {}
    for name in (t for t in cell_pos if t in CELL_TYPES):
        clusters = cell_clusters[name]
        counts[name] = len(cell_pos[name]) + sum(c.cells for c in clusters)
        pixels = np.sum([np.int64(c.area) for c in cell_pos[name]])  # np.sum([]) is 0.0, like the reference's
        for c in clusters:
            pixels = pixels + c["area"]
        densities[name] = round(counts[name] / particle_um2, 5)
        ratios[name] = round((pixels / um2) / particle_um2, 5)
    return counts, densities, ratios


def combine_cell_positions_and_clusters(dapi_channel, other_channel):
    """tiff_analysis.py:252-287: drop DAPI cells that overlap cells of the other channel by > 10 %."""
    d_dev, was = _to_dev_u8(dapi_channel)
    o_dev, _ = _to_dev_u8(other_channel)
    out = ops.remove_overlapping(d_dev, o_dev, DAPI_RFP_OVERLAP_THRESHOLD)
    res = _back(out[0], was)
    if not was:
        res = res.astype(np.asarray(dapi_channel).dtype, copy=False)
    return res


def get_rfp_base_arr(rfp_arr, cell_strains):
    """Lift a per-channel RFP class map into the 5-class numbering, in place (tiff_analysis.py:224-231): without a
    3D05 strain the RFP channel holds (particle, background) = (1, 2), otherwise (cell, particle, background)."""
    particle, background = (1, 2) if cell_strains in (["6B07"], ["6B07", "C3M10"]) else (2, 3)
    rfp_arr[rfp_arr == particle] = 4   # order matters: the particle value is remapped before the background value
    rfp_arr[rfp_arr == background] = 5
    return rfp_arr


def combine_channels(rfp_base, channel_ds_arrs, cell_strains):
    """Paint the cells of every non-3D05 strain (value 1 of its own channel) into the RFP base map with the strain's
    value of BASE_TYPE_MAP (tiff_analysis.py:233-249)."""
    value_of = {name: val for val, name in BASE_TYPE_MAP.items()}
    for strain in cell_strains:
        if strain != "3D05":
            rfp_base[channel_ds_arrs[STRAIN_MAP[strain]] == 1] = value_of[strain]
    return rfp_base


def normalize_ds_arr(ds_arr):
    """Squeeze an ilastik export to (H, W) (tiff_analysis.py:727-737): (H, W, 1), (1, H, W) or exactly 2048 x 2048."""
    shape = ds_arr.shape
    if shape[-1] == 1:
        return np.squeeze(ds_arr)
    if shape[0] == 1:
        return ds_arr[0]
    if shape[0] == 2048 and shape[1] == 2048:
        return ds_arr
    raise ValueError(f"DS arr shape is not (2048,2048,1) or (1,2048,2048) or (2048,2048). Shape: {ds_arr.shape}")


def get_strains_from_file(file_name):
    """Strain names (in CELL_TYPES order) that occur in the upper-cased path (tiff_analysis.py:673-678)."""
    upper = file_name.upper()
    return [strain for strain in CELL_TYPES if strain in upper]


def get_channel_from_file(file_name):
    """The one fluorescence channel named in the file (tiff_analysis.py:680-687); IndexError without any."""
    upper = file_name.upper()
    found = [channel for channel in CHANNELS if channel in upper]
    if len(found) > 1:
        raise ValueError("More than one channel found in file path")
    return found[0]


def get_cell_type_map(file_path):
    """{1..k: strains named in the file, k+1: "Particle", k+2: "Background"} (tiff_analysis.py:694-702).  Like the
    reference this needs at least one strain in the name (UnboundLocalError otherwise)."""
    strains = get_strains_from_file(file_path)
    if not strains:
        raise UnboundLocalError("local variable 'i' referenced before assignment")
    cell_type_map = dict(enumerate(strains, start=1))
    cell_type_map[len(strains) + 1] = "Particle"
    cell_type_map[len(strains) + 2] = "Background"
    return cell_type_map


def get_cell_type_map_from_channel(strain_types, channel):
    """Class values of one channel file of a multi-channel sample (tiff_analysis.py:709-712)."""
    if channel == "RFP" and strain_types in (["6B07"], ["6B07", "C3M10"]):
        return {1: "Particle", 2: "Background"}
    return {1: CHANNEL_MAP[channel], 2: "Particle", 3: "Background"}


def get_pos_and_density_file_names(cur_folder):
    """tiff_analysis.py:619-624: (<grandparent>_<parent>_cell_density_info.csv next to the folder, <folder>_cell_pos.csv in it)."""
    parts = cur_folder.split("/")
    density_csv = os.path.join(cur_folder, "..", "%s_%s_cell_density_info.csv" % (parts[-3], parts[-2]))
    return density_csv, os.path.join(cur_folder, parts[-1] + "_cell_pos.csv")


def _um2(area_px):
    """pixels^2 -> micrometres^2 with the reference's conversion factor (PX_TO_UM_CONV, tiff_analysis.py:82)."""
    return area_px / (PX_TO_UM_CONV ** 2)


def _write_rows(csv_output_file, header, rows):
    # csv.writer's default dialect on a file opened without newline="" -- the reference's output has \r\n line ends
    with open(csv_output_file, "w") as f:
        out = csv.writer(f)
        out.writerow(header)
        out.writerows(rows)


def write_cell_position_info(cell_positions, cell_clusters, csv_output_file, particle_area):
    """``<folder>_cell_pos.csv`` (tiff_analysis.py:1047-1063): one row per single cell (area rounded to 5 decimals,
    count 1) then one per cluster (area left unrounded, count = cluster.cells); x = column, y = row, 2 decimals."""
    particle_um2 = _um2(particle_area)

    def rows():
        for kind, table in (("cell", cell_positions), ("cluster", cell_clusters)):
            for strain_type, regions in table.items():
                for region in regions:
                    row_pos, col_pos = region.centroid
                    area = _um2(region.area)
                    shown_area = round(area, 5) if kind == "cell" else area
                    count = 1 if kind == "cell" else region.cells
                    yield [strain_type, kind, round(col_pos, 2), round(row_pos, 2), shown_area,
                           round(area / particle_um2, 8), count]

    _write_rows(csv_output_file, ["strain", "cell_type", "x_pos", "y_pos", "cell_area", "cell_area_ratio", "cell_count"],
                rows())


def write_merged_cell_position_info(merged_clusters, csv_output_file, particle_area):
    """``<folder>_merged_cell_pos.csv`` (tiff_analysis.py:1065-1075): one row per merged group."""
    particle_um2 = _um2(particle_area)

    def rows():
        for strain_type, groups in merged_clusters.items():
            for group in groups:
                row_pos, col_pos = group["centroid"]
                area = _um2(group["area"])
                yield [strain_type, round(col_pos, 2), round(row_pos, 2), round(area, 5), round(area / particle_um2, 8),
                       len(group["regions"])]

    _write_rows(csv_output_file, ["strain_type", "x_pos", "y_pos", "cell_area", "cell_area_ratio", "cell_num"], rows())


def write_density_info(csv_output_file, h5_folder, cell_density, cell_area_ratio, cell_count):
    """Per-parent-folder density summary (tiff_analysis.py:1078-1107): rows of ``h5_folder`` already in the file are
    dropped (the file is rewritten without them), then this run's rows are appended -- reruns stay idempotent."""
    header = ["folder", "strain", "cell_density", "cell_area_ratio", "cell_count"]
    is_new = not os.path.exists(csv_output_file)
    if not is_new:
        with open(csv_output_file, "r") as f:
            old_rows = list(csv.reader(f))[1:]
        kept = [row for row in old_rows if row[0] != h5_folder]
        if len(kept) != len(old_rows):
            _write_rows(csv_output_file, header, kept)
    with open(csv_output_file, "a") as f:
        out = csv.writer(f)
        if is_new:
            out.writerow(header)
        out.writerows([h5_folder, strain, cell_density[strain], cell_area_ratio[strain], cell_count[strain]]
                      for strain in cell_density)


def read_class_map(path):
    """The reference reads the first dataset of an ilastik .h5 (tiff_analysis.py:118-121, 639-642).
    h5py is optional here; ``.npy`` files are always accepted."""
    if path.endswith(".npy"):
        return np.load(path, allow_pickle=False)
    try:
        import h5py
    except ImportError as e:  # pragma: no cover
        raise ImportError("reading .h5 class maps needs h5py; save the class map as .npy instead") from e
    with h5py.File(path, "r") as f:
        return f[next(iter(f.keys()))][()]


def process_single_h5_file(cur_folder, file_path):
    """tiff_analysis.py:627-671 without the matplotlib figures: position CSV, merged-position CSV, density row."""
    cell_types = get_cell_type_map(file_path)
    if not cell_types:
        raise ValueError("Cell type not found in file path")
    density_csv, cell_pos_csv = get_pos_and_density_file_names(cur_folder)
    denoised = median_filter(normalize_ds_arr(read_class_map(os.path.join(cur_folder, file_path))), size=DENOISE_SIZE)
    positions, clusters, particle_area, merged = get_cell_positions_and_areas(denoised, cell_types, merged=True)
    counts, densities, ratios = get_cell_counts_and_densities(positions, clusters, particle_area)  # BEFORE the particle fill (:652)
    _, particle_area = recreate_particle_area(denoised, cell_types, particle_area)
    write_cell_position_info(positions, clusters, cell_pos_csv, particle_area)
    write_merged_cell_position_info(merged, cell_pos_csv.replace("_cell_pos.csv", "_merged_cell_pos.csv"), particle_area)
    write_density_info(density_csv, cur_folder.split("/")[-1], densities, ratios, counts)


def _analyse_channel_file(cur_folder, file, cell_strains):
    """One channel file of a multi-channel sample (loop body of tiff_analysis.py:107-157, no figures)."""
    channel = get_channel_from_file(file)
    cell_types = get_cell_type_map_from_channel(cell_strains, channel)
    denoised = median_filter(normalize_ds_arr(read_class_map(os.path.join(cur_folder, file))), size=DENOISE_SIZE)
    positions, clusters, particle_area, _ = get_cell_positions_and_areas(denoised, cell_types)
    return channel, cell_types, denoised, positions, clusters, particle_area


def process_multiple_h5_files(cur_folder, h5_files):
    """A sample exported as one class map per fluorescence channel (tiff_analysis.py:92-222), without the figures.

    Per file: denoise + region table; the RFP file also yields the (recreated) particle area every density refers to.
    Then the raw position CSV, DAPI cells that overlap the other channel by more than 10 % removed (:167-170), densities,
    the channels recombined into one 5-class map (:198-204) whose merged clusters go to ``*_merged_cell_pos.csv``.
    Returns (combined class map, merged clusters)."""
    density_csv, cell_pos_csv = get_pos_and_density_file_names(cur_folder)
    raw_csv = cell_pos_csv.replace("_cell_pos.csv", "_cell_pos_raw.csv")
    combined_csv = cell_pos_csv.replace("_cell_pos.csv", "_cell_pos_combined.csv")
    merged_csv = combined_csv.replace("_cell_pos_combined.csv", "_merged_cell_pos.csv")
    sample_name = cur_folder.split("/")[-1]
    cell_strains = get_strains_from_file(cur_folder)

    all_positions, all_clusters, denoised_by_channel = {}, {}, {}
    rfp_particle_area, dapi_cell_types = None, None
    for file in h5_files:
        channel, cell_types, denoised, positions, clusters, particle_area = _analyse_channel_file(cur_folder, file, cell_strains)
        denoised_by_channel[channel] = denoised
        first_type = cell_types[1]
        if channel == "RFP":
            _, rfp_particle_area = recreate_particle_area(denoised, cell_types, particle_area)
            if first_type == "Particle":  # an RFP channel that only shows the particle carries no cells (:131-132)
                continue
        elif channel == "DAPI":
            dapi_cell_types = cell_types
        if first_type not in CELL_TYPES:
            raise ValueError(f"Strain type not in cell types. {first_type}")
        all_positions.update(positions)
        all_clusters.update(clusters)
    if rfp_particle_area is None:
        raise ValueError("RFP particle area not found")
    write_cell_position_info(all_positions, all_clusters, raw_csv, rfp_particle_area)

    if len(cell_strains) > 1:
        other = "GFP" if cell_strains == ["6B07", "C3M10"] else "RFP"
        dapi_clean = combine_cell_positions_and_clusters(denoised_by_channel["DAPI"], denoised_by_channel[other])
        dapi_positions, dapi_clusters, _, _ = get_cell_positions_and_areas(dapi_clean, dapi_cell_types)
        all_positions["6B07"] = dapi_positions["6B07"]
        all_clusters["6B07"] = dapi_clusters["6B07"]

    counts, densities, area_ratios = get_cell_counts_and_densities(all_positions, all_clusters, rfp_particle_area)
    write_density_info(density_csv, sample_name, densities, area_ratios, counts)

    base = get_rfp_base_arr(denoised_by_channel["RFP"].copy(), cell_strains)
    combined_channels = combine_channels(base, denoised_by_channel, cell_strains)
    _, _, _, merged_clusters = get_cell_positions_and_areas(combined_channels, BASE_TYPE_MAP, merged=True)
    write_cell_position_info(all_positions, all_clusters, combined_csv, rfp_particle_area)
    write_merged_cell_position_info(merged_clusters, merged_csv, rfp_particle_area)
    return combined_channels, merged_clusters


def process_h5_folder(cur_folder, h5_files):
    """tiff_analysis.py:85-89."""
    if len(h5_files) == 1:
        process_single_h5_file(cur_folder, h5_files[0])
    else:
        process_multiple_h5_files(cur_folder, h5_files)


def get_h5_files_recursively(folder_path, suffixes=(".h5", ".npy")):
    """tiff_analysis.py:1113-1123 (also picks up .npy class maps)."""
    h5_files = {}
    for root, _, files in os.walk(folder_path):
        for file in files:
            if file.endswith(tuple(suffixes)):
                h5_files.setdefault(root, []).append(file)
    return h5_files

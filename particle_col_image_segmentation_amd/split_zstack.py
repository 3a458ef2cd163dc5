"""Frame enumeration in front of the GPU path: the reference's z-stack splitter as a drop-in.

``split_zstack.py`` of ssilverman16/particle_col_image_segmentation turns microscope stacks ``(Z, C, H, W)`` into one
single-page TIFF per slice and channel.  This module keeps its public names, arguments and every observable effect --
output names ``<stem>_z<i>_<CH>.tif`` (split_zstack.py:63), the per-sample and per-channel folders (:32-36, :44), the
input file being MOVED into its sample folder (:46-47), non-``.tif`` inputs only being moved (:48-49), and the rule
that a slice without exactly four channels is read as ``RFP, GFP`` (:53-55) -- but is organised around a small plan
object and the in-tree TIFF codec (``tifffile`` is not on the system python).  It is host I/O, not pixel work.

Additive: ``channel_map`` lets a 5-isotope stack be split under its own plane names; ``load_frames`` reads the
written pages back as one array for ``FramePipeline``.
"""
import os

from . import tiffio

FOUR_CHANNEL_NAMES = ("CY5", "RFP", "GFP", "DAPI")   # split_zstack.py:39
TWO_CHANNEL_NAMES = ("RFP", "GFP")                   # split_zstack.py:54
# (marker that must occur in the stem, tag that is cut out of names) -- split_zstack.py:21-26
_NAME_TAGS = (("CY5_RFP_GFP_DAPI_", "_CY5_RFP_GFP_DAPI"), ("RFP_GFP_", "_RFP_GFP"))
_STACK_SUFFIXES = ("_zstack.tif", "_mip.tif", "_mip.jpg")  # split_zstack.py:83


def create_folder(folder_name):
    os.makedirs(folder_name, exist_ok=True)


def get_clean_file_name(input_file):
    """(channel tag, sample path without tag / _zstack / _mip); names without a known tag are returned untouched."""
    stem = input_file.split(".")[0]
    for marker, tag in _NAME_TAGS:
        if marker in stem:
            sample = stem.replace(tag, "")
            for suffix in ("_zstack", "_mip"):
                sample = sample.replace(suffix, "")
            return (tag, sample)
    return ("", stem)


def create_channel_folder(destination, used_channels, channel_name):
    """Folder ``<moved file without .tif / _mip / tag>_<channel>``; created on demand."""
    root = destination
    for piece in (".tif", "_mip", used_channels):
        root = root.replace(piece, "")
    folder = "%s_%s" % (root, channel_name)
    create_folder(folder)
    return folder


class _ChannelPlan:
    """Which planes of a slice are written and under which names.  Once a slice shows a channel count the current map
    does not cover, the plan falls back to (RFP, GFP) for good, as the reference's loop variables do."""

    def __init__(self, channel_indices, channel_map):
        self.names = dict(channel_map) if channel_map else dict(enumerate(FOUR_CHANNEL_NAMES))
        self.indices = list(channel_indices)

    def select(self, n_channels):
        if n_channels != len(self.names) or (len(self.names) == 4 and n_channels != 4):
            self.names = dict(enumerate(TWO_CHANNEL_NAMES))
            self.indices = [0, 1]
        return [(idx, self.names[idx]) for idx in self.indices]


def process_tif(input_file, channel_indices, channel_map=None):
    """Move ``input_file`` into its sample folder and write one page per (slice, selected channel).
    Returns the list of written files (the reference returns nothing)."""
    stem_name = input_file.split("/")[-1].split(".")[0]
    tag, sample_folder = get_clean_file_name(input_file)
    create_folder(sample_folder)
    moved = os.path.join(sample_folder, os.path.basename(input_file))
    os.rename(input_file, moved)
    if not input_file.endswith(".tif"):
        return []
    plan = _ChannelPlan(channel_indices, channel_map)
    page_stem = stem_name.replace(tag, "")
    written = []
    for z_index, z_slice in enumerate(tiffio.imread(moved)):
        for plane, name in plan.select(z_slice.shape[0]):
            folder = create_channel_folder(moved, tag, name)
            target = os.path.join(folder, "%s_z%d_%s.tif" % (page_stem, z_index, name))
            tiffio.imwrite(target, z_slice[plane])
            written.append(target)
    return written


def create_output_folder(file):
    folder_name = file.split(".")[0]
    create_folder(folder_name)
    return folder_name


def process_folder(top_level_folder, channel_indices, channel_map=None):
    """Every ``*_zstack.tif`` / ``*_mip.tif`` / ``*_mip.jpg`` in the immediate, non-hidden sub-folders."""
    for entry in os.listdir(top_level_folder):
        sub = os.path.join(top_level_folder, entry)
        if entry.startswith(".") or not os.path.isdir(sub):
            continue
        for name in os.listdir(sub):
            if name.lower().endswith(_STACK_SUFFIXES):
                process_tif(os.path.join(sub, name), channel_indices, channel_map)


def load_frames(written_files):
    """Pages written by :func:`process_tif`, stacked in file order."""
    import numpy as np
    return np.stack([tiffio.imread(p) for p in written_files])


def main():
    folder_name = "3D05_6B07"  # split_zstack.py:94
    print("Processing folder: ", folder_name)
    process_folder(folder_name, [1, 2])  # 1 = RFP, 2 = GFP
    print("Processing complete")


if __name__ == "__main__":
    main()

"""Drop-in for the reference's ``split_zstack.py``: reorganise microscope z-stack TIFFs ``(Z, C, H, W)`` into
per-slice, per-channel single-page TIFFs and folders.  Same function names, arguments, file names
(``..._z{i}_{CH}.tif``, split_zstack.py:63), folder side effects (the input file is MOVED, :44-47) and the reference's
channel rule: a slice whose channel count is not 4 is treated as 2-channel {0: RFP, 1: GFP} (:53-55).  Pure host I/O:
this is the frame enumeration in front of the GPU path, not pixel work.  ``channel_map`` / ``five_channel`` are
additive options for 5-isotope stacks; the defaults reproduce the reference."""
import os

from . import tiffio


def create_folder(folder_name):
    """split_zstack.py:15-17."""
    if not os.path.exists(folder_name):
        os.makedirs(folder_name)


def get_clean_file_name(input_file):
    """split_zstack.py:19-30."""
    base_name = input_file.split(".")[0]
    if "CY5_RFP_GFP_DAPI_" in base_name:
        channels = "_CY5_RFP_GFP_DAPI"
        clean_file_name = base_name.replace(channels, "").replace("_zstack", "").replace("_mip", "")
    elif "RFP_GFP_" in base_name:
        channels = "_RFP_GFP"
        clean_file_name = base_name.replace(channels, "").replace("_zstack", "").replace("_mip", "")
    else:
        channels = ""
        clean_file_name = base_name
    return (channels, clean_file_name)


def create_channel_folder(destination, used_channels, channel_name):
    """split_zstack.py:32-36."""
    clean_name = destination.replace(".tif", "").replace("_mip", "").replace(used_channels, "")
    clean_name = clean_name + "_" + channel_name
    create_folder(clean_name)
    return clean_name


def process_tif(input_file, channel_indices, channel_map=None):
    """split_zstack.py:38-65."""
    channel_map = dict(channel_map or {0: "CY5", 1: "RFP", 2: "GFP", 3: "DAPI"})
    custom = len(channel_map) != 4 or set(channel_map) != {0, 1, 2, 3}
    input_file_end = input_file.split("/")[-1].split(".")[0]
    used_channels, clean_file_name = get_clean_file_name(input_file)
    create_folder(clean_file_name)
    destination = os.path.join(clean_file_name, os.path.basename(input_file))
    os.rename(input_file, destination)
    if not input_file.endswith(".tif"):
        return []
    zstack = tiffio.imread(destination)
    written = []
    for i, z_slice in enumerate(zstack):
        if z_slice.shape[0] != 4 and not (custom and z_slice.shape[0] == len(channel_map)):
            channel_map = {0: "RFP", 1: "GFP"}
            channel_indices = [0, 1]
        channel_names = [channel_map[channel_idx] for channel_idx in channel_indices]
        selected_channels = z_slice[channel_indices]
        for idx, channel in enumerate(selected_channels):
            channel_name = channel_names[idx]
            channel_folder = create_channel_folder(destination, used_channels, channel_name)
            channel_file_name = input_file_end.replace(used_channels, "")
            output_file = os.path.join(channel_folder, f"{channel_file_name}_z{i}_{channel_name}.tif")
            tiffio.imwrite(output_file, channel)
            written.append(output_file)
    return written


def create_output_folder(file):
    """split_zstack.py:67-71."""
    folder_name = file.split(".")[0]
    if not os.path.exists(folder_name):
        os.makedirs(folder_name)
    return folder_name


def process_folder(top_level_folder, channel_indices, channel_map=None):
    """split_zstack.py:73-89: immediate sub-directories only, *_zstack.tif | *_mip.tif | *_mip.jpg."""
    for folder in os.listdir(top_level_folder):
        folder_path = os.path.join(top_level_folder, folder)
        if not os.path.isdir(folder_path) or folder.startswith("."):
            continue
        for file in os.listdir(folder_path):
            if file.lower().endswith("_zstack.tif") or file.lower().endswith("_mip.tif") or file.lower().endswith("_mip.jpg"):
                process_tif(os.path.join(folder_path, file), channel_indices, channel_map)


def load_frames(written_files):
    """Frames produced by process_tif as one (N, H, W) array in file order (feeds FramePipeline after stacking)."""
    import numpy as np
    return np.stack([tiffio.imread(p) for p in written_files])


def main():
    """split_zstack.py:92-97."""
    channel_indices = [1, 2]
    folder_name = "3D05_6B07"
    print("Processing folder: ", folder_name)
    process_folder(folder_name, channel_indices)
    print("Processing complete")


if __name__ == "__main__":
    main()

"""MI355X-native (gfx950) implementation of the per-frame segmentation hot path
of ssilverman16/particle_col_image_segmentation.

Submodules (imported lazily so that ``import particle_col_image_segmentation_amd``
works on a machine without a GPU):

* ``tiff_analysis``      -- drop-in for the reference's ``tiff_analysis.py`` hot-path functions
* ``refine_boundaries``  -- ``refine_boundaries(boundary_map, threshold)`` (the reference's script as a function)
* ``pipeline``           -- batched device pipeline (frames -> label masks + ROI tables)
* ``ops``                -- thin torch wrappers over the C ABI of ``libpcseg.so`` (``include/pcseg.h``)
* ``synth``              -- seeded synthetic frames
"""
__version__ = "0.1.0"

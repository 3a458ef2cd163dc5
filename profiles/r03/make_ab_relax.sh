#!/bin/bash
# A/B trees for the split-layout relaxation (build container, repo root): the current tree built with
#   s0      PCSEG_WS_SPLIT=0 (pairs in LDS, round-2 sweep)
#   s1o2/3/4  PCSEG_WS_SPLIT=1 at 2 / 3 / 4 blocks per CU (175 / 168 / 128 registers)
set -e
rm -rf ab/s0 ab/s1o2 ab/s1o3 ab/s1o4
for v in "s0:-DPCSEG_WS_SPLIT=0" "s1o2:-DPCSEG_WS_SPLIT=1 -DPCSEG_WS_RELAX_OCC=2" "s1o3:-DPCSEG_WS_SPLIT=1 -DPCSEG_WS_RELAX_OCC=3" "s1o4:-DPCSEG_WS_SPLIT=1 -DPCSEG_WS_RELAX_OCC=4"; do
  name=${v%%:*}; flags=${v#*:}
  mkdir -p ab/$name
  cp -r particle_col_image_segmentation_amd include bench.py oracle ab/$name/
  rm -rf ab/$name/particle_col_image_segmentation_amd/build ab/$name/particle_col_image_segmentation_amd/libpcseg.so
  (cd ab/$name && PCSEG_EXTRA_FLAGS="$flags" python -c "from particle_col_image_segmentation_amd import build; build.build(force=True)")
done
ls -la ab/*/particle_col_image_segmentation_amd/libpcseg.so

#!/bin/bash
# Round-3 same-box A/B trees (run in the BUILD container, from the repo root; the trees travel to the GPU box with the
# snapshot, they are git-ignored):
#   ab/r2      the round-2 tree (commit d8be8f5), built as it was
#   ab/v0, v1  the current tree with the quadrant sweep's change tracking built as variant 0 / 1 (PCSEG_WS_SWEEP_MASKS)
set -e
rm -rf ab && mkdir -p ab/r2
git archive d8be8f5 | tar -x -C ab/r2
(cd ab/r2 && python -c "from particle_col_image_segmentation_amd import build; build.build(force=True)")
for v in 0 1; do
  mkdir -p ab/v$v
  cp -r particle_col_image_segmentation_amd include bench.py oracle ab/v$v/
  rm -rf ab/v$v/particle_col_image_segmentation_amd/build ab/v$v/particle_col_image_segmentation_amd/libpcseg.so
  (cd ab/v$v && PCSEG_EXTRA_FLAGS="-DPCSEG_WS_SWEEP_MASKS=$v" python -c "from particle_col_image_segmentation_amd import build; build.build(force=True)")
done
ls -la ab/*/particle_col_image_segmentation_amd/libpcseg.so

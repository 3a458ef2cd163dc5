#!/bin/bash
# A/B builds of libpcseg.so (build container, repo root): bash profiles/r04/make_ab.sh name:"flags" ...
#   -> ab/<name>/libpcseg.so, loaded with PCSEG_LIB=ab/<name>/libpcseg.so (see ab_run.sh)
set -e
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  rm -rf ab/$name; mkdir -p ab/$name
  python - "$name" "$flags" <<'PY'
import sys
sys.path.insert(0, ".")
from particle_col_image_segmentation_amd import build
print(build.build(out_dir="ab/" + sys.argv[1], extra_flags=sys.argv[2]))
PY
  rm -rf ab/$name/build
done
ls -la ab/*/libpcseg.so

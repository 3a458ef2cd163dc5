#!/bin/bash
# Same-box comparison of libpcseg.so builds with different -D tunables (PCSEG_EXTRA_FLAGS, see build.py):
#   bash profiles/ab_variants.sh "kernel name pattern" variants/a.so variants/b.so ...
# per build: ms/step of the serial bench, us/launch of the kernels matching the pattern, then ms/step of the default bench
P=particle_col_image_segmentation_amd
PAT=$1; shift
cp $P/libpcseg.so /tmp/libpcseg_keep.so
Q="--no-cpu-baseline --secondary-batch 0 --no-end-to-end"
for v in "$@"; do
  cp "$v" $P/libpcseg.so
  python bench.py $Q --serial --kernel-table --steps 4 2> /tmp/ab_err.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v serial', d['ms_per_step'])"
  grep -E "$PAT" /tmp/ab_err.txt | cut -c1-110
  python bench.py $Q 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v default', d['ms_per_step'])"
done
cp /tmp/libpcseg_keep.so $P/libpcseg.so

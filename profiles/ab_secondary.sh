#!/bin/bash
# Same-box comparison of builds on the quantised (tie) leg:  bash profiles/ab_secondary.sh FRAMES variants/a.so ...
P=particle_col_image_segmentation_amd
N=$1; shift
cp $P/libpcseg.so /tmp/libpcseg_keep.so
for v in "$@"; do
  cp "$v" $P/libpcseg.so
  python bench.py --no-cpu-baseline --no-end-to-end --steps 4 --warmup 2 --secondary-batch $N 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read())['secondary']; print('$v', d['frames'], d['value'], d['unit'], d['ms_per_step'], 'checked', d['parity_checked_frames'])"
done
cp /tmp/libpcseg_keep.so $P/libpcseg.so

#!/bin/bash
# Same-box A/B of two builds of libpcseg.so (box-to-box spread is larger than most single optimisations):
#   bash profiles/ab_compare.sh <a.so> <b.so> [rounds]
# alternates the two libraries, `rounds` bench runs each, prints ms/step per run.
P=particle_col_image_segmentation_amd
A=$1; B=$2; N=${3:-3}
cp $P/libpcseg.so /tmp/libpcseg_keep.so
for i in $(seq $N); do
  for v in "$A" "$B"; do
    cp "$v" $P/libpcseg.so
    python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'])"
  done
done
cp /tmp/libpcseg_keep.so $P/libpcseg.so

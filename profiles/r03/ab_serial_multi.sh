#!/bin/bash
# serial (single-stream) kernel tables of the current tree and of variant trees ab/<name>... on ONE box:
#   bash profiles/r03/ab_serial_multi.sh <out dir under gpurun_out> <kernel name pattern> <name>...
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; PAT=$2; shift 2
mkdir -p $OUT
Q="--no-cpu-baseline --secondary-batch 0 --batch64-frames 0 --graph-leg-steps 0 --no-end-to-end"
for V in main "$@" main; do
  D=$GRAFT_REPO_ROOT; [ $V != main ] && D=$GRAFT_REPO_ROOT/ab/$V
  (cd $D && timeout -k 10 200 python bench.py $Q --serial --steps 6 --kernel-table > $OUT/${V}_serial.json 2> $OUT/${V}_serial.err) || exit 1
  echo "== $V"; grep -E "$PAT" $OUT/${V}_serial.err
done

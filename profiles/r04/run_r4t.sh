#!/bin/bash
# twentieth GPU call of round 4: the default bench command after the roofline block prices both relaxation kernels together,
# then batches in flight: 6 / 8 / 10 / 12 lanes, twice each, interleaved
O=$GRAFT_REPO_ROOT/gpurun_out/r4t; mkdir -p $O; cd $GRAFT_REPO_ROOT
step() {
  local name=$1 t=$2; shift 2
  echo "=== $name"; timeout -k 10 $t "$@" > $O/$name.log 2> $O/$name.err; local rc=$?
  echo "=== $name rc=$rc"; tail -1 $O/$name.log | cut -c1-300
  if [ $rc -ge 124 ]; then echo "killed: stopping"; exit $rc; fi
}
step bench 600 python bench.py --kernel-table
Q="--no-shape-legs --no-cpu-baseline --secondary-batch 0 --batch64-frames 0 --graph-leg-steps 0 --no-end-to-end"
for rep in 1 2; do
  for L in 8 6 10 12; do
    step lanes${L}_$rep 200 python bench.py $Q --lanes $L
    python3 -c "import json;d=json.load(open('$O/lanes${L}_$rep.log'));print('== lanes $L #$rep', d['ms_per_step'], d['value'])"
  done
done

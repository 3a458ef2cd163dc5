#!/bin/bash
# twenty-third GPU call of round 4: the chain as hipGraph replays with 2 / 4 / 8 graphs in flight against eager launches from
# eight host threads, interleaved, twice
O=$GRAFT_REPO_ROOT/gpurun_out/r4w; mkdir -p $O; cd $GRAFT_REPO_ROOT
step() {
  local name=$1 t=$2; shift 2
  echo "=== $name"; timeout -k 10 $t "$@" > $O/$name.log 2> $O/$name.err; local rc=$?
  echo "=== $name rc=$rc"; tail -1 $O/$name.log | cut -c1-200
  if [ $rc -ge 124 ]; then echo "killed: stopping"; exit $rc; fi
}
Q="--no-shape-legs --no-cpu-baseline --secondary-batch 0 --batch64-frames 0 --graph-leg-steps 0 --no-end-to-end"
for rep in 1 2; do
  step eager_$rep 200 python bench.py $Q
  python3 -c "import json;d=json.load(open('$O/eager_$rep.log'));print('== eager #$rep', d['ms_per_step'], d['value'])"
  for L in 2 4 8; do
    step graph${L}_$rep 300 python bench.py $Q --graph --lanes $L
    python3 -c "import json;d=json.load(open('$O/graph${L}_$rep.log'));print('== graph lanes $L #$rep', d['ms_per_step'], d['value'])"
  done
done

#!/bin/bash
# thirty-first GPU call of round 4: graph replays with 3 / 4 / 5 / 6 in flight, interleaved, three times
O=$GRAFT_REPO_ROOT/gpurun_out/r5e; mkdir -p $O; cd $GRAFT_REPO_ROOT
Q="--no-shape-legs --no-cpu-baseline --secondary-batch 0 --batch64-frames 0 --graph-leg-steps 0 --no-end-to-end"
for rep in 1 2 3; do
  for L in 4 3 5 6; do
    timeout -k 10 300 python bench.py $Q --graph --lanes $L > $O/graph${L}_$rep.log 2> $O/graph${L}_$rep.err || exit 1
    python3 -c "import json;d=json.load(open('$O/graph${L}_$rep.log'));print('== graph lanes $L #$rep', d['ms_per_step'], d['value'])"
  done
done

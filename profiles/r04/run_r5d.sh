#!/bin/bash
# thirtieth GPU call of round 4: bench.py's launch modes -- auto (setup probe picks eager launches or graph replays), forced graph,
# forced eager; the full default command once, the short command three times each way
O=$GRAFT_REPO_ROOT/gpurun_out/r5d; mkdir -p $O; cd $GRAFT_REPO_ROOT
step() {
  local name=$1 t=$2; shift 2
  echo "=== $name"; timeout -k 10 $t "$@" > $O/$name.log 2> $O/$name.err; local rc=$?
  echo "=== $name rc=$rc"; tail -1 $O/$name.log | cut -c1-200
  if [ $rc -ne 0 ]; then tail -5 $O/$name.err; fi
  if [ $rc -ge 124 ]; then echo "killed: stopping"; exit $rc; fi
}
step bench 600 python bench.py
python3 -c "import json;d=json.load(open('$O/bench.log'));print('== default', d['ms_per_step'], d['value'], d['config']['launch'], d['config']['launch_probe_ms_per_step'], d.get('graph_replay',{}).get('ms_per_step'))"
Q="--no-shape-legs --no-cpu-baseline --secondary-batch 0 --batch64-frames 0 --graph-leg-steps 0 --no-end-to-end"
for rep in 1 2 3; do
  for M in auto eager graph; do
    step ${M}_$rep 300 python bench.py $Q --launch $M
    python3 -c "import json;d=json.load(open('$O/${M}_$rep.log'));print('== $M #$rep', d['ms_per_step'], d['value'], d['config']['launch'], d['config']['launch_probe_ms_per_step'])"
  done
done

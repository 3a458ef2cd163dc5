#!/bin/bash
# tenth GPU call of round 4: GPU tests on the committed tree (label pass at seven workgroups per CU),
# then A/B: EDT reach at 4 rows / 8 waves, region stats at 6 waves, label pass with three quads a thread
O=$GRAFT_REPO_ROOT/gpurun_out/r4j; mkdir -p $O; cd $GRAFT_REPO_ROOT
step() {
  local name=$1 t=$2; shift 2
  echo "=== $name"; timeout -k 10 $t "$@" > $O/$name.log 2> $O/$name.err; local rc=$?
  echo "=== $name rc=$rc"; tail -3 $O/$name.log
  if [ $rc -ge 124 ]; then echo "killed: stopping"; exit $rc; fi
}
step pytest 600 python -m pytest tests -m gpu -x -q
REPS=3 step ab 570 bash profiles/r04/ab_run.sh r4j/ab "watershed or fill_particle or dilate or region or edt" "edt_reach|region_stats|ws_uf_label4" reach4 stats6 l4q3
grep -v "^\.\.\.\|passed" $O/ab.log | tail -70

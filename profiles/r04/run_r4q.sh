#!/bin/bash
# seventeenth GPU call of round 4: GPU tests on the tree whose small-reach threshold epilogues (disk dilation, particle fill) run on the bit words
# (reach_bits_kernel), then A/B against the row-block pass (reachrows)
O=$GRAFT_REPO_ROOT/gpurun_out/r4q; mkdir -p $O; cd $GRAFT_REPO_ROOT
step() {
  local name=$1 t=$2; shift 2
  echo "=== $name"; timeout -k 10 $t "$@" > $O/$name.log 2> $O/$name.err; local rc=$?
  echo "=== $name rc=$rc"; tail -3 $O/$name.log
  if [ $rc -ge 124 ]; then echo "killed: stopping"; exit $rc; fi
}
step pytest 600 python -m pytest tests -m gpu -x -q
REPS=3 step ab 570 bash profiles/r04/ab_run.sh r4q/ab "fill_particle or dilate or threshold or edt" "edt_|reach_bits" reachrows
grep -v "^\.\.\.\|passed" $O/ab.log | tail -40

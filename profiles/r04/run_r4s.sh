#!/bin/bash
# nineteenth GPU call of round 4: GPU tests on the tree with the packed 16-bit EDT row search, then A/B against uint32 rows only
O=$GRAFT_REPO_ROOT/gpurun_out/r4s; mkdir -p $O; cd $GRAFT_REPO_ROOT
step() {
  local name=$1 t=$2; shift 2
  echo "=== $name"; timeout -k 10 $t "$@" > $O/$name.log 2> $O/$name.err; local rc=$?
  echo "=== $name rc=$rc"; tail -3 $O/$name.log
  if [ $rc -ge 124 ]; then echo "killed: stopping"; exit $rc; fi
}
step pytest 600 python -m pytest tests -m gpu -x -q
REPS=3 step ab 570 bash profiles/r04/ab_run.sh r4s/ab "edt or refine or local_max" "edt_" edt32
grep -v "^\.\.\.\|passed" $O/ab.log | tail -40

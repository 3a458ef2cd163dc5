#!/bin/bash
# twenty-second GPU call of round 4: GPU tests on the tree whose front end takes its medians from 6-bit histogram fields, two strips one below the other per thread (one
# subtraction and a bit count per pixel instead of five extract / compare / add triples), then A/B against the 5-bit form (med5)
O=$GRAFT_REPO_ROOT/gpurun_out/r4v; mkdir -p $O; cd $GRAFT_REPO_ROOT
step() {
  local name=$1 t=$2; shift 2
  echo "=== $name"; timeout -k 10 $t "$@" > $O/$name.log 2> $O/$name.err; local rc=$?
  echo "=== $name rc=$rc"; tail -3 $O/$name.log
  if [ $rc -ge 124 ]; then echo "killed: stopping"; exit $rc; fi
}
step pytest 600 python -m pytest tests -m gpu -x -q
REPS=3 step ab 570 bash profiles/r04/ab_run.sh r4v/ab "classmap or median or label" "classmap_median" med5
grep -v "^\.\.\.\|passed" $O/ab.log | tail -30

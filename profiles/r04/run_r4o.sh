#!/bin/bash
# fifteenth GPU call of round 4: GPU tests on the tree whose local-maxima stencil pass links plateau candidates from recorded
# link bits (no run-based tile pass over sparse keys), then A/B against the previous form (lmold)
O=$GRAFT_REPO_ROOT/gpurun_out/r4o; mkdir -p $O; cd $GRAFT_REPO_ROOT
step() {
  local name=$1 t=$2; shift 2
  echo "=== $name"; timeout -k 10 $t "$@" > $O/$name.log 2> $O/$name.err; local rc=$?
  echo "=== $name rc=$rc"; tail -3 $O/$name.log
  if [ $rc -ge 124 ]; then echo "killed: stopping"; exit $rc; fi
}
step pytest 600 python -m pytest tests -m gpu -x -q
REPS=3 step ab 570 bash profiles/r04/ab_run.sh r4o/ab "locmax or local_max or maxima or markers" "locmax" lmold
grep -v "^\.\.\.\|passed" $O/ab.log | tail -40

#!/bin/bash
# twenty-eighth GPU call of round 4: GPU tests on the tree whose fused sums pass reads the planes with non-temporal loads, then
# A/B: plain plane loads (sumsplain), the region passes' label images as non-temporal loads too (labnt)
O=$GRAFT_REPO_ROOT/gpurun_out/r5b; mkdir -p $O; cd $GRAFT_REPO_ROOT
step() {
  local name=$1 t=$2; shift 2
  echo "=== $name"; timeout -k 10 $t "$@" > $O/$name.log 2> $O/$name.err; local rc=$?
  echo "=== $name rc=$rc"; tail -3 $O/$name.log
  if [ $rc -ge 124 ]; then echo "killed: stopping"; exit $rc; fi
}
step pytest 600 python -m pytest tests -m gpu -x -q
REPS=3 step ab 570 bash profiles/r04/ab_run.sh r5b/ab "region" "region_sums2|region_stats" sumsplain labnt
grep -v "^\.\.\.\|passed" $O/ab.log | tail -34

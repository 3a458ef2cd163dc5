#!/bin/bash
# sixth GPU call of round 4: full GPU tests on the cleaned tree (+ the 2048 x 2048 sharded test), then A/B: relaxation blocks per
# CU (co-residency with the other batches' kernels) and the number of second-level grid rounds
O=$GRAFT_REPO_ROOT/gpurun_out/r4f; mkdir -p $O; cd $GRAFT_REPO_ROOT
step() {
  local name=$1 t=$2; shift 2
  echo "=== $name"; timeout -k 10 $t "$@" > $O/$name.log 2> $O/$name.err; local rc=$?
  echo "=== $name rc=$rc"; tail -3 $O/$name.log
  if [ $rc -ge 124 ]; then echo "killed: stopping"; exit $rc; fi
}
step pytest 600 python -m pytest tests -m gpu -x -q
REPS=3 step ab 560 bash profiles/r04/ab_run.sh r4f/ab "watershed" "ws_k2|ws_relax_kernel" relax3 relax2 k2r2 k2r1
grep -v "^\.\.\.\|passed" $O/ab.log | tail -60

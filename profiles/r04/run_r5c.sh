#!/bin/bash
# twenty-ninth GPU call of round 4: the class-map label image written with non-temporal stores (ntstore) against plain stores
O=$GRAFT_REPO_ROOT/gpurun_out/r5c; mkdir -p $O; cd $GRAFT_REPO_ROOT
step() {
  local name=$1 t=$2; shift 2
  echo "=== $name"; timeout -k 10 $t "$@" > $O/$name.log 2> $O/$name.err; local rc=$?
  echo "=== $name rc=$rc"; tail -3 $O/$name.log
  if [ $rc -ge 124 ]; then echo "killed: stopping"; exit $rc; fi
}
REPS=2 step ab 570 bash profiles/r04/ab_run.sh r5c/ab "label or classmap or region" "ccl_relabel|region_stats|region_sums2" ntstore
grep -v "^\.\.\.\|passed" $O/ab.log | tail -24

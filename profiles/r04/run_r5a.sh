#!/bin/bash
# twenty-seventh GPU call of round 4: the planes read with non-temporal loads (front end and fused sums pass) against plain loads
O=$GRAFT_REPO_ROOT/gpurun_out/r5a; mkdir -p $O; cd $GRAFT_REPO_ROOT
step() {
  local name=$1 t=$2; shift 2
  echo "=== $name"; timeout -k 10 $t "$@" > $O/$name.log 2> $O/$name.err; local rc=$?
  echo "=== $name rc=$rc"; tail -3 $O/$name.log
  if [ $rc -ge 124 ]; then echo "killed: stopping"; exit $rc; fi
}
REPS=3 step ab 700 bash profiles/r04/ab_run.sh r5a/ab "classmap or region_sums2 or median" "classmap_median|region_sums2" ntload
grep -v "^\.\.\.\|passed" $O/ab.log | tail -30

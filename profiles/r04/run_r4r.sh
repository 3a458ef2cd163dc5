#!/bin/bash
# eighteenth GPU call of round 4: the relaxation's statistics build (visits by iterations that changed something), then A/B of
# the parallel fixed-point test after an iteration that changed something (chk1: always, chk2: if at most two sweeps changed)
O=$GRAFT_REPO_ROOT/gpurun_out/r4r; mkdir -p $O; cd $GRAFT_REPO_ROOT
step() {
  local name=$1 t=$2; shift 2
  echo "=== $name"; timeout -k 10 $t "$@" > $O/$name.log 2> $O/$name.err; local rc=$?
  echo "=== $name rc=$rc"; tail -3 $O/$name.log
  if [ $rc -ge 124 ]; then echo "killed: stopping"; exit $rc; fi
}
PCSEG_LIB=$GRAFT_REPO_ROOT/ab/rstats/libpcseg.so step stats 200 python profiles/r04/time_ops.py refine 1
grep relax_stats $O/stats.log | sort | uniq -c | sort -k4,4n -k6,6n | tail -80
REPS=3 step ab 800 bash profiles/r04/ab_run.sh r4r/ab "watershed" "ws_relax" chk1 chk2
grep -v "^\.\.\.\|passed" $O/ab.log | tail -40

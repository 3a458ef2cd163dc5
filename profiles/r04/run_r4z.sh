#!/bin/bash
# twenty-sixth GPU call of round 4: iteration caps of the relaxation again, now that the late rounds walk lists: round 0 at
# 3 / 5 / 6 iterations (4 is the setting), later rounds at 8 (6)
O=$GRAFT_REPO_ROOT/gpurun_out/r4z; mkdir -p $O; cd $GRAFT_REPO_ROOT
step() {
  local name=$1 t=$2; shift 2
  echo "=== $name"; timeout -k 10 $t "$@" > $O/$name.log 2> $O/$name.err; local rc=$?
  echo "=== $name rc=$rc"; tail -3 $O/$name.log
  if [ $rc -ge 124 ]; then echo "killed: stopping"; exit $rc; fi
}
REPS=2 step ab 900 bash profiles/r04/ab_run.sh r4z/ab "watershed" "ws_relax" r0s3 r0s5 r0s6 rs8
grep -v "^\.\.\.\|passed" $O/ab.log | tail -50

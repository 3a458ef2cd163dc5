#!/bin/bash
# twenty-fourth GPU call of round 4: A/B of 32-bit in-frame offsets from uniform frame bases (scalar-base global accesses) in
# the watershed's union-find tile pass (off32) against 64-bit pixel addresses
O=$GRAFT_REPO_ROOT/gpurun_out/r4x; mkdir -p $O; cd $GRAFT_REPO_ROOT
step() {
  local name=$1 t=$2; shift 2
  echo "=== $name"; timeout -k 10 $t "$@" > $O/$name.log 2> $O/$name.err; local rc=$?
  echo "=== $name rc=$rc"; tail -3 $O/$name.log
  if [ $rc -ge 124 ]; then echo "killed: stopping"; exit $rc; fi
}
REPS=2 step ab 570 bash profiles/r04/ab_run.sh r4x/ab "watershed" "ws_uf_tile" off32
grep -v "^\.\.\.\|passed" $O/ab.log | tail -30

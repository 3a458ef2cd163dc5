#!/bin/bash
# first GPU call of round 4: GPU tests, quantised-component diagnostic, relaxation A/B, full default bench
O=$GRAFT_REPO_ROOT/gpurun_out/r4a; mkdir -p $O; cd $GRAFT_REPO_ROOT
step() { # name, timeout, command...: stops the whole script when a step was KILLED (timeout / signal), goes on after a plain failure
  local name=$1 t=$2; shift 2
  echo "=== $name"; timeout -k 10 $t "$@" > $O/$name.log 2> $O/$name.err; local rc=$?
  echo "=== $name rc=$rc"; tail -3 $O/$name.log
  if [ $rc -ge 124 ]; then echo "killed: stopping"; exit $rc; fi
}
step pytest 500 python -m pytest tests -m gpu -x -q
step diag 300 python profiles/r04/diag_quantised_components.py
step ab 600 bash profiles/r04/ab_run.sh r4a/ab "watershed" "ws_relax|ws_uf_tile" pipe min64 pipe64
step bench 400 python bench.py --kernel-table

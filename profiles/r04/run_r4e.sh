#!/bin/bash
# fifth GPU call of round 4: GPU tests (watershed + pipeline files) on the tree with the 1024-thread union-find K2 pass and the
# merged fills, then serial / overlapped A/B: pair-step relaxation sweep, directional-sweep K2
O=$GRAFT_REPO_ROOT/gpurun_out/r4e; mkdir -p $O; cd $GRAFT_REPO_ROOT
step() {
  local name=$1 t=$2; shift 2
  echo "=== $name"; timeout -k 10 $t "$@" > $O/$name.log 2> $O/$name.err; local rc=$?
  echo "=== $name rc=$rc"; tail -3 $O/$name.log
  if [ $rc -ge 124 ]; then echo "killed: stopping"; exit $rc; fi
}
step pytest 500 python -m pytest tests -m gpu -x -q
REPS=2 step ab 600 bash profiles/r04/ab_run.sh r4e/ab "watershed" "ws_k2|ws_relax_kernel|ws_uf_label_kernel|ws_pack" pair k2sweeps
grep -v "^\.\.\.\|passed" $O/ab.log | tail -60

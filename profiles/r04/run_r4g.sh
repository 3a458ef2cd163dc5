#!/bin/bash
# seventh GPU call of round 4: full GPU tests on the tree whose second-level kernels walk a device-side tile list, then serial
# and overlapped timings of it (three default-bench runs, legs off)
O=$GRAFT_REPO_ROOT/gpurun_out/r4g; mkdir -p $O; cd $GRAFT_REPO_ROOT
step() {
  local name=$1 t=$2; shift 2
  echo "=== $name"; timeout -k 10 $t "$@" > $O/$name.log 2> $O/$name.err; local rc=$?
  echo "=== $name rc=$rc"; tail -3 $O/$name.log
  if [ $rc -ge 124 ]; then echo "killed: stopping"; exit $rc; fi
}
step pytest 600 python -m pytest tests -m gpu -x -q
REPS=4 step ab 500 bash profiles/r04/ab_run.sh r4g/ab "watershed" "ws_k2|ws_relax_kernel|ws_pack|ws_uf_label|ws_uf_tile|ws_list" k2r1
grep -v "^\.\.\.\|passed" $O/ab.log | tail -60

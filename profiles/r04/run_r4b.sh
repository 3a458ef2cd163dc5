#!/bin/bash
# second GPU call of round 4: full GPU tests on the tree with the fused relabel + region table, then the A/B of the relaxation
# variants and of the unfused table pass
O=$GRAFT_REPO_ROOT/gpurun_out/r4b; mkdir -p $O; cd $GRAFT_REPO_ROOT
step() {
  local name=$1 t=$2; shift 2
  echo "=== $name"; timeout -k 10 $t "$@" > $O/$name.log 2> $O/$name.err; local rc=$?
  echo "=== $name rc=$rc"; tail -3 $O/$name.log
  if [ $rc -ge 124 ]; then echo "killed: stopping"; exit $rc; fi
}
step pytest 500 python -m pytest tests -m gpu -x -q
step ab 900 bash profiles/r04/ab_run.sh r4b/ab "watershed" "ws_relax|ws_uf_tile|relabel|region_stats|region_init|region_class" pipe min64 pipe64 nofold

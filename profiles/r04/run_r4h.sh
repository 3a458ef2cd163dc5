#!/bin/bash
# eighth GPU call of round 4: GPU tests on the tree with list-walking late relaxation rounds and branch-free fences, then A/B of
# the first list-walking round (4 = product, 2, 6, 12 = never)
O=$GRAFT_REPO_ROOT/gpurun_out/r4h; mkdir -p $O; cd $GRAFT_REPO_ROOT
step() {
  local name=$1 t=$2; shift 2
  echo "=== $name"; timeout -k 10 $t "$@" > $O/$name.log 2> $O/$name.err; local rc=$?
  echo "=== $name rc=$rc"; tail -3 $O/$name.log
  if [ $rc -ge 124 ]; then echo "killed: stopping"; exit $rc; fi
}
step pytest 600 python -m pytest tests -m gpu -x -q
REPS=4 step ab 560 bash profiles/r04/ab_run.sh r4h/ab "watershed" "ws_relax|ws_uf_label4|relabel_quads|ccl_border|ws_uf_border" nolist list2 list6
grep -v "^\.\.\.\|passed" $O/ab.log | tail -60

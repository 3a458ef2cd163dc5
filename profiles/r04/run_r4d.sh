#!/bin/bash
# fourth GPU call of round 4: full GPU tests on the tree with the union-find K2 pass and the leaner EDT threshold pass, then
# serial tables: the directional-sweep K2 build, the fused relabel + table variants (block heights, occupancy), the unfused one
O=$GRAFT_REPO_ROOT/gpurun_out/r4d; mkdir -p $O; cd $GRAFT_REPO_ROOT
step() {
  local name=$1 t=$2; shift 2
  echo "=== $name"; timeout -k 10 $t "$@" > $O/$name.log 2> $O/$name.err; local rc=$?
  echo "=== $name rc=$rc"; tail -3 $O/$name.log
  if [ $rc -ge 124 ]; then echo "killed: stopping"; exit $rc; fi
}
step pytest 500 python -m pytest tests -m gpu -x -q
REPS=1 step ab 800 bash profiles/r04/ab_run.sh r4d/ab "watershed or fill_particle or dilate or threshold" "ws_k2|edt_reach|relabel|region_stats|ws_relax_kernel" k2sweeps nofold fold8 fold16 foldocc5 fold8occ5
grep -v "^\.\.\.\|passed" $O/ab.log | tail -80

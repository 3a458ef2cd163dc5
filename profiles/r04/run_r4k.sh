#!/bin/bash
# eleventh GPU call of round 4: GPU tests on the tree with the clamped-base EDT row search, chunk kinds in the EDT reach pass, label pass at three quads a lane,
# then A/B: relaxation tiles loaded / stored twice (what do a revisited tile's loads and stores cost), EDT row with one guard cell (round 3 loop), EDT reach scanning every chunk
O=$GRAFT_REPO_ROOT/gpurun_out/r4k; mkdir -p $O; cd $GRAFT_REPO_ROOT
step() {
  local name=$1 t=$2; shift 2
  echo "=== $name"; timeout -k 10 $t "$@" > $O/$name.log 2> $O/$name.err; local rc=$?
  echo "=== $name rc=$rc"; tail -3 $O/$name.log
  if [ $rc -ge 124 ]; then echo "killed: stopping"; exit $rc; fi
}
step pytest 600 python -m pytest tests -m gpu -x -q
REPS=3 step ab 570 bash profiles/r04/ab_run.sh r4k/ab "watershed or fill_particle or dilate or edt" "ws_relax|edt_row|edt_reach|ws_uf_label4" loads2 stores2 guard1 kinds0
grep -v "^\.\.\.\|passed" $O/ab.log | tail -70

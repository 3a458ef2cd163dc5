#!/bin/bash
# thirteenth GPU call of round 4: ABLATION builds timed stage by stage (profiles/r04/time_ops.py): the watershed's union-find tile
# pass without its vertical unions (1), without its final finds (2), without both (3); the front end without its median (1),
# without its tile unions (2), without both (3); then A/B of the hook + compress form of the union-find tile pass (sv)
O=$GRAFT_REPO_ROOT/gpurun_out/r4m; mkdir -p $O; cd $GRAFT_REPO_ROOT
step() {
  local name=$1 t=$2; shift 2
  echo "=== $name"; timeout -k 10 $t "$@" > $O/$name.log 2> $O/$name.err; local rc=$?
  echo "=== $name rc=$rc"; tail -3 $O/$name.log
  if [ $rc -ge 124 ]; then echo "killed: stopping"; exit $rc; fi
}
lib() { [ "$1" = main ] && echo "" || echo "$GRAFT_REPO_ROOT/ab/$1/libpcseg.so"; }
for V in main uft1 uft2 uft3 main; do
  PCSEG_LIB=$(lib $V) step refine_$V 200 python profiles/r04/time_ops.py refine 4
  echo "== refine $V"; grep -E "ws_uf_tile|ws_uf_border|ws_uf_label4|ws_relax_kernel" $O/refine_$V.log
done
for V in main fe1 fe2 fe3 main; do
  PCSEG_LIB=$(lib $V) step classmap_$V 200 python profiles/r04/time_ops.py classmap 6
  echo "== classmap $V"; grep -E "classmap_median|ccl_border|relabel" $O/classmap_$V.log
done
REPS=3 step ab 570 bash profiles/r04/ab_run.sh r4m/ab "watershed" "ws_uf_tile|ws_uf_border|ws_uf_label4" sv
grep -v "^\.\.\.\|passed" $O/ab.log | tail -40

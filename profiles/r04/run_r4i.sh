#!/bin/bash
# ninth GPU call of round 4: GPU tests on the tree with launch bounds for residency (scalar registers) and the min-based fences,
# then A/B: label pass at seven workgroups per CU, front end at eight waves per SIMD, list-walking rounds from round 2 / 1
O=$GRAFT_REPO_ROOT/gpurun_out/r4i; mkdir -p $O; cd $GRAFT_REPO_ROOT
step() {
  local name=$1 t=$2; shift 2
  echo "=== $name"; timeout -k 10 $t "$@" > $O/$name.log 2> $O/$name.err; local rc=$?
  echo "=== $name rc=$rc"; tail -3 $O/$name.log
  if [ $rc -ge 124 ]; then echo "killed: stopping"; exit $rc; fi
}
step pytest 600 python -m pytest tests -m gpu -x -q
REPS=4 step ab 570 bash profiles/r04/ab_run.sh r4i/ab "watershed or label or classmap or edt" "ws_relax|ws_uf_label4|relabel_quads|ws_uf_border|edt_bits_kernel|classmap_median" l4occ7 fe8 list2 list1
grep -v "^\.\.\.\|passed" $O/ab.log | tail -70

#!/bin/bash
# fourteenth GPU call of round 4: ABLATION builds timed stage by stage (profiles/r04/time_ops.py): the local-maxima stencil pass
# without its "touches an equal non-candidate" test (1), without its tile unions (2), with a two-neighbour stencil (4), all
# three (7); the EDT threshold pass without its scans (1), without the bit scans of its staging (2), without both (3)
O=$GRAFT_REPO_ROOT/gpurun_out/r4n; mkdir -p $O; cd $GRAFT_REPO_ROOT
step() {
  local name=$1 t=$2; shift 2
  echo "=== $name"; timeout -k 10 $t "$@" > $O/$name.log 2> $O/$name.err; local rc=$?
  echo "=== $name rc=$rc"
  if [ $rc -ge 124 ]; then echo "killed: stopping"; exit $rc; fi
}
lib() { [ "$1" = main ] && echo "" || echo "$GRAFT_REPO_ROOT/ab/$1/libpcseg.so"; }
for V in main lm1 lm2 lm4 lm7 main; do
  PCSEG_LIB=$(lib $V) step locmax_$V 200 python profiles/r04/time_ops.py locmax 6
  echo "== locmax $V"; grep -E "locmax|ccl_border" $O/locmax_$V.log
done
for V in main reach1 reach2 reach3 main; do
  PCSEG_LIB=$(lib $V) step fill_$V 200 python profiles/r04/time_ops.py fill 6
  echo "== fill $V"; grep -E "edt_" $O/fill_$V.log
done

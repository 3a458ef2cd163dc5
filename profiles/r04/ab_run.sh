#!/bin/bash
# same-box A/B of library variants (GPU box):  bash profiles/r04/ab_run.sh <out> "<pytest -k expr>" "<kernel pattern>" name...
#   per variant (and the product library before and after): the GPU tests selected by -k against THAT library, the serial
#   kernel table (pattern lines), and two overlapped default-bench timings, interleaved
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; KEXPR=$2; PAT=$3; shift 3
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
Q="--no-shape-legs --no-cpu-baseline --secondary-batch 0 --batch64-frames 0 --graph-leg-steps 0 --no-end-to-end $Q_EXTRA"
lib() { [ "$1" = main ] && echo "" || echo "$GRAFT_REPO_ROOT/ab/$1/libpcseg.so"; }
for V in "$@"; do
  [ -n "$NOTEST" ] && continue   # ablation builds (wrong results by construction): timing only
  echo "== tests $V"
  PCSEG_LIB=$(lib $V) timeout -k 10 600 python -m pytest tests/test_gpu_primitives.py tests/test_gpu_edge_cases.py -q -x -m gpu -k "$KEXPR" 2>&1 | tail -2 || exit 1
done
for V in main "$@" main; do
  PCSEG_LIB=$(lib $V) timeout -k 10 200 python bench.py $Q --serial --steps 6 --kernel-table > $OUT/${V}_serial.json 2> $OUT/${V}_serial.err || exit 1
  echo "== serial $V: $(python3 -c "import json;d=json.load(open('$OUT/${V}_serial.json'));print(d['ms_per_step'])") ms/step"; grep -E "$PAT" $OUT/${V}_serial.err
done
for rep in $(seq 1 ${REPS:-2}); do
  for V in main "$@"; do
    PCSEG_LIB=$(lib $V) timeout -k 10 200 python bench.py $Q > $OUT/${V}_$rep.json 2> $OUT/${V}_$rep.err || exit 1
    echo "== overlapped $V #$rep: $(python3 -c "import json;d=json.load(open('$OUT/${V}_$rep.json'));print(d['ms_per_step'], d['value'])")"
  done
done

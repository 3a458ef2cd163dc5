#!/bin/bash
# same-box A/B (GPU box): round-2 tree, sweep variants 0 / 1 and the current tree, alternately, twice:
# overlapped ms per step and the serial per-kernel table (relaxation us per launch)
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-ab}
mkdir -p $OUT
Q2="--no-cpu-baseline --secondary-batch 0 --no-end-to-end"
Q3="--eager --no-cpu-baseline --secondary-batch 0 --batch64-frames 0 --no-end-to-end"
for rep in 1 2; do
  (cd $GRAFT_REPO_ROOT/ab/r2 && timeout -k 10 200 python bench.py $Q2 > $OUT/r2_$rep.json 2> $OUT/r2_$rep.err)
  for v in v0 v1; do
    (cd $GRAFT_REPO_ROOT/ab/$v && timeout -k 10 200 python bench.py $Q3 > $OUT/${v}_$rep.json 2> $OUT/${v}_$rep.err)
  done
  (cd $GRAFT_REPO_ROOT && timeout -k 10 200 python bench.py $Q3 > $OUT/main_$rep.json 2> $OUT/main_$rep.err)
done
(cd $GRAFT_REPO_ROOT/ab/r2 && timeout -k 10 200 python bench.py $Q2 --serial --steps 6 --kernel-table > $OUT/r2_serial.json 2> $OUT/r2_serial.err)
for v in v0 v1; do
  (cd $GRAFT_REPO_ROOT/ab/$v && timeout -k 10 200 python bench.py $Q3 --serial --steps 6 --kernel-table > $OUT/${v}_serial.json 2> $OUT/${v}_serial.err)
done
(cd $GRAFT_REPO_ROOT && timeout -k 10 200 python bench.py $Q3 --serial --steps 6 --kernel-table > $OUT/main_serial.json 2> $OUT/main_serial.err)
cd $OUT
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("*.json")):
    try:
        d = json.load(open(f))
        print("%-18s %8.3f ms/step  %9.1f Mpx/s" % (f, d["ms_per_step"], d["value"]))
    except Exception as e:
        print(f, "unreadable", e)
for f in sorted(glob.glob("*_serial.err")):
    tot = 0.0
    relax = None
    for l in open(f):
        p = l.split()
        if len(p) > 5 and p[1] == "ms" and p[3] == "launches":
            tot += float(p[0])
            if "ws_relax_kernel" in l:
                relax = float(p[4])
    print("%-18s serial kernel ms/step %.3f  ws_relax us/launch %s" % (f, tot / 6, relax))
PY

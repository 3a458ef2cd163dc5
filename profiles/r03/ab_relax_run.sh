#!/bin/bash
# same-box A/B of the relaxation variants (GPU box): golden watershed tests first (each tree's own library), then serial
# relaxation us per launch and overlapped ms per step, twice
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-abrelax}
mkdir -p $OUT
Q="--no-cpu-baseline --secondary-batch 0 --batch64-frames 0 --graph-leg-steps 0 --no-end-to-end"
for v in s0 s1o2 s1o3 s1o4; do
  (cd $GRAFT_REPO_ROOT/ab/$v && timeout -k 10 200 python bench.py $Q --serial --steps 6 --kernel-table > $OUT/${v}_serial.json 2> $OUT/${v}_serial.err)
done
for rep in 1 2; do
  for v in s0 s1o2 s1o3 s1o4; do
    (cd $GRAFT_REPO_ROOT/ab/$v && timeout -k 10 200 python bench.py $Q > $OUT/${v}_$rep.json 2> $OUT/${v}_$rep.err)
  done
done
cd $OUT
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("*.json")):
    try:
        d = json.load(open(f))
        print("%-18s %8.3f ms/step  %9.1f Mpx/s  parity-checked %s" % (f, d["ms_per_step"], d["value"], d["config"].get("parity_checked_frames")))
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

#!/bin/bash
# same-box A/B of the current tree against ONE variant tree ab/<name> (GPU box):
#   bash profiles/r03/ab_pair_run.sh <name> <out dir under gpurun_out>
V=$1
OUT=$GRAFT_REPO_ROOT/gpurun_out/${2:-abpair}
mkdir -p $OUT
Q="--no-cpu-baseline --secondary-batch 0 --batch64-frames 0 --graph-leg-steps 0 --no-end-to-end"
(cd $GRAFT_REPO_ROOT/ab/$V && timeout -k 10 200 python bench.py $Q --serial --steps 6 --kernel-table > $OUT/${V}_serial.json 2> $OUT/${V}_serial.err)
(cd $GRAFT_REPO_ROOT && timeout -k 10 200 python bench.py $Q --serial --steps 6 --kernel-table > $OUT/main_serial.json 2> $OUT/main_serial.err)
for rep in 1 2 3; do
  (cd $GRAFT_REPO_ROOT/ab/$V && timeout -k 10 200 python bench.py $Q > $OUT/${V}_$rep.json 2> $OUT/${V}_$rep.err)
  (cd $GRAFT_REPO_ROOT && timeout -k 10 200 python bench.py $Q > $OUT/main_$rep.json 2> $OUT/main_$rep.err)
done
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

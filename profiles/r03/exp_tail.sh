#!/bin/bash
# How much of `ms_per_step` at the driver's K = 20 is the pipeline's ramp (the last batches run with nothing beside
# them)?  Same box: K = 20 against K = 80 for 8 / 4 eager lanes and for 2 graph replays in flight.
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-tail}
mkdir -p $OUT
Q="--no-cpu-baseline --secondary-batch 0 --batch64-frames 0 --no-end-to-end"
cd $GRAFT_REPO_ROOT
for steps in 20 80; do
  for mode in "--eager --lanes 8" "--eager --lanes 4" "--eager --lanes 6" "--graph --lanes 2" "--graph --lanes 1"; do
    tag=$(echo "$mode" | tr -d ' -')
    timeout -k 10 200 python bench.py $mode --steps $steps $Q > $OUT/${tag}_k$steps.json 2> $OUT/${tag}_k$steps.err
  done
done
cd $OUT
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("*.json")):
    try:
        d = json.load(open(f))
        print("%-24s %8.3f ms/step  %9.1f Mpx/s" % (f, d["ms_per_step"], d["value"]))
    except Exception as e:
        print(f, "unreadable", e)
PY

#!/bin/bash
# lanes on one box (default bench, no side legs):  bash profiles/ab_lanes.sh 3 8 12 ...
Q="--no-cpu-baseline --secondary-batch 0 --no-end-to-end --steps ${STEPS:-40}"
for l in "$@"; do
  python bench.py $Q --lanes $l 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lanes $l', d['value'], d['ms_per_step'])"
done

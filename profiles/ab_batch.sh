#!/bin/bash
# Sub-batch size against lanes on one box: Mpixels/s of the default (overlapped) bench
Q="--no-cpu-baseline --secondary-batch 0 --no-end-to-end"
for b in 64 32 16; do
  for l in 2 3 4; do
    steps=$((20 * 64 / b)); warm=$((4 * 64 / b))
    python bench.py $Q --batch $b --lanes $l --steps $steps --warmup $warm 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('batch $b lanes $l', d['value'], d['unit'], d['ms_per_step'])"
  done
done

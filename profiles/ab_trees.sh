#!/bin/bash
# Same-box A/B of two source trees (box-to-box spread is larger than most single optimisations):
#   ARGS_A="..." ARGS_B="..." bash profiles/ab_trees.sh <tree A> <tree B> [rounds]
A=$1; B=$2; N=${3:-2}
for i in $(seq $N); do
  (cd $A && python bench.py --no-cpu-baseline $ARGS_A 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('A $A', d['ms_per_step'])")
  (cd $B && python bench.py --no-cpu-baseline $ARGS_B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('B $B', d['ms_per_step'])")
done

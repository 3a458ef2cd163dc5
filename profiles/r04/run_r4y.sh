#!/bin/bash
# twenty-fifth GPU call of round 4: does the host's OpenMP thread pool (torch intra-op threads) get in the way of the eight launch
# threads?  default environment against OMP_NUM_THREADS=1 (what torchrun sets), interleaved, four times
O=$GRAFT_REPO_ROOT/gpurun_out/r4y; mkdir -p $O; cd $GRAFT_REPO_ROOT
Q="--no-shape-legs --no-cpu-baseline --secondary-batch 0 --batch64-frames 0 --graph-leg-steps 0 --no-end-to-end"
nproc; python3 -c "import torch;print('torch threads', torch.get_num_threads())"
for rep in 1 2 3 4; do
  timeout -k 10 200 python bench.py $Q > $O/default_$rep.log 2> $O/default_$rep.err || exit 1
  python3 -c "import json;d=json.load(open('$O/default_$rep.log'));print('== default env #$rep', d['ms_per_step'], d['value'])"
  OMP_NUM_THREADS=1 timeout -k 10 200 python bench.py $Q > $O/omp1_$rep.log 2> $O/omp1_$rep.err || exit 1
  python3 -c "import json;d=json.load(open('$O/omp1_$rep.log'));print('== OMP_NUM_THREADS=1 #$rep', d['ms_per_step'], d['value'])"
done

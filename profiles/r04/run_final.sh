#!/bin/bash
# final collection of round 4 (two GPU calls: `run_final.sh a`, then `run_final.sh b`): GPU tests, the default bench command (all legs), rocprofv3 kernel stats + HBM PMC passes
# (profiles/collect_r04.sh), SQ counter passes (profiles/pmc_sq.sh), one rank under torchrun over RCCL
O=$GRAFT_REPO_ROOT/gpurun_out/r4final; mkdir -p $O; cd $GRAFT_REPO_ROOT
step() {
  local name=$1 t=$2; shift 2
  echo "=== $name"; timeout -k 10 $t "$@" > $O/$name.log 2> $O/$name.err; local rc=$?
  echo "=== $name rc=$rc"; tail -2 $O/$name.log | cut -c1-600
  if [ $rc -ge 124 ]; then echo "killed: stopping"; exit $rc; fi
}
if [ "$1" = a ]; then
step pytest 500 python -m pytest tests -m gpu -x -q
step bench 600 python bench.py --kernel-table
exit 0
fi
step collect 600 bash profiles/collect_r04.sh
step pmc_sq 300 bash profiles/pmc_sq.sh "ws_relax_kernel"
step torchrun 250 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu-baseline --no-shape-legs --secondary-batch 0 --batch64-frames 0 --graph-leg-steps 0

#!/bin/bash
# SQ / LDS counter passes of the serial bench (what are the waves of a kernel waiting for?):
#   bash profiles/pmc_sq.sh "kernel name pattern"
# one rocprofv3 --pmc run per counter group (no tracing besides --kernel-trace), then per-kernel sums
set -x
PAT=${1:-ws_relax_kernel}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_sq
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
Q="--no-shape-legs --no-cpu-baseline --secondary-batch 0 --batch64-frames 0 --graph-leg-steps 0 --no-end-to-end --serial --steps 2 --warmup 1"
i=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INSTS_LDS" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_INSTS_VALU"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/g$i -o t -- python3 $GRAFT_REPO_ROOT/bench.py $Q > /dev/null 2> $OUT/g$i.log
done
cd $GRAFT_REPO_ROOT
python profiles/pmc_sq_sum.py gpurun_out/pmc_sq "$PAT" > gpurun_out/pmc_sq/summary.txt
rm -rf gpurun_out/pmc_sq/g1 gpurun_out/pmc_sq/g2 gpurun_out/pmc_sq/g3
cat gpurun_out/pmc_sq/summary.txt

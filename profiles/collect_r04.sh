#!/bin/bash
# Round-4 profile collection on the GPU box (run from the repo root through gpurun):
#   bash profiles/collect_r04.sh
# 1. rocprofv3 --kernel-trace --stats of the DEFAULT bench command (what the driver runs; its legs switched off)
# 2. the same for the --serial command (clean per-kernel durations: one stream, nothing alongside)
# 3. two PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs as the guide prescribes) of the serial command
# Everything lands under gpurun_out/prof_r04; profiles/summarize_rocprof.py turns it into the tracked summaries.
# (the program itself follows `--`: no env / shell hop between rocprofv3 and python3)
set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_r04
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
Q="--no-shape-legs --no-cpu-baseline --secondary-batch 0 --batch64-frames 0 --graph-leg-steps 0 --no-end-to-end"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/default -o t -- python3 $GRAFT_REPO_ROOT/bench.py $Q > $OUT/bench_default_under_rocprof.json 2> $OUT/default.log
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 $GRAFT_REPO_ROOT/bench.py --serial --steps 6 $Q > $OUT/bench_serial_under_rocprof.json 2> $OUT/trace.log
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o t -- python3 $GRAFT_REPO_ROOT/bench.py --serial --steps 3 --warmup 0 $Q > /dev/null 2> $OUT/fetch.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o t -- python3 $GRAFT_REPO_ROOT/bench.py --serial --steps 3 --warmup 0 $Q > /dev/null 2> $OUT/write.log
cd $GRAFT_REPO_ROOT
python profiles/summarize_rocprof.py gpurun_out/prof_r04 r04 gpurun_out/prof_r04/summary
ls -la gpurun_out/prof_r04/summary
# raw traces and counter dumps are large (gpurun merges at most 64 MiB back): keep the summaries and rocprofv3's own
# per-kernel stats of the two traced runs
mkdir -p gpurun_out/prof_r04/summary
for d in default trace; do
  f=$(find gpurun_out/prof_r04/$d -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" gpurun_out/prof_r04/summary/r04_rocprofv3_kernel_stats_${d}.csv
done
cp gpurun_out/prof_r04/bench_default_under_rocprof.json gpurun_out/prof_r04/bench_serial_under_rocprof.json gpurun_out/prof_r04/summary/
rm -rf gpurun_out/prof_r04/default gpurun_out/prof_r04/trace gpurun_out/prof_r04/fetch gpurun_out/prof_r04/write

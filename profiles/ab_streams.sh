mkdir -p gpurun_out/r2e
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2e/pytest.log 2>&1; tail -5 gpurun_out/r2e/pytest.log
Q="--no-cpu-baseline --secondary-batch 0 --no-end-to-end"
run() { python bench.py $Q "$@" 2> gpurun_out/r2e/err_$N.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$N', d['value'], d['ms_per_step'])"; }
N=a_l2_multi run
N=b_l2_single run --single-class-stream
N=c_l1_multi run --lanes 1
N=d_l1_single run --lanes 1 --single-class-stream
N=e_l3_multi run --lanes 3
N=f_l2_multi_again run
N=g_host_overhead run --batch 1 --size 64 --steps 100
N=h_serial run --serial --kernel-table --steps 6
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2e/trace -o t -- python3 $GRAFT_REPO_ROOT/bench.py --serial --steps 2 --warmup 1 $Q > $GRAFT_REPO_ROOT/gpurun_out/r2e/trace.log 2>&1; cd $GRAFT_REPO_ROOT && python profiles/relax_rounds.py gpurun_out/r2e/trace | tee gpurun_out/r2e/rounds.txt; rm -rf gpurun_out/r2e/trace

#!/bin/bash
# third GPU call of round 4: sensitivity builds of the relaxation sweep (LDS instruction count), plain stores, fold block heights
O=$GRAFT_REPO_ROOT/gpurun_out/r4c; mkdir -p $O; cd $GRAFT_REPO_ROOT
REPS=1 timeout -k 10 1000 bash profiles/r04/ab_run.sh r4c/ab "watershed" "ws_relax|relabel|region_stats" plainst addatom addread fold8 fold16 nofold > $O/ab.log 2> $O/ab.err
echo "ab rc=$?"; grep -v "^\.\.\.\|passed" $O/ab.log | tail -60

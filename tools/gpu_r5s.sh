#!/bin/bash
# round 5, call 19: the per-launch sweeps again WITHOUT the order bias (untimed pass + three interleaved rounds): schedule, tile order, narrow tile
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
( PF_S3_T192=2 timeout 600 python tools/persist_probe.py envsweep:PF_S3_FLAGS=0,8,0 ) > $O/r5s_narrow_sweep.md 2>&1
echo "== narrow (0, 8, 0 = the same arm twice)"; cat $O/r5s_narrow_sweep.md
( PF_S3_T192=2 timeout 600 python tools/persist_probe.py envsweep:PF_S3_BLOAD=0,1 ) > $O/r5s_bload_sweep.md 2>&1
echo "== bload"; cat $O/r5s_bload_sweep.md
( timeout 600 python tools/persist_probe.py envsweep:PF_S3_ORDER=1,2 ) > $O/r5s_order_sweep.md 2>&1
echo "== order"; cat $O/r5s_order_sweep.md

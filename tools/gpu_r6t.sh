#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp; O=gpurun_out
( timeout 600 python -m pytest tests/test_hip_ops_gpu.py -m gpu -q -x -k "conv1x1_split3" -s 2>&1 | tail -30 ) > $O/r6t_checks_new.log 2>&1
( PF_C1_BM=128 timeout 600 python -m pytest tests/test_hip_ops_gpu.py -m gpu -q -x -k "conv1x1_split3" -s 2>&1 | tail -30 ) > $O/r6t_checks_new128.log 2>&1
( PF_C1_WREG=0 timeout 600 python -m pytest tests/test_hip_ops_gpu.py -m gpu -q -x -k "conv1x1_split3" -s 2>&1 | tail -30 ) > $O/r6t_checks_dma.log 2>&1
python tools/conv1x1_time.py > $O/r6t_conv1x1.log 2>&1
PF_C1_BM=128 python tools/conv1x1_time.py > $O/r6t_conv1x1_bm128.log 2>&1
PF_C1_WREG=0 python tools/conv1x1_time.py > $O/r6t_conv1x1_dma.log 2>&1
tail -n 30 $O/r6t_checks_new.log $O/r6t_checks_new128.log $O/r6t_checks_dma.log $O/r6t_conv1x1.log $O/r6t_conv1x1_bm128.log $O/r6t_conv1x1_dma.log

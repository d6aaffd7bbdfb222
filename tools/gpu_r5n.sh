#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
( timeout 600 python tools/op_roofline.py fp32 $O/r5n_op_roofline_fp32.md $O/r5n_op_roofline_fp32.json ) > $O/r5n_oproof.log 2>&1
tail -3 $O/r5n_oproof.log; head -45 $O/r5n_op_roofline_fp32.md

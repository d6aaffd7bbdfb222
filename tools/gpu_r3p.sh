#!/bin/bash
# round-3 GPU session P: per-(op,shape) table of the f32 pass, whole-pass kernel stats
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
( timeout 400 python tools/op_roofline.py fp32 $O/r3_op_roofline_fp32.md $O/r3_op_roofline_fp32.json ) > $O/r3p_op_roofline.log 2>&1
sed -n 1,75p $O/r3_op_roofline_fp32.md

#!/bin/bash
mkdir -p gpurun_out
timeout 600 python tools/op_roofline.py fp32 gpurun_out/r3ad_op_roofline_fp32.md gpurun_out/r3ad_op_roofline_fp32.json > gpurun_out/r3ad_op.log 2>&1
tail -5 gpurun_out/r3ad_op.log

#!/bin/bash
mkdir -p gpurun_out
timeout 300 python tools/dvfs_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3_dvfs_probe.md

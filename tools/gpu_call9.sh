#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
for m in 0 1 256128 256256; do ( PF_GEMM_PERSIST=$m timeout 100 python bench.py --gemm-sweep --dtype bf16 --only qkv,proj,fc1,fc2 ) > $O/r2c9_sweep_persist_$m.log 2>&1; done
( timeout 400 python -m pytest tests/test_persistent_gemm_gpu.py -m gpu -q -x 2>&1 | tail -4 ) > $O/r2c9_persist_tests.log 2>&1
( timeout 200 python -m pytest tests/test_hip_ops_gpu.py -m gpu -q -x -k "resize" 2>&1 | tail -4 ) > $O/r2c9_resize.log 2>&1
for m in 0 1 256128 256256; do echo "== PF_GEMM_PERSIST=$m"; grep -v amdgpu $O/r2c9_sweep_persist_$m.log; done; tail -n 3 $O/r2c9_persist_tests.log $O/r2c9_resize.log

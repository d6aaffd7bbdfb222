#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 600 python tools/wino_decomp.py ) > gpurun_out/r6m_wino_n256_decomp.md 2>&1; grep -v amdgpu.ids gpurun_out/r6m_wino_n256_decomp.md

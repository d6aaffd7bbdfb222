#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 600 python tools/subbatch_probe.py ) > gpurun_out/r5l_subbatch_probe.md 2>&1
cat gpurun_out/r5l_subbatch_probe.md

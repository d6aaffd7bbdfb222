#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
( timeout 120 ./tools/coissue_probe ) > gpurun_out/r6n_coissue_probe.log 2>&1; cat gpurun_out/r6n_coissue_probe.log

#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp; O=gpurun_out
( timeout 1500 python tools/image_ab.py --steps 4 --rounds 3 "" "PF_CONV1X1_SPLIT3=0" "PF_SWIN_MFMA=0" "PF_CONV1X1_SPLIT3=0,PF_SWIN_MFMA=0" ) > $O/r6u_image_ab.md 2> $O/r6u_image_ab.err
( PF_LIB_PATH=$PWD/patchfusion_amd/libpf_geluocml.so timeout 600 python tools/image_ab.py --steps 4 --rounds 2 "" ) > $O/r6u_image_geluocml.md 2>> $O/r6u_image_ab.err
( timeout 600 python tools/image_ab.py --steps 4 --rounds 2 "" ) > $O/r6u_image_gelufast.md 2>> $O/r6u_image_ab.err
( PF_LIB_PATH=$PWD/patchfusion_amd/libpf_geluocml.so timeout 600 python tools/image_ab.py --steps 4 --rounds 2 "" ) > $O/r6u_image_geluocml2.md 2>> $O/r6u_image_ab.err
cat $O/r6u_image_ab.md $O/r6u_image_geluocml.md $O/r6u_image_gelufast.md $O/r6u_image_geluocml2.md
( timeout 2400 python -m pytest tests -m gpu -q -x --durations=5 2>&1 | tail -15 ) > $O/r6u_pytest_gpu.log 2>&1; cat $O/r6u_pytest_gpu.log

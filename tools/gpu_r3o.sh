#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
( timeout 300 python -m pytest tests/test_hip_ops_gpu.py -q -k "vit_attention" 2>&1 | tail -3 ) > $O/r3o_pytest_attn.log 2>&1; cat $O/r3o_pytest_attn.log
( PF_ATTN_QG=1 timeout 100 python tools/attn_probe.py fp32 20; PF_ATTN_QG=2 timeout 100 python tools/attn_probe.py fp32 20 ) 2>&1 | grep vit_attention > $O/r3o_attn_time.log; cat $O/r3o_attn_time.log
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-roofline"
( timeout 300 $B ) > $O/r3o_bench.json 2> $O/r3o_bench.err; python -c "import json; print(json.load(open('$O/r3o_bench.json'))['ms_per_step'])"

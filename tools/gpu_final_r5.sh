#!/bin/bash
# round 5, final measurement session (every file lands in gpurun_out/r5_*; the ones judged are copied to profiles/)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
bash tools/gpu_session.sh r5 bench:--steps,20,--warmup,5 stats:fp32 pmc:fp32:gemm_split3_persist192 pmc:bf16:conv3x3_halo oproof \
     bench:--encoder,vits,--process-num,4,--steps,10,--warmup,2,--no-cpu-baseline \
     bench:--split,2x4,--steps,10,--warmup,3,--no-secondary,--no-cpu-baseline,--no-roofline \
     probe:mem_probe 2>&1 | tail -c 6000
export TMPDIR=/tmp
rm -rf /tmp/rp_pass
( timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_pass -o ro -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary ) > gpurun_out/r5_pass_stats.log 2>&1
f=$(find /tmp/rp_pass -name '*kernel_trace.csv' | head -1)
python tools/rocprof_summary.py "$f" gpurun_out/r5_image_kernel_stats.md "one f32 image pass x3 (1 warm-up + 2 timed), final tree" > /dev/null 2>&1
head -20 gpurun_out/r5_image_kernel_stats.md

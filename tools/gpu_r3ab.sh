#!/bin/bash
# session AB: log-binomial table check + time, schedule variants (process_num x streams)
mkdir -p gpurun_out
timeout 300 python tools/gpu_selfcheck.py bins_ops > gpurun_out/r3ab_check.log 2>&1
grep -E "PASS|FAIL" gpurun_out/r3ab_check.log | cut -c1-250
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3ab_logbinom.log
import torch
from patchfusion_amd.hip_ops import ops
pt = torch.rand(8, 392, 518, 4, device="cuda") + 0.1
cen = torch.rand(8, 224, 296, 64, device="cuda").sort(-1).values
d = torch.empty(8, 392, 518, device="cuda")
ops.logbinom_depth(pt, cen, d, 0.0212, 50.0)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): ops.logbinom_depth(pt, cen, d, 0.0212, 50.0)
e1.record(); torch.cuda.synchronize()
print(f"logbinom_depth (8,392,518) centres (224,296): {e0.elapsed_time(e1) / 20 * 1e3:.1f} us (was 285.1)")
PY
for v in "8 2" "4 2" "4 4" "8 3" "16 2"; do
  set -- $v
  PF_STREAMS=$2 timeout 300 python bench.py --process-num $1 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-secondary 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('process_num=$1 streams=$2', d['ms_per_step'], d['value'])" | tee -a gpurun_out/r3ab_schedule.log
done

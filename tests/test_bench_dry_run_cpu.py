"""The driver's multi-GPU command line, executed end to end once WITHOUT GPUs (round-4 review item 7c): `python -m torch.distributed.run --nnodes=1
--nproc-per-node 8 --master-addr 127.0.0.1 --master-port P bench.py --gpus 8 --steps K --warmup W` with --dry-run -- gloo instead of RCCL, CPU tensors,
the torch stand-in op set, the SAME 8 x 8 tile grid sharded 8 ways (8 tiles per rank): rendezvous from the environment, shard_patches model, same-image
guard, the all-gather of the tile depths, barrier + max-over-ranks timing, ONE JSON line from rank 0, common exit."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("n", [8, 2])
def test_scale_command_line_dry_run(n):
    env = dict(os.environ, OMP_NUM_THREADS="1", PF_BENCH_DRY_RUN="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), "bench.py", "--gpus", str(n), "--steps", "1", "--warmup", "0"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                      # rank 0 prints ONE JSON line
    j = json.loads(lines[0])
    grid = {8: (8, 8), 2: (4, 8)}[n]
    assert j["dry_run"] is True and j["n_gpus"] == n and j["steps"] == 1 and j["warmup"] == 0 and len(j["rank_seconds"]) == n
    assert j["depth_shape"] == [1, 1, grid[0] * 112, grid[1] * 154]
    assert abs(j["ms_per_step"] - max(j["rank_seconds"]) * 1e3) < 1.0 and j["value"] > 0
    assert j["scaling"] == ("strong" if n == 8 else "weak")
    # the communicator's self-diagnosis (bench.gather_diagnosis; "rccl" on GPUs, gloo here): its size after a real collective, rows in tile order, the gather's own time
    c = j["rccl"]
    assert c["ranks"] == n and c["backend"] == "gloo" and c["rows_in_tile_order"] is True and c["tile_gather_ms"] > 0
    assert c["gathered_shape"] == [grid[0] * grid[1], 112, 154]

"""The N > 1 launch of bench.py on real RCCL, as far as a single-GPU box can take it: `python -m torch.distributed.run --nproc-per-node 1`
with NR_DIST_FORCE=1 runs the SAME calls as N > 1 (init_process_group("nccl"), plan + export on rank 0, broadcast_object_list of the
manifest, ONE device broadcast of each packed bf16 arena, barrier-bracketed timing, MAX all-reduce of the elapsed time) with a world of one.
The receiving side (import into a fresh network, bit-identical pipeline) is tests/test_fullsize_gpu.py's replay; the rank logic at
world_size 2 runs on gloo in tests/test_distributed_gloo.py."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_single_rank_through_rccl():
    import socket
    env = dict(os.environ, NR_DIST_FORCE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    with socket.socket() as sk:                      # a free rendezvous port (a fixed one collides when suites share a box)
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1", "--ddim-steps", "4",
           "--no-cpu-baseline", "--no-psnr"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["config"]["output_finite"]
    assert d["scaling"] == "weak"


@pytest.mark.gpu
def test_bench_self_launch_world_of_one_through_rccl():
    """`python bench.py --gpus N` without a launcher (the driver's SCALE command form): bench.self_launch starts the rank processes itself
    before any GPU call.  One GPU here, so --self-launch forces the launcher at N = 1; the rank takes the RCCL path (NR_DIST_FORCE)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--self-launch", "--steps", "1", "--warmup", "1", "--ddim-steps", "4",
           "--no-cpu-baseline", "--no-psnr", "--no-end-to-end", "--launch-timeout", "800"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["rccl_ranks_seen"] == 1 and d["value"] > 0 and d["config"]["output_finite"]

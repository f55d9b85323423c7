"""`python bench.py --gpus N` with no launcher around it (the driver's SCALE command form) starts its own N rank processes before any
GPU call (bench.self_launch; reference layout: `accelerate launch`, one process per GPU, train_neurons.sh:92-96, rank-strided clips
scripts/neuroclips_video.py:39-40,323).  CPU test of the launcher + rank logic on gloo; the RCCL variant is tests/test_dist_gpu.py."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env=None, n=2):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra_env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "3", "--launch-dry-run",
                           "--launch-timeout", "240"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=400)


def test_self_launch_two_ranks_one_json_line():
    r = _run()
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout                       # ONE line on stdout, rank 0's
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2
    assert [x["rank"] for x in d["ranks"]] == [0, 1] and [x["local_rank"] for x in d["ranks"]] == [0, 1]
    assert d["ranks"][0]["clips"] == [0, 2, 4] and d["ranks"][1]["clips"] == [1, 3, 5]     # rank-strided, every clip once
    assert d["slowest_rank_time"] == 2.0                   # MAX over ranks


def test_self_launch_reports_a_failed_rank():
    r = _run({"NR_LAUNCH_DRY_RUN_FAIL_RANK": "1"})
    assert r.returncode != 0
    assert "rank 1 exited with code 3" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_parent_makes_no_gpu_call_before_the_ranks_exist():
    """the launcher branch sits before torch.cuda.set_device / any HIP call of main()"""
    src = open(os.path.join(ROOT, "bench.py")).read()
    main_src = src[src.index("def main():"):]
    assert main_src.index("return self_launch(args)") < main_src.index("torch.cuda.set_device")
    body = src[src.index("def self_launch(args):"):src.index("def launch_dry_run(args):")]
    assert "os.exec" not in body and "set_device" not in body and "is_available" not in body
    assert "torch.cuda" not in body and "device_count" not in body      # not even a device count through the runtime


def test_gpu_count_comes_from_the_kfd_topology(tmp_path, monkeypatch):
    """bench.count_gpus_without_runtime: nodes with simd_count > 0 are GPUs (the CPU node has 0); *_VISIBLE_DEVICES cut the count down;
    an unreadable topology gives None (the launcher then skips its pre-check and lets a rank fail)."""
    sys.path.insert(0, ROOT)
    import bench
    for i, simd in enumerate([0, 1024, 1024, 1024]):
        d = tmp_path / str(i)
        d.mkdir()
        (d / "properties").write_text(f"cpu_cores_count {16 if simd == 0 else 0}\nsimd_count {simd}\nmem_banks_count 1\n")
    for v in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(v, raising=False)
    assert bench.count_gpus_without_runtime(str(tmp_path)) == 3
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,2")
    assert bench.count_gpus_without_runtime(str(tmp_path)) == 2
    assert bench.count_gpus_without_runtime(str(tmp_path / "missing")) is None

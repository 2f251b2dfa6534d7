"""`python bench.py --gpus N` must start N ranks by itself (the driver's N > 1 command form when no launcher is
used) and print ONE JSON line from rank 0 with n_gpus == N.  Rehearsed on CPUs with `--dry --backend gloo`: the
launcher, the rendezvous, the sharding arithmetic and the status fold are the real code; only the solver is absent
(it has no CPU path)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(*argv, env=None):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    p = subprocess.run([sys.executable, BENCH, *argv], cwd=ROOT, env=e, capture_output=True, text=True, timeout=240)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    return p, lines


def test_gpus_2_launches_two_ranks_weak():
    p, lines = _run("--gpus", "2", "--dry", "--backend", "gloo", "--steps", "3", "--warmup", "1")
    assert p.returncode == 0, p.stderr[-2000:]
    assert len(lines) == 1, "rank 0 prints exactly one JSON line"
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["config"]["batch_per_gpu"] == 65536 and d["config"]["batch_total"] == 131072
    assert d["metric"] == "qp_solves_per_sec" and d["unit"] == "solves/s" and d["dry"] is True


def test_gpus_3_strong_scaling_is_config5_sharded():
    p, lines = _run("--gpus", "3", "--dry", "--backend", "gloo", "--scaling", "strong", "--steps", "2", "--warmup", "1")
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads(lines[-1])
    assert d["n_gpus"] == 3 and d["scaling"] == "strong" and d["config"]["family"] == "quadrotor"
    assert d["config"]["batch_total"] == 2 ** 20 and d["config"]["batch_per_gpu"] == 349526   # rank 0 of a ragged split


@pytest.mark.gpu
def test_gpus_2_on_two_real_devices():
    """`python bench.py --gpus 2` on hardware: two ranks, one per GPU, RCCL status fold, one JSON line with the whole-job
    rate.  Needs two devices: skipped on the one-GPU test box, runs on the first multi-GPU lease."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip(f"needs 2 GPUs, {torch.cuda.device_count()} present")
    p, lines = _run("--gpus", "2", "--steps", "5", "--warmup", "2", "--no-extras", "--no-cpu-baseline")
    assert p.returncode == 0, p.stderr[-2000:]
    assert len(lines) == 1, "rank 0 prints exactly one JSON line"
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["batch_total"] == 131072
    assert d["value"] > 1.0e8 and not d.get("dry")            # two GPUs' worth of the one-GPU rate (1.8e8 each)


@pytest.mark.gpu
def test_sharded_capi_mode_rehearsed_on_one_device():
    """`bench.py --sharded-capi`: ONE process drives every device through tinympc_create_sharded (the path a Julia host
    uses).  Rehearsed on the one GPU of this box with two shards placed on device 0 (host fold): the same JSON contract,
    plus the per-shard kernel times and the fold backend; tolerance-terminated config 5 style and the fixed headline."""
    p, lines = _run("--sharded-capi", "--gpus", "2", "--devices", "0,0", "--steps", "3", "--warmup", "1", "--batch", "24576")
    assert p.returncode == 0, p.stderr[-2000:]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["mode"] == "sharded_capi" and d["fold_backend"] == "host" and d["scaling"] == "weak"
    assert d["config"]["batch_total"] == 49152 and d["config"]["batch_per_gpu"] == 24576 and d["config"]["devices"] == [0, 0]
    assert d["value"] > 0 and len(d["shard_kernel_ms"]["all"]) == 2 and d["shard_kernel_ms"]["min"] > 0
    assert d["solve_status"] == 1 and d["mean_iters"] == 100
    p, lines = _run("--sharded-capi", "--gpus", "2", "--devices", "0,0", "--scaling", "strong", "--batch", "8192", "--tol", "1e-3",
                    "--steps", "2", "--warmup", "1")
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads(lines[-1])
    assert d["config"]["family"] == "quadrotor" and d["config"]["batch_total"] == 8192 and d["config"]["batch_per_gpu"] == 4096
    assert 10 <= d["mean_iters"] < 100 and d["config"]["kernel"] == "mfma<12,4,30>"
    # a real multi-device run of this mode (RCCL fold) needs >= 2 GPUs: first multi-GPU lease
    import torch
    if torch.cuda.device_count() >= 2:
        p, lines = _run("--sharded-capi", "--gpus", "2", "--steps", "3", "--warmup", "1")
        assert p.returncode == 0, p.stderr[-2000:]
        assert json.loads(lines[-1])["fold_backend"] == "rccl"


def test_sharded_capi_refuses_dry_and_bad_device_lists():
    p, _ = _run("--sharded-capi", "--gpus", "2", "--dry")
    assert p.returncode != 0 and "no dry mode" in (p.stderr + p.stdout)
    p, _ = _run("--sharded-capi", "--gpus", "2", "--devices", "0,0,0")
    assert p.returncode != 0


def test_world_size_mismatch_is_refused():
    # a launcher that started a different number of ranks than --gpus says
    p, _ = _run("--gpus", "2", "--dry", "--backend", "gloo", env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode != 0 and "WORLD_SIZE" in (p.stderr + p.stdout)


def test_no_gpu_is_refused_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    p, lines = _run("--steps", "1", "--warmup", "0")
    assert p.returncode != 0 and not lines and "no CPU fallback" in p.stderr


def test_bench_library_hash_matches_package():
    """bench.py labels replayed counters with the library's source hash; it must be the package's own digest"""
    import importlib.util
    import tinympc_julia_amd as t
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.LIB_HASH == t.source_hash()

"""bench.py --gpus N must start N ranks itself when it is not under torchrun (`--gpus` was parsed and ignored)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_worker_envs_one_rank_per_gpu():
    import bench
    envs = bench.worker_envs(4, 29555, base={})
    assert [e["RANK"] for e in envs] == ["0", "1", "2", "3"]
    assert [e["LOCAL_RANK"] for e in envs] == ["0", "1", "2", "3"]            # LOCAL_RANK -> HIP device
    assert all(e["WORLD_SIZE"] == "4" and e["MASTER_ADDR"] == "127.0.0.1" and e["MASTER_PORT"] == "29555" for e in envs)
    assert all(e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" for e in envs)          # dmabuf IPC for RCCL


def _run(cmd, env=None):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        e.pop(k, None)
    e.update(env or {})
    p = subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    return json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])


def test_launcher_spawns_n_ranks_and_they_rendezvous():
    out = _run([sys.executable, "bench.py", "--gpus", "2", "--launch-check"])
    assert out["n_gpus"] == 2 and out["ranks"] == [0, 1]


def test_under_torchrun_it_is_one_of_the_ranks():
    out = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", "29613", "bench.py", "--gpus", "2", "--launch-check"])
    assert out["n_gpus"] == 2 and out["ranks"] == [0, 1]


def test_rank_count_must_equal_gpus():
    e = dict(os.environ); e.update({"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "4", "--launch-check"], cwd=ROOT, env=e, capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "ranks were launched" in p.stderr


def test_launcher_fails_fast_when_fewer_devices_than_ranks():
    """`bench.py --gpus 8` on a box with fewer than 8 visible devices stops before any rank is started; the
    dry run on one GPU is the explicit ORBHIP_BENCH_SHARED_GPU=1."""
    import subprocess
    env = dict(os.environ); env.pop("ORBHIP_BENCH_SHARED_GPU", None); env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 2, (r.returncode, r.stderr[-500:])
    assert "HIP device(s) visible" in r.stderr and r.stdout.strip() == ""

"""GPU: the N > 1 code path of bench.py on a REAL RCCL communicator before the driver's 8-GPU run meets it: one rank under
torch.distributed.run with the `nccl` backend (= RCCL on ROCm) forced at world size 1 (ORBHIP_BENCH_FORCE_DIST=1), process
group bound to the device, barrier + max-over-ranks all-reduce + the landmark all-gather on device tensors."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_allgather_landmarks_on_rccl_world_size_1():
    import torch
    import torch.distributed as dist
    from ceres_mono_orb_slam2_amd import sharding
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = "29533"
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        assert dist.get_backend() == "nccl"
        pts = torch.from_numpy(np.random.default_rng(0).normal(size=(1234, 3))).to(dev)
        ids = torch.arange(1234, dtype=torch.int64, device=dev) + 7
        allp, alli, counts = sharding.allgather_landmarks(pts, ids)
        torch.cuda.synchronize()
        assert counts == [1234] and torch.equal(allp, pts) and torch.equal(alli, ids) and allp.is_cuda
        assert sharding.max_over_ranks(1.5, device=dev) == 1.5
    finally:
        dist.destroy_process_group()


def test_bench_under_torchrun_uses_rccl():
    env = dict(os.environ, ORBHIP_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", "29537",
           os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--batches-per-step", "2", "--no-cpu", "--no-pipelined"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    j = json.loads(line)
    assert j["n_gpus"] == 1 and j["value"] > 0
    c = j["collective"]
    assert c["backend"] == "nccl (RCCL)" and c["world_size"] == 1 and c["merged_points"] == c["points_per_rank"][0] > 0

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


_CABI_SCRIPT = r"""
import sys, json, numpy as np, torch
sys.path.insert(0, sys.argv[1])
from ceres_mono_orb_slam2_amd import sharding
from ceres_mono_orb_slam2_amd._lib import OrbHipError
dev = torch.device("cuda", 0)
lc = sharding.LandmarkCommunicator(0, 1, 0)
rng = np.random.default_rng(3)
pts = torch.from_numpy(rng.normal(size=(4321, 3))).to(dev); ids = torch.arange(4321, dtype=torch.int64, device=dev) * 3 + 11
out = {}
p, i, c = lc.allgather_landmarks(pts, ids, cap_per_rank=5000)
out["with_ids"] = bool(c == [4321] and torch.equal(p, pts) and torch.equal(i, ids) and p.is_cuda)
p, i, c = lc.allgather_landmarks(pts, None, cap_per_rank=4321)
out["no_ids"] = bool(c == [4321] and torch.equal(p, pts) and i is None)
p, i, c = lc.allgather_landmarks(pts[:0], None, cap_per_rank=16)
out["empty"] = bool(c == [0] and p.shape[0] == 0)
try:
    lc.allgather_landmarks(pts, ids, cap_per_rank=100); out["cap_refused"] = False
except OrbHipError as e:
    out["cap_refused"] = "cap_per_rank" in str(e)
try:
    lc.allgather_landmarks(pts, ids, cap_per_rank=5000, cap_all=1000); out["cap_all_refused"] = False
except OrbHipError as e:
    out["cap_all_refused"] = True
# the torch.distributed twin on the same data (RCCL at world size 1)
import os, torch.distributed as dist
os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = "29541"
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
tp, ti, tc = sharding.allgather_landmarks(pts, ids)
p, i, c = lc.allgather_landmarks(pts, ids, cap_per_rank=4321)
out["equals_torch_distributed"] = bool(torch.equal(tp, p) and torch.equal(ti, i) and tc == c)
dist.destroy_process_group()
lc.close()
print("RESULT " + json.dumps(out))
"""


def test_c_abi_landmark_merge_on_rccl_world_size_1():
    """orbhip_comm_create + orbhip_allgather_landmarks (csrc/orb_comm.hip): the one collective of the hot path from the C ABI - RCCL
    looked up at run time, ONE ncclAllGather of fixed-size slots, a dense rank-ordered output - against sharding.allgather_landmarks on
    the same data; ragged / empty / over-capacity inputs.  (World size 1: the GPU box has one GPU and RCCL refuses two ranks on a device;
    bench.py runs the same entry point at N > 1 beside the torch.distributed merge and records whether they agree.)"""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", _CABI_SCRIPT, ROOT], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1][len("RESULT "):])
    assert all(out.values()), out


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
    assert "rccl" in c and (c["rccl"].get("nranks") == 1 or "error" in c["rccl"] or c["rccl"]["channel_lines"] >= 0), c.get("rccl")
    assert c["c_abi"].get("equals_torch_distributed") is True and c["c_abi"]["points_per_rank"] == c["points_per_rank"], c["c_abi"]


def test_bench_two_ranks_shared_gpu_merges_both_ranks_landmarks(tmp_path):
    """The N = 2 line of bench.py on a one-GPU box (ORBHIP_BENCH_SHARED_GPU=1: both ranks on device 0, gloo carries the collectives):
    two ranks really ran (`collective.world_size` 2), the merged landmark array is the concatenation of BOTH ranks' GlobalBA points in
    rank order, and the N > 1 line still carries `roofline` and `cpu_baseline`.  --ba-quick: shortened BA legs."""
    env = dict(os.environ, ORBHIP_BENCH_SHARED_GPU="1", ORBHIP_BENCH_DUMP_DIR=str(tmp_path), HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batches-per-step", "2", "--batch", "64",
           "--ba-quick", "--cpu-sample", "4", "--cpu-all-seconds", "1"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and j["value"] > 0
    assert j["config"]["frames_per_gpu_per_step"] == 128
    c = j["collective"]
    assert c["world_size"] == 2 and len(c["points_per_rank"]) == 2 and c["merged_points"] == sum(c["points_per_rank"])
    p0 = np.load(tmp_path / "final_points_rank0.npy"); p1 = np.load(tmp_path / "final_points_rank1.npy"); merged = np.load(tmp_path / "merged_points.npy")
    assert len(p0) == c["points_per_rank"][0] and len(p1) == c["points_per_rank"][1]
    assert np.array_equal(merged, np.concatenate([p0, p1]))
    assert not np.array_equal(p0, p1)                                    # every rank solved ITS OWN sub-map
    assert j["roofline"]["frac"] > 0 and "kernel" in j["roofline"] and j["localba"]["roofline"]["cases"]["c5"]["frac"] > 0
    assert j["cpu_baseline"]["value"] > 0 and "N = 2" in j["cpu_baseline"]["note"]
    assert j["localba"].get("quick") is True

"""N > 1 path on CPU: world_size-2 gloo processes exercise frame sharding, the max-over-ranks timing
reduction and the single landmark all-gather."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ceres_mono_orb_slam2_amd import sharding
    lo, hi = sharding.shard_range(11, rank, world)
    t = sharding.max_over_ranks(1.0 + rank)
    rng = np.random.default_rng(rank)
    n_local = 5 + 3 * rank                                     # ragged sub-maps
    pts = torch.from_numpy(rng.normal(size=(n_local, 3)))
    ids = torch.arange(n_local, dtype=torch.int64) + 1000 * rank
    allp, alli, counts = sharding.allgather_landmarks(pts, ids)
    ret[rank] = dict(lo=lo, hi=hi, t=t, pts=allp.numpy(), ids=alli.numpy(), counts=counts, mine=pts.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_and_allgather():
    world, port = 2, 29517
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    r0, r1 = ret[0], ret[1]
    assert (r0["lo"], r0["hi"], r1["lo"], r1["hi"]) == (0, 6, 6, 11)          # contiguous, covers all frames
    assert r0["t"] == r1["t"] == 2.0                                            # MAX over ranks
    assert r0["counts"] == r1["counts"] == [5, 8]
    exp = np.concatenate([r0["mine"], r1["mine"]])
    assert np.array_equal(r0["pts"], exp) and np.array_equal(r1["pts"], exp)    # every rank holds the merged map
    assert np.array_equal(r0["ids"], np.concatenate([np.arange(5), 1000 + np.arange(8)]))


def test_shard_range_covers_everything():
    sys.path.insert(0, ROOT)
    from ceres_mono_orb_slam2_amd import sharding
    for n in (0, 1, 7, 8, 64, 1000):
        for w in (1, 2, 3, 8):
            spans = [sharding.shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1

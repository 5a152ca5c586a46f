"""Multi-GPU sharding of the hot path.

Frames (front-end) and independent BA sub-problems shard embarrassingly: one process per GPU, no
data-path collective.  The only exchange step is the merge of landmark (and pose) updates after a
batched GlobalBA so that every rank holds the whole map: ONE all-gather (RCCL over xGMI on GPUs,
gloo in the CPU tests).  Payload at C5 size is 8 x ~50k points x 24 B = 9.6 MB: latency-bound."""
import torch
import torch.distributed as dist


def shard_range(n_items, rank, world):
    """Contiguous sub-sequence [lo, hi) of rank `rank` (consecutive frames stay co-resident so that the
    frame t / t+1 match never crosses ranks)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def max_over_ranks(seconds, device=None):
    """bench.py contract: the step time is the MAX over ranks."""
    if not (dist.is_available() and dist.is_initialized()):
        return float(seconds)
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def allgather_landmarks(local_pts, local_ids=None):
    """Merge the landmark updates of every rank's sub-map with a single all-gather.
    local_pts: (n_local, 3) float64 tensor (n_local may differ per rank); local_ids: optional (n_local,) int64
    global landmark ids.  Returns (pts_all (sum n, 3), ids_all or None, counts per rank)."""
    if not dist.is_initialized():                    # no process group: a single process holds the whole map already
        return local_pts, local_ids, [local_pts.shape[0]]
    world = dist.get_world_size()                    # (a group of ONE rank still runs the collective: the one-GPU RCCL check)
    dev = local_pts.device
    n = torch.tensor([local_pts.shape[0]], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n)                       # 8-byte size exchange (needed because sub-maps are ragged)
    counts = [int(c.item()) for c in counts]
    cap = max(counts)
    width = 4 if local_ids is not None else 3        # ids ride in the same payload (bit-cast to float64)
    buf = torch.zeros((cap, width), dtype=torch.float64, device=dev)
    buf[:local_pts.shape[0], :3] = local_pts
    if local_ids is not None:
        buf[:local_pts.shape[0], 3] = local_ids.to(torch.int64).view(torch.float64)
    out = torch.empty((world, cap, width), dtype=torch.float64, device=dev)
    dist.all_gather_into_tensor(out.view(-1), buf.view(-1))      # THE collective (ncclAllGather over xGMI)
    pts = torch.cat([out[r, :counts[r], :3] for r in range(world)])
    ids = None
    if local_ids is not None:
        ids = torch.cat([out[r, :counts[r], 3].contiguous().view(torch.int64) for r in range(world)])
    return pts, ids, counts


class LandmarkCommunicator:
    """The C-ABI form of the landmark merge (include/orbslam_hip.h: orbhip_comm_create / orbhip_allgather_landmarks): what a C++
    embedding of the reference calls - RCCL directly, no torch.distributed in the data path.  The 128-byte RCCL id is made by rank 0
    and reaches the other ranks through `exchange` (default: torch.distributed's object broadcast, whatever its backend - the id is
    128 bytes of host data); the collective itself is ONE ncclAllGather of fixed-size slots of `cap_per_rank` landmarks."""

    def __init__(self, rank, world, device, exchange=None):
        import ctypes as C
        import numpy as np
        from . import _lib
        self._lib = _lib; self._C = C; self._L = _lib.load()
        ident = np.zeros(128, np.uint8)
        if rank == 0:
            _lib.check(self._L.orbhip_comm_get_unique_id(_lib.ptr(ident)), "orbhip_comm_get_unique_id")
        if world > 1:
            if exchange is None:
                box = [ident.tobytes()]
                dist.broadcast_object_list(box, src=0)
                ident = np.frombuffer(box[0], np.uint8).copy()
            else:
                ident = np.frombuffer(exchange(ident.tobytes()), np.uint8).copy()
        h = C.c_void_p()
        _lib.check(self._L.orbhip_comm_create(_lib.ptr(ident), world, rank, device, C.byref(h)), "orbhip_comm_create")
        self._h = h; self.rank = rank; self.world = world; self.device = device

    def allgather_landmarks(self, local_pts, local_ids=None, cap_per_rank=None, cap_all=None):
        """local_pts (n, 3) float64 CUDA tensor, local_ids optional (n,) int64.  Returns (pts_all, ids_all or None, counts per rank)."""
        _lib, C = self._lib, self._C
        n = int(local_pts.shape[0])
        cap = int(cap_per_rank if cap_per_rank is not None else max(n, 1))
        cap_all = int(cap_all if cap_all is not None else cap * self.world)
        dev = local_pts.device
        pts = local_pts.contiguous()
        out = torch.empty((cap_all, 3), dtype=torch.float64, device=dev)
        ids = local_ids.contiguous() if local_ids is not None else None
        ids_out = torch.empty((cap_all,), dtype=torch.int64, device=dev) if local_ids is not None else None
        counts = (C.c_int32 * self.world)(); total = C.c_int(0)
        st = torch.cuda.current_stream(dev).cuda_stream
        _lib.check(self._L.orbhip_allgather_landmarks(self._h, _lib.ptr(pts), _lib.ptr(ids) if ids is not None else None, n, cap, _lib.ptr(out),
                                                      _lib.ptr(ids_out) if ids_out is not None else None, cap_all, counts, C.byref(total), C.c_void_p(st)),
                   "orbhip_allgather_landmarks")
        m = total.value
        return out[:m], (ids_out[:m] if ids_out is not None else None), list(counts)

    def close(self):
        if self._h:
            self._L.orbhip_comm_destroy(self._h); self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

"""One process per GPU: the little the host needs from torch.distributed.

The data path of the slab decomposition (halo exchange, batched dot-product all-reduce) runs inside
libthermalporous_hip.so directly on RCCL (csrc/tp_api.hip); torch.distributed is only plumbing:
  * bootstrap: broadcast RCCL's 128-byte unique id from rank 0 (tp_comm_unique_id -> tp_comm_init),
  * the time loop's global saturation guard (thermalmodel.py:195-199 of the reference: comm.reduce MAX
    + bcast of a 1-tuple) as an all-reduce of (min, max),
  * gathering the slab-distributed state on the host when the user asks for it (diagnostics, output).
Backend "nccl" (= RCCL) on GPUs, "gloo" in the CPU tests.
"""
import os

import numpy as np


def world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def local_device():
    """GPU index of this rank: LOCAL_RANK, unless TP_LOCAL_DEVICE pins every rank to one card (rehearsing the
    N-rank path on a one-GPU box)."""
    return int(os.environ.get("TP_LOCAL_DEVICE", os.environ.get("LOCAL_RANK", "0")))


def _dist():
    import torch.distributed as dist
    return dist


def is_initialized():
    try:
        d = _dist()
        return d.is_available() and d.is_initialized()
    except Exception:
        return False


def init(backend=None):
    """Initialise the default process group from the torchrun environment (idempotent)."""
    rank, size = world()
    if size == 1 or is_initialized():
        return rank, size
    import torch
    dist = _dist()
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(local_device())
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(backend=backend, rank=rank, world_size=size)
    return rank, size


def _device():
    import torch
    dist = _dist()
    if dist.get_backend() == "nccl":
        return torch.device("cuda", local_device())
    return torch.device("cpu")


def rccl_bootstrap(make_id):
    """comm_bootstrap callable for HipEngine: rank 0 creates the RCCL unique id, everyone receives it."""
    import torch
    dist = _dist()
    rank = dist.get_rank()
    buf = torch.zeros(128, dtype=torch.uint8)
    if rank == 0:
        buf = torch.frombuffer(bytearray(make_id()), dtype=torch.uint8).clone()
    buf = buf.to(_device())
    dist.broadcast(buf, src=0)
    return bytes(buf.cpu().numpy().tobytes())


def allreduce_minmax(lo, hi):
    if not is_initialized():
        return lo, hi
    import torch
    dist = _dist()
    t = torch.tensor([-lo, hi], dtype=torch.float64, device=_device())
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    t = t.cpu()
    return -float(t[0]), float(t[1])


def allgather_slabs(local, counts):
    """Concatenate per-rank arrays (fields, n2_local, n1, n0) along the slab axis on every rank."""
    if not is_initialized():
        return local
    import torch
    dist = _dist()
    dev = _device()
    nmax = max(counts)
    f, _, n1, n0 = local.shape
    pad = np.zeros((f, nmax, n1, n0))
    pad[:, :local.shape[1]] = local
    mine = torch.from_numpy(pad).to(dev)
    outs = [torch.empty_like(mine) for _ in counts]
    dist.all_gather(outs, mine)
    return np.concatenate([o.cpu().numpy()[:, :c] for o, c in zip(outs, counts)], axis=1)


def barrier():
    if is_initialized():
        _dist().barrier()

"""world_size-2 tests of the N>1 host path on CPU (gloo): slab partition, halo slicing, ownership of
wells, the torch.distributed plumbing (bootstrap broadcast, min/max all-reduce, slab gather) and the
distributed algorithm itself -- every rank applies its slab of the operator after a halo exchange and
reduces dot products, exactly the pattern csrc/tp_api.hip:halo_exchange / allreduce_sum implement
with RCCL -- checked against the single-process oracle."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _spawn(fn, world=2):
    port = _free_port()
    mp.spawn(_entry, args=(world, port, fn.__name__), nprocs=world, join=True)


def _entry(rank, world, port, fname):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from thermalporous_amd import parallel
    parallel.init("gloo")
    try:
        globals()[fname](rank, world)
    finally:
        dist.destroy_process_group()


# ---- workers ------------------------------------------------------------------------------------------
def _w_plumbing(rank, world):
    from thermalporous_amd import parallel
    ident = parallel.rccl_bootstrap(lambda: bytes(range(128)))
    assert ident == bytes(range(128))
    lo, hi = parallel.allreduce_minmax(0.1*(rank + 1), 0.5 + rank)
    assert np.isclose(lo, 0.1) and np.isclose(hi, 0.5 + world - 1)
    counts = [3, 2]
    local = np.full((2, counts[rank], 2, 2), float(rank))
    full = parallel.allgather_slabs(local, counts)
    assert full.shape == (2, 5, 2, 2) and (full[:, :3] == 0).all() and (full[:, 3:] == 1).all()


def _halo_exchange(x, rank, world):
    """x: (nf, n2+2, n1, n0) slab with halo planes; same pattern as csrc/tp_api.hip:halo_exchange."""
    reqs = []
    bufs = {}
    for nb, send_plane, recv_plane in ((rank - 1, 1, 0), (rank + 1, -2, -1)):
        if 0 <= nb < world:
            s = torch.from_numpy(np.ascontiguousarray(x[:, send_plane]))
            r = torch.empty_like(s)
            bufs[recv_plane] = r
            reqs += [dist.isend(s, nb), dist.irecv(r, nb)]
    for q in reqs:
        q.wait()
    for plane, r in bufs.items():
        x[:, plane] = r.numpy()


def _w_distributed_operator(rank, world):
    import cases
    import oracle.linalg as la
    from oracle.engine import OracleEngine
    from thermalporous_amd.engine import slab_range
    spec, u0, *_ = cases.c4_spe10_3d(6, 10, 5)
    o = OracleEngine(spec, dict(pc="cptr"))
    u = cases.perturbed_state(spec, seed=1, amp=0.3)
    o.set_old(u0)
    o.set_dt(500.0)
    o.set_state(u)
    J = o.jacobian()                                   # every rank builds the global operator (test only)
    x = np.random.default_rng(5).standard_normal(J.shape[1:2] + J.shape[3:])
    y_ref = la.spmv_block(J, x)
    n2 = J.shape[3]
    lo, hi = slab_range(n2, rank, world)
    # my slab of x with halo planes, halos filled by the exchange only
    xs = np.zeros((x.shape[0], hi - lo + 2) + x.shape[2:])
    xs[:, 1:-1] = x[:, lo:hi]
    _halo_exchange(xs, rank, world)
    if lo > 0:
        assert np.array_equal(xs[:, 0], x[:, lo - 1])
    if hi < n2:
        assert np.array_equal(xs[:, -1], x[:, hi])
    # local rows of J applied to the haloed slab (the +-a2 slots reach into the halo planes)
    idx = np.clip(np.arange(lo - 1, hi + 1), 0, n2 - 1)
    Js = J[:, :, :, idx]
    ys = la.spmv_block(Js, xs)[:, 1:-1]
    assert np.allclose(ys, y_ref[:, lo:hi], rtol=1e-13, atol=1e-13*np.abs(y_ref).max())
    # dot products: local partial + all-reduce == global
    t = torch.tensor([float(np.vdot(ys, ys))], dtype=torch.float64)
    dist.all_reduce(t)
    assert np.isclose(t.item(), float(np.vdot(y_ref, y_ref)), rtol=1e-12)


def _w_host_slicing(rank, world):
    """HipEngine's slab bookkeeping (no GPU call): ranges, halo slicing, ownership of source entries."""
    import cases
    from thermalporous_amd import engine as E
    spec, u0, *_ = cases.c4_spe10_3d(6, 11, 5)
    gn2 = spec["n"][2]
    ranges = [E.slab_range(gn2, r, world) for r in range(world)]
    assert ranges[0][0] == 0 and ranges[-1][1] == gn2 and all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
    h = E.HipEngine.__new__(E.HipEngine)               # bookkeeping only, no library / GPU
    h.b, h.rank, h.nranks = 3, rank, world
    h.gn = tuple(spec["n"])
    h.lo, h.hi = ranges[rank]
    h.n = (h.gn[0], h.gn[1], h.hi - h.lo)
    h.np_ = h.gn[0]*h.gn[1]
    h.ntot = h.np_*(h.n[2] + 2)
    a = h._with_halo(spec["phi"]).reshape(h.n[2] + 2, h.gn[1], h.gn[0])
    assert np.array_equal(a[1:-1], spec["phi"][h.lo:h.hi])
    assert np.array_equal(a[0], spec["phi"][max(h.lo - 1, 0)]) and np.array_equal(a[-1], spec["phi"][min(h.hi, gn2 - 1)])
    cells = np.asarray(spec["sources"]["cell"])
    mine = (cells//h.np_ >= h.lo) & (cells//h.np_ < h.hi)
    owned = torch.tensor([int(mine.sum())])
    dist.all_reduce(owned)
    assert owned.item() == len(cells)                  # every entry has exactly one owner


# ---- tests ----------------------------------------------------------------------------------------------
def test_gloo_plumbing():
    _spawn(_w_plumbing)


def test_gloo_distributed_operator_matches_global():
    _spawn(_w_distributed_operator)


def test_gloo_host_slicing():
    _spawn(_w_host_slicing)

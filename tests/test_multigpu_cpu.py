"""CPU, world_size 2 on gloo: the N>1 plumbing of bench.py / tensor_stream.parallel --
stream sharding and the one-off coefficient broadcast (the path's only collective)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class FakeVPP:
    """Stands in for the HIP context: holds a coefficient block (no GPU in this test)."""

    def __init__(self, c):
        self.c = list(c)

    def get_coeffs(self):
        return list(self.c)

    def set_coeffs(self, v):
        self.c = list(v)


def _worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, "tensor-stream_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tensor_stream import parallel
    from tensor_stream.vpp import default_coeffs
    ref = default_coeffs()
    # rank 1 starts with garbage: after the broadcast it must hold rank 0's (== default) block
    vpp = FakeVPP(ref if rank == 0 else [0.0] * 8)
    got = parallel.broadcast_coeffs(vpp, dist, device=torch.device("cpu"))
    ok = vpp.get_coeffs() == ref and got == ref
    # a rank-0 block that differs from the literals must be refused everywhere
    bad = FakeVPP([1.0] * 8)
    refused = False
    try:
        parallel.broadcast_coeffs(bad, dist, device=torch.device("cpu"))
    except RuntimeError:
        refused = True
    mine = parallel.shard_streams(64, rank, world)
    allc = [None] * world
    dist.all_gather_object(allc, mine)
    q.put((rank, ok, refused, mine, sorted(sum(allc, []))))
    dist.barrier()
    dist.destroy_process_group()


def test_world2_gloo_broadcast_and_sharding():
    from tensor_stream import _native
    if not os.path.exists(_native.LIB_PATH):
        pytest.skip("libtsvpp.so not built")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, refused, mine, union in res:
        assert ok and refused
        assert mine == list(range(rank, 64, 2))      # stream s -> rank s % world
        assert union == list(range(64))              # every stream served exactly once


def test_shard_streams_single_and_eight():
    sys.path.insert(0, os.path.join(ROOT, "tensor-stream_amd"))
    from tensor_stream import parallel
    assert parallel.shard_streams(8, 0, 1) == list(range(8))
    seen = sorted(s for r in range(8) for s in parallel.shard_streams(64, r, 8))
    assert seen == list(range(64))
    assert all(len(parallel.shard_streams(64, r, 8)) == 8 for r in range(8))   # C5: 8 consumers per GPU
    with pytest.raises(ValueError):
        parallel.shard_streams(8, 2, 2)

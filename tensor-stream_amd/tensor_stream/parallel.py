"""Multi-GPU plumbing for the VPP path: one process per GPU, frames shard by stream, NO data-path
collective (every frame is independent).  The reference's own multi-GPU model is "one TensorStream
instance per stream, each with its own cudaDevice" (reference src/Wrappers/WrapperPython.cpp:18-29,
README.md:193-196); this module is that, plus the one collective north_star asks for: a one-off
broadcast of the colour coefficient block from rank 0 (RCCL over xGMI when the backend is "nccl").

Works with any torch.distributed backend -- the CPU tests run it on gloo with world_size 2.
"""
import torch

_LAST = {"collective": None}  # how the last coefficient broadcast travelled (bench.py reports it; the GPU tests assert on it)


def last_broadcast_info():
    return dict(_LAST)


def shard_streams(n_streams, rank, world):
    """Stream s is served by rank s % world.  Returns this rank's stream ids."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    return [s for s in range(n_streams) if s % world == rank]


def broadcast_coeff_block(block, dist, device=None):
    """Broadcast rank 0's 8-float block; returns the received list.  `dist` is torch.distributed
    (initialised) or None for a single process."""
    if dist is None or not dist.is_initialized():
        _LAST.update(collective=None, backend=None, device=None, world=1)
        return [float(x) for x in block]
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    t = torch.tensor(list(block), dtype=torch.float32, device=device)
    if dist.get_rank() != 0:
        t.zero_()  # prove the values really travel
    dist.broadcast(t, src=0)
    _LAST.update(collective="torch.distributed.broadcast", backend=str(dist.get_backend()), device=str(t.device), world=int(dist.get_world_size()))
    return [float(x) for x in t.cpu().tolist()]


def broadcast_coeffs(vpp, dist, device=None):
    """Rank 0's coefficients -> every rank's context.  Must not change results: each receiver checks
    the block bit-for-bit against its compiled-in literals (reference src/ColorConversion.cu:23-35
    has them as literals) and refuses a job whose ranks disagree."""
    from .vpp import default_coeffs
    mine = vpp.get_coeffs()
    got = broadcast_coeff_block(mine, dist, device)
    ref = default_coeffs()
    t_got = torch.tensor(got, dtype=torch.float32)
    t_ref = torch.tensor(ref, dtype=torch.float32)
    if not torch.equal(t_got.view(torch.int32), t_ref.view(torch.int32)):
        raise RuntimeError(f"colour coefficient block from rank 0 differs from the built-in constants: {got} vs {ref}")
    vpp.set_coeffs(got)
    return got

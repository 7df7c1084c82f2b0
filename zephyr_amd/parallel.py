"""Rank-level sharding of the embarrassingly parallel axes (frequencies, source batches).

One process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in
CPU tests).  Forward modelling needs no data-path collective: each rank solves its own
frequencies; the small receiver data / the N-element gradient are combined with ONE all-reduce,
the counterpart of the reference's `reduce(np.add, ...)` over frequencies
(zephyr/middleware/problem.py:152,162).
"""
import numpy as np


def _dist():
    try:
        import torch.distributed as dist
    except Exception:          # pragma: no cover
        return None
    return dist if (dist.is_available() and dist.is_initialized()) else None


def rank_and_size():
    d = _dist()
    if d is None:
        return 0, 1
    return d.get_rank(), d.get_world_size()


def owned_indices(n, rank=None, size=None):
    """Round-robin ownership of n work items: rank r owns r, r+size, ...  (frequency-major work
    items keep a GPU on the coefficients it has already assembled)."""
    if rank is None or size is None:
        rank, size = rank_and_size()
    return list(range(rank, n, size))


def allreduce_sum(arr):
    """Sum a numpy array (real or complex) over all ranks; returns a numpy array on every rank."""
    d = _dist()
    arr = np.ascontiguousarray(arr)
    if d is None or d.get_world_size() == 1:
        return arr
    import torch
    is_complex = np.iscomplexobj(arr)
    flat = arr.view(np.float64) if is_complex else arr.astype(np.float64, copy=False)
    t = torch.from_numpy(np.array(flat, copy=True))
    if d.get_backend() == 'nccl':
        # the rank's own GPU (LOCAL_RANK / HELM_DEVICE, the one its operator handles live on), not torch's current device
        from .discretization import default_device
        t = t.cuda(default_device())
    d.all_reduce(t, op=d.ReduceOp.SUM)
    out = t.cpu().numpy()
    return out.view(np.complex128).reshape(arr.shape) if is_complex else out.reshape(arr.shape)


def allreduce_sum_device(t):
    """In-place sum of a CUDA torch tensor over ranks (complex tensors are reduced as float pairs)."""
    d = _dist()
    if d is None or d.get_world_size() == 1:
        return t
    import torch
    v = torch.view_as_real(t) if t.is_complex() else t
    d.all_reduce(v, op=d.ReduceOp.SUM)
    return t

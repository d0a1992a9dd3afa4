"""Multi-GPU helpers: the env path shards trivially.

One process per GPU (torch.distributed, backend "nccl" = RCCL on ROCm, "gloo" on CPU for tests).
Every cube / walk / leaf is independent, so ranks own contiguous ranges and rank-distinct RNG
streams and NEVER exchange cube data: there is no collective on the env path.  The only
communication is reporting (a barrier and a MAX / SUM of scalars), done here.
"""
from __future__ import annotations

import os

# RCCL and tensor sharing between processes need dmabuf IPC on this driver stack.  The variable is read when the HIP
# runtime initialises, so it is set at IMPORT time (import this module before the first GPU call of the process; the
# launcher normally exports it already) -- setting it inside init() would come too late in a process that already
# touched the GPU.
_IPC_WAS_SET = os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY") == "0"    # before the setdefault below: what the HIP runtime may already have read
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

# a process that initialised the GPU before importing this module and without the variable exported started its HIP runtime in
# legacy IPC mode: setting the variable now does not change that
_TOO_LATE = (not _IPC_WAS_SET) and torch.cuda.is_initialized()


def world():
    """(rank, world_size, local_rank) from the launcher's environment (torch.distributed.run)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def shard(n_total: int, rank: int, world_size: int):
    """Contiguous range [lo, hi) of the n_total cubes / walks owned by `rank`; sizes differ by <= 1."""
    base, extra = divmod(n_total, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def rng_stream(rank: int, base_stream: int = 0) -> int:
    """stream_id for rc_scramble / rc_adi_generate: distinct per rank, reproducible for a fixed world size."""
    return base_stream * 65536 + rank


def init(backend=None, device=None):
    """Initialise the default process group when launched with WORLD_SIZE > 1 (reporting only)."""
    rank, ws, local = world()
    if ws > 1 and not dist.is_initialized():
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            # the import-time flag catches "set too late"; the live value catches an explicitly exported value other than 0 (which the
            # setdefault above leaves alone) -- either way RCCL would start in legacy IPC mode and fail in hipIpcGetMemHandle
            if _TOO_LATE or os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY") != "0":
                raise RuntimeError("HSA_ENABLE_IPC_MODE_LEGACY=0 must be exported (or this module imported) before the process first "
                                   f"touches the GPU: RCCL needs dmabuf IPC here (now: {os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY')!r})")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        kw = {}
        if backend == "nccl" and device is not None:
            kw["device_id"] = device
        dist.init_process_group(backend, **kw)
    return rank, ws, local


def barrier():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def reduce_scalars(values, op="max", device=None):
    """MAX or SUM of a list of Python floats over all ranks (identity when not distributed)."""
    if not (dist.is_available() and dist.is_initialized()):
        return [float(v) for v in values]
    dev = device if device is not None else ("cuda" if dist.get_backend() == "nccl" else "cpu")
    t = torch.tensor(values, dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX if op == "max" else dist.ReduceOp.SUM)
    return [float(x) for x in t.cpu()]

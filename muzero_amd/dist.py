"""Multi-GPU layer: one process per GPU, games sharded across ranks, RCCL used
only to broadcast network weights.

Games are independent (own tree, own min/max statistics, own frame ring), so the
search needs no collective: rank r of R owns games [r*B/R, (r+1)*B/R).  The one
exchange on the path is the weight hand-over from the rank that holds fresh
parameters to the others after a weight update - a single flat broadcast of the
~62 MB float32 state (the counterpart of `self.network` being shared by every
game in the reference's single process, train.py:155).
"""
import hashlib
import os
from collections import OrderedDict
from typing import Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: str = None) -> Tuple[int, int, int]:
    """(rank, world_size, local_rank); initialises torch.distributed when WORLD_SIZE > 1.
    backend defaults to RCCL ("nccl") when a GPU is visible, else gloo."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("MZ_FORCE_DEVICE") is not None:   # several ranks rehearsing on one GPU (with gloo)
        local = int(os.environ["MZ_FORCE_DEVICE"])
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # MZ_DIST_BACKEND=gloo lets several ranks rehearse on one GPU (RCCL wants one GPU per rank)
            backend = os.environ.get("MZ_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced [lo, hi) slice of `total` games for `rank`."""
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def state_checksum(sd) -> str:
    h = hashlib.sha256()
    for k, v in sd.items():
        h.update(k.encode())
        h.update(v.detach().cpu().contiguous().numpy().tobytes())
    return h.hexdigest()


class WeightComm:
    """The weight broadcast through this library's own RCCL binding (``mz_comm_*`` in the C ABI)
    instead of torch.distributed's.  The 128-byte RCCL id travels over the existing process group
    (any backend) once; afterwards ``broadcast`` is one in-place RCCL broadcast of a flat device
    buffer on the current stream."""

    def __init__(self):
        import ctypes as C
        from . import _lib
        rank = dist.get_rank() if dist.is_initialized() else 0
        world = dist.get_world_size() if dist.is_initialized() else 1
        ident = (C.c_char * 128)()
        if rank == 0:
            _lib.call("mz_comm_unique_id", ident)
        box = [bytes(ident)]
        if world > 1:
            dist.broadcast_object_list(box, src=0)
        handle = C.c_void_p()
        _lib.call("mz_comm_init", box[0], rank, world, C.byref(handle))
        self._lib, self.handle, self.rank, self.world = _lib, handle, rank, world

    def broadcast(self, flat: torch.Tensor, src: int = 0) -> None:
        if not flat.is_cuda or not flat.is_contiguous():
            raise ValueError("WeightComm.broadcast takes a contiguous device tensor")
        self._lib.call("mz_comm_bcast_weights", self.handle, self._lib.ptr(flat), flat.numel() * flat.element_size(), src,
                       self._lib.current_stream())

    def close(self) -> None:
        if self.handle:
            self._lib.call("mz_comm_destroy", self.handle)
            self.handle = None


_weight_comm = None


def _flat_broadcast(flat: torch.Tensor, src: int) -> None:
    """One broadcast of a flat buffer: torch.distributed's (RCCL for device tensors under the "nccl" backend), or
    this library's own RCCL binding when MZ_COMM=rccl."""
    global _weight_comm
    if flat.is_cuda and os.environ.get("MZ_COMM") == "rccl":
        if _weight_comm is None:
            _weight_comm = WeightComm()
        _weight_comm.broadcast(flat, src)
    else:
        dist.broadcast(flat, src=src)


def broadcast_state_dict(sd, src: int = 0, device=None):
    """Broadcast a state_dict from `src` as ONE flat float32 buffer (plus the integer
    entries as a second small one).  Every rank must pass a state_dict with the same
    keys and shapes (contents are ignored off-source).  Returns the received dict: under RCCL its
    tensors are views into the flat DEVICE buffer (``MuZeroNetwork.load_state_dict`` folds and packs
    them there; nothing visits the host), under gloo host tensors.  A no-op without an initialised
    process group."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return sd
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    fkeys = [k for k, v in sd.items() if v.is_floating_point()]
    ikeys = [k for k, v in sd.items() if not v.is_floating_point()]
    flat = torch.cat([sd[k].detach().reshape(-1).float() for k in fkeys]).to(device)
    _flat_broadcast(flat, src)
    out = OrderedDict()
    off = 0
    for k in fkeys:
        n = sd[k].numel()
        out[k] = flat[off:off + n].view(sd[k].shape).to(sd[k].dtype)      # float32 stays a view
        off += n
    if ikeys:
        iflat = torch.cat([sd[k].detach().reshape(-1).long() for k in ikeys]).to(device)
        _flat_broadcast(iflat, src)
        off = 0
        for k in ikeys:
            n = sd[k].numel()
            out[k] = iflat[off:off + n].view(sd[k].shape).to(sd[k].dtype)
            off += n
    return OrderedDict((k, out[k]) for k in sd)


def gather_checksums(sd) -> list:
    """SHA-256 of the state_dict on every rank, in rank order (one all-gather of the digests)."""
    digest = state_checksum(sd)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [digest]
    gathered = [None] * dist.get_world_size()
    dist.all_gather_object(gathered, digest)
    return gathered


def verify_same_everywhere(sd) -> bool:
    """True when every rank holds bit-identical parameters (checksum all-gather)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return True
    gathered = gather_checksums(sd)
    return all(g == gathered[0] for g in gathered)


def max_over_ranks(value: float) -> float:
    """MAX all-reduce of a host scalar (on the GPU for RCCL, on the CPU for gloo)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return value
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    t = torch.tensor([value], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def average_gradients(parameters) -> None:
    """Data-parallel training: ONE all-reduce of all gradients as a flat float32 buffer
    (62 MB for the reference architecture), then the mean is written back in place.  A no-op
    without an initialised process group.  Parameters without a gradient count as zeros."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    params = [p for p in parameters]
    if not params:
        return
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).detach().reshape(-1).float()
                      for p in params]).to(dev)
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat /= dist.get_world_size()
    off = 0
    for p in params:
        n = p.numel()
        g = flat[off:off + n].view_as(p).to(p.device, p.dtype)
        if p.grad is None:
            p.grad = g.clone()
        else:
            p.grad.copy_(g)
        off += n

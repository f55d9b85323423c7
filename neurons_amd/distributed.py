"""Multi-GPU plumbing for the denoising path: one process per GPU, clips sharded across ranks, ONE weight
broadcast at start-up, no collective inside the denoising loop.

Reference behaviour: ``accelerate`` shards a ``DataLoader(batch_size=1)`` over processes
(scripts/neuroclips_video.py:206,238) and maps a local index back with
``org_idx = machine_id + local_index * interval`` (scripts/neuroclips_video.py:39-40,323): rank-strided clips.
"""
from typing import Dict, List

import torch


def clip_indices_for_rank(num_clips: int, rank: int, world_size: int) -> List[int]:
    """Global clip indices handled by ``rank``: rank, rank + world, rank + 2*world, ... (reference :39-40)."""
    if not (0 <= rank < world_size):
        raise ValueError(f"rank {rank} outside world of {world_size}")
    return list(range(rank, num_clips, world_size))


def global_clip_index(rank: int, local_index: int, world_size: int) -> int:
    """``get_index``-equivalent of scripts/neuroclips_video.py:39-40."""
    return rank + local_index * world_size


def broadcast_state_dict(schema: Dict[str, tuple], state_dict, src: int = 0, device=None) -> Dict[str, torch.Tensor]:
    """Rank ``src`` holds ``state_dict``; everyone returns an identical fp32 copy.  The whole network travels as
    ONE flat buffer = one RCCL broadcast over xGMI (backend "nccl" on ROCm) or one gloo broadcast on CPU.
    ``schema`` (name -> shape, same on every rank) fixes the packing order."""
    import torch.distributed as dist
    rank = dist.get_rank() if dist.is_initialized() else 0
    total = sum(int(torch.Size(s).numel()) for s in schema.values())
    dev = device if device is not None else torch.device("cpu")
    flat = torch.empty(total, dtype=torch.float32, device=dev)
    if rank == src:
        off = 0
        for k, s in schema.items():
            t = state_dict[k]
            if tuple(t.shape) != tuple(s):
                raise ValueError(f"{k}: shape {tuple(t.shape)} != schema {tuple(s)}")
            n = t.numel()
            flat[off:off + n] = t.reshape(-1).to(device=dev, dtype=torch.float32)
            off += n
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(flat, src=src)
    flat = flat.cpu()
    out, off = {}, 0
    for k, s in schema.items():
        n = int(torch.Size(s).numel())
        out[k] = flat[off:off + n].view(s)
        off += n
    return out


def _collectives_active() -> bool:
    """True when the collectives below should run: an initialised process group of more than one rank, or of ONE rank when
    NR_DIST_FORCE=1 (tests/test_dist_gpu.py: the RCCL code path of a single-GPU box: init, object broadcast, 2.5 GB device
    broadcast, MAX all-reduce, barrier)."""
    import os
    import torch.distributed as dist
    return dist.is_initialized() and (dist.get_world_size() > 1 or os.environ.get("NR_DIST_FORCE") == "1")


def broadcast_native_weights(net, src: int = 0):
    """rank ``src`` has loaded + planned ``net`` (its converted bf16 weights exist on its GPU); every other rank holds a
    FRESH network of the same config.  The manifest (a few hundred KB of text) goes through ``broadcast_object_list``, the packed
    arena (2.55 GB U-Net / 0.99 GB SparseCtrl) through ONE device-to-device broadcast (RCCL over xGMI); receivers import it without
    ever holding fp32 host weights or re-running the conversion.  Shape constraint: the arena holds exactly the converted weights of the
    plan(s) rank ``src`` made before exporting, so ``src`` must have planned the shape every rank will run (bench.py does); a receiver
    asked for another shape raises NR_ERR_STATE naming the missing conversion (it has no host weights to convert from)."""
    import torch.distributed as dist
    if not _collectives_active():
        return net
    rank = dist.get_rank()
    if rank == src:
        manifest, arena = net.export_weights()
        meta = [manifest, int(arena.numel())]
    else:
        meta = [None, None]
    dist.broadcast_object_list(meta, src=src)
    manifest, nbytes = meta
    if rank != src:
        arena = torch.empty(nbytes, dtype=torch.uint8, device=net.device)
    dist.broadcast(arena, src=src)
    if rank != src:
        net.import_weights(manifest, arena)
    return net


def max_over_ranks(value: float, device=None) -> float:
    """MAX all-reduce of a timing scalar (bench.py's elapsed time)."""
    import torch.distributed as dist
    if not _collectives_active():
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())

"""Pair sharding across the GPUs of one node and the final pose gather.

Pairs are independent (the reference never chains estimates: batch_processor.py:82-92
takes R_prev from ground truth), so each rank processes a contiguous block with no
data-path collective; the only exchange is one all-gather of fixed-size 128-byte pose
records at the end (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU tests).
"""
import numpy as np

RECORD_DTYPE = np.dtype([("R", "<f8", (9,)), ("t", "<f8", (3,)), ("inliers", "<i4"), ("status", "<i4"),
                         ("n_matches", "<i4"), ("pair", "<i4"), ("_pad", "<u1", (16,))])
assert RECORD_DTYPE.itemsize == 128


def shard_bounds(total, rank, world):
    """Contiguous block [lo, hi) of ceil(total/world) pairs for this rank."""
    per = -(-total // world)
    lo = min(rank * per, total)
    return lo, min(lo + per, total)


def pack_records(R, t, inliers, status, n_matches, first_pair=0):
    n = len(inliers)
    rec = np.zeros(n, RECORD_DTYPE)
    rec["R"] = np.asarray(R, np.float64).reshape(n, 9)
    rec["t"] = np.asarray(t, np.float64).reshape(n, 3)
    rec["inliers"] = inliers; rec["status"] = status; rec["n_matches"] = n_matches
    rec["pair"] = first_pair + np.arange(n)
    return rec


def gather_pose_records(local, per_rank, device=None):
    """All-gather `local` (<= per_rank records, padded) from every rank; returns the
    concatenated records of all ranks on every rank.  Requires an initialised
    torch.distributed process group; with world size 1 it is the identity."""
    import torch
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return local.copy()
    world = dist.get_world_size()
    buf = np.zeros(per_rank, RECORD_DTYPE)
    buf["pair"] = -1
    buf[:len(local)] = local
    src = torch.from_numpy(buf.view(np.uint8).reshape(-1).copy())
    if device is not None:
        src = src.to(device)
    out = torch.empty(world * src.numel(), dtype=torch.uint8, device=src.device)
    dist.all_gather_into_tensor(out, src)
    rec = out.cpu().numpy().view(RECORD_DTYPE)
    return rec[rec["pair"] >= 0].copy()

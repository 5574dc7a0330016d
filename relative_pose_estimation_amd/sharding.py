"""Sharding across the GPUs of one node and the final pose gather.

Pairs are independent (the reference never chains estimates: batch_processor.py:82-92 takes R_prev from
ground truth), so every rank processes a contiguous block with no data-path collective.  The only exchange
is one all-gather of fixed-size 128-byte pose records at the end of a step:

  * `PoseComm` -- the product path: `rpe_gather_poses` in librpe_amd.so, an ncclAllGather over RCCL / xGMI on
    the engine's HIP stream, records packed on the device.  No torch: ranks bootstrap through the RCCL unique
    id that rank 0 writes to a file (one node, so a local path is visible to every rank).
  * `gather_pose_records` -- the same exchange through an already initialised torch.distributed group; used
    by the CPU tests (backend "gloo", world size 2) and by `bench.py --dist-backend gloo` rehearsals.

Stream mode (BASELINE configs[4]; caller loop batch_processor.py:71-109): a sequence of F frames gives F-1
consecutive pairs; `shard_stream_bounds` cuts it into contiguous frame ranges with a ONE-FRAME HALO, so each rank
extracts features once per frame it owns plus one shared frame, and pair p = (frame p, frame p+1) is computed
by exactly one rank.
"""
import ctypes as C
import os
import time

import numpy as np

RECORD_DTYPE = np.dtype([("R", "<f8", (9,)), ("t", "<f8", (3,)), ("inliers", "<i4"), ("status", "<i4"),
                         ("n_matches", "<i4"), ("pair", "<i4"), ("_pad", "<u1", (16,))])
assert RECORD_DTYPE.itemsize == 128


def shard_bounds(total, rank, world):
    """Contiguous block [lo, hi) of ceil(total/world) pairs for this rank."""
    per = -(-total // world)
    lo = min(rank * per, total)
    return lo, min(lo + per, total)


def shard_stream_bounds(n_frames, rank, world):
    """Stream of n_frames consecutive frames -> n_frames-1 pairs (p, p+1), split over `world` ranks.
    Returns (frame_lo, frame_hi, pair_lo, pair_hi): this rank loads frames [frame_lo, frame_hi) -- its pairs'
    first frames plus ONE halo frame -- and produces pairs [pair_lo, pair_hi).  Ranks without pairs get
    (x, x, x, x).  Every pair belongs to exactly one rank; interior boundary frames are extracted twice in
    total (once per neighbour) instead of once per pair as in the reference (batch_processor.py:79,92)."""
    pairs = max(n_frames - 1, 0)
    lo, hi = shard_bounds(pairs, rank, world)
    if hi <= lo:
        return lo, lo, lo, lo
    return lo, hi + 1, lo, hi


def pack_records(R, t, inliers, status, n_matches, first_pair=0):
    n = len(inliers)
    rec = np.zeros(n, RECORD_DTYPE)
    rec["R"] = np.asarray(R, np.float64).reshape(n, 9)
    rec["t"] = np.asarray(t, np.float64).reshape(n, 3)
    rec["inliers"] = inliers; rec["status"] = status; rec["n_matches"] = n_matches
    rec["pair"] = first_pair + np.arange(n)
    return rec


# ----------------------------------------------------------------------------- native RCCL path (no torch)
def _id_path(tag=None):
    # MASTER_PORT + run id + the launcher's pid (every rank is a child of the same torch.distributed.run agent): a file
    # left behind by a crashed earlier launch on the same port can never be mistaken for this launch's id
    tag = "{}_{}_{}_{}".format(os.environ.get("MASTER_PORT", "0"), os.environ.get("TORCHELASTIC_RUN_ID", "none"), os.getppid(), tag or "")
    return os.path.join(os.environ.get("RPE_COMM_DIR", "/tmp"), f"rpe_comm_{tag}.id")


def exchange_unique_id(lib, rank, world, tag=None, timeout=300.0):
    """Rank 0 asks RCCL for a unique id and publishes it atomically (write + rename); the others poll for the file.
    The launcher contract (python -m torch.distributed.run, one node) provides RANK / WORLD_SIZE / MASTER_PORT in the
    environment; nothing else of torch is used."""
    path = _id_path(tag)
    if rank == 0:
        buf = (C.c_uint8 * 128)()
        rc = lib.rpe_comm_unique_id(buf)
        if rc != 0:
            raise RuntimeError(f"rpe_comm_unique_id failed ({rc}): {lib.rpe_comm_last_error().decode()}")
        tmp = path + f".tmp{os.getpid()}"
        with open(tmp, "wb") as fh:
            fh.write(bytes(buf))
        os.replace(tmp, path)
        return bytes(buf)
    t0 = time.time()
    while True:
        try:
            with open(path, "rb") as fh:
                data = fh.read()
            if len(data) == 128:
                return data
        except FileNotFoundError:
            pass
        if time.time() - t0 > timeout:
            raise TimeoutError(f"rank {rank}: no RCCL unique id at {path} after {timeout} s")
        time.sleep(0.05)


class PoseComm:
    """RCCL communicator bound to one Engine (one GPU): gather of pose records, barrier, max-reduce of a scalar."""

    def __init__(self, engine, rank, world, tag=None):
        self.eng, self.rank, self.world = engine, rank, world
        lib = engine.lib
        self._path = _id_path(tag)
        uid = exchange_unique_id(lib, rank, world, tag)
        idbuf = (C.c_uint8 * 128).from_buffer_copy(uid)
        c = C.c_void_p()
        rc = lib.rpe_comm_create(engine.h, rank, world, idbuf, C.byref(c))
        if rc != 0:
            raise RuntimeError(f"rpe_comm_create failed ({rc}): {lib.rpe_comm_last_error().decode()}")
        self.c = c
        self.barrier()                                   # every rank has read the id: rank 0 may remove the file
        if rank == 0:
            try:
                os.remove(self._path)
            except OSError:
                pass

    def _chk(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} failed ({rc}): {self.eng.lib.rpe_comm_last_error().decode()}")

    def gather(self, n_local, per_rank, first_pair):
        """All ranks' records of the engines' last batch (padding removed), on every rank."""
        rec = np.zeros(self.world * per_rank, RECORD_DTYPE)
        self._chk(self.eng.lib.rpe_gather_poses(self.eng.h, self.c, n_local, per_rank, first_pair,
                                                rec.ctypes.data_as(C.c_void_p)), "rpe_gather_poses")
        return rec[rec["pair"] >= 0]

    def barrier(self):
        self._chk(self.eng.lib.rpe_comm_barrier(self.c), "rpe_comm_barrier")

    def max(self, value):
        v = C.c_double(float(value))
        self._chk(self.eng.lib.rpe_comm_allreduce_max(self.c, C.byref(v)), "rpe_comm_allreduce_max")
        return float(v.value)

    def close(self):
        if getattr(self, "c", None):
            self.eng.lib.rpe_comm_destroy(self.c)
            self.c = None


# ----------------------------------------------------------------------------- torch.distributed path (tests / rehearsal)
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

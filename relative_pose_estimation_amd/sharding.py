"""Sharding across the GPUs of one node and the final pose gather.

Pairs are independent (the reference never chains estimates: batch_processor.py:82-92 takes R_prev from
ground truth), so every rank processes a contiguous block with no data-path collective.  The only exchange
is one all-gather of fixed-size 128-byte pose records at the end of a step:

  * `PoseComm` -- the product path: `rpe_gather_poses` in librpe_amd.so, an ncclAllGather over RCCL / xGMI on
    the engine's HIP stream, records packed on the device.  No torch: ranks bootstrap through the RCCL unique
    id that rank 0 publishes in a per-user rendezvous directory (one node, so a local path is visible to every
    rank), and they AGREE on the outcome of every set-up phase, so a failure on one rank is a failure on all.
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
class CommUnavailable(RuntimeError):
    """The native pose gather cannot be set up on this node.  Raised by EVERY rank of the job (the outcome of each
    set-up phase is agreed through the rendezvous directory), so all ranks take the same way out."""


class Rendezvous:
    """File rendezvous of the ranks of ONE node (the sharded path is one process per GPU of one node, so a local
    directory is visible to every rank).  One directory per job inside a per-user 0700 directory; every file is
    created exclusively with mode 0600 and published by rename, so a reader sees a whole file or none.

    The job tag must be the same on all ranks and different from any other job of this user on this node:
      * RPE_COMM_TAG, if set (any launcher: mpirun, srun, a wrapper shell per rank);
      * else MASTER_ADDR, MASTER_PORT and TORCHELASTIC_RUN_ID plus -- only under `python -m torch.distributed.run`,
        which sets TORCHELASTIC_RUN_ID and is the parent of every rank -- that launcher's pid, which keeps files
        left behind by a crashed earlier launch on the same port from being mistaken for this launch's."""

    def __init__(self, rank, world, tag=None, timeout=300.0, root=None):
        self.rank, self.world, self.timeout = int(rank), int(world), float(timeout)
        env = os.environ
        job = env.get("RPE_COMM_TAG")
        if not job:
            job = "{}_{}_{}".format(env.get("MASTER_ADDR", "127.0.0.1"), env.get("MASTER_PORT", "0"), env.get("TORCHELASTIC_RUN_ID", "none"))
            if "TORCHELASTIC_RUN_ID" in env:
                job += f"_p{os.getppid()}"
        job = "".join(ch if ch.isalnum() or ch in "._-" else "_" for ch in f"{job}_{tag or ''}")
        root = root or env.get("RPE_COMM_DIR") or os.path.join("/tmp", f"rpe_comm_{os.getuid()}")
        os.makedirs(root, mode=0o700, exist_ok=True)
        st = os.stat(root)
        if st.st_uid != os.getuid() or (st.st_mode & 0o077):
            raise CommUnavailable(f"rendezvous directory {root} is not private to this user")
        self.dir = os.path.join(root, job)
        os.makedirs(self.dir, mode=0o700, exist_ok=True)
        # files of an EARLIER launch under the same tag (an explicit RPE_COMM_TAG reused, a crashed run) must not be read as
        # this launch's: nothing older than STALE_S before this process came up is accepted, and rank 0 sweeps such files away.
        # (Reusing an explicit tag within STALE_S of a crashed launch is the one case this cannot tell apart: use a fresh tag.)
        self.t_min = time.time() - self.STALE_S
        if self.rank == 0:
            for f in os.listdir(self.dir):
                try:
                    if os.stat(os.path.join(self.dir, f)).st_mtime < self.t_min:
                        os.remove(os.path.join(self.dir, f))
                except OSError:
                    pass

    STALE_S = 120.0

    def publish(self, name, payload=b""):
        final = os.path.join(self.dir, name)
        tmp = f"{final}.tmp{os.getpid()}"
        fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL, 0o600)
        try:
            os.write(fd, payload)
        finally:
            os.close(fd)
        os.replace(tmp, final)

    def wait(self, name):
        path = os.path.join(self.dir, name)
        t0 = time.time()
        while True:
            try:
                with open(path, "rb") as fh:
                    if os.fstat(fh.fileno()).st_mtime >= self.t_min:
                        return fh.read()
            except FileNotFoundError:
                pass
            if time.time() - t0 > self.timeout:
                raise CommUnavailable(f"rank {self.rank}: nothing at {path} after {self.timeout:g} s")
            time.sleep(0.02)

    def agree(self, phase, ok, message=""):
        """Every rank reports the outcome of `phase`; returns normally on ALL ranks if all succeeded, raises
        CommUnavailable on ALL ranks otherwise (a rank that died is a timeout, i.e. also a raise everywhere)."""
        self.publish(f"{phase}.{self.rank}", (b"ok" if ok else b"fail:" + str(message).encode()))
        bad = []
        for r in range(self.world):
            data = self.wait(f"{phase}.{r}")
            if data != b"ok":
                bad.append(f"rank {r}: {data[5:].decode(errors='replace') or 'failed'}")
        if bad:
            raise CommUnavailable(f"{phase}: " + "; ".join(bad))

    def cleanup(self):
        """rank 0, after the last agreement: the job's files are of no further use"""
        try:
            for f in os.listdir(self.dir):
                os.remove(os.path.join(self.dir, f))
            os.rmdir(self.dir)
        except OSError:
            pass


def exchange_unique_id(lib, rdv):
    """Rank 0 asks RCCL for a unique id and publishes it -- or publishes WHY it could not, so that the other ranks stop
    waiting; the others read it.  Raises CommUnavailable on every rank when rank 0 failed."""
    if rdv.rank == 0:
        buf = (C.c_uint8 * 128)()
        rc = lib.rpe_comm_unique_id(buf)
        if rc != 0:
            msg = f"rpe_comm_unique_id failed ({rc}): {lib.rpe_comm_last_error().decode()}"
            rdv.publish("id", b"FAIL" + msg.encode())
            raise CommUnavailable(msg)
        rdv.publish("id", bytes(buf))
        return bytes(buf)
    data = rdv.wait("id")
    if data[:4] == b"FAIL" or len(data) != 128:
        raise CommUnavailable("rank 0: " + (data[4:].decode(errors="replace") if data[:4] == b"FAIL" else "malformed unique id"))
    return data


class PoseComm:
    """RCCL communicator bound to one Engine (one GPU): gather of pose records, barrier, max-reduce of a scalar.
    Set-up is collective in its OUTCOME: the ranks agree after the local phase (librccl loaded, device buffers allocated)
    and again after ncclCommInitRank; if any rank failed a phase every rank raises CommUnavailable and nobody is left
    waiting inside a collective."""

    def __init__(self, engine, rank, world, tag=None, timeout=300.0):
        self.eng, self.rank, self.world = engine, rank, world
        self.c = None
        lib = engine.lib
        rdv = Rendezvous(rank, world, tag, timeout)
        try:
            uid = exchange_unique_id(lib, rdv)
            c = C.c_void_p()
            rc = lib.rpe_comm_prepare(engine.h, rank, world, C.byref(c))          # local: dlopen librccl, hipMalloc
            if rc == 0:
                self.c = c
            rdv.agree("prepare", rc == 0, "" if rc == 0 else f"rpe_comm_prepare ({rc}): {lib.rpe_comm_last_error().decode()}")
            idbuf = (C.c_uint8 * 128).from_buffer_copy(uid)
            rc = lib.rpe_comm_connect(self.c, idbuf)                                # collective: ncclCommInitRank
            rdv.agree("connect", rc == 0, "" if rc == 0 else f"rpe_comm_connect ({rc}): {lib.rpe_comm_last_error().decode()}")
            self.barrier()
            rdv.agree("barrier", True)
            self.barrier()                                   # every rank is past its last read of the rendezvous files
            if rank == 0:
                rdv.cleanup()
        except BaseException:
            self.close()
            if rank == 0:                                    # the others poll every 20 ms: give them time to read the verdict
                time.sleep(2.0)
                rdv.cleanup()
            raise

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

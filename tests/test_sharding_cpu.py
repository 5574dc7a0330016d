"""world_size-2 gloo test of the N>1 path: contiguous pair sharding, no data-path
collective, one all-gather of 128-byte pose records (RCCL on the GPU box)."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys, numpy as np
sys.path.insert(0, os.environ["RPE_ROOT"])
import torch.distributed as dist
from relative_pose_estimation_amd import sharding, synthetic, geometry
from oracle import oracle
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
total = 3
lo, hi = sharding.shard_bounds(total, rank, world)
K = geometry.default_camera_matrix(320, 240)
i1, i2, _, _ = synthetic.make_batch(hi - lo, K, 320, 240, cfg=4, first=lo)
res = oracle.estimate_pose_batch(i1, i2, K, 300, 200, nthreads=1)      # stands in for the per-rank engine on a CPU box
rec = sharding.pack_records(res["R"], res["t"], res["inliers"], res["status"], res["n_matches"], first_pair=lo)
allrec = sharding.gather_pose_records(rec, per_rank=-(-total // world))
if rank == 0:
    np.save(os.environ["RPE_OUT"], allrec)
dist.destroy_process_group()
"""


def test_shard_bounds():
    from relative_pose_estimation_amd.sharding import shard_bounds
    assert [shard_bounds(32768, r, 8) for r in (0, 7)] == [(0, 4096), (28672, 32768)]
    cover = [shard_bounds(10, r, 4) for r in range(4)]
    assert cover == [(0, 3), (3, 6), (6, 9), (9, 10)]
    assert shard_bounds(2, 3, 4) == (2, 2)


def test_shard_stream_bounds():
    """configs[4]: one sequence over the ranks with a one-frame halo -- every pair exactly once, every rank's frame
    range = its pairs' first frames + one shared frame."""
    from relative_pose_estimation_amd.sharding import shard_stream_bounds
    for F, world in ((4541, 8), (880, 8), (10, 4), (3, 8), (2, 2), (1, 2)):
        seen = []
        for r in range(world):
            flo, fhi, plo, phi = shard_stream_bounds(F, r, world)
            if phi > plo:
                assert (flo, fhi) == (plo, phi + 1)          # frames p .. p_last + 1: one halo frame
                assert fhi <= F
            else:
                assert flo == fhi
            seen += list(range(plo, phi))
        assert seen == list(range(max(F - 1, 0)))
    assert shard_stream_bounds(4541, 0, 8) == (0, 569, 0, 568) and shard_stream_bounds(4541, 7, 8) == (3976, 4541, 3976, 4540)


def test_stream_shards_equal_whole_stream(oracle):
    """the halo split changes nothing: pair p computed inside its shard (frames re-indexed from the shard's first
    frame) equals pair p of the unsharded sequence (oracle as the per-rank engine on a CPU box)"""
    from relative_pose_estimation_amd import synthetic, geometry
    from relative_pose_estimation_amd.sharding import shard_stream_bounds
    K = geometry.default_camera_matrix(320, 240)
    frames, _, _ = synthetic.make_stream(6, K, 320, 240, seed=77, workers=1)
    whole = oracle.estimate_pose_batch(frames[:-1], frames[1:], K, 300, 200, nthreads=2)
    for r in range(2):
        flo, fhi, plo, phi = shard_stream_bounds(6, r, 2)
        sh = frames[flo:fhi]
        part = oracle.estimate_pose_batch(sh[:-1], sh[1:], K, 300, 200, nthreads=2)
        assert len(part) == phi - plo and np.array_equal(part["R"], whole["R"][plo:phi]) and np.array_equal(part["inliers"], whole["inliers"][plo:phi])


def test_gloo_world2_gather(tmp_path, oracle):
    out = str(tmp_path / "rec.npy")
    env = dict(os.environ, RPE_ROOT=ROOT, RPE_OUT=out, MASTER_ADDR="127.0.0.1")
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29531", str(script)]
    subprocess.run(cmd, check=True, env=env, timeout=600, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    rec = np.load(out)
    from relative_pose_estimation_amd import synthetic, geometry, sharding
    assert rec.dtype == sharding.RECORD_DTYPE and rec["pair"].tolist() == [0, 1, 2]
    K = geometry.default_camera_matrix(320, 240)
    i1, i2, _, _ = synthetic.make_batch(3, K, 320, 240, cfg=4, first=0)
    ref = oracle.estimate_pose_batch(i1, i2, K, 300, 200, nthreads=1)
    assert np.array_equal(rec["R"], ref["R"]) and np.array_equal(rec["t"], ref["t"])
    assert np.array_equal(rec["inliers"], ref["inliers"]) and np.array_equal(rec["status"], ref["status"])

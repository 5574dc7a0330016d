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


# ------------------------------------------------------------------ native rendezvous (no GPU: a stub library object)
RDV_WORKER = r"""
import os, sys, ctypes as C
sys.path.insert(0, os.environ["RPE_ROOT"])
from relative_pose_estimation_amd import sharding

class StubLib:                                   # the three entry points exchange_unique_id / agree use
    def rpe_comm_unique_id(self, buf):
        if os.environ.get("RPE_TEST_FAIL_ID"):
            return -2
        for i in range(128):
            buf[i] = (i * 7 + 3) & 255
        return 0
    def rpe_comm_last_error(self):
        return b"stub: no librccl"

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
rdv = sharding.Rendezvous(rank, world, tag="t", timeout=20.0, root=os.environ["RPE_TEST_ROOT"])
out = []
try:
    uid = sharding.exchange_unique_id(StubLib(), rdv)
    out.append("id:" + str(sum(uid)))
    fail_rank = int(os.environ.get("RPE_TEST_FAIL_PREPARE", "-1"))
    rdv.agree("prepare", rank != fail_rank, "hipMalloc failed")
    out.append("prepared")
    rdv.agree("connect", True)
    out.append("connected")
except sharding.CommUnavailable as exc:
    out.append("unavailable:" + str(exc))
print("|".join(out), flush=True)
"""


def _run_ranks(tmp_path, style, extra_env=None, world=2, tag="42"):
    """style 'torchrun': python -m torch.distributed.run starts the ranks (RANK / WORLD_SIZE / MASTER_* / TORCHELASTIC_RUN_ID
    from the launcher); style 'plain': one subprocess per rank with RANK / WORLD_SIZE and an explicit RPE_COMM_TAG, as under
    mpirun / srun / a wrapper shell (different parents: nothing may depend on the parent pid)."""
    script = tmp_path / f"rdv_{style}.py"
    script.write_text(RDV_WORKER)
    root = tmp_path / f"root_{style}"
    env = dict(os.environ, RPE_ROOT=ROOT, RPE_TEST_ROOT=str(root), **(extra_env or {}))
    if style == "torchrun":
        logs = tmp_path / f"logs_{style}"
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
               "--master-port", "29641", "--redirects", "3", "--log-dir", str(logs), str(script)]
        subprocess.run(cmd, env=env, check=True, timeout=120, capture_output=True)
        outs = []
        for r in range(world):
            f = [p for p in logs.rglob("stdout.log") if f"/{r}/" in str(p)]
            outs.append(f[0].read_text().strip().splitlines()[-1])
        return outs
    procs = []
    for r in range(world):
        e = dict(env, RANK=str(r), WORLD_SIZE=str(world), RPE_COMM_TAG="job" + tag)
        e.pop("TORCHELASTIC_RUN_ID", None)
        # every rank behind its own wrapper shell: the ranks do NOT share a parent process
        procs.append(subprocess.Popen(["/bin/sh", "-c", f"exec {sys.executable} {script}"] if r % 2 else [sys.executable, str(script)],
                                      env=e, stdout=subprocess.PIPE, text=True))
    return [p.communicate(timeout=120)[0].strip().splitlines()[-1] for p in procs]


def test_rendezvous_two_ranks_both_launch_styles(tmp_path):
    want = "id:" + str(sum((i * 7 + 3) & 255 for i in range(128))) + "|prepared|connected"
    for style in ("plain", "torchrun"):
        assert _run_ranks(tmp_path, style) == [want, want], style
    # the rendezvous directory is private to the user
    st = os.stat(tmp_path / "root_plain")
    assert (st.st_mode & 0o077) == 0


def test_rendezvous_failures_are_agreed(tmp_path):
    """a failure on ONE rank is CommUnavailable on ALL ranks -- nobody is left polling or waiting in a collective"""
    outs = _run_ranks(tmp_path, "plain", {"RPE_TEST_FAIL_ID": "1"}, tag="43")
    assert all(o.startswith("unavailable:") and "stub: no librccl" in o for o in outs), outs
    outs = _run_ranks(tmp_path, "plain", {"RPE_TEST_FAIL_PREPARE": "1"}, tag="44")        # a fresh tag per launch (Rendezvous docstring)
    assert all("|unavailable:prepare: rank 1: hipMalloc failed" in o for o in outs), outs


def test_bench_refuses_a_mislaunch():
    """bench.py --gpus N never prints a line for another N: a launcher whose WORLD_SIZE disagrees is an error (without a
    launcher it starts the N ranks itself -- exercised on the GPU box, bench.self_launch)."""
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "--gpus 4 but WORLD_SIZE=2" in (p.stderr + p.stdout) and '"n_gpus"' not in p.stdout
    import bench
    assert callable(bench.self_launch)

#!/usr/bin/env python3
"""Benchmark of the MI355X relative-pose hot path (BASELINE.json metric:
image-pairs/s end-to-end + median rotation-angle error, 640x480 pairs).

A step = one pass of feature-extract -> match -> essential RANSAC -> pose over one
batch of synthetic pairs whose images are already resident in HBM; the step ends when
the pose records are in host memory.  Workload (config.workload) = BASELINE configs[1]:
1024 VGA pairs per GPU, ORB(1000 kp) + BF-Hamming crossCheck + 5-pt RANSAC + recoverPose.
Weak scaling: every rank processes its own 1024-pair shard; the only collective is the
RCCL all-gather of 128-byte pose records at the end of each step (rpe_gather_poses in
librpe_amd.so: no torch on the path; the launcher's RANK / WORLD_SIZE / MASTER_PORT
environment is all that is used).

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

The default single-GPU run appends `extra`: BASELINE configs[2]'s shape (128 HD pairs,
SIFT(2048) + BF-L2) and the consecutive-frame stream (configs[4] stand-in), each with its
own roofline and cpu_baseline (--no-extra skips them).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6290 measured
# vector-instruction issue peak (MI355X_MICROARCH.md 'Wave scheduling' / cycle constants: a wave64 VALU
# instruction takes 2 cycles on the 32-wide SIMD with >= 2 waves resident): 256 CUs x 4 SIMDs x 2.4 GHz / 2
VALU_PEAK_WAVE_INSTS = 256 * 4 * 2.4e9 / 2
# dense int8 MFMA peak: 2x the bf16 rate per clock (MI355X_MICROARCH.md 'Matrix cores'), bf16 dense ~2.5 PFLOP/s
MFMA_I8_PEAK_TOPS = 5000.0
COUNTERS = os.path.join(ROOT, "profiles", "r03_counters.json")
if not os.path.exists(COUNTERS):
    COUNTERS = os.path.join(ROOT, "profiles", "r02_counters.json")


def orb_levels(W, H, nlevels=12, scale=1.1):
    """ORB pyramid level sizes (orb.cpp: cvRound(cols * (1.0f / scale_l)), scale_l = (float)pow(1.1f, l));
    sum(w*h) must equal rpe_orb_pyramid_pixels()."""
    out = []
    sf = np.float64(np.float32(scale))
    for l in range(nlevels):
        inv = np.float32(1.0) / np.float32(sf ** l)
        out.append((int(np.rint(np.float32(W) * inv)), int(np.rint(np.float32(H) * inv))))
    return out


def sift_octaves(W, H):
    """SIFT octave sizes (sift.dispatch.cpp: base = 2x upsampled image, nOctaves = round(log2(min) - 2) + 1)."""
    bw, bh = 2 * W, 2 * H
    noct = min(12, int(np.rint(np.log(min(bw, bh)) / np.log(2.0) - 2)) + 1)
    out = []
    for o in range(noct):
        out.append((bw, bh))
        bw, bh = bw // 2, bh // 2
    return out


def stage_bytes(stage, W, H, pyr_px, nkp, mm, method="ORB"):
    """ALGORITHMIC (compulsory) HBM bytes of one stage for ONE PAIR (= 2 images),
    DESIGN.md 'Algorithmic bytes'.  pyr_px = pixels of one 12-level pyramid."""
    img = W * H
    if method == "SIFT":
        px = sum(w * h for w, h in sift_octaves(W, H))
        per_image = {
            "pyramid": img + 6 * 4 * px,                 # read the u8 image, write the 6 Gaussian f32 levels per octave (DoG is never stored)
            "fast": 6 * 4 * px + 3 * px // 8,            # extrema scan: read every Gaussian level once (DoG on the fly), write the 1-bit hit mask
            "select": 27 * 4 * nkp * 8,                  # adjustLocalExtrema: 3x3x3 DoG block per seed (~8 seeds per kept keypoint)
            "harris": 4 * 4 * 400 * nkp,                 # orientation: 4 gradient taps x ~400 window samples per keypoint
            "keypoints": 24 * nkp * 4,
            "describe": 4 * 4 * 4000 * nkp + 128 * nkp,  # descriptor window gradients + 128-B descriptor
        }
        if stage in per_image:
            return 2 * per_image[stage]
        if stage == "match":
            return 2 * nkp * 128 + nkp * 8               # u8 descriptors: (N1+N2)*128 + N1*8
        if stage in ("nms", "angle", "blur"):
            return 0
    lv = orb_levels(W, H)
    assert sum(w * h for w, h in lv) == pyr_px, "level-size formula disagrees with the library"
    live = [(w, h) for w, h in lv if w > 62 and h > 62]
    nms_survivors = 5 * nkp                      # ~0.3 % of the pyramid pixels survive NMS (measured on the synthetic batch)
    per_image = {
        "pyramid": img + pyr_px,                 # read level 0, write the 12-level pyramid
        # FAST+NMS reads the border-filtered region once (ring reads reach 4 px past the NMS output [31, w-31)) and
        # writes one 4-byte list entry per surviving keypoint + the histograms
        "fast": sum((w - 54) * (h - 54) for w, h in live) + 4 * nms_survivors + 12 * 256 * 4,
        "nms": 0,                                # fused into fast
        "select": 4 * nms_survivors + 12 * 256 * 4 + 4 * 2 * nkp,   # tile lists in, histogram in, ~2*quota candidates out
        "harris": 81 * 2 * nkp + 8 * 2 * nkp,    # 9x9 patch per candidate (~2*quota kept) + record
        "keypoints": 16 * 2 * nkp,
        "angle": 45 * 45 * nkp + 4 * nkp + 32 * nkp,   # fused orientation + descriptor: one 45x45 patch per keypoint in, angle + 32-B descriptor out
        "blur": 0,                               # fused into angle (per-keypoint patch blur)
        "describe": 0,                           # fused into angle
    }
    if stage in per_image:
        return 2 * per_image[stage]
    if stage == "match":
        return 2 * nkp * 32 + nkp * 8            # (N1+N2)*32 + N1*8
    if stage == "ransac":
        return mm * 2 * 2 * 4 + mm               # matched points in, mask out
    if stage == "pose":
        return mm * 2 * 2 * 4 + 128
    raise KeyError(stage)


STAGE_KERNEL = {"pyramid": "pyr_resize_kernel", "fast": "fast_nms_kernel", "angle": "orient_describe_kernel",
                "select": "raster_corners_kernel + retain_fast_kernel", "harris": "harris_kernel", "keypoints": "retain_harris_kernel + compact_keypoints_kernel",
                "match": "match_hamming_mfma_kernel", "ransac": "ransac_*_kernel (group)", "pose": "recover_pose_kernel"}
# SIFT reuses the stage slots (csrc/sift_kernels.hip rpe_sift_run)
SIFT_STAGE_KERNEL = {"pyramid": "sift_blur_fused_kernel", "fast": "sift_extrema_mask_kernel",
                     "select": "sift_adjust_kernel", "harris": "sift_orient_kernel", "keypoints": "sift_finalize_kernel",
                     "describe": "sift_describe_kernel", "match": "match_l2_mfma_kernel",
                     "ransac": "ransac_*_kernel (group)", "pose": "recover_pose_kernel"}
# instruction kind of rpe_calibrate_valu that a stage's inner loop is made of (the measured issue roof it is priced against)
STAGE_CALIB_KIND = {"fast": 4, "pyramid": 5, "angle": 3, "match": 3, "ransac": 6, "pose": 6, "harris": 5, "select": 5, "keypoints": 5,
                    "describe": 5, "nms": 5, "blur": 5}


def counters_for(workload_key):
    """Per-kernel PMC numbers of the committed rocprofv3 passes (profiles/r03_counters.json, written by
    profiles/make_counters.py from separate --pmc runs: FETCH_SIZE, WRITE_SIZE, SQ_*): HBM bytes per launch
    (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE) and SQ_INSTS_VALU per launch, or {} when no profile exists for
    this workload."""
    if not os.path.exists(COUNTERS):
        return {}
    return json.load(open(COUNTERS)).get(workload_key, {}).get("kernels", {})


def kernel_counter(ctr, kernel, field):
    """sum over the kernels whose name starts with `kernel` (template instances, kernel groups)"""
    vals = [v[field] * v.get("launches_per_step", 1) for k, v in ctr.items() if k.startswith(kernel) and field in v]
    return float(sum(vals)) if vals else None


def cpu_baseline(i1, i2, K, nfeatures, max_matches, sample, method="ORB"):
    """The oracle (a scalar C port of the reference's cv2 calls) timed on this host's
    cores on a bounded sample of the same workload."""
    from oracle import oracle
    oracle.build()
    # a 1-GPU box owns a 16-core share of the host (RPE_CPU_SHARE overrides)
    cores = min(os.cpu_count() or 1, int(os.environ.get("RPE_CPU_SHARE", "16")))
    n = min(sample, len(i1))
    threads = min(cores, n)
    t0 = time.perf_counter()
    res = oracle.estimate_pose_batch(i1[:n], i2[:n], K, nfeatures, max_matches, nthreads=threads, method=method)
    dt = time.perf_counter() - t0
    # single-thread latency of one pair: the first pair alone
    t1 = time.perf_counter()
    oracle.estimate_pose_batch(i1[:1], i2[:1], K, nfeatures, max_matches, nthreads=1, method=method)
    lat = time.perf_counter() - t1
    return res, {"value": n / dt, "unit": "pairs/s", "cores": threads, "kind": "port",
                 "single_thread_pair_latency_ms": lat * 1e3,
                 "sample": f"first {n} pairs of the same synthetic batch ({i1.shape[2]}x{i1.shape[1]}, {method} {nfeatures}), "
                           f"{threads} pthreads over pairs, {dt:.1f} s wall"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1024, help="pairs per GPU per step")
    ap.add_argument("--nfeatures", type=int, default=1000)
    ap.add_argument("--max-matches", type=int, default=500)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--cpu-sample", type=int, default=1024, help="pairs of the batch the CPU oracle is timed on (about 10 s on a 16-core share)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the configs[2] / stream lines appended to the default single-GPU run")
    ap.add_argument("--no-calibrate", action="store_true", help="skip the live VALU / HBM calibration kernels")
    ap.add_argument("--gen-workers", type=int, default=0)
    ap.add_argument("--config", type=int, default=2, choices=[2, 3],
                    help="BASELINE configs[]: 2 = 1024 VGA pairs ORB(1000)+Hamming (the metric's config, default); "
                         "3 = 1920x1080 pairs SIFT(2048)+L2, processed in sub-batches")
    ap.add_argument("--stream", action="store_true",
                    help="BASELINE configs[4] stand-in: consecutive-frame stream, features once per frame; with --gpus N ONE "
                         "sequence of N*batch+1 frames is cut into per-rank frame ranges with a one-frame halo")
    ap.add_argument("--sub-batch", type=int, default=0, help="pairs per enqueue (config 3 default 128: 0.66 GB of pyramid per pair)")
    ap.add_argument("--streams", type=int, default=1, help="split the batch over S engine handles (S HIP streams) so latency-bound stages overlap")
    ap.add_argument("--dist-backend", default="rccl", choices=["rccl", "gloo"],
                    help="rccl (default) = rpe_gather_poses in librpe_amd.so (ncclAllGather, no torch); "
                         "gloo = torch.distributed rehearsal of the N>1 path on one GPU / CPU")
    ap.add_argument("--data-cache", default="", help="npz file to load/save the synthetic batch (keeps forks out of profiled runs)")
    ap.add_argument("--unique", type=int, default=0, help="generate only this many distinct pairs and tile them to --batch "
                    "(full-size config 3 runs: rendering 4096 HD pairs takes longer than measuring them)")
    return ap.parse_args()


def resolve(args):
    """fills the per-config defaults; returns the workload descriptor"""
    method = "ORB"
    if args.config == 3:
        method = "SIFT"
        if args.width == 640 and args.height == 480:
            args.width, args.height = 1920, 1080
        if args.nfeatures == 1000:
            args.nfeatures = 2048
        if args.batch == 1024:
            args.batch = 128
        if args.cpu_sample == 1024:
            args.cpu_sample = 48
    sub = args.sub_batch or (128 if method == "SIFT" else args.batch)
    return method, min(sub, args.batch)


def generate(args, method, rank, world, workers):
    """synthetic data of one workload -- called BEFORE any GPU / process-group initialisation because the
    generator forks worker processes"""
    from relative_pose_estimation_amd import synthetic, geometry, sharding
    W, H, B = args.width, args.height, args.batch
    K = geometry.default_camera_matrix(W, H)
    if args.stream:
        # ONE sequence of world*B + 1 frames; this rank renders its frame range (its pairs' first frames + 1 halo frame)
        F = world * B + 1
        flo, fhi, plo, phi = sharding.shard_stream_bounds(F, rank, world)
        frames, Rgt, tgt = synthetic.make_stream(F, K, W, H, seed=5_000_011, workers=workers, frame_range=(flo, fhi))
        return dict(K=K, frames=frames, i1=frames[:-1], i2=frames[1:], Rgt=Rgt, tgt=tgt, first_pair=plo, distinct=B)
    cache = f"{args.data_cache}.r{rank}.npz" if args.data_cache else ""
    if cache and os.path.exists(cache):
        z = np.load(cache)
        i1, i2, Rgt, tgt = z["i1"], z["i2"], z["R"], z["t"]
        assert i1.shape == (B, H, W), "data cache does not match the requested workload"
        return dict(K=K, i1=i1, i2=i2, Rgt=Rgt, tgt=tgt, first_pair=rank * B, distinct=(min(args.unique, B) if args.unique > 0 else B))
    U = min(args.unique, B) if args.unique > 0 else B
    i1, i2, Rgt, tgt = synthetic.make_batch(U, K, W, H, cfg=2, first=rank * B, workers=workers)
    if U < B:
        reps = -(-B // U)
        i1, i2 = np.concatenate([i1] * reps)[:B], np.concatenate([i2] * reps)[:B]
        Rgt, tgt = np.concatenate([Rgt] * reps)[:B], np.concatenate([tgt] * reps)[:B]
    if cache:
        np.savez(cache, i1=i1, i2=i2, R=Rgt, t=tgt)
    return dict(K=K, i1=i1, i2=i2, Rgt=Rgt, tgt=tgt, first_pair=rank * B, distinct=U)


def run_workload(args, method, sub, data, rank, world, device, comm_kind):
    """times args.steps steps of one workload on this rank's GPU; returns the JSON object (rank 0) or None"""
    from relative_pose_estimation_amd import _capi, geometry, sharding
    W, H, B = args.width, args.height, args.batch
    K, i1, i2, Rgt, tgt = data["K"], data["i1"], data["i2"], data["Rgt"], data["tgt"]
    S = max(1, args.streams)
    bounds = [sharding.shard_bounds(B, s, S) for s in range(S)]
    fm, nt = (_capi.FEATURE_SIFT, _capi.NORM_L2) if method == "SIFT" else (_capi.FEATURE_ORB, _capi.NORM_HAMMING)
    engs = [_capi.Engine(W, H, max_batch=min(sub, hi - lo), nfeatures=args.nfeatures, max_matches=args.max_matches, device=device,
                         feature_method=fm, norm_type=nt)
            for lo, hi in bounds]
    eng = engs[0]
    comm = None
    dist = None
    if world > 1:
        assert S == 1 and sub >= B, "multi-GPU runs use one handle and one launch group per rank"
        if comm_kind == "rccl":
            try:
                comm = sharding.PoseComm(eng, rank, world, tag=f"{args.config}{int(args.stream)}")
            except sharding.CommUnavailable as exc:
                # raised by EVERY rank (the ranks agree on the outcome of each set-up phase, sharding.PoseComm): all of them
                # say so and deliver the 128-byte records through torch.distributed instead
                sys.stderr.write(f"[bench] rank {rank}: native RCCL pose gather unavailable ({exc}); using torch.distributed gloo for the 128-B records\n")
                comm_kind = "gloo (RCCL gather unavailable)"
        if comm is None:
            import torch.distributed as dist
            if not dist.is_initialized():
                dist.init_process_group("gloo")
    if args.stream:
        assert S == 1 and sub >= B, "--stream uses one handle and one launch group"
        d_frames = eng.upload(data["frames"])
        dbuf = [(d_frames, d_frames)]
    else:
        dbuf = [(e.upload(i1[lo:hi]), e.upload(i2[lo:hi])) for e, (lo, hi) in zip(engs, bounds)]   # inputs resident in HBM
    for e in engs:
        e.set_profiling(True)

    def barrier():
        for e in engs:
            e.synchronize()
        if comm is not None:
            comm.barrier()
        elif dist is not None:
            dist.barrier()

    sub_acc = {}

    def step():
        if args.stream:
            eng.enqueue_stream_device(dbuf[0][0], B + 1, K)
            parts = None if comm is not None else [eng.fetch_results(B)]
        elif sub >= B:
            for e, (a, b), (lo, hi) in zip(engs, dbuf, bounds):
                e.enqueue_batch_device(a, b, hi - lo, K)
            parts = None if comm is not None else [e.fetch_results(hi - lo) for e, (lo, hi) in zip(engs, bounds)]
        else:                       # sub-batched (workspace-bound configs): one engine, consecutive slices
            import ctypes
            parts = []
            (a, b), e = dbuf[0], engs[0]
            for lo in range(0, B, sub):
                n = min(sub, B - lo)
                off = lo * W * H
                e.enqueue_batch_device(ctypes.c_void_p(a.value + off), ctypes.c_void_p(b.value + off), n, K)
                parts.append(e.fetch_results(n))
                for k, v in e.stage_ms().items():
                    sub_acc[k] = sub_acc.get(k, 0.0) + v
        if comm is not None:
            # the pose records of every rank, packed on the device and all-gathered over RCCL / xGMI
            rec = comm.gather(B, B, data["first_pair"])
            mine = rec[(rec["pair"] >= data["first_pair"]) & (rec["pair"] < data["first_pair"] + B)]
            return rec, (mine["R"].reshape(-1, 3, 3), mine["t"].reshape(-1, 3, 1), mine["inliers"], mine["n_matches"], mine["status"])
        R, t, inl, nm, st = (np.concatenate([p[k] for p in parts]) for k in range(5))
        rec = sharding.pack_records(R, t, inl, st, nm, first_pair=data["first_pair"])
        if dist is not None:
            rec = sharding.gather_pose_records(rec, B)
        return rec, (R, t, inl, nm, st)

    for _ in range(args.warmup):
        step()
    barrier()
    stage_acc = {}
    t0 = time.perf_counter()
    for _ in range(args.steps):
        rec, local = step()
        if sub >= B:
            for e in engs:
                for k, v in e.stage_ms().items():
                    stage_acc[k] = stage_acc.get(k, 0.0) + v
    barrier()
    elapsed = time.perf_counter() - t0
    if comm is not None:
        elapsed = comm.max(elapsed)
    elif dist is not None:
        import torch
        tt = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    assert len(rec) == world * B and sorted(rec["pair"].tolist()) == list(range(world * B)), "pose gather incomplete"

    R, t, inl, nm, st = local
    ok = st == 0
    errs = np.array([geometry.rotation_error(R[i], Rgt[i]) for i in range(B) if ok[i]])
    # translation is recovered up to scale: direction error against the ground-truth direction (pose_evaluator.py:111-116)
    terrs = np.array([geometry.translation_direction_error(t[i], tgt[i]) for i in range(B) if ok[i] and np.linalg.norm(tgt[i]) > 0])
    # per-LAUNCH averages: every step launches each kernel group once per stream on B/S pairs; fused-away slots are dropped
    stage_ms = {k: v / (args.steps * S) for k, v in stage_acc.items()}
    Bl = B // S if B % S == 0 else B / S          # pairs per launch
    if sub < B:
        Bl = sub
        launches = (args.steps + args.warmup) * (-(-B // sub))
        stage_ms = {k: v / launches for k, v in sub_acc.items()}
    fused = ("nms", "blur", "describe") if method == "ORB" else ("nms", "angle", "blur")
    stage_ms = {k: v for k, v in stage_ms.items() if k not in fused}
    pyr_px = eng.lib.rpe_orb_pyramid_pixels(eng.h)

    out = None
    if rank == 0:
        names = SIFT_STAGE_KERNEL if method == "SIFT" else STAGE_KERNEL
        wkey = f"{method}_{W}x{H}_{args.nfeatures}_{int(Bl)}" + ("_stream" if args.stream else "")
        ctr = counters_for(wkey)
        calib = {}

        def issue_rate(kind):
            if args.no_calibrate:
                return None
            if kind not in calib:
                calib[kind] = eng.calibrate_valu(kind, 8)
            return calib[kind]

        def roof(stage):
            """both roofs of one stage's kernel: HBM (algorithmic bytes / spec peak) and VALU issue (SQ_INSTS_VALU of the
            committed PMC pass / measured issue rate of the instruction kind its inner loop is made of)"""
            ms = stage_ms[stage]
            b = stage_bytes(stage, W, H, pyr_px, args.nfeatures, args.max_matches, method) * Bl
            ach = b / (ms * 1e-3) / 1e9
            r = {"kernel": names.get(stage, stage), "stage": stage, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                 "frac": ach / HBM_PEAK_GBS, "traffic": kernel_counter(ctr, names.get(stage, stage).split(" ")[0].replace("_*_kernel", "_"), "hbm_bytes_per_launch"),
                 "algorithmic_bytes_per_launch": b, "avg_launch_ms": ms}
            insts = kernel_counter(ctr, names.get(stage, stage).split(" ")[0].replace("_*_kernel", "_"), "valu_insts_per_launch")
            rate = issue_rate(STAGE_CALIB_KIND.get(stage, 5))
            if insts is not None:
                v = {"insts": insts, "insts_per_s": insts / (ms * 1e-3), "spec_peak_insts_per_s": VALU_PEAK_WAVE_INSTS,
                     "frac_of_spec": insts / (ms * 1e-3) / VALU_PEAK_WAVE_INSTS}
                if rate is not None:
                    v.update({"measured_peak_insts_per_s": rate[1], "calibration_instruction": rate[0], "frac": insts / (ms * 1e-3) / rate[1]})
                else:
                    v["frac"] = v["frac_of_spec"]
                r["valu"] = v
                r["bound"] = "valu" if v["frac"] > r["frac"] else "hbm"
            else:
                r["bound"] = "hbm"
            return r

        dom = max(stage_ms, key=stage_ms.get)
        rl = roof(dom)
        rl["matcher"] = roof("match")              # the stage north_star attaches a roofline target to: always reported
        if method == "ORB":
            # the crossCheck Hamming matcher runs its O(N^2) part on the matrix cores: d = |q| + |t| - 2 q.t over 256 0/1 bytes,
            # 32x32 tiles of v_mfma_i32_32x32x32_i8 (8 per tile); N1 = N2 = nfeatures is the algorithmic problem size
            ops = 2.0 * 256 * args.nfeatures * args.nfeatures * Bl
            tops = ops / (stage_ms["match"] * 1e-3) / 1e12
            rl["matcher"]["mfma"] = {"ops": ops, "achieved": tops, "peak": MFMA_I8_PEAK_TOPS, "unit": "TOP/s (int8, dense)", "frac": tops / MFMA_I8_PEAK_TOPS}
            if rl["matcher"]["mfma"]["frac"] > max(rl["matcher"]["frac"], rl["matcher"].get("valu", {}).get("frac", 0.0)):
                rl["matcher"]["bound"] = "mfma"
        else:
            # the L2 matcher: |a-b|^2 = |a'|^2 + |b'|^2 - 2 a'.b' with a' = a - 128 as int8, 4 x v_mfma_i32_32x32x32_i8 per 32 x 32
            # tile (K = 128), two crossCheck passes; the stage time includes the norms and the select kernels
            ops = 2 * 2.0 * 128 * args.nfeatures * args.nfeatures * Bl
            tops = ops / (stage_ms["match"] * 1e-3) / 1e12
            rl["matcher"]["mfma"] = {"ops": ops, "achieved": tops, "peak": MFMA_I8_PEAK_TOPS, "unit": "TOP/s (int8, dense)", "frac": tops / MFMA_I8_PEAK_TOPS}
        if not args.no_calibrate:
            try:
                rl["hbm_measured_gbs"] = eng.calibrate_hbm() / 1e9
            except _capi.RpeError:
                rl["hbm_measured_gbs"] = None
        out = {
            "metric": "image-pairs/s end-to-end (640x480 pairs), median rotation-angle error alongside",
            "value": world * B * args.steps / elapsed,
            "unit": "pairs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8/i32 (ORB, Hamming) + f64 (RANSAC, pose)" if method == "ORB" else "f32 (SIFT) + u8/i32 (exact L2) + f64 (RANSAC, pose)",
            "data": "synthetic",
            "config": {"workload": (f"{B} {W}x{H} pairs per GPU, ORB({args.nfeatures}kp)+BF-Hamming crossCheck top-{args.max_matches}"
                                    "+5pt-RANSAC(0.999,1px)+recoverPose (BASELINE configs[1])" +
                                    (f"; ONE consecutive-frame stream of {world * B + 1} frames cut into {world} frame ranges with a one-frame halo, "
                                     "features once per frame (configs[4] stand-in)" if args.stream else "")) if method == "ORB" else
                                   (f"{B} {W}x{H} pairs per GPU in sub-batches of {sub}, SIFT(cap {args.nfeatures})+BF-L2 crossCheck "
                                    f"top-{args.max_matches}+5pt-RANSAC+recoverPose (BASELINE configs[2]" + ("" if B >= 4096 else f" shape, {B} of its 4096 pairs") + ")"),
                       "pairs_per_gpu": B, "distinct_pairs": data["distinct"], "global_pairs": world * B, "streams_per_gpu": S, "pairs_per_launch": Bl,
                       "sharding": f"pairs x{world}, " + ("rpe_gather_poses: ncclAllGather (RCCL) of 128-B pose records, no torch" if comm_kind == "rccl" or world == 1
                                                         else f"torch.distributed all-gather of 128-B pose records: {comm_kind}")},
            "median_rotation_error_deg": float(np.median(errs)) if len(errs) else None,
            "median_translation_dir_error_deg": float(np.median(terrs)) if len(terrs) else None,
            "pairs_ok": int(ok.sum()),
            "stage_ms_per_launch": {k: round(v, 4) for k, v in stage_ms.items()},
            "roofline": rl,
        }
        if not args.no_cpu_baseline and world == 1:
            res, cb = cpu_baseline(i1, i2, K, args.nfeatures, args.max_matches, args.cpu_sample, method)
            n = len(res)
            cerr = [geometry.rotation_error(res["R"][i].reshape(3, 3), Rgt[i]) for i in range(n) if res["status"][i] == 0]
            cb["median_rotation_error_deg"] = float(np.median(cerr)) if cerr else None
            gerr = [geometry.rotation_error(R[i], Rgt[i]) for i in range(n) if ok[i]]
            cb["gpu_median_rotation_error_deg_same_sample"] = float(np.median(gerr)) if gerr else None
            same = all(int(res["status"][i]) == int(st[i]) and int(res["inliers"][i]) == int(inl[i]) and
                       np.linalg.norm(res["R"][i].reshape(3, 3) - R[i]) <= 1e-4 for i in range(n))
            cb["gpu_matches_cpu_on_sample"] = bool(same)
            out["cpu_baseline"] = cb

    if comm is not None:
        comm.barrier()
        comm.close()
    for e in engs:
        e.close()
    return out


def dropin_operating_points(data, device, workers):
    """The drop-in's own operating points, driver-timed in the same run (rank 0, N = 1):
    single_pair: PoseEstimator.estimate(img1, img2) -- host images in, (R, t) out -- at the reference's defaults (ORB 4000,
      top-500: src/run_single_pair.py:49-57), 640x480 and 1920x1080, median of 30 calls;
    host_entry: rpe_estimate_batch on the 1024 headline pairs held in HOST memory (the boundary the reference's callers see,
      src/utils/image_loader.py:23-28): pageable numpy arrays and page-locked ones (Engine.pinned_empty), uploads pipelined
      behind the kernels in four chunks; the device-resident rate is the headline `value`."""
    import statistics
    from relative_pose_estimation_amd import PoseEstimator, _capi, geometry, synthetic
    out = {"single_pair": {}, "host_entry": {}}
    for (W, H) in ((640, 480), (1920, 1080)):
        K = geometry.default_camera_matrix(W, H)
        i1, i2, _, _ = synthetic.make_batch(1, K, W, H, cfg=5, workers=1)
        pe = PoseEstimator(K, device=device)                      # reference defaults: ORB, Hamming, nfeatures 4000, max_matches 500
        for _ in range(3):
            pe.estimate(i1[0], i2[0])
        ts = []
        for _ in range(30):
            t0 = time.perf_counter(); pe.estimate(i1[0], i2[0]); ts.append(time.perf_counter() - t0)
        eng = pe._engine(H, W, 1)
        eng.set_profiling(True); pe.estimate(i1[0], i2[0])
        out["single_pair"][f"{W}x{H}"] = {"ms_per_estimate_call": round(statistics.median(ts) * 1e3, 3), "min_ms": round(min(ts) * 1e3, 3),
                                          "config": "PoseEstimator(K).estimate(img1, img2): ORB 4000, Hamming crossCheck, top-500, host images in, (R, t) out",
                                          "stage_ms": {k: round(v, 3) for k, v in eng.stage_ms().items() if v > 0.0005}}
        pe.close()
        if W == 1920:
            # feature_method="SIFT" as the reference builds it: SIFT_create() without a cap (pose_estimator.py:93-94); this
            # textured frame holds ~12.7k keypoints per image
            pe = PoseEstimator(K, feature_method="SIFT", norm_type="L2", device=device)
            for _ in range(2):
                d = pe.estimate_with_debug(i1[0], i2[0])
            ts = []
            for _ in range(10):
                t0 = time.perf_counter(); pe.estimate(i1[0], i2[0]); ts.append(time.perf_counter() - t0)
            eng = pe._engine(H, W, 1)
            eng.set_profiling(True); pe.estimate(i1[0], i2[0])
            out["single_pair"][f"{W}x{H}_sift"] = {"ms_per_estimate_call": round(statistics.median(ts) * 1e3, 3), "min_ms": round(min(ts) * 1e3, 3),
                                                   "config": "PoseEstimator(K, 'SIFT', 'L2').estimate(img1, img2): SIFT without a cap, L2 crossCheck, top-500",
                                                   "keypoints": [int(c) for c in eng.sift_detect_and_compute(np.stack([i1[0], i2[0]]))[2]], "matches": int(d["num_matches"]),
                                                   "overflow_flags": int(pe.last_overflow()[0]),
                                                   "stage_ms": {k: round(v, 3) for k, v in eng.stage_ms().items() if v > 0.0005}}
            pe.close()
    K, i1, i2 = data["K"], data["i1"], data["i2"]
    B, H, W = i1.shape
    e = _capi.Engine(W, H, max_batch=B, nfeatures=1000, max_matches=500, device=device)
    p1 = e.pinned_empty(i1.shape); p2 = e.pinned_empty(i2.shape)
    p1[...] = i1; p2[...] = i2
    nbytes = 2 * i1.nbytes
    for name, (a, b) in (("pageable", (i1, i2)), ("pinned", (p1, p2))):
        for _ in range(2):
            e.estimate_batch(a, b, K)
        t0 = time.perf_counter()
        for _ in range(5):
            e.estimate_batch(a, b, K)
        dt = (time.perf_counter() - t0) / 5
        out["host_entry"][name] = {"pairs_per_s": round(B / dt, 1), "ms_per_batch": round(dt * 1e3, 3), "host_bytes_per_batch": int(nbytes),
                                   "h2d_gbs_if_serial": round(nbytes / dt / 1e9, 2)}
    # the upload alone, for the H2D share
    d = e.device_malloc(i1.nbytes)
    for name, a in (("pageable", i1), ("pinned", p1)):
        e.lib.rpe_memcpy_h2d(e.h, d, a.ctypes.data_as(__import__("ctypes").c_void_p), a.nbytes)
        t0 = time.perf_counter()
        for _ in range(3):
            e.lib.rpe_memcpy_h2d(e.h, d, a.ctypes.data_as(__import__("ctypes").c_void_p), a.nbytes)
        dt = (time.perf_counter() - t0) / 3
        out["host_entry"][name]["h2d_alone_ms_per_batch"] = round(2 * dt * 1e3, 3)
        out["host_entry"][name]["h2d_alone_gbs"] = round(a.nbytes / dt / 1e9, 2)
    e.device_free(d)
    e.close()
    out["host_entry"]["note"] = "1024 VGA pairs = 629 MB per batch; PCIe Gen5 x16 ~ 63 GB/s = 10 ms = a ceiling of ~100 k pairs/s for this entry"
    out["reference_rows"] = reference_rows_on_gpu(device)
    return out


def reference_rows_on_gpu(device):
    """BASELINE's accuracy half on the reference's OWN inputs: every committed result row of the reference (147 image pairs of its
    three evaluation runs, tests/golden/reference_rows/: the frames, ground truth, camera matrices and the CSV's est_* columns)
    through PoseEstimator.estimate_batch with the reference's parameters (ORB 4000, top-500; pipeline.py:94-101).  Reported per
    run: rows whose forward Euler triple (batch_processor.py:82-101) agrees with the CSV within 1e-6 / 1e-3 / 0.01 / 0.1 / 0.5 deg,
    and the median rotation error beside the reference's.  keypoint_order = the C++ runtime of the cv2 build that produced
    the file.  Reads fixtures only; the oracle is not involved."""
    from relative_pose_estimation_amd import PoseEstimator, geometry
    from tests import reference_rows as rr
    res = {}
    for name, order in (("sim", "msvc"), ("salah", "libstdc++"), ("phone", "libstdc++")):
        ds = rr.load(name)
        B = len(ds["frames2"])
        pe = PoseEstimator(ds["K"], nfeatures=4000, max_matches=500, max_batch=B, device=device, keypoint_order=order)
        t0 = time.perf_counter()
        R, t, inl, st = pe.estimate_batch(ds["img1"], ds["img2"])
        dt = time.perf_counter() - t0
        pe.close()
        diff = rr.euler_agreement(ds, R, geometry)
        err = rr.rotation_errors(ds, R, geometry)
        ref = ds["ref_rotation_error"]
        res[name] = {"pairs": B, "image": f"{ds['img1'].shape[2]}x{ds['img1'].shape[1]}", "keypoint_order": order, "status_ok": int((st == 0).sum()),
                     "rows_agreeing_within_deg": {f"{e:g}": c for e, c in zip(rr.AGREE_EDGES, rr.agreement_counts(diff))},
                     "median_rotation_error_deg": round(float(np.median(err)), 4), "reference_median_rotation_error_deg": round(float(np.median(ref)), 4),
                     "flips_over_90_deg": int((err > 90).sum()), "reference_flips_over_90_deg": int((ref > 90).sum()),
                     "first_call_s_incl_workspace": round(dt, 3)}
    res["total_rows_within_1e-6_deg"] = sum(v["rows_agreeing_within_deg"]["1e-06"] for v in res.values() if isinstance(v, dict))
    return res


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves, exactly as the driver would
    (python -m torch.distributed.run, one process per GPU, rendezvous on 127.0.0.1), as a CHILD process -- nothing in this
    process has touched the GPU yet, and nothing will -- and hand its exit code on.  A 1-GPU line for an N-GPU request
    is never printed."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.stderr.write(f"[bench] --gpus {args.gpus} without WORLD_SIZE in the environment: launching {' '.join(cmd[1:8])} ...\n")
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus} (or without a launcher)")
    method, sub = resolve(args)
    cores = os.cpu_count() or 1
    workers = args.gen_workers or max(1, min(16, cores // max(1, world)))

    # ---- all synthetic data first (the generator forks): headline + the two extra workloads of the default run
    data = generate(args, method, rank, world, workers)
    want_extra = (world == 1 and not args.no_extra and args.config == 2 and not args.stream and args.streams == 1 and
                  args.batch == 1024 and (args.width, args.height, args.nfeatures) == (640, 480, 1000))
    extras = []
    if want_extra:
        a3 = argparse.Namespace(**vars(args))
        a3.config, a3.steps, a3.warmup, a3.unique, a3.cpu_sample, a3.data_cache = 3, 3, 1, 128, 16, ""      # 128 distinct rendered pairs (about a minute of the run on 16 cores)
        m3, s3 = resolve(a3)
        d3 = generate(a3, m3, 0, 1, workers)
        extras.append(("config3_128pairs", a3, m3, s3, d3))
        # the same 128 HD pairs as three launch groups on three handles / HIP streams (43 + 43 + 42 pairs): the large SIFT
        # kernels of one group overlap those of another where they are bound by different resources (about -6 %)
        a3s = argparse.Namespace(**vars(a3))
        a3s.streams, a3s.no_cpu_baseline, a3s.no_calibrate = 3, True, True
        extras.append(("config3_three_streams", a3s, m3, s3, d3))
        ast = argparse.Namespace(**vars(args))
        ast.stream, ast.cpu_sample, ast.data_cache = True, 256, ""
        extras.append(("stream", ast, "ORB", ast.batch, generate(ast, "ORB", 0, 1, workers)))
        # the headline batch again on TWO handles (two HIP streams, half the batch each): the latency-bound RANSAC / pose
        # kernels of one half overlap the issue-bound ORB kernels of the other.  Not the headline because per-kernel
        # durations of overlapping streams are not clean launch times.
        a2 = argparse.Namespace(**vars(args))
        a2.streams, a2.no_cpu_baseline, a2.no_calibrate = 2, True, True
        extras.append(("two_streams", a2, "ORB", a2.batch, data))

    from relative_pose_estimation_amd import _capi
    ndev = _capi.load().rpe_device_count()
    device = local_rank % max(ndev, 1)
    out = run_workload(args, method, sub, data, rank, world, device, args.dist_backend)
    if rank == 0 and extras:
        out["extra"] = {}
        for name, a, m, s, d in extras:
            o = run_workload(a, m, s, d, 0, 1, device, "rccl")
            out["extra"][name] = {k: o[k] for k in ("value", "unit", "steps", "ms_per_step", "config", "median_rotation_error_deg", "pairs_ok",
                                                    "stage_ms_per_launch", "roofline", "cpu_baseline") if k in o}
    if rank == 0 and want_extra:
        try:
            out["extra"].update(dropin_operating_points(data, device, workers))
        except Exception as exc:                   # never lose the headline line to an extra
            out["extra"]["dropin_operating_points_error"] = repr(exc)
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Benchmark of the MI355X relative-pose hot path (BASELINE.json metric:
image-pairs/s end-to-end + median rotation-angle error, 640x480 pairs).

A step = one pass of feature-extract -> match -> essential RANSAC -> pose over one
batch of synthetic pairs whose images are already resident in HBM; the step ends when
the pose records are in host memory.  Workload (config.workload) = BASELINE configs[1]:
1024 VGA pairs per GPU, ORB(1000 kp) + BF-Hamming crossCheck + 5-pt RANSAC + recoverPose.
Weak scaling: every rank processes its own 1024-pair shard; the only collective is the
RCCL all-gather of 128-byte pose records at the end of each step.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
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


def orb_levels(W, H, nlevels=12, scale=1.1):
    """ORB pyramid level sizes (orb.cpp: cvRound(size / scale^l)); sum(w*h) must equal rpe_orb_pyramid_pixels()."""
    out = []
    for l in range(nlevels):
        s = scale ** l
        out.append((int(np.rint(W / s)), int(np.rint(H / s))))
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
    per_image = {
        "pyramid": img + pyr_px,                 # read level 0, write the 12-level pyramid
        # FAST+NMS run on the border-filtered region only: ring reads reach 4 px past the NMS output [31, w-31)
        "fast": sum((w - 54) * (h - 54) + (w - 62) * (h - 62) for w, h in live) + 12 * 256 * 4,
        "nms": 0,                                # fused into fast
        "select": sum(w * (h - 62) for w, h in live),   # linear scan of the NMS map rows that can hold keypoints
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
                "select": "select_candidates_kernel", "harris": "harris_kernel", "keypoints": "select_keypoints_kernel",
                "match": "match_hamming_kernel", "ransac": "ransac_*_kernel (group)", "pose": "recover_pose_kernel"}
# SIFT reuses the stage slots (csrc/sift_kernels.hip rpe_sift_run)
SIFT_STAGE_KERNEL = {"pyramid": "sift_upsample + sift_blur_fused<R> x16/octave-set (group)", "fast": "sift_extrema_mask_kernel (+scan, emit)",
                     "select": "sift_adjust_kernel", "harris": "sift_orient_kernel", "keypoints": "sift_prefilter/sort/finalize (group)",
                     "describe": "sift_describe_kernel", "match": "match_l2_nearest_kernel (+select)",
                     "ransac": "ransac_*_kernel (group)", "pose": "recover_pose_kernel"}


def pmc_traffic(stage, pairs_per_launch, W, H, nfeatures):
    """HBM bytes per launch of the stage's kernel from the committed rocprofv3 PMC passes
    (profiles/r01_traffic.json: FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, separate passes),
    or None when no profile exists for this kernel / workload."""
    path = os.path.join(ROOT, "profiles", "r01_traffic.json")
    if stage not in STAGE_KERNEL or not os.path.exists(path) or (W, H, nfeatures, pairs_per_launch) != (640, 480, 1000, 1024):
        return None
    k = json.load(open(path))["kernels"].get(STAGE_KERNEL[stage])
    return k["hbm_bytes_per_launch"] if k else None


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
    return res, {"value": n / dt, "unit": "pairs/s", "cores": threads, "kind": "port",
                 "sample": f"first {n} pairs of the same synthetic batch ({i1.shape[2]}x{i1.shape[1]}, {method} {nfeatures}), "
                           f"{threads} pthreads over pairs, {dt:.1f} s wall"}


def main():
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
    ap.add_argument("--gen-workers", type=int, default=0)
    ap.add_argument("--config", type=int, default=2, choices=[2, 3],
                    help="BASELINE configs[]: 2 = 1024 VGA pairs ORB(1000)+Hamming (the metric's config, default); "
                         "3 = 1920x1080 pairs SIFT(2048)+L2, processed in sub-batches")
    ap.add_argument("--stream", action="store_true",
                    help="BASELINE configs[4] stand-in: consecutive-frame stream (batch+1 frames -> batch pairs), features once per frame")
    ap.add_argument("--sub-batch", type=int, default=0, help="pairs per enqueue (config 3 default 32: 1.1 GB of pyramid per pair)")
    ap.add_argument("--streams", type=int, default=1, help="split the batch over S engine handles (S HIP streams) so latency-bound stages overlap")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, default) or gloo (single-GPU rehearsal of the N>1 path)")
    ap.add_argument("--data-cache", default="", help="npz file to load/save the synthetic batch (keeps forks out of profiled runs)")
    ap.add_argument("--unique", type=int, default=0, help="generate only this many distinct pairs and tile them to --batch "
                    "(full-size config 3 runs: rendering 4096 HD pairs takes longer than measuring them)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    from relative_pose_estimation_amd import _capi, synthetic, geometry, sharding

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
    W, H, B = args.width, args.height, args.batch
    sub = args.sub_batch or (32 if method == "SIFT" else B)
    K = geometry.default_camera_matrix(W, H)
    cores = os.cpu_count() or 1
    workers = args.gen_workers or max(1, min(16, cores // max(1, world)))
    # global pair index space: rank r owns pairs [r*B, (r+1)*B) (weak scaling)
    if args.stream:
        frames, Rgt, tgt = synthetic.make_stream(B + 1, K, W, H, seed=5_000_011 + rank, workers=workers)
        i1, i2 = frames[:-1], frames[1:]
    cache = f"{args.data_cache}.r{rank}.npz" if args.data_cache and not args.stream else ""
    if cache and os.path.exists(cache):
        z = np.load(cache)
        i1, i2, Rgt, tgt = z["i1"], z["i2"], z["R"], z["t"]
        assert i1.shape == (B, H, W), "data cache does not match the requested workload"
    elif not args.stream:
        U = min(args.unique, B) if args.unique > 0 else B
        i1, i2, Rgt, tgt = synthetic.make_batch(U, K, W, H, cfg=2, first=rank * B, workers=workers)
        if U < B:
            reps = -(-B // U)
            i1, i2 = np.concatenate([i1] * reps)[:B], np.concatenate([i2] * reps)[:B]
            Rgt, tgt = np.concatenate([Rgt] * reps)[:B], np.concatenate([tgt] * reps)[:B]
        if cache:
            np.savez(cache, i1=i1, i2=i2, R=Rgt, t=tgt)

    # the process group (and with it the HIP runtime) comes up only AFTER the synthetic batch exists:
    # the generator forks worker processes, which must not inherit an initialised GPU context
    dist = None
    torch = None
    if world > 1:
        import torch
        import torch.distributed as dist
        if args.dist_backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.dist_backend)

    ndev = _capi.load().rpe_device_count()
    device = local_rank % max(ndev, 1)
    gather_dev = None
    if dist is not None and args.dist_backend == "nccl":
        gather_dev = torch.device("cuda", local_rank)
    S = max(1, args.streams)
    bounds = [sharding.shard_bounds(B, s, S) for s in range(S)]
    fm, nt = (_capi.FEATURE_SIFT, _capi.NORM_L2) if method == "SIFT" else (_capi.FEATURE_ORB, _capi.NORM_HAMMING)
    engs = [_capi.Engine(W, H, max_batch=min(sub, hi - lo), nfeatures=args.nfeatures, max_matches=args.max_matches, device=device,
                         feature_method=fm, norm_type=nt)
            for lo, hi in bounds]
    eng = engs[0]
    if args.stream:
        assert S == 1 and sub >= B, "--stream uses one handle and one launch group"
        d_frames = engs[0].upload(frames)
        dbuf = [(d_frames, d_frames)]
    else:
        dbuf = [(e.upload(i1[lo:hi]), e.upload(i2[lo:hi])) for e, (lo, hi) in zip(engs, bounds)]   # inputs resident in HBM
    for e in engs:
        e.set_profiling(True)

    def barrier():
        for e in engs:
            e.synchronize()
        if dist is not None:
            dist.barrier()
            if gather_dev is not None:
                torch.cuda.synchronize()

    sub_acc = {}

    def step():
        if args.stream:
            engs[0].enqueue_stream_device(dbuf[0][0], B + 1, K)
            parts = [engs[0].fetch_results(B)]
        elif sub >= B:
            for e, (a, b), (lo, hi) in zip(engs, dbuf, bounds):
                e.enqueue_batch_device(a, b, hi - lo, K)
            parts = [e.fetch_results(hi - lo) for e, (lo, hi) in zip(engs, bounds)]
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
        R, t, inl, nm, st = (np.concatenate([p[k] for p in parts]) for k in range(5))
        rec = sharding.pack_records(R, t, inl, st, nm, first_pair=rank * B)
        if dist is not None:
            rec = sharding.gather_pose_records(rec, B, device=gather_dev)
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
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=gather_dev if gather_dev is not None else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    R, t, inl, nm, st = local
    ok = st == 0
    errs = np.array([geometry.rotation_error(R[i], Rgt[i]) for i in range(B) if ok[i]])
    # translation is recovered up to scale: direction error against the ground-truth direction (pose_evaluator.py:111-116)
    terrs = np.array([geometry.translation_direction_error(t[i], tgt[i]) for i in range(B) if ok[i] and np.linalg.norm(tgt[i]) > 0])
    # per-LAUNCH averages: every step launches each kernel group once per stream on B/S pairs
    stage_ms = {k: v / (args.steps * S) for k, v in stage_acc.items()}
    Bl = B // S if B % S == 0 else B / S          # pairs per launch
    if sub < B:
        Bl = sub
        launches = (args.steps + args.warmup) * (-(-B // sub))
        stage_ms = {k: v / launches for k, v in sub_acc.items()}
    pyr_px = eng.lib.rpe_orb_pyramid_pixels(eng.h)

    if rank == 0:
        dom = max(stage_ms, key=stage_ms.get)
        names = SIFT_STAGE_KERNEL if method == "SIFT" else STAGE_KERNEL
        dom_bytes = stage_bytes(dom, W, H, pyr_px, args.nfeatures, args.max_matches, method) * Bl
        achieved = dom_bytes / (stage_ms[dom] * 1e-3) / 1e9
        # the matcher is the stage north_star attaches a roofline target to: always report it too
        m_bytes = stage_bytes("match", W, H, pyr_px, args.nfeatures, args.max_matches, method) * Bl
        m_achieved = m_bytes / (stage_ms["match"] * 1e-3) / 1e9
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
                                    (f"; consecutive-frame stream of {B + 1} frames, features once per frame (configs[4] stand-in)" if args.stream else "")) if method == "ORB" else
                                   (f"{B} {W}x{H} pairs per GPU in sub-batches of {sub}, SIFT(cap {args.nfeatures})+BF-L2 crossCheck "
                                    f"top-{args.max_matches}+5pt-RANSAC+recoverPose (BASELINE configs[2]" + ("" if B >= 4096 else f" shape, {B} of its 4096 pairs") + ")"),
                       "pairs_per_gpu": B, "distinct_pairs": (min(args.unique, B) if args.unique > 0 else B), "global_pairs": world * B, "streams_per_gpu": S, "pairs_per_launch": Bl, "sharding": f"pairs x{world}, RCCL all-gather of 128-B pose records"},
            "median_rotation_error_deg": float(np.median(errs)) if len(errs) else None,
            "median_translation_dir_error_deg": float(np.median(terrs)) if len(terrs) else None,
            "pairs_ok": int(ok.sum()),
            "stage_ms_per_launch": {k: round(v, 4) for k, v in stage_ms.items()},
            "roofline": {"kernel": names.get(dom, dom), "stage": dom, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(dom, Bl, W, H, args.nfeatures),
                         "algorithmic_bytes_per_launch": dom_bytes, "avg_launch_ms": stage_ms[dom],
                         "matcher": {"achieved": m_achieved, "frac": m_achieved / HBM_PEAK_GBS,
                                     "algorithmic_bytes_per_launch": m_bytes, "avg_launch_ms": stage_ms["match"]}},
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
        print(json.dumps(out), flush=True)

    for e in engs:
        e.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

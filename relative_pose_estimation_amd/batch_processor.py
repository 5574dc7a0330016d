"""Sequence front-end: the immediate caller of the hot path.

Same contract as reference src/core/batch_processor.py: BatchProcessor(images_dir,
pose_estimator, ground_truth_loader, euler_convention).process_sequence(frame_indices) -> dict
with 'frames', 'roll', 'pitch', 'yaw', 'R', 't' (:38-116); world rotation of the second frame
R_new = R_prev_GT @ R_rel (:94-97, R_prev always from ground truth, :82-89), Euler columns through
rotation_to_euler; process_at_interval (:118-129); a failing pair raises the estimator's
RuntimeError (:60-61).

MI355X-first differences: the frames of the list are decoded once, converted to gray in HBM and run
as ONE consecutive-frame stream (rpe_enqueue_stream_device): features are extracted once per frame
(the reference loads and extracts every interior frame twice, :79,:92) and all pairs go through
the matcher / RANSAC / pose kernels in a single launch group.
"""
from pathlib import Path

import numpy as np

from . import _capi, image_loader
from .geometry import euler_to_rotation, rotation_to_euler

CONVENTION_YUP = "yup"


class BatchProcessor:
    def __init__(self, images_dir, pose_estimator, ground_truth_loader, euler_convention=CONVENTION_YUP):
        self.images_dir = Path(images_dir) if images_dir is not None else None
        self.pose_estimator = pose_estimator
        self.gt_loader = ground_truth_loader
        self.euler_convention = euler_convention

    def get_image_path(self, frame_idx):
        return self.images_dir / f"{frame_idx:06d}.png"

    # ---- stream over already decoded frames (gray uint8 [F,H,W] or colour [F,H,W,3])
    def process_frames(self, frame_indices, frames, order=_capi.ORDER_RGB):
        frame_indices = [int(f) for f in frame_indices]
        if len(frame_indices) < 2:
            raise ValueError("Need at least 2 frames to process")
        frames = np.asarray(frames)
        est = self.pose_estimator
        F, H, W = frames.shape[:3]
        if F != len(frame_indices):
            raise ValueError("one frame per frame index expected")
        eng = est._engine(H, W, F - 1)
        want_vp = bool(getattr(est, "use_vp_refinement", False))
        gray = None
        if frames.ndim == 4 and not want_vp:                   # colour: gray conversion in HBM, no host round trip
            d_gray = eng.upload_bgr_as_gray(frames, order=order)
        else:
            # the VP post-step (host code, as in the reference) needs the gray frames on the host as well
            gray = eng.bgr_to_gray(frames, order=order) if frames.ndim == 4 else np.ascontiguousarray(frames, np.uint8)
            d_gray = eng.upload(gray)
        try:
            eng.enqueue_stream_device(d_gray, F, est.K)
            R_rel, t_rel, inl, nm, st = eng.fetch_results(F - 1)
        finally:
            eng.device_free(d_gray)
        out = {"frames": [], "roll": [], "pitch": [], "yaw": [], "R": [], "t": []}
        for i in range(F - 1):
            est._raise_for(int(st[i]), int(nm[i]))
            gt = self.gt_loader.get_pose(frame_indices[i])
            R_prev_world = euler_to_rotation(gt["yaw"], gt["pitch"], gt["roll"], convention=self.euler_convention)
            R_i = R_rel[i]
            if want_vp:                                        # estimate(img1, img2, R_prev=R_prev_world) (:92)
                R_i, _, _ = est._vp_refine(R_i, R_prev_world, gray[i], gray[i + 1])
            R_new_world = R_prev_world @ R_i                    # camera1 -> camera2 composed on the right (:97)
            yaw, pitch, roll = rotation_to_euler(R_new_world, convention=self.euler_convention)
            out["frames"].append(frame_indices[i + 1])
            out["roll"].append(roll); out["pitch"].append(pitch); out["yaw"].append(yaw)
            out["R"].append(R_new_world); out["t"].append(t_rel[i])
        for k in ("roll", "pitch", "yaw"):
            out[k] = np.array(out[k])
        out["inliers"] = inl
        return out

    def process_sequence(self, frame_indices):
        frame_indices = [int(f) for f in frame_indices]
        if len(frame_indices) < 2:
            raise ValueError("Need at least 2 frames to process")
        rgb = np.stack([image_loader.decode_rgb(str(self.get_image_path(f))) for f in frame_indices])
        return self.process_frames(frame_indices, rgb, order=_capi.ORDER_RGB)

    def process_at_interval(self, step=15):
        return self.process_sequence(self.gt_loader.get_frame_indices(step=step))

"""Accuracy metrics over a processed sequence: what turns the hot path's output
into BASELINE's second metric (median rotation-angle error).

Same contract as reference src/core/pose_evaluator.py: evaluate_sequence (:32-135) consumes the
dict of BatchProcessor.process_sequence and returns per-frame roll/pitch/yaw errors (absolute,
wrapped to [0, 180], :183-195), rotation_error (geometry.py:128-149), translation direction error
against the ground-truth position delta (0.0 for the first evaluated frame, :111-119);
compute_summary_statistics (:137-157) gives mean/std/median/max/min per metric;
create_comparison_dataframe (:159-181) keeps the reference's CSV column order.
"""
import numpy as np

from .geometry import euler_to_rotation, rotation_error, translation_direction_error

CONVENTION_YUP = "yup"
METRICS = ("roll_error", "pitch_error", "yaw_error", "rotation_error", "translation_dir_error")
CSV_COLUMNS = ("frame", "gt_roll", "gt_pitch", "gt_yaw", "est_roll", "est_pitch", "est_yaw",
               "roll_error", "pitch_error", "yaw_error", "rotation_error", "translation_dir_error")


class PoseEvaluator:
    def __init__(self, ground_truth_loader, euler_convention=CONVENTION_YUP):
        self.gt_loader = ground_truth_loader
        self.euler_convention = euler_convention

    @staticmethod
    def _wrap_angle_error(error_deg):
        return np.abs((np.asarray(error_deg, dtype=np.float64) + 180.0) % 360.0 - 180.0)

    def evaluate_sequence(self, estimated_results):
        frames = estimated_results["frames"]
        poses = [self.gt_loader.get_pose(f) for f in frames]
        gt = {k: np.array([p[k] for p in poses], dtype=np.float64) for k in ("roll", "pitch", "yaw")}
        pos = np.array([[p["x"], p["y"], p["z"]] for p in poses], dtype=np.float64).reshape(len(poses), 3)
        res = {"frames": frames}
        for k in ("roll", "pitch", "yaw"):
            est = np.asarray(estimated_results[k], dtype=np.float64)
            res[f"{k}_error"] = self._wrap_angle_error(np.abs(est - gt[k]))
        res["rotation_error"] = np.array([
            rotation_error(R, euler_to_rotation(p["yaw"], p["pitch"], p["roll"], convention=self.euler_convention))
            for R, p in zip(estimated_results["R"], poses)])
        tde = np.zeros(len(poses))
        for i in range(1, len(poses)):
            tde[i] = translation_direction_error(estimated_results["t"][i], pos[i] - pos[i - 1])
        res["translation_dir_error"] = tde
        for k in ("roll", "pitch", "yaw"):
            res[f"gt_{k}"] = gt[k]
            res[f"est_{k}"] = estimated_results[k]
        return res

    def compute_summary_statistics(self, evaluation_results):
        stats = {}
        for metric in METRICS:
            e = np.asarray(evaluation_results[metric], dtype=np.float64)
            for name, fn in (("mean", np.mean), ("std", np.std), ("median", np.median), ("max", np.max), ("min", np.min)):
                stats[f"{metric}_{name}"] = fn(e)
        return stats

    def create_comparison_dataframe(self, evaluation_results):
        import pandas as pd
        src = dict(evaluation_results)
        src["frame"] = src["frames"]
        return pd.DataFrame({c: src[c] for c in CSV_COLUMNS})

    def print_summary(self, evaluation_results):
        s = self.compute_summary_statistics(evaluation_results)
        print(f"frames evaluated: {len(evaluation_results['frames'])}")
        for metric in METRICS:
            print(f"  {metric:24s} mean {s[metric + '_mean']:.2f}  std {s[metric + '_std']:.2f}  "
                  f"median {s[metric + '_median']:.2f}  max {s[metric + '_max']:.2f}  min {s[metric + '_min']:.2f}")

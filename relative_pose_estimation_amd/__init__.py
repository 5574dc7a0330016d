"""MI355X-native relative-pose engine: drop-in for the hot path of
ofekm5/relative-pose-estimation (src/core/pose_estimator.py PoseEstimator.estimate)."""
from .pose_estimator import PoseEstimator, estimate_relative_pose  # noqa: F401
from .geometry import rotation_error, translation_direction_error  # noqa: F401

__all__ = ["PoseEstimator", "estimate_relative_pose", "rotation_error", "translation_direction_error"]

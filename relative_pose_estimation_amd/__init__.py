"""MI355X-native relative-pose engine: drop-in for the hot path of
ofekm5/relative-pose-estimation (src/core/pose_estimator.py PoseEstimator.estimate) and the callers /
data formats either side of it (image ingest, sequence front-end, evaluator: SURVEY 8(f))."""
from .pose_estimator import PoseEstimator, estimate_relative_pose  # noqa: F401
from .geometry import rotation_error, translation_direction_error  # noqa: F401
from .batch_processor import BatchProcessor  # noqa: F401
from .pose_evaluator import PoseEvaluator  # noqa: F401
from .ground_truth_loader import GroundTruthLoader  # noqa: F401

__all__ = ["PoseEstimator", "estimate_relative_pose", "rotation_error", "translation_direction_error",
           "BatchProcessor", "PoseEvaluator", "GroundTruthLoader"]

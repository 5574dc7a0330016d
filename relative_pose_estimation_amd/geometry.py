"""Accuracy metrics of the reference's evaluator, restated for the benchmark harness.

rotation_error follows reference src/utils/geometry.py:128-149 (geodesic angle of
R_est @ R_gt.T, degrees); translation_direction_error follows :152-174;
default_camera_matrix follows src/core/camera_calibration.py:17-25,65-87.
"""
import numpy as np


def rotation_error(R_est, R_gt):
    R_diff = np.asarray(R_est) @ np.asarray(R_gt).T
    cos_angle = np.clip((np.trace(R_diff) - 1) / 2, -1.0, 1.0)
    return float(np.rad2deg(np.arccos(cos_angle)))


def translation_direction_error(t_est, t_gt):
    a = np.asarray(t_est, float).flatten(); b = np.asarray(t_gt, float).flatten()
    a = a / np.linalg.norm(a); b = b / np.linalg.norm(b)
    return float(np.rad2deg(np.arccos(np.clip(np.dot(a, b), -1.0, 1.0))))


def default_camera_matrix(width, height):
    """CameraCalibration().get_matrix(width, height): base intrinsics scaled to the image size."""
    sx, sy = width / 960, height / 720
    return np.array([[924.82939686 * sx, 0, 468.24930789 * sx],
                     [0, 920.4766382 * sy, 353.65863024 * sy],
                     [0, 0, 1]], dtype=np.float64)

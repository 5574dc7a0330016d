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


# ---- Euler conventions of the reference's evaluator (harness only) -------------
# follows src/utils/geometry.py:48-125 ('yup': R = Ry(yaw) Rx(pitch) Rz(roll); NB the
# reference's yup pair is not self-inverse, use forward only) and :188-237 ('zyx').
def euler_to_rotation(yaw_deg, pitch_deg, roll_deg, convention="yup"):
    y, p, r = np.deg2rad(yaw_deg), np.deg2rad(pitch_deg), np.deg2rad(roll_deg)
    cy, sy, cp, sp, cr, sr = np.cos(y), np.sin(y), np.cos(p), np.sin(p), np.cos(r), np.sin(r)
    if convention == "zyx":
        return np.array([[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
                         [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
                         [-sp, cp * sr, cp * cr]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rx = np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]])
    Rz = np.array([[cr, -sr, 0], [sr, cr, 0], [0, 0, 1]])
    return Ry @ Rx @ Rz


def rotation_to_euler(R, convention="yup"):
    """(yaw, pitch, roll) in degrees."""
    R = np.asarray(R, float)
    if convention == "zyx":
        sy = np.sqrt(R[0, 0] ** 2 + R[1, 0] ** 2)
        if sy >= 1e-6:
            roll = np.arctan2(R[2, 1], R[2, 2]); pitch = np.arctan2(-R[2, 0], sy); yaw = np.arctan2(R[1, 0], R[0, 0])
        else:
            roll = np.arctan2(-R[1, 2], R[1, 1]); pitch = np.arctan2(-R[2, 0], sy); yaw = 0.0
        return float(np.rad2deg(yaw)), float(np.rad2deg(pitch)), float(np.rad2deg(roll))
    pitch = np.arcsin(R[2, 1])
    if abs(R[2, 1]) > 0.9999:
        roll = np.arctan2(-R[1, 2], R[1, 1]); yaw = 0.0
    else:
        yaw = np.arctan2(-R[2, 0], R[0, 0]); roll = np.arctan2(R[1, 0], R[1, 1])
    return float(np.rad2deg(yaw)), float(np.rad2deg(pitch)), float(np.rad2deg(roll))


def bgr_to_gray(rgb):
    """cv2.cvtColor(BGR2GRAY) fixed-point formula applied to an RGB uint8 array
    (image_loader.py:23-28 path): (B*3735 + G*19235 + R*9798 + 16384) >> 15."""
    a = np.asarray(rgb).astype(np.int64)
    return ((a[..., 2] * 3735 + a[..., 1] * 19235 + a[..., 0] * 9798 + 16384) >> 15).astype(np.uint8)

"""VP-refinement post-step: host code (LSD restatement in librpe_amd.so + numpy algebra),
tested on the CPU with synthetic Manhattan scenes whose lines, vanishing directions and rotations are known.
Parity with cv2's LSD is unpinned (no cv2 offline); these tests pin geometry, gates and the optimiser."""
import numpy as np
import pytest
from PIL import Image, ImageDraw


def _Ry(a):
    c, s = np.cos(a), np.sin(a); return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])


def _Rx(a):
    c, s = np.cos(a), np.sin(a); return np.array([[1, 0, 0], [0, c, -s], [0, s, c]])


def _Rz(a):
    c, s = np.cos(a), np.sin(a); return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])


def _angle(R1, R2):
    return np.rad2deg(np.arccos(np.clip((np.trace(R1 @ R2.T) - 1) / 2, -1, 1)))


def _segments_world():
    """axis-aligned wireframe: a room, windows on its walls, a floor grid"""
    segs = []
    X, Y, Z0, Z1 = 2.5, 1.6, 1.0, 9.0
    for x in (-X, X):
        for y in (-Y, Y):
            segs.append(((x, y, Z0), (x, y, Z1)))
    for z in (Z1,):
        for x in (-X, X):
            segs.append(((x, -Y, z), (x, Y, z)))
        for y in (-Y, Y):
            segs.append(((-X, y, z), (X, y, z)))
    for x in (-X, X):                                             # windows on the side walls
        for z in (2.5, 4.5, 6.5):
            for y0, y1 in ((-0.8, 0.4),):
                segs += [((x, y0, z), (x, y1, z)), ((x, y0, z + 1.2), (x, y1, z + 1.2)),
                         ((x, y0, z), (x, y0, z + 1.2)), ((x, y1, z), (x, y1, z + 1.2))]
    for xx in np.linspace(-1.5, 1.5, 4):                          # door / board on the back wall
        segs.append(((xx, -1.0, Z1), (xx, 0.9, Z1)))
    for yy in (-1.0, 0.9):
        segs.append(((-1.5, yy, Z1), (1.5, yy, Z1)))
    for yy in (Y, -Y):                                            # floor planks and ceiling beams
        for xx in np.linspace(-2.0, 2.0, 9):
            segs.append(((xx, yy, Z0 + 0.5), (xx, yy, Z1)))
        for zz in np.linspace(2.0, 8.0, 7):
            segs.append(((-X, yy, zz), (X, yy, zz)))
    for x in (-X, X):                                             # skirting / picture rails along the side walls
        for yy in (-1.2, 1.2):
            segs.append(((x, yy, Z0), (x, yy, Z1)))
    return segs


def _render(K, R_iw, size=(640, 480), C=(0.2, -0.1, 0.0)):
    im = Image.new("L", size, 200)
    d = ImageDraw.Draw(im)
    for a, b in _segments_world():
        pa, pb = R_iw @ (np.array(a, float) - C), R_iw @ (np.array(b, float) - C)
        if pa[2] < 0.2 and pb[2] < 0.2:
            continue
        if pa[2] < 0.2:
            pa = pb + (pa - pb) * (pb[2] - 0.2) / (pb[2] - pa[2])
        if pb[2] < 0.2:
            pb = pa + (pb - pa) * (pa[2] - 0.2) / (pa[2] - pb[2])
        ua, ub = K @ pa, K @ pb
        d.line([(ua[0] / ua[2], ua[1] / ua[2]), (ub[0] / ub[2], ub[1] / ub[2])], fill=40, width=3)
    return np.asarray(im, dtype=np.uint8)


@pytest.fixture(scope="module")
def K():
    from relative_pose_estimation_amd.geometry import default_camera_matrix
    return default_camera_matrix(640, 480)


def test_lsd_finds_drawn_segments():
    from relative_pose_estimation_amd import _capi
    im = Image.new("L", (320, 240), 210)
    d = ImageDraw.Draw(im)
    truth = [((30, 30), (280, 55)), ((40, 210), (200, 130)), ((290, 90), (300, 225))]          # non-crossing strokes
    for a, b in truth:
        d.line([a, b], fill=30, width=5)
    L = _capi.lsd_detect(np.asarray(im, dtype=np.uint8))
    assert 6 <= len(L) <= 40                                   # two long edges per stroke (+ short caps)
    ang = np.arctan2(L[:, 3] - L[:, 1], L[:, 2] - L[:, 0])
    ln = np.hypot(L[:, 2] - L[:, 0], L[:, 3] - L[:, 1])
    for a, b in truth:
        ta = np.arctan2(b[1] - a[1], b[0] - a[0]); tl = np.hypot(b[0] - a[0], b[1] - a[1])
        mid = np.array([(a[0] + b[0]) / 2, (a[1] + b[1]) / 2])
        dm = np.hypot((L[:, 0] + L[:, 2]) / 2 - mid[0], (L[:, 1] + L[:, 3]) / 2 - mid[1])
        da = np.abs((ang - ta + np.pi / 2) % np.pi - np.pi / 2)
        hit = (dm < 6) & (da < np.deg2rad(1.5)) & (np.abs(ln - tl) < 12)
        assert hit.sum() >= 2, (a, b)
    flat = np.full((120, 160), 77, np.uint8)
    assert len(_capi.lsd_detect(flat)) == 0
    assert np.array_equal(_capi.lsd_detect(np.asarray(im, dtype=np.uint8)), L)     # deterministic


def test_manhattan_directions_of_a_rendered_room(K):
    from relative_pose_estimation_amd import vp_refinement as vp
    R_iw = _Rz(np.deg2rad(3)) @ _Rx(np.deg2rad(-8)) @ _Ry(np.deg2rad(17))
    img = _render(K, R_iw)
    Delta, ok, dbg = vp.estimate_manhattan_dirs(img, K, rng_seed=0)
    assert ok and dbg["num_lines"] >= 40 and dbg["lines_used"] <= 120 and dbg["acc_max"] > 8e5 and dbg["vp2_score"] > 8000
    assert np.allclose(Delta.T @ Delta, np.eye(3), atol=1e-9)
    # every detected direction is one of the world axes seen from the camera (up to sign), within the 1-degree grid
    cosines = np.abs(Delta.T @ R_iw)                           # |delta_k . R e_j|
    assert sorted(np.argmax(cosines, axis=1)) == [0, 1, 2]
    assert np.all(np.rad2deg(np.arccos(np.clip(cosines.max(axis=1), -1, 1))) < 2.5)


def test_so3_exp_and_optimizer():
    from scipy.spatial.transform import Rotation
    from relative_pose_estimation_amd import vp_refinement as vp
    rng = np.random.default_rng(5)
    for _ in range(5):
        w = rng.normal(size=3) * 0.7
        assert np.allclose(vp.so3_exp(w), Rotation.from_rotvec(w).as_matrix(), atol=1e-12)
    assert np.array_equal(vp.so3_exp(np.zeros(3)), np.eye(3))
    R_true = Rotation.from_euler("yxz", [25, -10, 4], degrees=True).as_matrix()
    D_world = np.eye(3)
    Delta = R_true @ D_world
    R0 = Rotation.from_rotvec(np.deg2rad([3.0, -2.0, 1.5])).as_matrix() @ R_true
    assert vp.vp_cost(R_true, Delta, D_world) < 1e-6
    assert vp.vp_cost(R0, Delta, D_world) == pytest.approx(sum(np.arccos(np.clip(Delta[:, k] @ R0[:, k], -1, 1)) for k in range(3)))
    # one LM step = the reference's formula (pose_estimator.py:452-475): J_k = -(delta_k x R d_k) / sqrt(1 - s_k^2),
    # dw = -(J^T J + lambda I)^-1 J^T r, R <- exp(dw) R
    U = R0 @ D_world
    s = np.array([Delta[:, k] @ U[:, k] for k in range(3)])
    J = np.stack([-np.cross(Delta[:, k], U[:, k]) / np.sqrt(1 - s[k] ** 2) for k in range(3)])
    dw = -np.linalg.solve(J.T @ J + 1e-2 * np.eye(3), J.T @ np.arccos(s).reshape(3, 1)).reshape(3)
    assert np.allclose(vp.optimize_rotation_from_vps(R0, Delta, D_world, iters=1, lm_lambda=1e-2), vp.so3_exp(dw) @ R0, atol=1e-12)
    # NB the reference's Jacobian has the opposite sign of d arccos(delta . exp(w) R d)/dw = +(delta x R d)/sqrt(1-s^2),
    # so its step climbs; the port keeps it (results must equal the reference's) and the acceptance gate rejects the result
    assert vp.vp_cost(vp.so3_exp(dw) @ R0, Delta, D_world) > vp.vp_cost(R0, Delta, D_world)
    assert vp.vp_cost(vp.so3_exp(-dw) @ R0, Delta, D_world) < 0.2 * vp.vp_cost(R0, Delta, D_world)


def test_refinement_gates_and_acceptance(K):
    from scipy.spatial.transform import Rotation
    from relative_pose_estimation_amd import vp_refinement as vp
    R_prev = _Rz(np.deg2rad(3)) @ _Rx(np.deg2rad(-8)) @ _Ry(np.deg2rad(17))    # world rotation of frame 1 (delta ~ R d)
    R_rel_true = _Ry(np.deg2rad(5)) @ _Rx(np.deg2rad(1))
    R_new = R_prev @ R_rel_true
    img1, img2 = _render(K, R_prev), _render(K, R_new)
    R_rel_bad = R_rel_true @ Rotation.from_rotvec(np.deg2rad([1.5, -2.0, 1.0])).as_matrix()
    R_ref, used, dbg = vp.refine_relative_rotation(R_rel_bad, R_prev, img1, img2, K)
    assert dbg["vp_extracted"] and dbg["reliability"] == {"prev_reliable": True, "new_reliable": True}     # default gates pass
    opt = dbg["optimization"]
    assert used == opt["cost_improved"] == (opt["cost_opt"] < opt["cost_init"] - 1e-3)
    if used:
        assert _angle(R_prev @ R_ref, R_new) <= _angle(R_prev @ R_rel_bad, R_new) + 5.0
    else:
        assert R_ref is R_rel_bad                                        # rejected: R_rel untouched (:563-565)
    # the Manhattan frames of the two renders are consistent with the true motion: D2 ~ R_rel^T-rotated D1
    D1, _, _ = vp.estimate_manhattan_dirs(img1, K, rng_seed=0)
    D2, _, _ = vp.estimate_manhattan_dirs(img2, K, rng_seed=1)
    c = np.abs((R_prev.T @ D1).T @ (R_new.T @ D2))                        # both are world axes up to sign / order
    assert np.all(c.max(axis=1) > np.cos(np.deg2rad(3.5)))
    # gates: a texture without long lines leaves R untouched and never reaches the optimiser
    noise = np.random.default_rng(1).integers(0, 256, (480, 640), dtype=np.uint8)
    R_same, used2, dbg2 = vp.refine_relative_rotation(R_rel_bad, R_prev, noise, noise, K)
    assert not used2 and R_same is R_rel_bad and "optimization" not in dbg2


def test_estimator_applies_vp_only_with_flag_and_R_prev(K, monkeypatch):
    """call-site contract of pose_estimator.py:536: both `use_vp_refinement` and R_prev are required"""
    from relative_pose_estimation_amd import PoseEstimator
    calls = []
    pe = PoseEstimator(K, use_vp_refinement=True)
    monkeypatch.setattr(pe, "_vp_refine", lambda R, Rp, a, b: (calls.append(1), (R, False, {"stub": True}))[1])

    class _Eng:
        def estimate_batch(self, a, b, K):
            return np.eye(3)[None], np.ones((1, 3, 1)), np.array([9]), np.array([50]), np.array([0])

        def fetch_matched_points(self, B):
            return np.zeros((1, 500, 2), np.float32), np.zeros((1, 500, 2), np.float32)
    monkeypatch.setattr(pe, "_engine", lambda h, w, b: _Eng())
    g = np.zeros((480, 640), np.uint8)
    pe.estimate(g, g); assert calls == []
    pe.estimate(g, g, R_prev=np.eye(3)); assert calls == [1]
    d = pe.estimate_with_debug(g, g, R_prev=np.eye(3))
    assert d["vp_used"] is False and d["vp_debug"] == {"stub": True} and calls == [1, 1]
    pe.use_vp_refinement = False
    pe.estimate(g, g, R_prev=np.eye(3)); assert calls == [1, 1]

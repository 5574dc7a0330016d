"""CPU tests of the host logic either side of the hot path: ground-truth table,
evaluator metrics against the reference's own result rows, ingest decoding.  No GPU call."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class _GT:
    """ground-truth stand-in with the loader's get_pose contract"""
    def __init__(self, rows):                  # rows: frame x y z roll pitch yaw
        self.rows = {int(r[0]): r for r in rows}

    def get_pose(self, f):
        r = self.rows[int(f)]
        return {"frame": int(r[0]), "x": r[1], "y": r[2], "z": r[3], "roll": r[4], "pitch": r[5], "yaw": r[6]}


def test_evaluator_reproduces_reference_rows():
    """angle-error columns and column order of the reference's evaluation_results.csv (first 16 rows of
    the simulator run, tests/golden/evaluator_rows.npz) from its own est_* / gt_* columns"""
    from relative_pose_estimation_amd import PoseEvaluator, geometry as g
    from relative_pose_estimation_amd.pose_evaluator import CSV_COLUMNS
    z = np.load(os.path.join(GOLD, "evaluator_rows.npz"))
    cols = [str(c) for c in z["columns"]]
    assert list(CSV_COLUMNS) == cols
    T = {c: z["table"][:, i] for i, c in enumerate(cols)}
    n = len(T["frame"])
    # R / t are not in the CSV: rotations equal to ground truth (error 0), unit translations along the GT step
    gt = z["gt"]
    R = [g.euler_to_rotation(r[6], r[5], r[4], "yup") for r in gt]
    pos = gt[:, 1:4]
    t = [np.array([[0.], [0.], [1.]])] + [(pos[i] - pos[i - 1]).reshape(3, 1) for i in range(1, n)]
    est = {"frames": [int(f) for f in T["frame"]], "roll": T["est_roll"], "pitch": T["est_pitch"], "yaw": T["est_yaw"], "R": R, "t": t}
    ev = PoseEvaluator(_GT(gt), euler_convention="yup")
    with np.errstate(invalid="ignore"):
        res = ev.evaluate_sequence(est)
    for k in ("roll", "pitch", "yaw"):
        assert np.allclose(res[f"{k}_error"], T[f"{k}_error"], rtol=0, atol=1e-9), k
        assert np.array_equal(res[f"gt_{k}"], T[f"gt_{k}"])
    assert np.all(res["rotation_error"] < 1e-5)
    assert res["translation_dir_error"][0] == 0.0 and T["translation_dir_error"][0] == 0.0    # first evaluated frame (:117-119)
    moving = np.linalg.norm(np.diff(pos, axis=0), axis=1) > 0
    assert np.all(res["translation_dir_error"][1:][moving] < 1e-5)
    stats = ev.compute_summary_statistics({k: np.nan_to_num(v) if k.endswith("error") else v for k, v in res.items()})
    assert stats["yaw_error_median"] == pytest.approx(np.median(T["yaw_error"]))
    assert set(stats) == {f"{m}_{s}" for m in ("roll_error", "pitch_error", "yaw_error", "rotation_error", "translation_dir_error")
                          for s in ("mean", "std", "median", "max", "min")}
    df = ev.create_comparison_dataframe(res)
    assert list(df.columns) == cols and len(df) == n


def test_wrap_angle_error():
    from relative_pose_estimation_amd import PoseEvaluator
    w = PoseEvaluator._wrap_angle_error
    assert w(350.0) == pytest.approx(10.0) and w(190.0) == pytest.approx(170.0) and w(10.0) == pytest.approx(10.0)
    assert w(359.0 + 0.5) == pytest.approx(0.5) and w(180.0) == pytest.approx(180.0)


def test_ground_truth_loader(tmp_path):
    from relative_pose_estimation_amd import GroundTruthLoader
    p = tmp_path / "camera_poses.txt"
    p.write_text("frame x y z roll pitch yaw\n0 0 0 0 0.1 0.2 0.3\n15 1 2 3 1.5 2.5 3.5\n30 2 4 6 -1 -2 179.5\n")
    gl = GroundTruthLoader(p)
    with pytest.raises(RuntimeError, match="Ground truth not loaded"):
        gl.get_pose(0)
    gl.load()
    assert gl.get_pose(15) == {"frame": 15, "x": 1.0, "y": 2.0, "z": 3.0, "roll": 1.5, "pitch": 2.5, "yaw": 3.5}
    with pytest.raises(KeyError, match="Frame 7 not found"):
        gl.get_pose(7)
    assert list(gl.get_frame_indices(step=30)) == [0, 30] and list(gl.get_all_frames()) == [0, 15, 30]
    assert gl.get_trajectory(step=15).shape == (3, 3) and np.array_equal(gl.get_orientations()[2], [-1, -2, 179.5])


def test_decode_and_errors(tmp_path):
    from PIL import Image
    from relative_pose_estimation_amd import image_loader, BatchProcessor
    rgb = np.random.default_rng(3).integers(0, 256, (20, 30, 3), dtype=np.uint8)
    Image.fromarray(rgb).save(tmp_path / "000001.png")
    assert np.array_equal(image_loader.decode_rgb(str(tmp_path / "000001.png")), rgb)
    assert np.array_equal(image_loader.load_image(tmp_path / "000001.png", to_gray=False), rgb[..., ::-1])   # cv2.imread layout
    with pytest.raises(FileNotFoundError, match="Could not read image from"):
        image_loader.decode_rgb(str(tmp_path / "missing.png"))
    bp = BatchProcessor(tmp_path, pose_estimator=None, ground_truth_loader=None)
    assert bp.get_image_path(7).name == "000007.png"
    with pytest.raises(ValueError, match="at least 2 frames"):
        bp.process_sequence([3])

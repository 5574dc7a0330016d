"""CPU tests of the boundary: the C-ABI library loads and exports every symbol the
header declares; no compute call is made without a GPU; the product fails loudly
when no HIP device is present; the drop-in class mirrors the reference's errors."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    src = open(os.path.join(ROOT, "include", "rpe_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rpe_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from relative_pose_estimation_amd import _capi
    lib = _capi.load()
    syms = _header_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/rpe_amd.h but not exported"
    assert sorted(_capi.EXPORTS) == syms


def test_no_cpu_fallback():
    from relative_pose_estimation_amd import _capi
    lib = _capi.load()
    if lib.rpe_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(_capi.RpeError, match="no HIP device"):
        _capi.Engine(640, 480)
    from relative_pose_estimation_amd import PoseEstimator
    pe = PoseEstimator(np.eye(3))
    with pytest.raises(_capi.RpeError):
        pe.estimate(np.zeros((480, 640), np.uint8), np.zeros((480, 640), np.uint8))


def test_configuration_limits_are_checked_before_the_device():
    """rpe_create validates the configuration first (no GPU needed): SIFT accepts nfeatures = 0 -- the reference's
    uncapped cv2.SIFT_create(), pose_estimator.py:93-94 -- and caps up to 16320; ORB needs 1 .. 8000."""
    from relative_pose_estimation_amd import _capi
    if _capi.load().rpe_device_count() > 0:
        pytest.skip("a GPU is present")
    sift = dict(feature_method=_capi.FEATURE_SIFT, norm_type=_capi.NORM_L2)
    for kw, msg in ((dict(nfeatures=0), "out of supported range"), (dict(nfeatures=8001), "out of supported range"),
                    (dict(nfeatures=16321, **sift), "NORM_L2: nfeatures must be <= 16320"), (dict(nfeatures=-1, **sift), "SIFT: nfeatures must be 0"),
                    (dict(nfeatures=8000, norm_type=_capi.NORM_L2), "no HIP device"),
                    (dict(nfeatures=0, **sift), "no HIP device"), (dict(nfeatures=16320, **sift), "no HIP device"),
                    (dict(max_matches=8065), "out of supported range"), (dict(stl_runtime=2), "unknown stl_runtime")):
        with pytest.raises(_capi.RpeError, match=msg):
            _capi.Engine(640, 480, **kw)


def test_product_never_imports_oracle():
    """the shipped path must not import, link, include or dlopen anything under oracle/"""
    pkg = os.path.join(ROOT, "relative_pose_estimation_amd")
    bad = re.compile(r"^\s*(import\s+oracle|from\s+oracle)|liboracle|#\s*include\s*[\"<][^\">]*oracle|-loracle|orc_[a-z_]+\s*\(", re.M)
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".inc")) or f == "Makefile":
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert not bad.search(txt), f


def test_constructor_mirrors_reference():
    from relative_pose_estimation_amd import PoseEstimator
    import inspect
    sig = inspect.signature(PoseEstimator.__init__)
    names = list(sig.parameters)[1:14]
    assert names == ["camera_matrix", "feature_method", "norm_type", "max_matches", "nfeatures", "use_vp_refinement",
                     "vp_max_lines", "vp_max_pairs", "vp_acc_min", "vp_vp2_min", "vp_iters", "vp_lm_lambda",
                     "vp_cost_improve_eps"]                                      # pose_estimator.py:19-32
    d = {k: v.default for k, v in sig.parameters.items()}
    assert (d["feature_method"], d["norm_type"], d["max_matches"], d["nfeatures"], d["use_vp_refinement"]) == ("ORB", "Hamming", 500, 4000, False)
    with pytest.raises(ValueError, match="Unknown feature extraction method: FOO"):  # pose_estimator.py:96
        PoseEstimator(np.eye(3), feature_method="foo")
    with pytest.raises(ValueError, match="Unknown norm type: L7"):                   # pose_estimator.py:129
        PoseEstimator(np.eye(3), norm_type="l7")
    with pytest.raises(RuntimeError, match=r"Insufficient matches: 3 \(minimum 5 required\)"):  # :514-515
        PoseEstimator._raise_for(2, 3)
    with pytest.raises(RuntimeError, match="Could not estimate Essential matrix."):   # :529-530
        PoseEstimator._raise_for(3, 9)

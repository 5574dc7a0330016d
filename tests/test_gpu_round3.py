"""GPU parity tests added in round 3 (through the C-ABI, against the CPU oracle and against the reference's own
committed answers): every one of the 147 result rows of the reference on the GPU; the 4096-pair shard of BASELINE
configs[3] on one GPU; capacity flags after a chunked host batch."""
import os

import numpy as np
import pytest

from tests import reference_rows as rr

pytestmark = pytest.mark.gpu
NTHREADS = max(1, min(16, os.cpu_count() or 1))
RUNTIME = {"sim": "msvc", "salah": "libstdc++", "phone": "libstdc++"}     # tests/test_reference_rows_cpu.py
FLOOR = {"sim": 45, "salah": 69, "phone": 8}


@pytest.fixture(scope="module")
def oracle():
    from oracle import oracle as o
    yield o
    o.set_stl("libstdc++")


@pytest.mark.parametrize("name", rr.NAMES)
def test_every_reference_row_on_the_gpu(oracle, name):
    """All rows of one reference result file (simulator 58 pairs at 640x480, Salah 80 at 1920x1080, phone 9 at 848x478:
    pitch != width) with the reference's parameters (ORB 4000, top-500: pipeline.py:94-101) through estimate_batch:
    GPU == oracle bit for bit, and the GPU's own answers against the CSV's est_* columns."""
    from relative_pose_estimation_amd import PoseEstimator, geometry
    ds = rr.load(name)
    B = len(ds["frames2"])
    pe = PoseEstimator(ds["K"], nfeatures=4000, max_matches=500, max_batch=B, keypoint_order=RUNTIME[name])
    R, t, inl, st = pe.estimate_batch(ds["img1"], ds["img2"])
    nm = pe._last_n_matches.copy()
    ovf = pe.last_overflow()
    pe.close()
    oracle.set_stl(RUNTIME[name])
    out = oracle.estimate_pose_batch(ds["img1"], ds["img2"], ds["K"], 4000, 500, nthreads=NTHREADS)
    oracle.set_stl("libstdc++")
    assert np.array_equal(st, out["status"]) and np.all(st == 0)
    assert np.array_equal(nm, out["n_matches"]) and np.array_equal(inl, out["inliers"])
    assert np.array_equal(ovf, out["overflow"]) and not ovf.any()
    assert np.array_equal(R.reshape(B, 9), out["R"]) and np.array_equal(t.reshape(B, 3), out["t"])      # bit-exact (tolerance of the path: 1e-4 Frobenius)
    diff = rr.euler_agreement(ds, R, geometry)
    err = rr.rotation_errors(ds, R, geometry)
    ref = ds["ref_rotation_error"]
    counts = rr.agreement_counts(diff)
    print(f"\n[{name}] GPU, {B} rows: est_* agree within " + ", ".join(f"{e:g} deg: {c}" for e, c in zip(rr.AGREE_EDGES, counts))
          + f"; median rotation error GPU {np.median(err):.3f} deg, reference {np.median(ref):.3f}")
    assert counts[0] >= FLOOR[name]
    agree = diff < 1e-6
    assert np.allclose(err[agree], ref[agree], atol=1e-5)


def test_reference_stream_equals_pairs(oracle):
    """The phone rows as ONE consecutive-frame stream through BatchProcessor (features once per frame,
    batch_processor.py:71-109): the est_* columns it produces are the ones estimate_batch produces."""
    from relative_pose_estimation_amd import BatchProcessor, PoseEstimator, geometry
    ds = rr.load("phone")
    frames = np.concatenate([ds["img1"][:1], ds["img2"]])
    idx = [int(ds["frames1"][0])] + [int(f) for f in ds["frames2"]]

    class GT:                                   # ground_truth_loader.get_pose of the committed rows
        def get_pose(self, f):
            i = idx.index(f)
            g = ds["gt1"][i] if i < len(ds["gt1"]) else ds["gt2"][-1]
            return {"roll": g[3], "pitch": g[4], "yaw": g[5]}
    pe = PoseEstimator(ds["K"], nfeatures=4000, max_matches=500, max_batch=len(idx))
    out = BatchProcessor(None, pe, GT(), euler_convention=ds["convention"]).process_frames(idx, frames)
    pe.close()
    cols = ds["columns"]
    ref = ds["table"][:, [cols.index("est_yaw"), cols.index("est_pitch"), cols.index("est_roll")]]
    got = np.stack([out["yaw"], out["pitch"], out["roll"]], 1)
    d = np.abs((got - ref + 180.0) % 360.0 - 180.0).max(1)
    assert (d < 1e-6).sum() >= FLOOR["phone"], d


def test_shard_of_4096_pairs(oracle, K_vga):
    """BASELINE configs[3] gives every GPU 4096 VGA pairs: that shard on ONE GPU -- the workspace for 8192 images is
    allocated, and the results of pair i equal those of the same pair in a 64-pair batch (and the oracle's for a sample)."""
    from relative_pose_estimation_amd import _capi, synthetic
    i1, i2, _, _ = synthetic.make_batch(64, K_vga, cfg=7)
    small = _capi.Engine(640, 480, max_batch=64, nfeatures=1000, max_matches=500)
    Rs, ts, inls, nms, sts = small.estimate_batch(i1, i2, K_vga)
    small.close()
    big = _capi.Engine(640, 480, max_batch=4096, nfeatures=1000, max_matches=500)
    a = np.tile(i1, (64, 1, 1)); b = np.tile(i2, (64, 1, 1))
    d1 = big.upload(a); d2 = big.upload(b)
    big.enqueue_batch_device(d1, d2, 4096, K_vga)
    R, t, inl, nm, st = big.fetch_results(4096)
    big.device_free(d1); big.device_free(d2)
    big.close()
    for rep in range(64):
        s = slice(64 * rep, 64 * rep + 64)
        assert np.array_equal(R[s], Rs) and np.array_equal(t[s], ts) and np.array_equal(inl[s], inls) and np.array_equal(st[s], sts)
    for n in (0, 17, 63):
        r = oracle.estimate_pose(i1[n], i2[n], K_vga, 1000, 500)
        assert st[n] == r["status"] and nm[n] == r["n_matches"] and inl[n] == r["inliers"] and np.array_equal(R[n], r["R"])


def test_capacity_flags_survive_a_chunked_host_batch(oracle, K_vga):
    """rpe_estimate_batch runs 512 host pairs in four chunks (uploads behind kernels); the per-image flag words are
    reused chunk after chunk, so the flags are collected per pair as the chunks finish: a dots image (thousands of tied
    FAST scores) in the middle of chunk 1 still reports its truncation, nobody else does."""
    from relative_pose_estimation_amd import _capi, synthetic
    i1, i2, _, _ = synthetic.make_batch(8, K_vga, cfg=8)
    a = np.tile(i1, (64, 1, 1)); b = np.tile(i2, (64, 1, 1))
    dots = np.full((480, 640), 40, np.uint8)
    dots[40:440:8, 40:600:8] = 220
    a[200] = dots; b[200] = dots
    e = _capi.Engine(640, 480, max_batch=512, nfeatures=1000, max_matches=500)
    R, t, inl, nm, st = e.estimate_batch(a, b, K_vga)
    ovf = e.fetch_overflow(512)
    e.close()
    _, _, fo = oracle.orb_detect_and_compute(dots, 1000, return_flags=True)
    assert fo == _capi.OVF_ORB_CANDIDATES | _capi.OVF_ORB_KEYPOINTS
    assert ovf[200] == fo and not np.delete(ovf, 200).any()
    assert np.array_equal(R[8:16], R[:8]) and np.array_equal(R[504:512], R[:8])       # copies of a pair, whichever chunk they sit in


@pytest.mark.parametrize("order", ["libstdc++", "msvc"])
def test_retain_best_replay_on_long_and_truncated_lists(oracle, K_vga, order):
    """The selection replay beyond the ordinary case, under both runtimes: a dots grid of pitch 8 gives level 0 a corner list of
    3500 entries (the long-list launch of retain_fast; all FAST scores tied, so the first retainBest keeps everything and the
    4 quota + 256 candidate capacity truncates in cv2's order), pitch 4 gives 14 000 corners (the raster list itself is
    truncated at clamp(w h / 64) = 4800 entries, first in raster order).  Keypoints, their ORDER, descriptors and flags equal
    the oracle's."""
    from relative_pose_estimation_amd import _capi
    stl = {"libstdc++": _capi.STL_LIBSTDCXX, "msvc": _capi.STL_MSVC}[order]
    oracle.set_stl(order)
    try:
        e = _capi.Engine(640, 480, max_batch=1, nfeatures=1000, max_matches=500, stl_runtime=stl)
        for pitch in (8, 4):
            dots = np.full((480, 640), 40, np.uint8)
            dots[40:440:pitch, 40:600:pitch] = 220
            kps, desc, cnt = e.orb_detect_and_compute(dots[None])
            ko, do, fo = oracle.orb_detect_and_compute(dots, 1000, return_flags=True)
            assert fo & _capi.OVF_ORB_CANDIDATES
            assert cnt[0] == len(ko)
            kg = kps[0, :cnt[0]]
            assert np.array_equal(kg["lx"], ko["lx"]) and np.array_equal(kg["ly"], ko["ly"]) and np.array_equal(kg["octave"], ko["octave"])
            assert np.array_equal(desc[0, :cnt[0]], do)
            R, t, inl, nm, st = e.estimate_batch(dots[None], dots[None], K_vga)
            assert int(e.fetch_overflow(1)[0]) == fo
        e.close()
    finally:
        oracle.set_stl("libstdc++")


def test_largest_keypoint_capacity(oracle, K_vga):
    """nfeatures = 8000 (the largest the handle accepts): 8064 keypoints per image -- the Hamming matcher's election words no
    longer fit the fused kernel's LDS and live in HBM for every batch size; the selection replays run their long-list
    launches; GPU == oracle end to end, crossCheck and Lowe ratio."""
    from relative_pose_estimation_amd import PoseEstimator, synthetic
    i1, i2, _, _ = synthetic.make_batch(2, K_vga, cfg=6)
    for ratio in (None, 0.8):
        pe = PoseEstimator(K_vga, nfeatures=8000, max_matches=500, max_batch=2, ratio=ratio)
        R, t, inl, st = pe.estimate_batch(i1, i2)
        nm = pe._last_n_matches.copy()
        pe.close()
        for n in range(2):
            r = oracle.estimate_pose(i1[n], i2[n], K_vga, 8000, 500, ratio=ratio)
            assert st[n] == r["status"] == 0 and nm[n] == r["n_matches"] and inl[n] == r["inliers"], (ratio, n, nm[n], r["n_matches"])
            assert np.array_equal(R[n], r["R"]) and np.array_equal(t[n], r["t"])


def test_orb_on_a_3840x2160_image(oracle):
    """A UHD frame (a reference HD frame doubled): the raster corner kernel needs 66 KB of dynamic LDS per workgroup here, more
    than the 64 KB default -- keypoints in cv2's order, descriptors and flags still equal the oracle's."""
    from relative_pose_estimation_amd import _capi
    ds = rr.load("salah", rows=[3])
    big = np.ascontiguousarray(np.kron(ds["img1"][0], np.ones((2, 2), np.uint8)))
    assert big.shape == (2160, 3840)
    e = _capi.Engine(3840, 2160, max_batch=1, nfeatures=4000, max_matches=500)
    kps, desc, cnt = e.orb_detect_and_compute(big[None])
    e.close()
    ko, do, fo = oracle.orb_detect_and_compute(big, 4000, return_flags=True)
    assert cnt[0] == len(ko) > 3000
    kg = kps[0, :cnt[0]]
    assert np.array_equal(kg["lx"], ko["lx"]) and np.array_equal(kg["ly"], ko["ly"]) and np.array_equal(kg["octave"], ko["octave"])
    assert np.array_equal(kg["angle"].view(np.uint32), ko["angle"].view(np.uint32)) and np.array_equal(desc[0, :cnt[0]], do)


def test_sift_without_a_cap(oracle):
    """feature_method="SIFT" as the reference builds it: cv2.SIFT_create() with no arguments (pose_estimator.py:93-94), i.e.
    no retainBest.  A textured 1920x1080 pair holds ~12.7k keypoints per image, three times what the capped path of
    rounds 1-2 could keep: keypoints, descriptors, match count and pose of the drop-in class equal the oracle's run with
    nfeatures = 0, and no capacity flag is raised."""
    from relative_pose_estimation_amd import PoseEstimator, _capi, synthetic, geometry
    W, H = 1920, 1080
    K = geometry.default_camera_matrix(W, H)
    i1, i2, _, _ = synthetic.make_batch(1, K, W, H, cfg=5)
    pe = PoseEstimator(K, feature_method="SIFT", norm_type="L2", nfeatures=4000, max_matches=500)   # nfeatures: "ORB only" (:41)
    d = pe.estimate_with_debug(i1[0], i2[0])
    assert int(pe.last_overflow()[0]) == 0
    eng = pe._engines[(H, W)]
    assert eng.kcap == _capi.SIFT_UNCAPPED_CAPACITY + 64
    kps, desc, cnt = eng.sift_detect_and_compute(np.stack([i1[0], i2[0]]))
    for n, img in enumerate((i1[0], i2[0])):
        ko, do, fo = oracle.sift_detect_and_compute(img, nfeatures=0, cap=eng.kcap, return_flags=True)
        assert fo == 0 and cnt[n] == len(ko) > 8192, (cnt[n], len(ko))
        kg = kps[n, :cnt[n]]
        for f in ("x", "y", "size", "angle", "response"):
            assert np.array_equal(kg[f].view(np.uint32), ko[f].view(np.uint32)), f
        assert np.array_equal(kg["octave"], ko["octave"]) and np.array_equal(desc[n, :cnt[n]], do)
    r = oracle.estimate_pose_batch(i1, i2, K, 0, 500, nthreads=1, method="SIFT")[0]
    assert r["status"] == 0 and int(r["overflow"]) == 0
    assert d["num_matches"] == r["n_matches"] and d["inliers"] == r["inliers"]
    assert np.array_equal(d["R"], r["R"].reshape(3, 3)) and np.array_equal(d["t"].ravel(), r["t"].ravel())
    pe.close()


def test_sift_cap_2048_on_a_textured_hd_frame(oracle):
    """BASELINE configs[2]'s extractor on its own kind of frame: ~10 k refined extrema per image, of which only the
    nfeatures * 5/4 + 256 strongest get an orientation histogram on the GPU (sift_select_kernel) -- keypoints, their order,
    descriptors and the cap flag still equal the oracle's, which orients every one of them and cuts afterwards."""
    from relative_pose_estimation_amd import _capi, synthetic, geometry
    W, H = 1920, 1080
    K = geometry.default_camera_matrix(W, H)
    i1, _, _, _ = synthetic.make_batch(1, K, W, H, cfg=5)
    e = _capi.Engine(W, H, max_batch=1, nfeatures=2048, max_matches=500, feature_method=_capi.FEATURE_SIFT, norm_type=_capi.NORM_L2)
    kps, desc, cnt = e.sift_detect_and_compute(i1)
    ko, do, fo = oracle.sift_detect_and_compute(i1[0], nfeatures=2048, cap=e.kcap, return_flags=True)
    assert cnt[0] == len(ko) >= 2048, (cnt[0], len(ko))
    kg = kps[0, :cnt[0]]
    for f in ("x", "y", "size", "angle", "response"):
        assert np.array_equal(kg[f].view(np.uint32), ko[f].view(np.uint32)), f
    assert np.array_equal(kg["octave"], ko["octave"]) and np.array_equal(desc[0, :cnt[0]], do)
    R, t, inl, nm, st = e.estimate_batch(i1, i1, K)                      # flags of a whole run: the cap removed keypoints
    assert int(e.fetch_overflow(1)[0]) == _capi.OVF_SIFT_CAP and (fo & _capi.OVF_SIFT_CAP)
    e.close()


def test_sift_second_round_when_the_strongest_do_not_fill_the_cap(oracle, monkeypatch):
    """The survivor selection is exact only while the selected ones produce >= nfeatures unique keypoints; when they do not,
    the image is oriented again with every survivor.  RPE_SIFT_SEL_K = 40 (instead of nfeatures * 5/4 + 256) forces that on
    ordinary images: keypoints, descriptors, flags and pose still equal the oracle's."""
    from relative_pose_estimation_amd import _capi, synthetic, geometry
    K = geometry.default_camera_matrix(320, 240)
    i1, i2, _, _ = synthetic.make_batch(2, K, 320, 240, cfg=6)
    monkeypatch.setenv("RPE_SIFT_SEL_K", "40")
    e = _capi.Engine(320, 240, max_batch=2, nfeatures=300, max_matches=200, feature_method=_capi.FEATURE_SIFT, norm_type=_capi.NORM_L2)
    monkeypatch.delenv("RPE_SIFT_SEL_K")
    imgs = np.concatenate([i1, i2])
    kps, desc, cnt = e.sift_detect_and_compute(imgs)
    for n in range(len(imgs)):
        ko, do = oracle.sift_detect_and_compute(imgs[n], nfeatures=300, cap=e.kcap)
        assert cnt[n] == len(ko) >= 300, (n, cnt[n], len(ko))
        kg = kps[n, :cnt[n]]
        for f in ("x", "y", "size", "angle", "response"):
            assert np.array_equal(kg[f].view(np.uint32), ko[f].view(np.uint32)), (n, f)
        assert np.array_equal(desc[n, :cnt[n]], do)
    R, t, inl, nm, st = e.estimate_batch(i1, i2, K)
    ovf = e.fetch_overflow(2)
    for n in range(2):
        r = oracle.estimate_pose_batch(i1[n:n + 1], i2[n:n + 1], K, 300, 200, nthreads=1, method="SIFT")[0]
        assert st[n] == r["status"] and nm[n] == r["n_matches"] and inl[n] == r["inliers"] and np.array_equal(R[n], r["R"].reshape(3, 3))
        assert int(ovf[n]) == int(r["overflow"])
    e.close()


def test_marching_pyramid_is_bit_identical(oracle, monkeypatch):
    """RPE_SIFT_MARCH=1 builds levels 1-3 and 4-5 of the large octaves with sift_march_kernel (one pass over the source level
    each, rings of row-filtered rows in LDS) instead of one tile kernel launch per level: the whole Gaussian pyramid, the
    keypoints and the descriptors must not change by a bit (640x480 and 320x240 octaves march here, the rest take the tiles)."""
    from relative_pose_estimation_amd import _capi, synthetic, geometry
    K = geometry.default_camera_matrix(320, 240)
    i1, i2, _, _ = synthetic.make_batch(1, K, 320, 240, cfg=6)
    imgs = np.concatenate([i1, i2])
    monkeypatch.setenv("RPE_SIFT_MARCH", "1")
    e = _capi.Engine(320, 240, max_batch=1, nfeatures=600, max_matches=300, feature_method=_capi.FEATURE_SIFT, norm_type=_capi.NORM_L2)
    monkeypatch.delenv("RPE_SIFT_MARCH")
    kps, desc, cnt = e.sift_detect_and_compute(imgs)
    for n in range(2):
        go, dims = oracle.sift_gauss_pyramid(imgs[n])
        gg = e.sift_debug_gauss(n)
        assert gg.shape == go.shape and np.array_equal(gg.view(np.uint32), go.view(np.uint32)), "gaussian pyramid differs"
        ko, do = oracle.sift_detect_and_compute(imgs[n], nfeatures=600, cap=e.kcap)
        assert cnt[n] == len(ko) and np.array_equal(desc[n, :cnt[n]], do)
    e.close()

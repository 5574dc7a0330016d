"""GPU parity tests added in round 2 (through the C-ABI, against the CPU oracle): the reference's HD frames with
SIFT + L2 (BASELINE configs[2] shape), every per-pair status produced end to end from images, the reference's
'no truncation' max_matches=None, capacity-overflow flags, ORB + L2, the opt-in Lowe ratio matcher, the FAST tile
lists, and size-independent properties at BASELINE's full batch size."""
import numpy as np
import pytest

from tests import reference_rows as rr

pytestmark = pytest.mark.gpu

TOL_RT = 1e-4   # north_star: R/t within 1e-4 Frobenius


@pytest.fixture(scope="module")
def capi():
    from relative_pose_estimation_amd import _capi
    assert _capi.load().rpe_device_count() > 0, "no HIP device visible"
    return _capi


def _blobs(seed, n, lo, hi, W=640, H=480):
    """a few bright rectangles on a dark background: a handful of FAST corners per image"""
    rng = np.random.default_rng(seed)
    img = np.full((H, W), 40, np.uint8)
    for _ in range(n):
        w, h = rng.integers(lo, hi, 2)
        x = rng.integers(70, W - 70 - w); y = rng.integers(70, H - 70 - h)
        img[y:y + h, x:x + w] = rng.integers(150, 255)
    return img


def _dots(W=640, H=480, pitch=8):
    """isolated bright pixels on a grid: thousands of identical strict FAST maxima with identical Harris responses"""
    img = np.full((H, W), 40, np.uint8)
    img[40:H - 40:pitch, 40:W - 40:pitch] = 220
    return img


# ------------------------------------------------------------------ BASELINE configs[2]: HD, SIFT(2048) + BF-L2
def test_hd_sift_l2_reference_frames(capi, oracle):
    """Two 1920x1080 frames of the reference's Salah run (vo_dataset_salah, CSV rows 2-3) through SIFT (cap 2048)
    + BFMatcher(NORM_L2, crossCheck) + RANSAC + recoverPose: GPU == oracle on keypoints, descriptors and pose.
    These frames hold fewer than 2048 SIFT keypoints, so the cap does not bite and no capacity flag is raised:
    the result is what the reference's uncapped SIFT_create() (pose_estimator.py:93-94) would extract."""
    ds = rr.load("salah", rows=slice(1, 2))
    K, a, b = ds["K"], ds["img1"][0], ds["img2"][0]
    H, W = a.shape
    assert (W, H) == (1920, 1080)
    e = capi.Engine(W, H, max_batch=1, nfeatures=2048, max_matches=500, feature_method=capi.FEATURE_SIFT, norm_type=capi.NORM_L2)
    kps, desc, cnt = e.sift_detect_and_compute(np.stack([a, b]))
    flags_o = 0
    for n, img in enumerate((a, b)):
        ko, do, fo = oracle.sift_detect_and_compute(img, nfeatures=2048, cap=e.kcap, return_flags=True)
        flags_o |= fo
        assert cnt[n] == len(ko), (cnt[n], len(ko))
        kg = kps[n, :cnt[n]]
        for f in ("x", "y", "size", "angle", "response"):
            assert np.array_equal(kg[f].view(np.uint32), ko[f].view(np.uint32)), f
        assert np.array_equal(kg["octave"], ko["octave"]) and np.array_equal(desc[n, :cnt[n]], do)
    R, t, inl, nm, st = e.estimate_batch(a[None], b[None], K)
    r = oracle.estimate_pose_batch(a[None], b[None], K, 2048, 500, nthreads=1, method="SIFT")[0]
    assert st[0] == r["status"] == 0 and nm[0] == r["n_matches"] and inl[0] == r["inliers"]
    assert np.linalg.norm(R[0] - r["R"].reshape(3, 3)) <= TOL_RT and np.linalg.norm(t[0].ravel() - r["t"]) <= TOL_RT
    assert np.array_equal(R[0], r["R"].reshape(3, 3)), "R not bit-identical"
    assert int(e.fetch_overflow(1)[0]) == flags_o == int(r["overflow"]) == 0
    e.close()


def test_sift_cap_is_reported(capi, oracle):
    """SIFT_create(nfeatures) through the C-ABI (an extension: the reference's SIFT_create() has no cap).  When the cap
    removes keypoints the pair carries RPE_OVF_SIFT_CAP (GPU == oracle), so a caller can tell that the feature set
    differs from the reference's.  The drop-in class ignores nfeatures for SIFT, as the reference does, and never
    raises that flag (tests/test_gpu_round3.py::test_sift_without_a_cap)."""
    from relative_pose_estimation_amd import PoseEstimator, synthetic, geometry
    K = geometry.default_camera_matrix(320, 240)
    i1, i2, _, _ = synthetic.make_batch(1, K, 320, 240, cfg=6)
    mask = capi.OVF_SIFT_SEEDS | capi.OVF_SIFT_CAP | capi.OVF_SIFT_KEYPOINTS
    for nf, want in ((100, capi.OVF_SIFT_CAP), (4000, 0), (0, 0)):
        e = capi.Engine(320, 240, max_batch=1, nfeatures=nf, max_matches=80, feature_method=capi.FEATURE_SIFT, norm_type=capi.NORM_L2)
        R, t, inl, nm, st = e.estimate_batch(i1, i2, K)
        r = oracle.estimate_pose_batch(i1, i2, K, nf, 80, nthreads=1, method="SIFT")[0]
        ovf = int(e.fetch_overflow(1)[0])
        assert (ovf & mask) == (int(r["overflow"]) & mask) == want, (nf, ovf, r["overflow"])
        assert st[0] == r["status"] == 0 and inl[0] == r["inliers"] and np.array_equal(R[0], r["R"].reshape(3, 3))
        e.close()
    pe = PoseEstimator(K, feature_method="SIFT", norm_type="L2", nfeatures=100, max_matches=80)      # nfeatures: "ORB only"
    R, t, inl, st = pe.estimate_batch(i1, i2)
    assert int(pe.last_overflow()[0]) == 0 and inl[0] == r["inliers"] and np.array_equal(R[0], r["R"].reshape(3, 3))
    pe.close()


def test_hd_orb_reference_defaults(capi, oracle):
    """the same HD pair with the reference's own configuration (ORB 4000, Hamming, top 500): bit-exact pose"""
    from relative_pose_estimation_amd import PoseEstimator
    ds = rr.load("salah", rows=slice(1, 2))
    pe = PoseEstimator(ds["K"])
    d = pe.estimate_with_debug(ds["img1"][0], ds["img2"][0])
    r = oracle.estimate_pose(ds["img1"][0], ds["img2"][0], ds["K"], 4000, 500)
    assert np.array_equal(d["R"], r["R"]) and np.array_equal(d["t"], r["t"]) and d["inliers"] == r["inliers"] and d["num_matches"] == r["n_matches"]
    pe.close()


# ------------------------------------------------------------------ every per-pair status, end to end from images
def test_statuses_end_to_end(capi, oracle, K_vga):
    """INSUFFICIENT_MATCHES (4 matches), AMBIGUOUS_ESSENTIAL (exactly 5 matches: cv2 hands recoverPose stacked
    models), NO_ESSENTIAL (degenerate camera matrix: no sample yields a model), NO_DESCRIPTORS and OK in ONE batch;
    statuses, match counts and inlier counts equal the oracle's, and the drop-in class raises the reference's texts."""
    from relative_pose_estimation_amd import PoseEstimator, synthetic
    ok1, ok2, _, _ = synthetic.make_batch(1, K_vga, cfg=2)
    flat = np.full((480, 640), 128, np.uint8)
    cases = [
        (_blobs(7, 1, 14, 40), _blobs(1007, 1, 14, 40), capi.PAIR_INSUFFICIENT_MATCHES),
        (_blobs(23, 1, 14, 40), _blobs(1023, 1, 14, 40), capi.PAIR_AMBIGUOUS_ESSENTIAL),
        (flat, ok2[0], capi.PAIR_NO_DESCRIPTORS),
        (ok1[0], ok2[0], capi.PAIR_OK),
    ]
    e = capi.Engine(640, 480, max_batch=len(cases), nfeatures=1000, max_matches=500)
    R, t, inl, nm, st = e.estimate_batch(np.stack([c[0] for c in cases]), np.stack([c[1] for c in cases]), K_vga)
    for i, (a, b, want) in enumerate(cases):
        r = oracle.estimate_pose(a, b, K_vga, 1000, 500)
        assert r["status"] == want, (i, r["status"], want)
        assert st[i] == r["status"] and nm[i] == r["n_matches"] and inl[i] == r["inliers"], (i, st[i], nm[i], inl[i], r)
        if want == capi.PAIR_OK:
            assert np.array_equal(R[i], r["R"]) and np.array_equal(t[i], r["t"])
    # NO_ESSENTIAL: a camera matrix with zero focal lengths normalises every point to inf / nan -> no model ever
    Kbad = np.diag([0., 0., 1.])
    R, t, inl, nm, st = e.estimate_batch(ok1, ok2, Kbad)
    r = oracle.estimate_pose(ok1[0], ok2[0], Kbad, 1000, 500)
    assert r["status"] == capi.PAIR_NO_ESSENTIAL and st[0] == capi.PAIR_NO_ESSENTIAL and nm[0] == r["n_matches"]
    e.close()
    pe = PoseEstimator(K_vga, nfeatures=1000)
    with pytest.raises(RuntimeError, match=r"Insufficient matches: 4 \(minimum 5 required\)"):      # pose_estimator.py:514-515
        pe.estimate(cases[0][0], cases[0][1])
    with pytest.raises(RuntimeError, match=r"E.cols == 3 && E.rows == 3"):                            # cv2.recoverPose on stacked E
        pe.estimate(cases[1][0], cases[1][1])
    pe.close()
    pb = PoseEstimator(Kbad, nfeatures=1000)
    with pytest.raises(RuntimeError, match="Could not estimate Essential matrix."):                   # :529-530
        pb.estimate(ok1[0], ok2[0])
    pb.close()


# ------------------------------------------------------------------ max_matches = None ("no truncation", :150-151)
def test_max_matches_none(capi, oracle, K_vga):
    from relative_pose_estimation_amd import PoseEstimator, synthetic
    i1, i2, _, _ = synthetic.make_batch(1, K_vga, cfg=2)
    pe = PoseEstimator(K_vga, nfeatures=1000, max_matches=None)
    d = pe.estimate_with_debug(i1[0], i2[0])
    r = oracle.estimate_pose(i1[0], i2[0], K_vga, 1000, None)
    assert d["num_matches"] == r["n_matches"] > 300          # every mutual match, no cut
    assert np.array_equal(d["R"], r["R"]) and np.array_equal(d["t"], r["t"]) and d["inliers"] == r["inliers"]
    pe.close()
    # reference defaults (nfeatures=4000): max_matches=None must construct and run (capacity 4064 matches per pair)
    pe = PoseEstimator(K_vga, max_matches=None)
    d = pe.estimate_with_debug(i1[0], i2[0])
    r = oracle.estimate_pose(i1[0], i2[0], K_vga, 4000, None)
    assert d["num_matches"] == r["n_matches"] and np.array_equal(d["R"], r["R"]) and d["inliers"] == r["inliers"]
    pe.close()


# ------------------------------------------------------------------ capacity overflow is reported, and truncation is canonical
def test_orb_capacity_overflow_flags(capi, oracle, K_vga):
    """A grid of identical dots ties thousands of FAST scores and Harris responses: level 0 overflows its
    candidate list (4*quota+256) and the image its keypoint list (nfeatures+64).  Both are flagged, GPU and
    oracle truncate identically (first entries in raster / level-major order), and an ordinary image flags nothing."""
    from relative_pose_estimation_amd import synthetic
    dots = _dots()
    i1, i2, _, _ = synthetic.make_batch(1, K_vga, cfg=2)
    e = capi.Engine(640, 480, max_batch=2, nfeatures=1000, max_matches=500)
    kps, desc, cnt = e.orb_detect_and_compute(np.stack([dots, i1[0]]))
    ko, do, fo = oracle.orb_detect_and_compute(dots, 1000, return_flags=True)
    assert fo == capi.OVF_ORB_CANDIDATES | capi.OVF_ORB_KEYPOINTS
    assert cnt[0] == len(ko) == e.kcap
    kg = kps[0, :cnt[0]]
    assert np.array_equal(kg["lx"], ko["lx"]) and np.array_equal(kg["ly"], ko["ly"]) and np.array_equal(kg["octave"], ko["octave"])
    assert np.array_equal(desc[0, :cnt[0]], do)
    R, t, inl, nm, st = e.estimate_batch(np.stack([dots, i1[0]]), np.stack([dots, i2[0]]), K_vga)
    ovf = e.fetch_overflow(2)
    assert ovf[0] == fo and ovf[1] == 0
    r = oracle.estimate_pose(dots, dots, K_vga, 1000, 500)
    assert r["overflow"] == fo and st[0] == r["status"] and nm[0] == r["n_matches"] and inl[0] == r["inliers"]
    e.close()


# ------------------------------------------------------------------ ORB + NORM_L2 (legal in the reference, :115-131)
def test_orb_l2(capi, oracle, K_vga):
    from relative_pose_estimation_amd import PoseEstimator, synthetic
    i1, i2, _, _ = synthetic.make_batch(2, K_vga, cfg=2)
    pe = PoseEstimator(K_vga, feature_method="ORB", norm_type="L2", nfeatures=1000)
    R, t, inl, st = pe.estimate_batch(i1, i2)
    for n in range(2):
        r = oracle.estimate_pose(i1[n], i2[n], K_vga, 1000, 500, norm="L2")
        assert st[n] == r["status"] == 0 and pe._last_n_matches[n] == r["n_matches"] and inl[n] == r["inliers"]
        assert np.array_equal(R[n], r["R"]) and np.array_equal(t[n], r["t"])
    pe.close()
    # stage level: byte descriptors through rpe_match_l2, heavy ties
    rng = np.random.default_rng(5)
    a = rng.integers(0, 256, (900, 32), dtype=np.uint8); b = rng.integers(0, 256, (1000, 32), dtype=np.uint8)
    b[:300] = a[rng.permutation(900)[:300]]
    e = capi.Engine(640, 480, max_batch=1, nfeatures=1000, max_matches=500, norm_type=capi.NORM_L2)
    q, tt, d, nm = e.match_l2([a.astype(np.float32)], [900], [b.astype(np.float32)], [1000])
    qo, to, do = oracle.match_l2(a.astype(np.float32), b.astype(np.float32), 500)
    assert nm[0] == len(qo) and np.array_equal(q[0, :nm[0]], qo) and np.array_equal(tt[0, :nm[0]], to)
    assert np.array_equal(d[0, :nm[0]].view(np.uint32), do.view(np.uint32))
    e.close()
    # SIFT + Hamming: constructs like the reference (pose_estimator.py:115-131) and fails where the reference's match() fails (:144)
    pe = PoseEstimator(K_vga, feature_method="SIFT", norm_type="Hamming")
    with pytest.raises(RuntimeError, match="normType=6 is not supported in function 'batchDistance'"):
        pe.estimate(i1[0], i2[0])
    with pytest.raises(RuntimeError, match="batchDistance"):
        pe.estimate_batch(i1, i2)


# ------------------------------------------------------------------ Lowe ratio (opt-in extension; the reference has none)
def test_lowe_ratio_mode(capi, oracle, K_vga):
    from relative_pose_estimation_amd import PoseEstimator, synthetic
    rng = np.random.default_rng(9)
    a = rng.integers(0, 256, (1000, 32), dtype=np.uint8); b = rng.integers(0, 256, (937, 32), dtype=np.uint8)
    src = rng.permutation(1000)[:400]
    b[:400] = a[src]
    b[np.arange(400), rng.integers(0, 32, 400)] ^= (1 << rng.integers(0, 8, 400)).astype(np.uint8)   # near duplicates pass the ratio
    b[400:450] = b[:50]                                                                              # two equal best trains: ratio fails
    e = capi.Engine(640, 480, max_batch=2, nfeatures=1000, max_matches=500, match_mode=capi.MATCH_RATIO, match_ratio=0.75)
    q, t, d, nm = e.match_hamming([a, a[:1]], [1000, 1], [b, b[:1]], [937, 1])
    qo, to, do = oracle.match_hamming_ratio(a, b, 0.75, 500)
    assert nm[0] == len(qo) > 300 and nm[1] == 0                     # one train only: no second neighbour, no match
    assert np.array_equal(q[0, :nm[0]], qo) and np.array_equal(t[0, :nm[0]], to) and np.array_equal(d[0, :nm[0]], do)
    e.close()
    i1, i2, Rgt, _ = synthetic.make_batch(2, K_vga, cfg=2)
    for norm in ("Hamming", "L2"):
        pe = PoseEstimator(K_vga, nfeatures=1000, norm_type=norm, ratio=0.8)
        R, tt, inl, st = pe.estimate_batch(i1, i2)
        for n in range(2):
            r = oracle.estimate_pose(i1[n], i2[n], K_vga, 1000, 500, norm=norm, ratio=0.8)
            assert st[n] == r["status"] == 0 and pe._last_n_matches[n] == r["n_matches"] and inl[n] == r["inliers"]
            assert np.array_equal(R[n], r["R"]) and np.array_equal(tt[n], r["t"])
        pe.close()
    # default stays the reference's crossCheck
    assert PoseEstimator(K_vga).ratio is None


# ------------------------------------------------------------------ FAST tile lists / NMS map
def test_fast_nms_lists(capi, oracle, K_vga):
    """The fused FAST kernel emits per-tile keypoint lists; rpe_orb_debug_fetch(which=2) rebuilds the NMS map from
    them: equal to the oracle's FAST-score -> 3x3 NMS -> 31-px border filter on every level, for a textured image and
    for the dots pattern (tiles holding hundreds of survivors); which=1 (never materialised) is rejected."""
    from relative_pose_estimation_amd import synthetic
    i1, _, _, _ = synthetic.make_batch(1, K_vga, cfg=3)
    imgs = np.stack([i1[0], _dots(pitch=4)])
    e = capi.Engine(640, 480, max_batch=1, nfeatures=1000)
    e.orb_detect_and_compute(imgs)
    L = oracle.orb_layout(640, 480, 1000)
    for n in range(2):
        pyr_o, _ = oracle.build_pyramid(imgs[n], 1000)
        nms_g = e.orb_debug_fetch(n, 2)
        off, total = 0, 0
        for l in range(12):
            w, h = L.w[l], L.h[l]
            nm = oracle.nms_map(oracle.fast_score_map(pyr_o[off:off + w * h].reshape(h, w), 15))
            assert np.array_equal(nm, nms_g[off:off + w * h].reshape(h, w)), (n, l)
            total += int((nm > 0).sum())
            off += w * h
        assert total > 1000
    with pytest.raises(capi.RpeError):
        e.orb_debug_fetch(0, 1)
    e.close()


# ------------------------------------------------------------------ size-independent properties at BASELINE's batch size
def test_full_batch_properties(capi, oracle, K_vga):
    """1024 pairs (BASELINE configs[1] batch) built from 8 distinct pairs: every copy of a pair gives bit-identical
    output wherever it sits in the batch (no cross-pair interference), a failing pair planted in the middle stays
    confined, and the distinct pairs equal the oracle."""
    from relative_pose_estimation_amd import synthetic
    U, B = 8, 1024
    i1, i2, _, _ = synthetic.make_batch(U, K_vga, cfg=2)
    rng = np.random.default_rng(3)
    idx = rng.integers(0, U, B)
    a, b = i1[idx].copy(), i2[idx].copy()
    a[511] = 77                                                      # flat image: NO_DESCRIPTORS
    e = capi.Engine(640, 480, max_batch=B, nfeatures=1000, max_matches=500)
    da, db = e.upload(a), e.upload(b)
    R, t, inl, nm, st = e.estimate_batch_device(da, db, B, K_vga)
    assert st[511] == capi.PAIR_NO_DESCRIPTORS and (np.delete(st, 511) == 0).all()
    assert (e.fetch_overflow(B) == 0).all()
    for u in range(U):
        sel = np.nonzero((idx == u) & (np.arange(B) != 511))[0]
        assert np.all(R[sel] == R[sel[0]]) and np.all(t[sel] == t[sel[0]]) and np.all(inl[sel] == inl[sel[0]]) and np.all(nm[sel] == nm[sel[0]])
        r = oracle.estimate_pose(i1[u], i2[u], K_vga, 1000, 500)
        assert np.array_equal(R[sel[0]], r["R"]) and np.array_equal(t[sel[0]], r["t"]) and inl[sel[0]] == r["inliers"]
    e.close()


# ------------------------------------------------------------------ the one collective: rpe_gather_poses over RCCL
def test_native_pose_gather_world1(capi, K_vga):
    """rpe_gather_poses (records packed on the device, ncclAllGather on the handle's stream, no torch) with a
    one-rank communicator on the 1-GPU box: the gathered records are the batch's results, padding is dropped,
    pair indices carry the shard offset; the scalar max-reduce / barrier round-trips."""
    from relative_pose_estimation_amd import sharding, synthetic
    i1, i2, _, _ = synthetic.make_batch(3, K_vga, cfg=2)
    i1 = np.concatenate([i1, np.full((1, 480, 640), 60, np.uint8)]); i2 = np.concatenate([i2, i2[:1]])   # a failing pair too
    e = capi.Engine(640, 480, max_batch=8, nfeatures=1000, max_matches=500)
    da, db = e.upload(i1), e.upload(i2)
    e.enqueue_batch_device(da, db, 4, K_vga)
    comm = sharding.PoseComm(e, 0, 1, tag=f"pytest_{__import__('os').getpid()}")
    rec = comm.gather(4, 8, first_pair=1000)                      # per_rank 8 > n_local 4: four padding records dropped
    R, t, inl, nm, st = e.fetch_results(4)
    assert rec.dtype == sharding.RECORD_DTYPE and rec["pair"].tolist() == [1000, 1001, 1002, 1003]
    assert np.array_equal(rec["R"].reshape(4, 3, 3), R) and np.array_equal(rec["t"].reshape(4, 3, 1), t)
    assert np.array_equal(rec["inliers"], inl) and np.array_equal(rec["status"], st) and np.array_equal(rec["n_matches"], nm)
    assert st[3] == capi.PAIR_NO_DESCRIPTORS and (st[:3] == 0).all()
    assert comm.max(3.25) == 3.25
    comm.barrier()
    comm.close()
    e.close()


def test_level0_in_place_and_copied_agree(capi, K_vga):
    """Level 0 of the ORB pyramid is read in place from the caller's device batches when the width is a multiple of 16
    and the batches are 16-byte aligned; a batch at an unaligned device address takes the copy into the pyramid buffer.
    Both routes give the same bits, and the pyramid debug fetch returns level 0 either way."""
    import ctypes
    from relative_pose_estimation_amd import synthetic
    i1, i2, _, _ = synthetic.make_batch(2, K_vga, cfg=2)
    e = capi.Engine(640, 480, max_batch=2, nfeatures=1000, max_matches=500)
    da, db = e.upload(i1), e.upload(i2)
    e.enqueue_batch_device(da, db, 2, K_vga)
    ref = e.fetch_results(2)
    pyr = e.orb_debug_fetch(2, 0)                                   # slot 2 = first image of the second batch
    assert np.array_equal(pyr[:640 * 480].reshape(480, 640), i2[0])
    pad1 = np.concatenate([np.zeros(4, np.uint8), i1.reshape(-1)]); pad2 = np.concatenate([np.zeros(4, np.uint8), i2.reshape(-1)])
    ua, ub = e.upload(pad1), e.upload(pad2)
    e.enqueue_batch_device(ctypes.c_void_p(ua.value + 4), ctypes.c_void_p(ub.value + 4), 2, K_vga)
    got = e.fetch_results(2)
    for a, b in zip(ref, got):
        assert np.array_equal(a, b)
    assert (ref[4] == 0).all()
    assert np.array_equal(e.orb_debug_fetch(2, 0), pyr)
    e.close()


def test_result_block_fetch_paths(capi, K_vga):
    """rpe_fetch_results copies the single result block [R | t | inliers | status | n_matches] to pinned memory in one piece,
    or only the used part of every section when the batch is small against max_batch: both give what a tight handle gives."""
    from relative_pose_estimation_amd import synthetic
    i1, i2, _, _ = synthetic.make_batch(3, K_vga, cfg=2)
    tight = capi.Engine(640, 480, max_batch=3, nfeatures=1000, max_matches=500)
    ref = tight.estimate_batch(i1, i2, K_vga)
    tight.close()
    wide = capi.Engine(640, 480, max_batch=16, nfeatures=1000, max_matches=500)       # 3 * 4 < 16: section copies
    got = wide.estimate_batch(i1, i2, K_vga)
    got2 = wide.estimate_batch(np.concatenate([i1, i1]), np.concatenate([i2, i2]), K_vga)   # 6 * 4 >= 16: whole block
    wide.close()
    for a, b, c in zip(ref, got, got2):
        assert np.array_equal(a, b) and np.array_equal(a, c[:3]) and np.array_equal(a, c[3:])
    assert (ref[4] == 0).all()


# ------------------------------------------------------------------ roofline calibration entry points
def test_calibration_entry_points(capi):
    """rpe_calibrate_valu / rpe_calibrate_hbm run and give physically possible rates: integer and f64 instructions issue
    at most once per 2 cycles per SIMD (1.23e12 wave-instructions/s at 2.4 GHz), HBM reads below the 8 TB/s spec peak"""
    e = capi.Engine(640, 480, max_batch=512, nfeatures=1000)          # 1.6 GB of pyramid: larger than the Infinity Cache
    for kind in (0, 4, 6):
        name, rate = e.calibrate_valu(kind, 8)
        assert name and 1e11 < rate < 1.3e12, (name, rate)
    name1, r1 = e.calibrate_valu(0, 1)
    assert r1 <= e.calibrate_valu(0, 8)[1] * 1.05                     # one wave per SIMD never beats eight
    bw = e.calibrate_hbm()
    assert 1e12 < bw < 8e12, bw
    e.close()

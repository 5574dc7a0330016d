"""Drop-in PoseEstimator backed by librpe_amd.so (HIP kernels on MI355X).

Mirrors reference src/core/pose_estimator.py: same constructor signature and
defaults (:19-32), estimate(img1, img2, R_prev=None) -> (R, t) (:487-569),
estimate_with_debug(...) -> dict with the reference's keys (:571-688, dict
:624-633), same exception types and messages (:96, :129, :508-509, :514-515,
:529-530).  Added: estimate_batch() for many pairs per call.

Scope: the feature -> match -> essential -> pose path on the GPU: ORB or SIFT features, Hamming or L2 matcher
(every combination cv2 can run: ORB + Hamming, ORB + L2, SIFT + L2; SIFT + Hamming constructs, as in the reference,
and fails at the first estimate with cv2's batchDistance message, where the reference's match() fails).  Added, opt-in
and NOT in the reference: `ratio` = Lowe ratio test instead of crossCheck; `keypoint_order` = which cv2 build's
keypoint order to reproduce ("libstdc++": the Linux wheels, default; "msvc": the Windows wheels).  VP refinement
(:160-481, :536-567) is the reference's CPU post-step on R; it is applied exactly where the
reference applies it -- `use_vp_refinement` set and `R_prev` given -- by vp_refinement.py on top of the
library's LSD restatement (host code, as in the reference).
"""
import numpy as np

from . import _capi, vp_refinement


class PoseEstimator:
    def __init__(self,
                 camera_matrix,
                 feature_method="ORB",
                 norm_type="Hamming",
                 max_matches=500,
                 nfeatures=4000,
                 use_vp_refinement=False,
                 vp_max_lines=120,
                 vp_max_pairs=3000,
                 vp_acc_min=8e5,
                 vp_vp2_min=8000.0,
                 vp_iters=12,
                 vp_lm_lambda=1e-2,
                 vp_cost_improve_eps=1e-3,
                 device=0,
                 max_batch=1,
                 ratio=None,
                 keypoint_order="libstdc++"):
        self.K = np.asarray(camera_matrix, dtype=np.float64)
        self.feature_method = feature_method
        self.norm_type = norm_type
        self.max_matches = max_matches
        self.nfeatures = nfeatures
        self.use_vp_refinement = use_vp_refinement
        self.vp_max_lines = vp_max_lines
        self.vp_max_pairs = vp_max_pairs
        self.vp_acc_min = vp_acc_min
        self.vp_vp2_min = vp_vp2_min
        self.vp_iters = vp_iters
        self.vp_lm_lambda = vp_lm_lambda
        self.vp_cost_improve_eps = vp_cost_improve_eps
        self.device = device
        self.max_batch = max_batch
        self.ratio = ratio            # None = the reference's crossCheck matcher; a float = knnMatch(k=2) + Lowe ratio (extension)
        # same validation order and messages as _create_feature_extractor / _create_matcher
        method = self.feature_method.upper()
        if method == "ORB":
            self._feature = _capi.FEATURE_ORB
        elif method == "SIFT":
            self._feature = _capi.FEATURE_SIFT
        else:
            raise ValueError(f"Unknown feature extraction method: {method}")
        norm = self.norm_type.upper()
        if norm == "HAMMING":
            self._norm = _capi.NORM_HAMMING
        elif norm == "L2":
            self._norm = _capi.NORM_L2
        else:
            raise ValueError(f"Unknown norm type: {norm}")
        # cv2 builds the SIFT + NORM_HAMMING matcher too (pose_estimator.py:131) and only its match() (:144) raises, on the
        # float descriptors: so does this class -- construction succeeds, the first estimate raises (_check_runnable)
        self._unrunnable = (self._feature, self._norm) == (_capi.FEATURE_SIFT, _capi.NORM_HAMMING)
        if keypoint_order not in ("libstdc++", "msvc"):
            raise ValueError(f"Unknown keypoint order: {keypoint_order}")
        self.keypoint_order = keypoint_order
        self._engines = {}

    def _check_runnable(self):
        if self._unrunnable:
            # cv2.error is not a RuntimeError; the text is batchDistance's (core/batch_distance.cpp): type CV_32F = 5,
            # dtype CV_32S = 4, NORM_HAMMING = 6
            raise RuntimeError("OpenCV: (-210:Unsupported format or combination of formats) The combination of type=5, dtype=4 "
                               "and normType=6 is not supported in function 'batchDistance'")

    def _engine(self, height, width, batch):
        self._check_runnable()
        key = (height, width)
        eng = self._engines.get(key)
        if eng is None or eng.max_batch < batch:
            if eng is not None:
                eng.close()
            # SIFT: the reference's cv2.SIFT_create() takes no arguments (pose_estimator.py:93-94; nfeatures is documented
            # "ORB only", :41), so nothing is removed by response: nfeatures = 0 asks the library for exactly that.  The
            # arrays hold SIFT_UNCAPPED_CAPACITY keypoints per image; an image with more carries OVF_SIFT_KEYPOINTS.
            sift = self._feature == _capi.FEATURE_SIFT
            nf = 0 if sift else self.nfeatures
            cap = (_capi.SIFT_UNCAPPED_CAPACITY if sift else nf) + 64
            # max_matches=None is the reference's "no truncation" (pose_estimator.py:150-151): every cross-checked
            # match is kept, at most one per keypoint = the keypoint capacity (the RANSAC tables stop at 8064 matches)
            mm = min(self.max_matches if self.max_matches is not None else cap, cap, _capi.MAX_MATCHES_LIMIT)
            eng = _capi.Engine(width, height, max_batch=max(batch, self.max_batch), nfeatures=nf,
                               max_matches=mm, device=self.device, feature_method=self._feature, norm_type=self._norm,
                               match_mode=_capi.MATCH_RATIO if self.ratio is not None else _capi.MATCH_CROSSCHECK,
                               match_ratio=float(self.ratio) if self.ratio is not None else 0.75,
                               stl_runtime=_capi.STL_MSVC if self.keypoint_order == "msvc" else _capi.STL_LIBSTDCXX)
            self._engines[key] = eng
        return eng

    @staticmethod
    def _raise_for(status, n_matches):
        if status == _capi.PAIR_NO_DESCRIPTORS:
            raise RuntimeError("Could not compute descriptors for one of the images.")
        if status == _capi.PAIR_INSUFFICIENT_MATCHES:
            raise RuntimeError(f"Insufficient matches: {n_matches} (minimum 5 required)")
        if status == _capi.PAIR_NO_ESSENTIAL:
            raise RuntimeError("Could not estimate Essential matrix.")
        if status == _capi.PAIR_AMBIGUOUS_ESSENTIAL:
            # exactly 5 matches, several five-point models: cv2.findEssentialMat returns them stacked (3n x 3) and the
            # reference's cv2.recoverPose(E, ...) call (pose_estimator.py:533) raises cv2.error with this assertion
            raise RuntimeError("OpenCV: (-215:Assertion failed) E.cols == 3 && E.rows == 3 in function 'decomposeEssentialMat' "
                               "(findEssentialMat returned several stacked models for exactly 5 matches)")

    @staticmethod
    def _gray(img):
        img = np.asarray(img)
        if img.ndim != 2 or img.dtype != np.uint8:
            raise ValueError("expected a 2-D uint8 grayscale image")
        return img

    def estimate_batch(self, imgs1, imgs2):
        """(R[B,3,3], t[B,3,1], inliers[B], status[B]); a failing pair never aborts the batch."""
        imgs1 = np.ascontiguousarray(imgs1, np.uint8); imgs2 = np.ascontiguousarray(imgs2, np.uint8)
        B, H, W = imgs1.shape
        eng = self._engine(H, W, B)
        R, t, inl, nm, st = eng.estimate_batch(imgs1, imgs2, self.K)
        self._last_n_matches = nm
        self._last_engine, self._last_pairs = eng, B
        return R, t, inl, st

    def last_overflow(self):
        """OVF_* capacity flags (see _capi) of the pairs of the last estimate_batch / estimate_sequence call: nonzero
        where a fixed-size GPU workspace truncated a list cv2 would have kept whole (status stays OK)."""
        return self._last_engine.fetch_overflow(self._last_pairs)

    def estimate_sequence(self, frames):
        """Relative poses of consecutive frames (frame i -> i+1): the pair loop of the reference's
        BatchProcessor.process_sequence (batch_processor.py:71-109) with features extracted once
        per frame.  Returns (R[F-1,3,3], t[F-1,3,1], inliers[F-1], status[F-1])."""
        frames = np.ascontiguousarray(frames, np.uint8)
        F, H, W = frames.shape
        eng = self._engine(H, W, F - 1)
        R, t, inl, nm, st = eng.estimate_stream(frames, self.K)
        self._last_n_matches = nm
        self._last_engine, self._last_pairs = eng, F - 1
        return R, t, inl, st

    def _vp_refine(self, R_rel, R_prev, img1, img2):
        """pose_estimator.py:536-567: (R_rel', vp_used, vp_debug)"""
        return vp_refinement.refine_relative_rotation(
            R_rel, R_prev, img1, img2, self.K, max_lines=self.vp_max_lines, max_pairs=self.vp_max_pairs,
            acc_min=self.vp_acc_min, vp2_min=self.vp_vp2_min, iters=self.vp_iters, lm_lambda=self.vp_lm_lambda,
            cost_improve_eps=self.vp_cost_improve_eps)

    def estimate(self, img1, img2, R_prev=None):
        img1 = self._gray(img1); img2 = self._gray(img2)
        eng = self._engine(img1.shape[0], img1.shape[1], 1)
        R, t, inl, nm, st = eng.estimate_batch(img1[None], img2[None], self.K)
        self._last_n_matches, self._last_engine, self._last_pairs = nm, eng, 1
        self._raise_for(int(st[0]), int(nm[0]))
        R_rel = R[0]
        if self.use_vp_refinement and R_prev is not None:
            R_rel, _, _ = self._vp_refine(R_rel, R_prev, img1, img2)
        return R_rel, t[0]

    def estimate_with_debug(self, img1, img2, R_prev=None):
        img1 = self._gray(img1); img2 = self._gray(img2)
        eng = self._engine(img1.shape[0], img1.shape[1], 1)
        R, t, inl, nm, st = eng.estimate_batch(img1[None], img2[None], self.K)
        self._last_n_matches, self._last_engine, self._last_pairs = nm, eng, 1
        self._raise_for(int(st[0]), int(nm[0]))
        p1, p2 = eng.fetch_matched_points(1)
        n = int(nm[0])
        info = {
            'R': R[0],
            't': t[0],
            'num_matches': n,
            'pts1': p1[0, :n].copy(),
            'pts2': p2[0, :n].copy(),
            'inliers': int(inl[0]),
            'vp_used': False,
            'vp_debug': {},
        }
        if self.use_vp_refinement and R_prev is not None:
            R_rel, used, dbg = self._vp_refine(R[0], R_prev, img1, img2)
            info['vp_debug'] = dbg
            if used:
                info['R'] = R_rel
                info['vp_used'] = True
        return info

    def close(self):
        for e in self._engines.values():
            e.close()
        self._engines = {}


def estimate_relative_pose(img1, img2, K, **kwargs):
    """north_star's call surface: (R, t, inliers) for one image pair."""
    est = PoseEstimator(K, **kwargs)
    try:
        d = est.estimate_with_debug(img1, img2)
    finally:
        est.close()
    return d['R'], d['t'], d['inliers']

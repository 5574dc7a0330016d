"""Latency of one rpe_estimate_batch call for small batches (B = 1, 8): the single-pair use of the drop-in class."""
import sys, time, numpy as np
sys.path.insert(0, '.')
from relative_pose_estimation_amd import _capi, synthetic, geometry
K = geometry.default_camera_matrix(640, 480)
i1, i2, _, _ = synthetic.make_batch(8, K, cfg=2)
for B in (1, 8):
    e = _capi.Engine(640, 480, max_batch=B, nfeatures=1000)
    a, b = e.upload(i1[:B]), e.upload(i2[:B])
    for _ in range(3): e.estimate_batch_device(a, b, B, K)
    t = time.perf_counter()
    for _ in range(20): e.estimate_batch_device(a, b, B, K)
    dt = (time.perf_counter() - t) / 20
    e.set_profiling(True); e.estimate_batch_device(a, b, B, K)
    print(B, f"{dt*1e3:.3f} ms per call", {k: round(v, 3) for k, v in e.stage_ms().items() if v > 0.01})
    t = time.perf_counter()
    for _ in range(20): e.estimate_batch(i1[:B], i2[:B], K)
    print("  host-image entry:", f"{(time.perf_counter()-t)/20*1e3:.3f} ms")
    e.close()

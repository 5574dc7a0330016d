"""PCIe-inclusive rate of the host-buffer boundary: rpe_estimate_batch (images in host memory) vs the
device-resident entry, 1024 VGA pairs."""
import sys, time, numpy as np
sys.path.insert(0, '.')
from relative_pose_estimation_amd import _capi, synthetic, geometry
K = geometry.default_camera_matrix(640, 480)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
i1, i2, _, _ = synthetic.make_batch(min(B, 128), K, cfg=2, workers=16)
reps = -(-B // len(i1))
i1 = np.ascontiguousarray(np.concatenate([i1] * reps)[:B]); i2 = np.ascontiguousarray(np.concatenate([i2] * reps)[:B])
e = _capi.Engine(640, 480, max_batch=B, nfeatures=1000)
a, b = e.upload(i1), e.upload(i2)
for name, fn in (("device-resident", lambda: e.estimate_batch_device(a, b, B, K)), ("host images", lambda: e.estimate_batch(i1, i2, K))):
    for _ in range(2): fn()
    t = time.perf_counter()
    for _ in range(5): fn()
    dt = (time.perf_counter() - t) / 5
    print(f"{name:16s} {dt*1e3:8.2f} ms per {B} pairs = {B/dt:9.0f} pairs/s")
e.close()

"""Image ingest in front of the hot path.

Same call surface as reference src/utils/image_loader.py (:9-30 load_image, :33-47
load_image_pair): decode a file into a colour array, optionally turn it into the uint8
grayscale image PoseEstimator.estimate() takes.  The reference decodes with cv2.imread and
converts with cv2.cvtColor(BGR2GRAY); here the file is decoded on the host (PIL, lossless
for PNG) and the gray conversion runs on the GPU with cv2's fixed-point weights
(rpe_bgr_to_gray, csrc/rpe_api.hip).  No CPU fallback for the conversion.
"""
import numpy as np

from . import _capi

_ctx = {}


def _engine(device=0):
    """A minimal handle (smallest legal workspace) that only owns a stream for the ingest kernel."""
    eng = _ctx.get(device)
    if eng is None:
        eng = _capi.Engine(96, 96, max_batch=1, nfeatures=64, max_matches=8, device=device)
        _ctx[device] = eng
    return eng


def close():
    for e in _ctx.values():
        e.close()
    _ctx.clear()


def decode_rgb(path):
    """File -> (H, W, 3) uint8 in R, G, B order.  Unreadable file: FileNotFoundError with the
    reference's message (image_loader.py:24-25)."""
    from PIL import Image
    try:
        with Image.open(path) as im:
            return np.asarray(im.convert("RGB"), dtype=np.uint8)
    except (OSError, ValueError):
        raise FileNotFoundError(f"Could not read image from: {path}") from None


def load_image(path, to_gray=True, device=0):
    rgb = decode_rgb(str(path))
    if not to_gray:
        return np.ascontiguousarray(rgb[..., ::-1])            # cv2.imread layout: B, G, R
    return _engine(device).bgr_to_gray(rgb, order=_capi.ORDER_RGB)


def load_image_pair(path1, path2, to_gray=True, device=0):
    return load_image(path1, to_gray=to_gray, device=device), load_image(path2, to_gray=to_gray, device=device)

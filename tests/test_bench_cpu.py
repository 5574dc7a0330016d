"""bench.py's byte accounting and workload arithmetic (no GPU): level sizes against the oracle's pyramid
layout, algorithmic bytes against SURVEY 8(d)'s per-unit figures."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@pytest.mark.parametrize("W,H", [(640, 480), (1920, 1080), (848, 478), (96, 96), (333, 257)])
def test_level_sizes_match_the_pyramid_layout(oracle, W, H):
    import bench
    L = oracle.orb_layout(W, H, 1000)
    assert bench.orb_levels(W, H) == [(L.w[l], L.h[l]) for l in range(12)]


def test_algorithmic_bytes():
    import bench
    P = 1590354                                               # VGA pyramid pixels
    assert sum(w * h for w, h in bench.orb_levels(640, 480)) == P
    assert bench.stage_bytes("match", 640, 480, P, 1000, 500) == 72000            # (N1+N2)*32 + N1*8
    assert bench.stage_bytes("pyramid", 640, 480, P, 1000, 500) == 2 * (307200 + P)
    fast = bench.stage_bytes("fast", 640, 480, P, 1000, 500)
    # list-emitting FAST: the border-filtered pyramid region of two images read ONCE (less than 2 P) + a few KB of
    # keypoint list entries and histograms; no dense NMS map is written any more
    assert 1.2 * P < fast < 2 * P
    assert bench.stage_bytes("select", 640, 480, P, 1000, 500) < 0.1 * P
    assert bench.stage_bytes("ransac", 640, 480, P, 1000, 500) == 8500
    oc = bench.sift_octaves(1920, 1080)
    assert oc[0] == (3840, 2160) and len(oc) == 10             # base 3840x2160, 10 octaves
    assert bench.stage_bytes("match", 1920, 1080, 0, 2048, 500, "SIFT") == 2 * 2048 * 128 + 2048 * 8
    px = sum(w * h for w, h in oc)
    assert bench.stage_bytes("pyramid", 1920, 1080, 0, 2048, 500, "SIFT") == 2 * (1920 * 1080 + 24 * px)
    with pytest.raises(KeyError):
        bench.stage_bytes("nope", 640, 480, P, 1000, 500)

#!/usr/bin/env python3
"""Per row: the reference's five sampled matches among the oracle's matches, the rank interval each may take under
level-major order with any within-level order, and the iteration of cv2's sample stream that comes closest
(max deviation from the intervals; 0 = consistent with an identical sorted list up to the within-level order)."""
import argparse
import os
import sys
from concurrent.futures import ProcessPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle                                    # noqa: E402
from tests import reference_rows as rr                       # noqa: E402
from tools.forensic import ref_rel_rotation, exact_fits      # noqa: E402

VARIANTS = []


def one(args):
    img1, img2, K, R = args
    for k, v in VARIANTS:
        oracle.set_variant(k, v)
    k1, d1 = oracle.orb_detect_and_compute(img1, 4000)
    k2, d2 = oracle.orb_detect_and_compute(img2, 4000)
    q, t, d = oracle.match_hamming(d1, d2, 100000)
    Mall = len(q)
    M = min(500, Mall)
    p1 = np.stack([k1["x"][q], k1["y"][q]], 1)[:M]; p2 = np.stack([k2["x"][t], k2["y"][t]], 1)[:M]
    n, info = exact_fits(R, p1, p2, K, 1e-9)
    if n < 5:
        return n, Mall, None
    pos = info[0][:5]
    iv = []
    for p in pos:
        lo = int((d[:Mall] < d[p]).sum())
        grp = np.nonzero(d[:Mall] == d[p])[0]
        lv = k1["octave"][q[grp]]; mylv = k1["octave"][q[p]]
        lo2 = lo + int((lv < mylv).sum()); hi2 = lo2 + int((lv == mylv).sum())
        iv.append((lo2, hi2 - 1))
    iv.sort()
    S = np.sort(oracle.ransac_subsets(M, 1000), 1)
    lo = np.array([a for a, b in iv]); hi = np.array([b for a, b in iv])
    dev = np.where(S < lo, S - lo, np.where(S > hi, S - hi, 0))
    m = np.abs(dev).max(1)
    k = int(m.argmin())
    return n, Mall, (pos.tolist(), iv, k, S[k].tolist(), dev[k].tolist(), d[pos].tolist())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--datasets", default="phone,sim")
    ap.add_argument("--variant", action="append", default=[])
    a = ap.parse_args()
    for kv in a.variant:
        k, v = kv.split("=")
        VARIANTS.append((int(k), int(v)))
        oracle.set_variant(int(k), int(v))
    for name in a.datasets.split(","):
        ds = rr.load(name)
        out = oracle.estimate_pose_batch(ds["img1"], ds["img2"], ds["K"], 4000, 500, nthreads=8)
        jobs = [(ds["img1"][i], ds["img2"][i], ds["K"], ref_rel_rotation(ds, i, out["R"][i].reshape(3, 3))) for i in range(len(out))]
        with ProcessPoolExecutor(8) as ex:
            res = list(ex.map(one, jobs))
        for i, (n, Mall, info) in enumerate(res):
            if info is None:
                print(f"{name} frame {int(ds['frames2'][i]):4d}: matches {Mall} fits {n}")
            else:
                pos, iv, k, S, dev, dd = info
                print(f"{name} frame {int(ds['frames2'][i]):4d}: matches {Mall} fits {n} pos {pos} dist {dd} intervals {iv}  nearest iteration {k} {S} dev {dev}")


if __name__ == "__main__":
    main()

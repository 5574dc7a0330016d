#!/usr/bin/env python3
"""Hypothesis scoring against the reference's committed answers: for every row, how many of the five matches cv2's
winning RANSAC sample used (recovered from the CSV rotation, tools/forensic.py) are among the oracle's first
max_matches matches, and do their sorted positions equal one of cv2's fixed samples for that M (= identical sorted
match list at those ranks).  Run under the oracle's convention knobs to compare hypotheses."""
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


def one(args):
    R, p1, p2, K, M = args
    n, info = exact_fits(R, p1, p2, K, 1e-9)
    pos = info[0].tolist() if info else []
    hit = -1
    if n >= 5:
        S = np.sort(oracle.ransac_subsets(M, 1000), 1)
        eq = np.nonzero((S == np.array(pos[:5])[None, :]).all(1))[0]
        if len(eq):
            hit = int(eq[0])
    return n, pos, hit


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--datasets", default="phone,sim,salah")
    ap.add_argument("--variant", action="append", default=[])
    ap.add_argument("--verbose", action="store_true")
    a = ap.parse_args()
    for kv in a.variant:
        k, v = kv.split("=")
        oracle.set_variant(int(k), int(v))
    tot = [0, 0, 0, 0]
    for name in a.datasets.split(","):
        ds = rr.load(name)
        out, pts = oracle.estimate_pose_batch(ds["img1"], ds["img2"], ds["K"], 4000, 500, nthreads=8, return_points=True)
        jobs = []
        for i in range(len(out)):
            M = int(out["n_matches"][i])
            jobs.append((ref_rel_rotation(ds, i, out["R"][i].reshape(3, 3)), pts[i, 0, :M].copy(), pts[i, 1, :M].copy(), ds["K"], M))
        with ProcessPoolExecutor(8) as ex:
            res = list(ex.map(one, jobs))
        found = sum(min(n, 5) for n, _, _ in res)
        full = sum(n >= 5 for n, _, _ in res)
        hits = sum(h >= 0 for _, _, h in res)
        print(f"{name:6s}: {len(res)} rows; sampled matches found among the oracle's first 500: {found}/{5 * len(res)} (2 per row are free);"
              f" rows with all five: {full}; rows whose five sit exactly at one of cv2's samples: {hits}", flush=True)
        if a.verbose:
            for i, (n, pos, h) in enumerate(res):
                print(f"   frame {int(ds['frames2'][i]):4d} n {n} pos {pos[:6]} iter {h}")
        tot[0] += found; tot[1] += 5 * len(res); tot[2] += full; tot[3] += hits
    print("total", tot)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Forensics, second step: rank INTERVALS.  For the reference's five sampled matches (tools/forensic.py) print the
oracle's Hamming distance and the interval of sorted positions [#matches with smaller distance, #matches with distance <=)
each could take under ANY tie order, and look for iterations of cv2's sample stream (depends only on M) whose five
indices fall into those intervals."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle                                    # noqa: E402
from tests import reference_rows as rr                       # noqa: E402
from tools.forensic import ref_rel_rotation, exact_fits      # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--datasets", default="phone")
    ap.add_argument("--variant", action="append", default=[])
    ap.add_argument("--rows", default="")
    ap.add_argument("--all", action="store_true", help="search the full match list, not only the first 500")
    a = ap.parse_args()
    for kv in a.variant:
        k, v = kv.split("=")
        oracle.set_variant(int(k), int(v))
    for name in a.datasets.split(","):
        rows = [int(x) for x in a.rows.split(",")] if a.rows else None
        ds = rr.load(name, rows=rows)
        for i in range(len(ds["frames2"])):
            k1, d1 = oracle.orb_detect_and_compute(ds["img1"][i], 4000)
            k2, d2 = oracle.orb_detect_and_compute(ds["img2"][i], 4000)
            q, t, d = oracle.match_hamming(d1, d2, 100000)
            Mall = len(q)
            M = Mall if a.all else min(500, Mall)
            p1 = np.stack([k1["x"][q], k1["y"][q]], 1)[:M]; p2 = np.stack([k2["x"][t], k2["y"][t]], 1)[:M]
            out = oracle.estimate_pose(ds["img1"][i], ds["img2"][i], ds["K"], 4000, 500)
            R = ref_rel_rotation(ds, i, out["R"])
            n, info = exact_fits(R, p1, p2, ds["K"], 1e-9)
            print(f"{name} frame {int(ds['frames2'][i])}: kp {len(k1)}/{len(k2)} matches {Mall} (used {M}); exact fits {n}")
            if not info:
                continue
            iv = []
            for pos in info[0]:
                lo = int((d[:Mall] < d[pos]).sum()); hi = int((d[:Mall] <= d[pos]).sum())
                # level-major order is certain (orb.cpp computeKeyPoints appends level by level): tie-group members of a lower
                # query level come first, of a higher level later; only the same-level members are order-dependent
                grp = np.nonzero(d[:Mall] == d[pos])[0]
                lv = k1["octave"][q[grp]]; mylv = k1["octave"][q[pos]]
                lo2 = lo + int((lv < mylv).sum()); hi2 = lo2 + int((lv == mylv).sum())
                iv.append((lo2, hi2))
                same = grp[lv == mylv]
                print(f"    pos {pos:4d} dist {d[pos]:3d} interval [{lo},{hi}) level-major [{lo2},{hi2})  q {q[pos]} (lvl {mylv}, y {k1['ly'][q[pos]]} x {k1['lx'][q[pos]]} resp {k1['response'][q[pos]]:.3e}) t {t[pos]}"
                      f"  same-level ties: " + " ".join(f"(q{q[g]} y{k1['ly'][q[g]]} x{k1['lx'][q[g]]} r{k1['response'][q[g]]:.2e})" for g in same))
            if n >= 5:
                for Mc in sorted(set([min(500, Mall), 500])):
                    S = np.sort(oracle.ransac_subsets(Mc, 1000), 1)
                    ivs = sorted(iv)[:5]
                    ok = np.ones(len(S), bool)
                    for c, (lo, hi) in enumerate(ivs):
                        ok &= (S[:, c] >= lo) & (S[:, c] < hi)
                    print(f"    cv2 stream M={Mc}: iterations inside all five intervals: {np.nonzero(ok)[0].tolist()}")


if __name__ == "__main__":
    main()

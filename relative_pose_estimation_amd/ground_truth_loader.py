"""Ground-truth pose table (frame x y z roll pitch yaw), the data format on the input side of
the sequence front-end.  Same method names, return shapes and error behaviour as reference
src/core/ground_truth_loader.py (:28-40 load, :42-69 get_pose, :71-133 index / trajectory
accessors); backed by a plain numpy table instead of a pandas frame."""
from pathlib import Path

import numpy as np

_COLUMNS = ("frame", "x", "y", "z", "roll", "pitch", "yaw")


class GroundTruthLoader:
    def __init__(self, gt_path):
        self.gt_path = Path(gt_path)
        self.df = None            # dict column -> array once loaded (the reference keeps a DataFrame here)

    def load(self):
        with open(self.gt_path) as f:
            header = f.readline().split()
            rows = [ln.split() for ln in f if ln.strip()]
        missing = [c for c in _COLUMNS if c not in header]
        if missing:
            raise KeyError(f"ground truth file lacks columns {missing}")
        table = np.array(rows, dtype=np.float64).reshape(len(rows), len(header))
        self.df = {name: table[:, header.index(name)] for name in header}
        self.df["frame"] = self.df["frame"].astype(np.int64)
        return self.df

    def _table(self):
        if self.df is None:
            raise RuntimeError("Ground truth not loaded. Call load() first.")
        return self.df

    def get_pose(self, frame_idx):
        t = self._table()
        hit = np.flatnonzero(t["frame"] == frame_idx)
        if hit.size == 0:
            raise KeyError(f"Frame {frame_idx} not found in ground truth data")
        i = hit[0]
        pose = {k: float(t[k][i]) for k in _COLUMNS[1:]}
        pose["frame"] = int(t["frame"][i])
        return pose

    def _every(self, step):
        t = self._table()
        return t["frame"] % step == 0

    def get_frame_indices(self, step=1):
        return self._table()["frame"][self._every(step)]

    def get_all_frames(self):
        return self._table()["frame"]

    def get_trajectory(self, step=1):
        t = self._table(); m = self._every(step)
        return np.stack([t["x"][m], t["y"][m], t["z"][m]], axis=1)

    def get_orientations(self, step=1):
        t = self._table(); m = self._every(step)
        return np.stack([t["roll"][m], t["pitch"][m], t["yaw"][m]], axis=1)

"""TUM RGB-D on-disk sequence reader: the layout and quirks of the reference's
semantic-slam/data/tum_dataset.py:210-273, without the training-pair / augmentation logic.

  <sequence>/rgb/*.png          sorted by file name           (tum_dataset.py:212)
  <sequence>/depth/*.png        uint16, metres = value / 5000 (tum_dataset.py:139); paired with rgb BY SORTED INDEX,
                                truncated to the shorter list (:219-223) - not by associate.py
  <sequence>/groundtruth.txt    "ts tx ty tz qx qy qz qw"; each frame takes the pose with the nearest timestamp (:248-252)

Quirk kept on purpose: the frame timestamp is float(name.split('.')[0]), i.e. the file name's WHOLE-second part
(:216), so all frames within one second share a timestamp and therefore a pose.
Host-side I/O only (PIL decode); frames are handed to the GPU as (n, H, W, 3) uint8 batches.
"""
from __future__ import annotations

import os
from pathlib import Path

import numpy as np


def quat_to_matrix(qx, qy, qz, qw, tx, ty, tz) -> np.ndarray:
    """4x4 pose from a (normalised here) quaternion and a translation (tum_dataset.py:257-272)."""
    n = np.sqrt(qx * qx + qy * qy + qz * qz + qw * qw)
    qx, qy, qz, qw = qx / n, qy / n, qz / n, qw / n
    T = np.eye(4)
    T[:3, :3] = [[1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qz * qw), 2 * (qx * qz + qy * qw)],
                 [2 * (qx * qy + qz * qw), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qx * qw)],
                 [2 * (qx * qz - qy * qw), 2 * (qy * qz + qx * qw), 1 - 2 * (qx * qx + qy * qy)]]
    T[:3, 3] = [tx, ty, tz]
    return T


class TUMSequence:
    def __init__(self, dataset_root: str, sequence: str = "", max_frames: int | None = None):
        root = Path(dataset_root)
        cand = root / sequence
        self.sequence_dir = cand if sequence and cand.exists() else root   # root may point at the sequence itself (:57-61)
        self.rgb_dir, self.depth_dir = self.sequence_dir / "rgb", self.sequence_dir / "depth"
        self.gt_file = self.sequence_dir / "groundtruth.txt"
        assert self.rgb_dir.exists(), f"RGB directory not found: {self.rgb_dir}"
        self.rgb_files = sorted(f for f in os.listdir(self.rgb_dir) if f.endswith(".png"))
        self.depth_files = sorted(f for f in os.listdir(self.depth_dir) if f.endswith(".png")) if self.depth_dir.exists() else []
        self.timestamps = [float(f.split(".")[0]) for f in self.rgb_files]
        if self.depth_files:
            n = min(len(self.rgb_files), len(self.depth_files))
            self.rgb_files, self.depth_files, self.timestamps = self.rgb_files[:n], self.depth_files[:n], self.timestamps[:n]
        self.poses = self._load_groundtruth() if self.gt_file.exists() else None
        if max_frames is not None:
            self.rgb_files, self.depth_files = self.rgb_files[:max_frames], self.depth_files[:max_frames]
            self.timestamps = self.timestamps[:max_frames]
            if self.poses is not None:
                self.poses = self.poses[:max_frames]

    def __len__(self):
        return len(self.rgb_files)

    def n_pairs(self, frame_spacing: int = 1) -> int:
        """What the reference's TUMDataset.__len__ returns (tum_dataset.py:119-120): frame pairs (i, i + spacing)."""
        return max(0, len(self.rgb_files) - frame_spacing)

    def _load_groundtruth(self) -> np.ndarray:
        ts_gt, poses = [], []
        with open(self.gt_file) as f:
            for line in f:
                if line.startswith("#"):
                    continue
                p = line.strip().split()
                if len(p) < 8:
                    continue
                v = [float(x) for x in p[:8]]
                ts_gt.append(v[0])
                poses.append(quat_to_matrix(v[4], v[5], v[6], v[7], v[1], v[2], v[3]))
        ts_gt = np.array(ts_gt)
        return np.array([poses[int(np.argmin(np.abs(ts_gt - ts)))] for ts in self.timestamps])

    def rgb_path(self, i: int) -> Path:
        return self.rgb_dir / self.rgb_files[i]

    def load_rgb(self, indices) -> np.ndarray:
        """(n, H, W, 3) uint8, PIL 'RGB' conversion as at visualize_matches_sequence.py:71."""
        from PIL import Image
        return np.stack([np.asarray(Image.open(self.rgb_path(i)).convert("RGB")) for i in indices])

    def load_depth(self, i: int) -> np.ndarray:
        """(H, W) float32 metres (loaded for completeness: no stage of the path consumes depth)."""
        from PIL import Image
        return np.asarray(Image.open(self.depth_dir / self.depth_files[i])).astype(np.float32) / 5000.0

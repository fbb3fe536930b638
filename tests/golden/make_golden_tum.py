#!/usr/bin/env python3
"""Golden fixture for the TUM directory reader, produced by RUNNING the reference's own
`data.tum_dataset.TUMDataset` (its __init__, `_load_associations` and `_load_groundtruth`, tum_dataset.py:27-95, 210-255)
on the tiny sequence directory that tests/synth.py:write_tum_sequence writes.  Only outputs are stored: file lists,
timestamps, poses, dataset lengths.  The third-party `torchvision.transforms` the module imports for its image pipeline
(unused by the loaders pinned here) is replaced by inert placeholders, as in make_golden.py.

Usage:  python tests/golden/make_golden_tum.py   ->  tests/golden/tum_reader.npz
"""
from __future__ import annotations

import os
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.dont_write_bytecode = True
REF = "/root/reference/semantic-slam"

import synth  # noqa: E402


class _Inert:
    def __init__(self, *a, **k):
        pass


def main():
    tv, tvt = types.ModuleType("torchvision"), types.ModuleType("torchvision.transforms")
    for attr in ["Compose", "Resize", "ToTensor", "Normalize", "ColorJitter", "GaussianBlur", "RandomApply"]:
        setattr(tvt, attr, _Inert)
    tv.transforms = tvt
    sys.modules["torchvision"], sys.modules["torchvision.transforms"] = tv, tvt
    sys.path.insert(0, REF)
    from data.tum_dataset import TUMDataset

    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        seq_name = "rgbd_dataset_freiburg1_desk"
        synth.write_tum_sequence(os.path.join(tmp, seq_name))
        for tag, kw in [("full", dict(dataset_root=tmp, sequence=seq_name)),
                        ("max4", dict(dataset_root=tmp, sequence=seq_name, max_frames=4, frame_spacing=2)),
                        ("direct", dict(dataset_root=os.path.join(tmp, seq_name), sequence="not_a_subdir"))]:
            ds = TUMDataset(input_size=448, is_train=False, **kw)
            out[f"{tag}_rgb"] = np.frombuffer(",".join(ds.rgb_files).encode(), np.uint8)
            out[f"{tag}_depth"] = np.frombuffer(",".join(ds.depth_files).encode(), np.uint8)
            out[f"{tag}_timestamps"] = np.asarray(ds.timestamps, np.float64)
            out[f"{tag}_poses"] = np.asarray(ds.poses, np.float64)
            out[f"{tag}_len"] = np.int64(len(ds))
        out["quat_pose"] = TUMDataset._quat_to_matrix(0.3, -0.1, 0.7, 1.2, 1.0, 2.0, 3.0)
    np.savez_compressed(os.path.join(HERE, "tum_reader.npz"), **out)
    print("wrote tum_reader.npz:", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()

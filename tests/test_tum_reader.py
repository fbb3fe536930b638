"""TUM RGB-D directory reader: layout and quirks of the reference's data/tum_dataset.py:210-273."""
import numpy as np
import pytest


def _write_sequence(root, n_rgb=7, n_depth=6):
    from PIL import Image
    (root / "rgb").mkdir(parents=True)
    (root / "depth").mkdir()
    rng = np.random.default_rng(0)
    names = [f"1305031{452 + i // 3}.{(791720 + 33000 * i) % 1000000:06d}.png" for i in range(n_rgb)]
    imgs = []
    for nm in names:
        a = rng.integers(0, 255, (24, 32, 3), dtype=np.uint8)
        Image.fromarray(a).save(root / "rgb" / nm)
        imgs.append(a)
    for i in range(n_depth):
        d = rng.integers(2500, 25000, (24, 32)).astype(np.uint16)
        Image.fromarray(d).save(root / "depth" / f"1305031{452 + i // 3}.{(800000 + 33000 * i) % 1000000:06d}.png")
    with open(root / "groundtruth.txt", "w") as f:
        f.write("# ground truth trajectory\n# timestamp tx ty tz qx qy qz qw\nshort line\n")
        for k in range(8):
            f.write(f"{1305031452.0 + 0.5 * k:.4f} {0.1 * k} 0 1 0 0 {np.sin(0.1 * k)} {np.cos(0.1 * k)}\n")
    return names, imgs


def test_tum_sequence_layout_and_quirks(tmp_path):
    from sslam_amd.tum import TUMSequence, quat_to_matrix
    names, imgs = _write_sequence(tmp_path / "rgbd_dataset_freiburg1_desk")
    seq = TUMSequence(str(tmp_path), "rgbd_dataset_freiburg1_desk")
    assert len(seq) == 6                                              # truncated to the shorter (depth) list
    assert seq.rgb_files == sorted(names)[:6]
    # the timestamp is the WHOLE-second part of the file name (tum_dataset.py:216)
    assert seq.timestamps == [float(n.split(".")[0]) for n in sorted(names)[:6]]
    assert len(set(seq.timestamps)) == 2
    assert np.array_equal(seq.load_rgb([0, 3])[1], imgs[names.index(sorted(names)[3])])
    d = seq.load_depth(2)
    assert d.dtype == np.float32 and 0.5 <= d.min() and d.max() <= 5.0
    # nearest-timestamp pose; frames sharing a whole-second stamp share the pose
    assert seq.poses.shape == (6, 4, 4)
    assert np.array_equal(seq.poses[0], seq.poses[1]) and not np.array_equal(seq.poses[0], seq.poses[5])
    T = quat_to_matrix(0, 0, np.sin(0.2), np.cos(0.2), 1, 2, 3)
    assert np.allclose(T[:3, :3] @ T[:3, :3].T, np.eye(3)) and np.allclose(T[:3, 3], [1, 2, 3])
    assert np.isclose(T[0, 0], np.cos(0.4))
    # dataset_root may point straight at the sequence; max_frames truncates everything
    seq2 = TUMSequence(str(tmp_path / "rgbd_dataset_freiburg1_desk"), max_frames=4)
    assert len(seq2) == 4 and seq2.poses.shape[0] == 4
    with pytest.raises(AssertionError):
        TUMSequence(str(tmp_path / "nope"))


def test_tum_sequence_equals_the_reference_loader(tmp_path):
    """Pinned: tests/golden/tum_reader.npz holds what the reference's own TUMDataset (data/tum_dataset.py:27-95, 210-255,
    run by tests/golden/make_golden_tum.py) loads from the directory synth.write_tum_sequence writes."""
    import os

    import synth
    from sslam_amd.tum import TUMSequence, quat_to_matrix
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tum_reader.npz"))
    name = "rgbd_dataset_freiburg1_desk"
    created = synth.write_tum_sequence(str(tmp_path / name))
    assert created != sorted(created), "the fixture directory is written out of order on purpose"
    cases = {"full": (TUMSequence(str(tmp_path), name), 1),
             "max4": (TUMSequence(str(tmp_path), name, max_frames=4), 2),
             "direct": (TUMSequence(str(tmp_path / name), "not_a_subdir"), 1)}
    for tag, (seq, spacing) in cases.items():
        assert seq.rgb_files == bytes(g[f"{tag}_rgb"]).decode().split(","), tag
        assert seq.depth_files == bytes(g[f"{tag}_depth"]).decode().split(","), tag
        assert np.array_equal(np.asarray(seq.timestamps, np.float64), g[f"{tag}_timestamps"]), tag
        assert np.array_equal(seq.poses, g[f"{tag}_poses"]), tag            # same arithmetic: bit-equal float64
        assert seq.n_pairs(spacing) == int(g[f"{tag}_len"]), tag
    assert np.array_equal(quat_to_matrix(0.3, -0.1, 0.7, 1.2, 1.0, 2.0, 3.0), g["quat_pose"])

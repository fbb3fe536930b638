#!/usr/bin/env python3
"""The reference's OWN scripts driven against the drop-in (BUILD CONTAINER ONLY: needs /root/reference; nothing of the reference
is stored - this script imports its modules where they lie and records outcomes).

north_star: "keeping the repo's model/matcher Python API as a drop-in so train.py ... and the visualize_* scripts run unchanged".
What runs here, with `semantic-slam-master_amd/` AHEAD of `/root/reference/semantic-slam` on sys.path, so that the reference's
`from models.dino_backbone import DinoBackbone` (etc.) resolves to this repository's classes while `data/`, `losses/`, the
scripts themselves and `configs/train_config.yaml` are the reference's files, unmodified:

  (i)   visualize_matches_sequence.SequenceMatcher(checkpoint, configs/train_config.yaml, device="cpu")   (:28-67)
        - the checkpoint is a dict written with tests/synth.py weights under the reference's key names (train.py:582-590);
        its .extract(path) (:69-104) on three synthetic PNGs with the third-party ViT replaced by a token stand-in (as
        make_golden.py does: the pretrained timm model is a remote fetch), and its .match_with_quality (:106-197):
        keypoints, scores, intensities, descriptors and match pairs must equal tests/golden/e2e.npz - the fixture the SAME
        inputs produced through the reference's own model classes.
  (ii)  train.SemanticSLAMTrainer(config) (:38-160, the real constructor: drop-in modules, the reference's seven losses, AdamW,
        the reference's TUMDataset on a synthetic TUM directory) and trainer.train() for one epoch (:501-575): train_epoch
        (B = 4, _forward_pass :292-408 with gradients through selector / grid_sample / refiner, _find_matches :410-449,
        backward, clip_grad_norm_, optimizer.step), validate, save_checkpoint('best_model.pth').
  (iii) SequenceMatcher again, on the checkpoint train.py just wrote, with the in-repo ViT (random weights) inside: extract +
        match run end to end from a PNG.
  (iv)  the other callers of the path, each through its own constructor (YAML + that checkpoint) and its own entry on the synthetic
        TUM directory: visualize_matches.MatchVisualizer.extract_features / find_matches (M2; :70-124), and the four evaluation
        classes of test/*.py, which call backbone.eval() (eval-mode BatchNorm): TrackingTester.track_frame_sequence
        (M5; test_tracking.py:87-197), DescriptorQualityTester.test_sequence (M4 + the pose-derived ground truth;
        test_descriptor_quality.py:233-305), RepeatabilityTester.test_sequence (test_repeatability.py:130-216),
        PerformanceTester.measure_component_times (the reference's timing protocol; test_performance.py:78-144).  Their
        matchers' outputs are compared with the CPU oracle's M2 / M4 / M5 on the same descriptors (matching.py is HIP-only).

Third-party modules absent from this image: `torchvision.transforms` is given a PIL stand-in (Resize / ToTensor / Normalize /
Compose - what torchvision does for a PIL input), `cv2` / `wandb` / `timm` inert placeholders (unused on these paths: wandb is
switched off in the config, timm's absence selects the in-repo ViT definition in the drop-in).

Usage:  python tests/golden/run_reference_scripts.py [--json OUT]      exit code 0 = every assertion held
"""
from __future__ import annotations

import argparse
import copy
import json
import os
import sys
import tempfile
import types
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
TESTS = os.path.dirname(HERE)
ROOT = os.path.dirname(TESTS)
PKG = os.path.join(ROOT, "semantic-slam-master_amd")
REF = "/root/reference/semantic-slam"
sys.dont_write_bytecode = True


# ---------------------------------------------------------------------------------------------- third-party stand-ins
class Compose:
    def __init__(self, transforms):
        self.transforms = list(transforms)

    def __call__(self, x):
        for t in self.transforms:
            x = t(x)
        return x


class Resize:
    """torchvision.transforms.Resize on a PIL image: Image.resize((w, h), BILINEAR) (antialiased by Pillow)."""

    def __init__(self, size):
        self.size = (size, size) if isinstance(size, int) else tuple(size)

    def __call__(self, img):
        from PIL import Image
        return img.resize((self.size[1], self.size[0]), Image.BILINEAR)


class ToTensor:
    def __call__(self, img):
        import torch
        a = np.asarray(img)
        if a.ndim == 2:
            a = a[:, :, None]
        t = torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1)))
        return t.to(torch.float32).div(255) if t.dtype == torch.uint8 else t.to(torch.float32)


class Normalize:
    def __init__(self, mean, std):
        import torch
        self.mean = torch.tensor(mean, dtype=torch.float32).view(-1, 1, 1)
        self.std = torch.tensor(std, dtype=torch.float32).view(-1, 1, 1)

    def __call__(self, t):
        return (t - self.mean) / self.std


class _Inert:
    def __init__(self, *a, **k):
        pass


def install_stand_ins():
    tv, tvt = types.ModuleType("torchvision"), types.ModuleType("torchvision.transforms")
    tvt.Compose, tvt.Resize, tvt.ToTensor, tvt.Normalize = Compose, Resize, ToTensor, Normalize
    for name in ("ColorJitter", "GaussianBlur", "RandomApply"):          # augmentation is switched off in the config used here
        setattr(tvt, name, _Inert)
    tv.transforms = tvt
    sys.modules["torchvision"], sys.modules["torchvision.transforms"] = tv, tvt
    for name in ("cv2", "wandb"):
        sys.modules.setdefault(name, types.ModuleType(name))
    import matplotlib
    matplotlib.use("Agg")
    # the drop-in FIRST, the reference's tree second: `models.*` resolves here, `data.*`, `losses.*`, the scripts there
    for p in (REF, PKG, TESTS, ROOT):
        if p in sys.path:
            sys.path.remove(p)
    sys.path[:0] = [PKG, REF, TESTS, ROOT]


def token_dino():
    """Stand-in for the third-party ViT: hands back the tokens it was given (the call site is dino_backbone.py:85)."""
    import torch

    class TokenDino(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.anchor = torch.nn.Parameter(torch.zeros(1), requires_grad=False)
            self.embed_dim, self.tokens = 384, None

        def forward_features(self, images):
            return self.tokens

    return TokenDino()


# ---------------------------------------------------------------------------------------------- (i)
def part_sequence_matcher(tmp: str) -> dict:
    import torch
    from PIL import Image

    import synth
    import visualize_matches_sequence as vms
    import models.dino_backbone as mbb
    assert os.path.realpath(mbb.__file__).startswith(os.path.realpath(PKG)), "models.* must resolve to the drop-in"
    assert os.path.realpath(vms.__file__).startswith("/root/reference/"), "the script must be the reference's file"

    ckpt = os.path.join(tmp, "synthetic_checkpoint.pth")
    torch.save({"epoch": 0, "loss": 0.0,
                "selector_state_dict": {k: torch.from_numpy(v) for k, v in synth.selector_state(0).items()},
                "refiner_state_dict": {k: torch.from_numpy(v) for k, v in synth.refiner_state(0).items()}}, ckpt)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")          # "timm is not installed: ... RANDOM weights" - replaced by the token stand-in below
        m = vms.SequenceMatcher(ckpt, os.path.join(REF, "configs", "train_config.yaml"), device="cpu")
    assert type(m.backbone).__module__ == "models.dino_backbone" and m.backbone.embed_dim == 384
    m.backbone.dino = token_dino()
    toks, imgs = synth.token_sequence(3, 28), synth.image_sequence(3)
    g = np.load(os.path.join(HERE, "e2e.npz"))
    frames = []
    for i in range(3):
        path = os.path.join(tmp, f"frame{i}.png")
        Image.fromarray(imgs[i], "RGB").save(path)
        m.backbone.dino.tokens = torch.from_numpy(toks[i:i + 1])
        f = m.extract(path)                                                    # the reference's method, unchanged
        assert set(f) == {"image", "saliency", "keypoints_pixel", "scores", "intensity", "descriptors"}
        assert f["keypoints_pixel"].shape == (500, 2) and f["descriptors"].shape == (500, 128) and f["saliency"].shape == (28, 28)
        kp_patch = (f["keypoints_pixel"] - 8.0) / 16.0
        assert np.array_equal(kp_patch, g[f"f{i}_kp"]), f"frame {i}: keypoints differ from e2e.npz"
        assert np.array_equal(f["intensity"], g[f"f{i}_intensity"]), f"frame {i}: intensities differ"
        np.testing.assert_allclose(f["scores"], g[f"f{i}_scores"], rtol=0, atol=1e-6)
        np.testing.assert_allclose(f["descriptors"][::5], g[f"f{i}_desc_sub"], rtol=0, atol=1e-5)
        frames.append(f)
    n_matches = 0
    for a, b in [(0, 1), (1, 2), (0, 2)]:
        fa, fb = frames[a], frames[b]
        mt, q = m.match_with_quality(fa["descriptors"], fb["descriptors"], fa["scores"], fb["scores"], saliency_weight=0.3,
                                     min_saliency=0.5, min_descriptor_sim=0.7, intensity1=fa["intensity"],
                                     intensity2=fb["intensity"], min_intensity=0.15)
        assert mt.dtype == np.int64 and q.dtype == np.float32
        assert np.array_equal(mt, g[f"pair{a}{b}_matches"]), f"pair ({a}, {b}): match pairs differ from e2e.npz"
        np.testing.assert_allclose(q, g[f"pair{a}{b}_quality"], rtol=0, atol=1e-5)
        n_matches += len(mt)
    bit_equal_desc = all(np.array_equal(frames[i]["descriptors"][::5], g[f"f{i}_desc_sub"]) for i in range(3))
    return {"script": "visualize_matches_sequence.py", "class": "SequenceMatcher", "device": "cpu", "frames": 3, "pairs": 3,
            "matches": int(n_matches), "keypoints_equal_e2e_npz": True, "intensities_equal_e2e_npz": True,
            "match_pairs_equal_e2e_npz": True, "descriptors_bit_equal_e2e_npz": bool(bit_equal_desc),
            "descriptors_max_abs_err": float(max(np.abs(frames[i]["descriptors"][::5] - g[f"f{i}_desc_sub"]).max() for i in range(3)))}


# ---------------------------------------------------------------------------------------------- (ii) + (iii)
def part_train_then_match(tmp: str) -> dict:
    import torch
    import yaml

    import synth
    import train as ref_train
    import visualize_matches_sequence as vms
    assert os.path.realpath(ref_train.__file__).startswith("/root/reference/")
    torch.manual_seed(0)
    torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))
    root = os.path.join(tmp, "tum_rgbd")
    seqs = {"rgbd_dataset_synth_train": 6, "rgbd_dataset_synth_val": 3}
    for k, (name, nfr) in enumerate(seqs.items()):
        frames = synth.image_sequence(nfr, seed=11 + k)
        synth.write_tum_rgb_sequence(os.path.join(root, name), frames, with_depth=True)
    with open(os.path.join(REF, "configs", "train_config.yaml")) as f:
        config = yaml.safe_load(f)                                   # the reference's config, then only what the data forces:
    config["dataset"].update(root=root, train_sequences=["rgbd_dataset_synth_train"], val_sequences=["rgbd_dataset_synth_val"],
                             augmentation=None)                       # ColorJitter / GaussianBlur are torchvision (absent)
    config["training"].update(epochs=1, num_workers=0, save_dir=os.path.join(tmp, "checkpoints"))
    config["logging"]["use_wandb"] = False
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")          # the drop-in's "timm is not installed: in-repo ViT with RANDOM weights"
        trainer = ref_train.SemanticSLAMTrainer(copy.deepcopy(config))          # the reference's constructor, unchanged
    for mod, name in ((trainer.backbone, "DinoBackbone"), (trainer.selector, "KeypointSelector"), (trainer.refiner, "DescriptorRefiner")):
        assert os.path.realpath(sys.modules[type(mod).__module__].__file__).startswith(os.path.realpath(PKG)), name
    before = {k: v.detach().clone() for k, v in list(trainer.selector.state_dict().items()) + list(trainer.refiner.state_dict().items())}
    assert len(trainer.train_loader) == 2 and len(trainer.val_loader) == 1          # 5 pairs at B = 4 -> 2 batches; 2 pairs -> 1

    # one step by hand first, to look at the gradients the reference's loop clips and applies (train.py:226-244)
    batch = next(iter(trainer.train_loader))
    trainer.selector.train()
    trainer.refiner.train()
    assert batch["rgb1"].shape == (4, 3, 448, 448)
    loss, comps, metrics = trainer._forward_pass(batch["rgb1"], batch["rgb2"])
    trainer.optimizer.zero_grad()
    loss.backward()
    gsel = float(sum(p.grad.abs().sum() for p in trainer.selector.parameters() if p.grad is not None))
    gref = float(sum(p.grad.abs().sum() for p in trainer.refiner.parameters() if p.grad is not None))
    assert np.isfinite(float(loss)) and gsel > 0 and gref > 0, (float(loss), gsel, gref)
    assert all(p.grad is not None for p in trainer.selector.parameters()) and all(p.grad is not None for p in trainer.refiner.parameters())
    assert all(p.grad is None for p in trainer.backbone.dino.parameters()), "the frozen ViT must not receive gradients"
    trainer.optimizer.zero_grad()

    trainer.train()                                                   # train_epoch + validate + save_checkpoint, unchanged
    after = dict(list(trainer.selector.state_dict().items()) + list(trainer.refiner.state_dict().items()))
    changed = sum(int(not torch.equal(before[k], after[k])) for k in before)
    assert changed == len(before), f"only {changed} of {len(before)} parameter tensors moved"
    ckpt = os.path.join(config["training"]["save_dir"], "best_model.pth")
    assert os.path.exists(ckpt)
    sd = torch.load(ckpt, map_location="cpu", weights_only=True)
    assert set(sd) == {"epoch", "loss", "selector_state_dict", "refiner_state_dict", "optimizer_state_dict", "scheduler_state_dict", "config"}
    assert set(sd["selector_state_dict"]) == {"conv.0.weight", "conv.0.bias", "conv.2.weight", "conv.2.bias"}
    assert len(sd["refiner_state_dict"]) == 20 and "residual_blocks.1.norm2.bias" in sd["refiner_state_dict"]

    # (iii) the visualize script on the checkpoint train.py wrote, ViT inside (in-repo definition, random weights), from PNGs
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = vms.SequenceMatcher(ckpt, os.path.join(REF, "configs", "train_config.yaml"), device="cpu")
    for k, v in m.selector.state_dict().items():
        assert torch.equal(v, sd["selector_state_dict"][k])
    rgb = os.path.join(root, "rgbd_dataset_synth_train", "rgb")
    files = sorted(os.listdir(rgb))
    f1, f2 = m.extract(os.path.join(rgb, files[0])), m.extract(os.path.join(rgb, files[1]))
    mt, q = m.match_with_quality(f1["descriptors"], f2["descriptors"], f1["scores"], f2["scores"], 0.3, 0.3, 0.5,
                                 f1["intensity"], f2["intensity"], 0.15)
    assert f1["descriptors"].shape == (500, 128) and np.allclose(np.linalg.norm(f1["descriptors"], axis=1), 1.0, atol=1e-5)
    assert mt.ndim == 2 and mt.shape[1] == 2 and mt.dtype == np.int64 and q.shape == (mt.shape[0],)
    other = part_other_callers(tmp, ckpt, root, config)
    return {"other_callers": other,
            "script": "train.py", "class": "SemanticSLAMTrainer", "device": "cpu", "batch": 4, "train_batches": 2, "val_batches": 1,
            "first_step_loss": float(loss), "loss_components": {k: float(v) for k, v in comps.items()},
            "first_step_matches_per_sample_padded": int(metrics["num_matches"]),
            "grad_abs_sum_selector": gsel, "grad_abs_sum_refiner": gref, "parameter_tensors_moved": f"{changed} / {len(before)}",
            "checkpoint_keys": sorted(sd), "best_val_loss": float(sd["loss"]),
            "then_visualize_matches_sequence_on_that_checkpoint": {"frames": 2, "matches": int(len(mt)), "vit": "in-repo DINOv3 ViT-S/16, random weights, eager fp32 on cpu"}}


# ---------------------------------------------------------------------------------------------- (iv)
def part_other_callers(tmp: str, ckpt: str, root: str, config: dict) -> dict:
    import torch
    import yaml

    from oracle import ora               # the CPU oracle's M2 / M4 / M5 (matching.py itself is HIP-only: no GPU in this container)
    sys.path.insert(2, os.path.join(REF, "test"))
    import test_descriptor_quality as tdq
    import test_performance as tperf
    import test_repeatability as trep
    import test_tracking as ttrack
    import visualize_matches as vm
    for mod in (tdq, tperf, trep, ttrack, vm):
        assert os.path.realpath(mod.__file__).startswith("/root/reference/"), mod.__file__
    cfg_path = os.path.join(tmp, "eval_config.yaml")            # the reference's YAML with the data paths of this run (a temp file)
    with open(cfg_path, "w") as f:
        yaml.safe_dump(config, f)
    seq = "rgbd_dataset_synth_train"
    rgb = os.path.join(root, seq, "rgb")
    files = sorted(os.listdir(rgb))
    out = {}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        # visualize_matches.py: M2 (ratio test against the runner-up by masking the best column)
        mv = vm.MatchVisualizer(ckpt, cfg_path, device="cpu")
        f1, f2 = mv.extract_features(os.path.join(rgb, files[0])), mv.extract_features(os.path.join(rgb, files[1]))
        m2 = mv.find_matches(f1["descriptors"], f2["descriptors"], ratio_thresh=0.8)
        mine = ora.find_matches_m2(f1["descriptors"], f2["descriptors"], ratio_thresh=0.8)
        assert [(i, j) for i, j, _ in m2] == [(i, j) for i, j, _ in mine], "M2: match pairs differ between the script and the oracle"
        assert np.allclose([s for _, _, s in m2], [s for _, _, s in mine], atol=1e-5)
        out["visualize_matches.MatchVisualizer"] = {"matches_M2": len(m2), "equal_to_oracle_M2": True}

        # test_tracking.py: M5 (count of rows whose best similarity exceeds the threshold), eval-mode backbone
        tt = ttrack.TrackingTester(ckpt, cfg_path, device="cpu")
        assert not tt.backbone.training and not tt.backbone.feature_norm.training
        r = tt.track_frame_sequence(seq, max_frames=4, min_matches=50, match_threshold=0.8, frame_spacing=1)
        assert r["total_frames"] == 2 and len(r["match_counts"]) == 2        # 4 frames: len(dataset) = 3, the loop visits frames 1 and 2
        ds = ttrack.TUMDataset(dataset_root=root, sequence=seq, input_size=448, frame_spacing=1, max_frames=4, augmentation=None, is_train=False)
        d0 = tt.extract_features(ds[0]["rgb1"].unsqueeze(0))[1]
        d1 = tt.extract_features(ds[1]["rgb1"].unsqueeze(0))[1]
        assert int(r["match_counts"][0]) == int((ora.sim_matrix(d0, d1).max(axis=1) > np.float32(0.8)).sum()), "M5: tracked count differs from the oracle"
        out["test_tracking.TrackingTester"] = {"pairs": 2, "match_counts": [int(c) for c in r["match_counts"]], "equal_to_oracle_M5": True,
                                               "tracking_success_rate": float(r["tracking_success_rate"])}

        # test_descriptor_quality.py: M4 (mutual NN + ratio by full row sort) and the pose-derived ground truth
        tq = tdq.DescriptorQualityTester(ckpt, cfg_path, device="cpu")
        r = tq.test_sequence(seq, num_pairs=2, frame_spacing=1)
        assert r["num_pairs"] == 2 and 0.0 <= r["mean_precision"] <= 1.0
        pm, pd = tq.find_mutual_nearest_neighbors(d0, d1)
        mm, md = ora.find_mnn_m4(d0, d1)
        assert np.array_equal(np.asarray(pm), np.asarray(mm)) and np.allclose(pd, md, atol=1e-5), "M4 differs from the oracle"
        out["test_descriptor_quality.DescriptorQualityTester"] = {"pairs": 2, "mean_num_matches": float(r["mean_num_matches"]),
                                                                  "matches_M4_first_pair": int(len(pm)), "equal_to_oracle_M4": True}

        # test_repeatability.py
        tr = trep.RepeatabilityTester(ckpt, cfg_path, device="cpu")
        r = tr.test_sequence(seq, num_pairs=2, frame_spacing=1, use_pose=True)
        assert r["num_pairs"] == 2 and 0.0 <= r["mean_repeatability"] <= 1.0
        out["test_repeatability.RepeatabilityTester"] = {"pairs": 2, "mean_repeatability": float(r["mean_repeatability"])}

        # test_performance.py: the reference's stage timers (10 warm-up + num_runs timed forward passes of one image)
        tp = tperf.PerformanceTester(ckpt, cfg_path, device="cpu")
        times = tp.measure_component_times(ds[0]["rgb1"].unsqueeze(0), num_runs=2)
        assert set(times) >= {"backbone", "selector", "selector_nms", "refiner", "total"} and times["total"]["mean"] > 0
        out["test_performance.PerformanceTester"] = {"stages": sorted(times), "total_ms_cpu_eager": float(times["total"]["mean"])}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--json", default=None, help="write the outcome record here")
    args = ap.parse_args()
    if not os.path.isdir(REF):
        print("run_reference_scripts.py: /root/reference is absent (this check runs in the build container only)")
        return 77
    install_stand_ins()
    with tempfile.TemporaryDirectory() as tmp:
        rec = {"sys_path_head": [PKG, REF], "stand_ins": {"torchvision.transforms": "PIL Resize / ToTensor / Normalize / Compose",
                                                           "cv2, wandb": "inert placeholders", "timm": "absent -> the drop-in's in-repo ViT"},
               "visualize_matches_sequence": part_sequence_matcher(tmp), "train": part_train_then_match(tmp)}
    line = json.dumps(rec, indent=1)
    print(line)
    if args.json:
        with open(args.json, "w") as f:
            f.write(line + "\n")
    print("reference scripts on the drop-in: ok")
    return 0


if __name__ == "__main__":
    sys.exit(main())

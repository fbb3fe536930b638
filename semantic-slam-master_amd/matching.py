"""Home of the reference's five script-resident matchers, with their original signatures and
return types, running on the HIP kernels sslam_sim_argmax / sslam_match_finalize.

    M1  match_with_quality             SequenceMatcher.match_with_quality   visualize_matches_sequence.py:106-197
    M2  find_matches                   MatchVisualizer.find_matches         visualize_matches.py:102-124
    M3  find_matches_batched           SemanticSLAMTrainer._find_matches    train.py:410-449
    M4  find_mutual_nearest_neighbors  DescriptorQualityTester.find_mutual_nearest_neighbors
                                                                             test/test_descriptor_quality.py:97-142
    M5  count_tracked                  track_frame_sequence                  test/test_tracking.py:158-161

numpy in -> numpy out like the originals (they pull descriptors to the host first); torch CUDA tensors are accepted
too and then nothing leaves the device except the result.  There is no CPU implementation here: without the GPU
library these functions raise.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from sslam_amd import lib


def _dev(a, dtype=torch.float32):
    if isinstance(a, torch.Tensor):
        return a.detach().to("cuda", dtype).contiguous()
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda", dtype).contiguous()


def _argmax(desc1, desc2, want_second=False):
    d1, d2 = _dev(desc1), _dev(desc2)
    if d1.shape[1] != lib.D_OUT or d2.shape[1] != lib.D_OUT:
        raise lib.SslamHipError("descriptor dimension must be 128")
    n, m = d1.shape[0], d2.shape[0]
    return (n, m) + lib.sim_argmax(d1, 0, n, d2, 0, m, 1, want_s21=False, want_second=want_second)


def match_with_quality(desc1, desc2, scores1, scores2, saliency_weight: float = 0.3, min_saliency: float = 0.2,
                       min_descriptor_sim: float = 0.7, intensity1: Optional[np.ndarray] = None,
                       intensity2: Optional[np.ndarray] = None, min_intensity: float = 0.1):
    """M1.  Returns (matches (K, 2) int64 ascending in idx1, quality (K,) float32); empty -> shapes (0, 2), (0,)."""
    if len(desc1) == 0 or len(desc2) == 0:
        return np.zeros((0, 2), dtype=np.int64), np.zeros((0,), dtype=np.float32)
    n, m, nn12, s12, nn21, _, _ = _argmax(desc1, desc2)
    use_int = intensity1 is not None and intensity2 is not None
    mt, q, cnt = lib.match_finalize(nn12, s12, nn21, n, m, 1, _dev(scores1), 0, _dev(scores2), 0,
                                    _dev(intensity1) if use_int else None, _dev(intensity2) if use_int else None,
                                    1.0 - saliency_weight, saliency_weight, min_saliency, min_descriptor_sim, min_intensity)
    c = int(cnt.item())
    if c == 0:
        return np.zeros((0, 2), dtype=np.int64), np.zeros((0,), dtype=np.float32)
    return mt[0, :c].cpu().numpy(), q[0, :c].cpu().numpy()


def find_matches(desc1, desc2, ratio_thresh: float = 0.8):
    """M2.  Mutual NN + 'best > second_best * ratio' (second best = best of the row with the winner removed).
    Returns a list of (i, j, sim) like the original."""
    n, m, nn12, s12, nn21, _, sec = _argmax(desc1, desc2, want_second=True)
    nn12, s12, nn21, sec = nn12[0].long(), s12[0], nn21[0].long(), sec[0]
    if m == 1:
        sec = torch.full_like(sec, -1.0)           # the original masks the winner with -1 and takes the max
    else:
        sec = torch.maximum(sec, torch.full_like(sec, -1.0))
    keep = (nn21[nn12] == torch.arange(n, device=nn12.device)) & (s12 > sec * ratio_thresh)
    idx = torch.nonzero(keep).squeeze(1)
    i, j, s = idx.cpu().numpy(), nn12[idx].cpu().numpy(), s12[idx].cpu().numpy()
    return [(int(a), int(b), c) for a, b, c in zip(i, j, s)]


def find_mutual_nearest_neighbors(desc1, desc2, ratio_threshold: float = 0.9):
    """M4.  Mutual NN and second/(best + 1e-8) < ratio.  Returns (matches (K, 2) int64, distances = 1 - sim (K,))."""
    n, m, nn12, s12, nn21, _, sec = _argmax(desc1, desc2, want_second=True)
    nn12, s12, nn21, sec = nn12[0].long(), s12[0], nn21[0].long(), sec[0]
    if m < 2:
        raise IndexError("index 1 is out of bounds for axis 1 with size 1")   # np.sort(...)[:, 1] in the original
    ratio = sec / (s12 + 1e-8)
    keep = (nn21[nn12] == torch.arange(n, device=nn12.device)) & (ratio < ratio_threshold)
    idx = torch.nonzero(keep).squeeze(1)
    matches = torch.stack([idx, nn12[idx]], dim=1).cpu().numpy().astype(np.int64)
    return matches, (1.0 - s12[idx]).cpu().numpy()


def find_matches_batched(desc1: torch.Tensor, desc2: torch.Tensor) -> torch.Tensor:
    """M3.  (B, N, D) x 2 on the device -> (B, Mmax, 2) int64 mutual-NN pairs, zero-padded to the longest;
    all-empty -> zeros(B, 1, 2) (train.py:439-440).  The padding rows are (0, 0), as in the original."""
    B, N, _ = desc1.shape
    d1, d2 = _dev(desc1), _dev(desc2)
    nn12, s12, nn21, _, _ = lib.sim_argmax(d1, N * lib.D_OUT, N, d2, N * lib.D_OUT, N, B)
    ones = torch.ones((B, N), dtype=torch.float32, device=d1.device)
    mt, _, cnt = lib.match_finalize(nn12, s12, nn21, N, N, B, ones, N, ones, N, None, None, 1.0, 0.0, -1e30, -1e30, -1e30)
    mmax = int(cnt.max().item())
    if mmax == 0:
        return torch.zeros(B, 1, 2, device=d1.device, dtype=torch.long)
    out = mt[:, :mmax].clone()
    pad = torch.arange(mmax, device=d1.device)[None, :] >= cnt[:, None].long()
    out[pad] = 0
    return out


def count_tracked(desc_prev, desc_curr, match_threshold: float = 0.8) -> int:
    """M5.  Number of rows of desc_prev whose best similarity in desc_curr exceeds the threshold."""
    _, _, _, s12, _, _, _ = _argmax(desc_prev, desc_curr)
    return int((s12[0] > match_threshold).sum().item())

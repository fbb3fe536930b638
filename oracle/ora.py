"""numpy front-end to oracle/_build/liboracle.so (the CPU restatement; see sslam_oracle.h).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by
the product package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")

REFINER_KEYS_HEAD = ["input_proj.weight", "input_proj.bias"]
REFINER_KEYS_BLOCK = ["norm1.weight", "norm1.bias", "fc1.weight", "fc1.bias",
                      "norm2.weight", "norm2.bias", "fc2.weight", "fc2.bias"]
REFINER_KEYS_TAIL = ["output_proj.weight", "output_proj.bias"]


def build(force: bool = False) -> str:
    src = [os.path.join(_HERE, f) for f in ("sslam_oracle.c", "sslam_oracle.h")]
    if force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def host_threads() -> int:
    """CPU threads this process may really use: affinity mask, capped by the cgroup CPU quota (the GPU box exposes
    256 logical CPUs but grants a 16-CPU share; 256 spinning OpenMP threads on 16 CPUs would be meaningless)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except (OSError, ValueError):
            pass
    return n


def lib():
    global _lib
    if _lib is None:
        os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
        os.environ.setdefault("OMP_NUM_THREADS", str(host_threads()))
        _lib = C.CDLL(build())
        _lib.ora_quantile.restype = C.c_float
        _lib.ora_quantile.argtypes = [C.c_void_p, C.c_int, C.c_double]
        _lib.ora_expf.restype = C.c_float
        _lib.ora_expf.argtypes = [C.c_float]
        _lib.ora_sigmoid.restype = C.c_float
        _lib.ora_sigmoid.argtypes = [C.c_float]
        _lib.ora_match_with_quality.restype = C.c_int
        # an OpenMP runtime that another library initialised first (torch: one thread per LOGICAL cpu, 256 on the GPU box) ignores
        # the environment default above: cap the team at what this process may really use (16 there), or every parallel region
        # runs 256 threads on 16 CPUs - ten times slower
        n = host_threads()
        if _lib.ora_num_threads() > n:
            _lib.ora_set_num_threads(C.c_int(n))
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def set_num_threads(n: int):
    lib().ora_set_num_threads(C.c_int(n))


def num_threads() -> int:
    return lib().ora_num_threads()


def bn_tokens(tokens, n_prefix=5, group=1, gamma=None, beta=None, run_mean=None, run_var=None, train=True,
              eps=1e-5):
    tokens = _f32(tokens)
    n, t, c = tokens.shape
    cells = t - n_prefix
    gamma = _f32(np.ones(c) if gamma is None else gamma)
    beta = _f32(np.zeros(c) if beta is None else beta)
    run_mean = _f32(np.zeros(c) if run_mean is None else run_mean)
    run_var = _f32(np.ones(c) if run_var is None else run_var)
    out = np.empty((n, cells, c), np.float32)
    mean = np.zeros((n // group, c), np.float32)
    var = np.zeros((n // group, c), np.float32)
    lib().ora_bn_tokens(_p(tokens), n, t, n_prefix, group, _p(gamma), _p(beta), _p(run_mean), _p(run_var),
                        int(bool(train)), C.c_float(eps), _p(out), _p(mean), _p(var))
    return out, mean, var


def selector_saliency(feat, sd):
    """feat (n, G, G, 384); sd: KeypointSelector state_dict as numpy."""
    feat = _f32(feat)
    n, g = feat.shape[0], feat.shape[1]
    w1, b1 = _f32(sd["conv.0.weight"]), _f32(sd["conv.0.bias"])
    w2, b2 = _f32(sd["conv.2.weight"]).reshape(-1), _f32(sd["conv.2.bias"])
    sal = np.empty((n, g, g), np.float32)
    lib().ora_selector_saliency(_p(feat), n, g, _p(w1), _p(b1), _p(w2), _p(b2), int(w1.shape[0]), _p(sal))
    return sal


def nms(sal, radius=2):
    sal = _f32(sal)
    out = np.empty_like(sal)
    lib().ora_nms(_p(sal), sal.shape[0], radius, _p(out))
    return out


def quantile(v, q):
    v = _f32(v).ravel()
    return np.float32(lib().ora_quantile(_p(v), v.size, C.c_double(q)))


def select_keypoints(sal, K=500, radius=2, pct=0.5):
    sal = _f32(sal)
    if sal.ndim == 2:
        sal = sal[None]
    n, g = sal.shape[0], sal.shape[1]
    kp = np.zeros((n, K, 2), np.float32)
    sc = np.zeros((n, K), np.float32)
    idx = np.zeros((n, K), np.int32)
    st = np.zeros((n,), np.int32)
    lib().ora_select_keypoints(_p(sal), n, g, K, radius, C.c_double(pct), _p(kp), _p(sc), _p(idx), _p(st))
    return kp, sc, idx, st


def gather(feat, kp):
    feat, kp = _f32(feat), _f32(kp)
    n, g = feat.shape[0], feat.shape[1]
    K = kp.shape[1]
    out = np.empty((n, K, feat.shape[-1]), np.float32)
    lib().ora_gather(_p(feat), n, g, _p(kp), K, _p(out))
    return out


def refiner_weight_list(sd, n_blocks=2):
    keys = list(REFINER_KEYS_HEAD)
    for i in range(n_blocks):
        keys += [f"residual_blocks.{i}.{k}" for k in REFINER_KEYS_BLOCK]
    keys += REFINER_KEYS_TAIL
    return [_f32(sd[k]) for k in keys]


def refine(x, sd, n_blocks=2):
    x = _f32(x)
    shape = x.shape
    x2 = x.reshape(-1, shape[-1])
    ws = refiner_weight_list(sd, n_blocks)
    d_out = ws[-2].shape[0]
    arr = (C.c_void_p * len(ws))(*[w.ctypes.data for w in ws])
    out = np.empty((x2.shape[0], d_out), np.float32)
    lib().ora_refine(_p(x2), x2.shape[0], arr, n_blocks, d_out, _p(out))
    return out.reshape(shape[:-1] + (d_out,))


def patch_to_pixel(kp):
    kp = _f32(kp)
    out = np.empty_like(kp)
    lib().ora_patch_to_pixel(_p(kp), kp.size, _p(out))
    return out


def pixel_to_patch(px):
    px = _f32(px)
    out = np.empty_like(px)
    lib().ora_pixel_to_patch(_p(px), px.size, _p(out))
    return out


def sim_argmax(d1, d2):
    d1, d2 = _f32(d1), _f32(d2)
    n, m, d = d1.shape[0], d2.shape[0], d1.shape[1]
    nn12, nn21 = np.empty(n, np.int32), np.empty(m, np.int32)
    s12, s21 = np.empty(n, np.float32), np.empty(m, np.float32)
    lib().ora_sim_argmax(_p(d1), n, _p(d2), m, d, _p(nn12), _p(s12), _p(nn21), _p(s21))
    return nn12, s12, nn21, s21


def sim_matrix(d1, d2):
    d1, d2 = _f32(d1), _f32(d2)
    S = np.empty((d1.shape[0], d2.shape[0]), np.float32)
    lib().ora_sim_matrix(_p(d1), d1.shape[0], _p(d2), d2.shape[0], d1.shape[1], _p(S))
    return S


def find_matches_m2(d1, d2, ratio_thresh=0.8):
    """M2 (visualize_matches.py:102-124) on the canonical similarity matrix."""
    S = sim_matrix(d1, d2)
    nn12, nn21 = S.argmax(axis=1), S.argmax(axis=0)
    out = []
    for i in range(S.shape[0]):
        j = nn12[i]
        if nn21[j] == i:
            sims = S[i].copy()
            sims[j] = -1
            if S[i, j] > sims.max() * np.float32(ratio_thresh):
                out.append((i, int(j), S[i, j]))
    return out


def find_mnn_m4(d1, d2, ratio_threshold=0.9):
    """M4 (test/test_descriptor_quality.py:97-142) on the canonical similarity matrix."""
    S = sim_matrix(d1, d2)
    nn12, nn21 = S.argmax(axis=1), S.argmax(axis=0)
    mutual = nn21[nn12] == np.arange(S.shape[0])
    srt = np.sort(S, axis=1)[:, ::-1]
    ratio = srt[:, 1] / (srt[:, 0] + np.float32(1e-8))
    idx1 = np.where(mutual & (ratio < np.float32(ratio_threshold)))[0]
    return np.stack([idx1, nn12[idx1]], axis=1).astype(np.int64), (np.float32(1.0) - S.max(axis=1)[idx1])


def match_with_quality(d1, d2, s1, s2, saliency_weight=0.3, min_saliency=0.2, min_descriptor_sim=0.7,
                       intensity1=None, intensity2=None, min_intensity=0.1):
    d1, d2, s1, s2 = _f32(d1), _f32(d2), _f32(s1), _f32(s2)
    n, m, d = d1.shape[0], d2.shape[0], d1.shape[1]
    i1 = None if intensity1 is None or intensity2 is None else _f32(intensity1)
    i2 = None if intensity1 is None or intensity2 is None else _f32(intensity2)
    matches = np.zeros((n, 2), np.int64)
    quality = np.zeros((n,), np.float32)
    cnt = lib().ora_match_with_quality(_p(d1), n, _p(d2), m, d, _p(s1), _p(s2), C.c_double(saliency_weight),
                                       C.c_double(min_saliency), C.c_double(min_descriptor_sim), _p(i1), _p(i2),
                                       C.c_double(min_intensity), _p(matches), _p(quality))
    return matches[:cnt].copy(), quality[:cnt].copy()


def resize_rgb(img, size, bicubic=False, want_chw=True):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape[:2]
    rs = np.empty((size, size, 3), np.uint8)
    chw = np.empty((3, size, size), np.float32) if want_chw else None
    lib().ora_resize_rgb(_p(img), h, w, size, int(bicubic), _p(rs), _p(chw))
    return rs, chw


def gray_resized(img, size):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape[:2]
    g = np.empty((size, size), np.uint8)
    lib().ora_gray_resized(_p(img), h, w, size, _p(g))
    return g


def intensity(img, size, kp_pixel):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    kp_pixel = _f32(kp_pixel)
    h, w = img.shape[:2]
    out = np.empty((kp_pixel.shape[0],), np.float32)
    lib().ora_intensity(_p(img), h, w, size, _p(kp_pixel), kp_pixel.shape[0], _p(out))
    return out


def expf(x):
    return np.float32(lib().ora_expf(C.c_float(float(x))))


def sigmoid(x):
    return np.float32(lib().ora_sigmoid(C.c_float(float(x))))

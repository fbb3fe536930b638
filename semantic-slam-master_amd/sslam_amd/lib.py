"""ctypes binding of libsslam_hip.so (include/sslam_hip.h) for torch tensors.

PyTorch is used for device memory and streams only: every wrapper passes raw device pointers and the current
HIP stream to the C ABI.  There is NO fallback: if the library is missing or a call fails, an exception is raised.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np
import torch

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(_PKG, "csrc")
SO_PATH = os.path.join(CSRC, "libsslam_hip.so")

OK, E_INVALID, E_UNSUPPORTED, E_LAUNCH = 0, -1, -2, -3
_ERR = {E_INVALID: "invalid argument", E_UNSUPPORTED: "unsupported shape", E_LAUNCH: "kernel launch failed"}

C_FEAT, HID, D_OUT = 384, 384, 128
MAX_TAPS = 32


class SslamHipError(RuntimeError):
    pass


class VitLayer(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("ln1_g", "ln1_b", "wqkv", "bqkv", "wo", "bo", "ln2_g", "ln2_b", "wup", "bup",
                                          "wdown", "bdown", "wmlp")]


class VitWeights(C.Structure):
    _fields_ = [("patch_w", C.c_void_p), ("patch_b", C.c_void_p), ("prefix", C.c_void_p), ("layer", VitLayer * 12),
                ("norm_g", C.c_void_p), ("norm_b", C.c_void_p), ("rope_cos", C.c_void_p), ("rope_sin", C.c_void_p)]


class VitLayerF32(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("ln1_g", "ln1_b", "wqkv", "bqkv", "wo", "bo", "ls1", "ln2_g", "ln2_b", "wup", "bup",
                                          "wdown", "bdown", "ls2")]


class VitWeightsF32(C.Structure):
    _fields_ = [("patch_w", C.c_void_p), ("patch_b", C.c_void_p), ("prefix", C.c_void_p), ("layer", VitLayerF32 * 12),
                ("norm_g", C.c_void_p), ("norm_b", C.c_void_p), ("rope_cos", C.c_void_p), ("rope_sin", C.c_void_p)]


class RefinerLayout(C.Structure):
    _fields_ = [("n_blocks", C.c_int), ("total", C.c_longlong), ("in_w", C.c_longlong), ("in_b", C.c_longlong),
                ("blk", (C.c_longlong * 8) * 8), ("out_w", C.c_longlong), ("out_b", C.c_longlong)]


def build(force: bool = False) -> str:
    """Compile the HIP library for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))]
    srcs.append(os.path.join(os.path.dirname(_PKG), "include", "sslam_hip.h"))
    stale = (not os.path.exists(SO_PATH)) or any(os.path.getmtime(s) > os.path.getmtime(SO_PATH) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", CSRC, "-s", "-j4"])
    return SO_PATH


_lib = None

EXPORTS = [
    "sslam_version", "sslam_arch", "sslam_launch_count", "sslam_pack_conv3x3_host", "sslam_pack_linear_host",
    "sslam_resample_table_host", "sslam_preprocess_u8", "sslam_bn_tokens", "sslam_selector_saliency",
    "sslam_select_keypoints", "sslam_gather", "sslam_refiner_layout", "sslam_refiner_pack_host", "sslam_refine",
    "sslam_gather_refine", "sslam_keypoint_intensity", "sslam_sim_argmax", "sslam_match_finalize",
    "sslam_vit_pack_linear_host", "sslam_vit_pack_mlp_host", "sslam_vit_workspace_bytes", "sslam_vit_forward",
    "sslam_f32_to_bf16", "sslam_pack_conv3x3_bf16_host", "sslam_selector_saliency_bf16",
    "sslam_bn_tokens_bf16copy", "sslam_refiner_bf16_bytes", "sslam_refiner_pack_bf16_host", "sslam_refine_bf16", "sslam_gather_refine_bf16",
    "sslam_workspace_bytes", "sslam_selector_saliency_workspace_bytes", "sslam_sim_argmax_workspace_bytes",
    "sslam_selector_saliency_ws", "sslam_sim_argmax_ws", "sslam_test_set_knob",
    "sslam_preprocess_u8_patches", "sslam_vit_forward_patches", "sslam_vit_f32_workspace_bytes", "sslam_vit_forward_f32",
    "sslam_vit_f32_pack_linear_host", "sslam_vit_forward_f32_form",
]


def lib():
    """Load libsslam_hip.so; raises if it has not been built (no silent fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise SslamHipError(f"{SO_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                                f"(or `make -C {CSRC}`); this package has no non-HIP execution path")
        L = C.CDLL(SO_PATH)
        L.sslam_arch.restype = C.c_char_p
        L.sslam_launch_count.restype = C.c_longlong
        p, i, ll, f, d = C.c_void_p, C.c_int, C.c_longlong, C.c_float, C.c_double
        L.sslam_pack_conv3x3_host.argtypes = [p, i, p]
        L.sslam_pack_linear_host.argtypes = [p, i, i, p]
        L.sslam_resample_table_host.argtypes = [i, i, i, p, p, i]
        L.sslam_preprocess_u8.argtypes = [p, i, i, i, i, p, p, i, p, p, i, p, p]
        L.sslam_preprocess_u8_patches.argtypes = [p, i, i, i, i, p, p, i, p, p, i, p, p]
        L.sslam_bn_tokens.argtypes = [p, i, i, i, i, p, p, p, p, i, f, p, p, p, p]
        L.sslam_selector_saliency.argtypes = [p, i, i, p, p, p, p, i, p, p]
        L.sslam_selector_saliency_ws.argtypes = [p, i, i, p, p, p, p, i, p, p, ll, p]
        for fn, at in ((L.sslam_workspace_bytes, [i, i, i, i]), (L.sslam_selector_saliency_workspace_bytes, [i, i]),
                       (L.sslam_sim_argmax_workspace_bytes, [i, i])):
            fn.restype, fn.argtypes = ll, at
        L.sslam_test_set_knob.argtypes = [C.c_char_p, ll, i]
        L.sslam_select_keypoints.argtypes = [p, i, i, i, i, d, p, p, p, p, p, p]
        L.sslam_gather.argtypes = [p, i, i, p, i, p, p]
        L.sslam_refiner_layout.argtypes = [i, C.POINTER(RefinerLayout)]
        L.sslam_refiner_pack_host.argtypes = [p, i, p]
        L.sslam_refine.argtypes = [p, ll, p, i, p, p]
        L.sslam_gather_refine.argtypes = [p, i, i, p, i, p, i, p, p]
        L.sslam_keypoint_intensity.argtypes = [p, i, i, i, i, p, p, i, p, p, i, p, i, p, p]
        L.sslam_sim_argmax.argtypes = [p, ll, i, p, ll, i, i, p, p, p, p, p, p]
        L.sslam_sim_argmax_ws.argtypes = [p, ll, i, p, ll, i, i, p, p, p, p, p, p, ll, p]
        L.sslam_match_finalize.argtypes = [p, p, p, i, i, i, p, ll, p, ll, p, p, f, f, f, f, f, p, p, p, p]
        L.sslam_f32_to_bf16.argtypes = [p, p, ll, p]
        L.sslam_pack_conv3x3_bf16_host.argtypes = [p, i, p]
        L.sslam_selector_saliency_bf16.argtypes = [p, i, i, p, p, p, p, i, p, p]
        L.sslam_bn_tokens_bf16copy.argtypes = [p, i, i, i, i, p, p, p, p, i, f, p, p, p, p, p]
        L.sslam_refiner_bf16_bytes.restype = C.c_longlong
        L.sslam_refiner_bf16_bytes.argtypes = [i]
        L.sslam_refiner_pack_bf16_host.argtypes = [p, i, p]
        L.sslam_refine_bf16.argtypes = [p, ll, p, i, p, p]
        L.sslam_gather_refine_bf16.argtypes = [p, i, i, p, i, p, i, p, p]
        L.sslam_vit_pack_linear_host.argtypes = [p, i, i, p]
        L.sslam_vit_pack_mlp_host.argtypes = [p, p, p, p]
        L.sslam_vit_workspace_bytes.restype = C.c_longlong
        L.sslam_vit_workspace_bytes.argtypes = [i, i]
        L.sslam_vit_forward.argtypes = [p, i, i, C.POINTER(VitWeights), p, ll, p, p]
        L.sslam_vit_forward_patches.argtypes = [p, i, i, C.POINTER(VitWeights), p, ll, p, p]
        L.sslam_vit_f32_workspace_bytes.restype = C.c_longlong
        L.sslam_vit_f32_workspace_bytes.argtypes = [i, i]
        L.sslam_vit_forward_f32.argtypes = [p, i, i, C.POINTER(VitWeightsF32), p, ll, p, p]
        L.sslam_vit_forward_f32_form.argtypes = [p, i, i, C.POINTER(VitWeightsF32), p, ll, p, i, p]
        L.sslam_vit_f32_pack_linear_host.argtypes = [p, i, i, p]
        _lib = L
    return _lib


def _check(rc: int, what: str):
    if rc != OK:
        if rc == E_INVALID:
            raise ValueError(f"{what}: {_ERR[rc]}")
        raise SslamHipError(f"{what}: {_ERR.get(rc, rc)}")


def common_device(*tensors):
    """The ONE CUDA device every tensor argument of a call lives on.  Raises ValueError for a CPU tensor or for tensors
    on different GPUs (the reference's torch ops raise a device-mismatch error there; handing a device-1 pointer to a
    device-0 stream would be a GPU memory fault instead)."""
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise ValueError(f"libsslam_hip takes CUDA tensors only, got a tensor on {t.device}")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise ValueError(f"tensor arguments on different devices: {dev} and {t.device}")
    if dev is None:
        raise ValueError("no device tensor among the arguments")
    return dev


def _run(what: str, fn, tensors, *args):
    """Call one C-ABI entry: all `tensors` must share a device; that device is made current for the call and the
    launch goes to ITS current stream (not the process-wide current device's)."""
    dev = common_device(*tensors)
    with torch.cuda.device(dev):
        _check(fn(*args, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), what)


def _dp(t):
    if t is None:
        return None
    if not (t.is_cuda and t.is_contiguous()):
        raise ValueError("device-resident contiguous tensor required")
    return C.c_void_p(t.data_ptr())


def launch_count() -> int:
    return int(lib().sslam_launch_count())


def workspace_bytes(n_frames: int, G: int, K: int, n_pairs: int) -> int:
    """Bytes of caller-owned scratch that serve every *_ws entry of one pipeline step on one stream (include/sslam_hip.h)."""
    b = int(lib().sslam_workspace_bytes(n_frames, G, K, n_pairs))
    if b < 0:
        _check(b, "workspace_bytes")
    return b


def _scratch(workspace, need: int, device):
    """The caller's workspace tensor if it is large enough, else a fresh torch allocation (the library itself never allocates)."""
    if need <= 0:
        return None, 0
    if workspace is not None and workspace.numel() * workspace.element_size() >= need:
        return workspace, workspace.numel() * workspace.element_size()
    ws = torch.empty((need,), dtype=torch.uint8, device=device)
    return ws, need


class knobs:
    """TEST-ONLY: `with lib.knobs(SSLAM_CONV_TAIL=4): ...` overrides load-time knobs of the library for the block
    (sslam_test_set_knob) and restores their load-time values (environment at load, else the built-in defaults) afterwards.  Product code never uses it."""

    def __init__(self, **kv):
        self.kv = kv

    def __enter__(self):
        for k, v in self.kv.items():
            _check(lib().sslam_test_set_knob(k.encode(), int(v), 0), f"set_knob({k})")
        return self

    def __exit__(self, *exc):
        for k in self.kv:
            lib().sslam_test_set_knob(k.encode(), 0, 1)
        return False


# ------------------------------------------------------------------------------------------ host-side packing
def pack_conv3x3(w: np.ndarray) -> np.ndarray:
    w = np.ascontiguousarray(w, np.float32)
    hs = w.shape[0]
    assert w.shape == (hs, C_FEAT, 3, 3)
    out = np.empty(9 * C_FEAT * hs, np.float32)
    _check(lib().sslam_pack_conv3x3_host(w.ctypes.data, hs, out.ctypes.data), "pack_conv3x3")
    return out


def refiner_layout(n_blocks: int) -> RefinerLayout:
    lay = RefinerLayout()
    _check(lib().sslam_refiner_layout(n_blocks, C.byref(lay)), "refiner_layout")
    return lay


def pack_refiner(weights: list, n_blocks: int) -> np.ndarray:
    """weights: 4 + 8*n_blocks fp32 arrays in state_dict order (input_proj, blocks, output_proj)."""
    ws = [np.ascontiguousarray(w, np.float32) for w in weights]
    assert len(ws) == 4 + 8 * n_blocks
    lay = refiner_layout(n_blocks)
    out = np.empty(lay.total, np.float32)
    arr = (C.c_void_p * len(ws))(*[w.ctypes.data for w in ws])
    _check(lib().sslam_refiner_pack_host(arr, n_blocks, out.ctypes.data), "refiner_pack")
    return out


def pack_refiner_bf16(weights: list, n_blocks: int) -> np.ndarray:
    """bf16-mode image of the same weight list (LayerNorm folded into fc1 / fc2): uint8 buffer."""
    ws = [np.ascontiguousarray(w, np.float32) for w in weights]
    assert len(ws) == 4 + 8 * n_blocks
    out = np.empty(int(lib().sslam_refiner_bf16_bytes(n_blocks)), np.uint8)
    arr = (C.c_void_p * len(ws))(*[w.ctypes.data for w in ws])
    _check(lib().sslam_refiner_pack_bf16_host(arr, n_blocks, out.ctypes.data), "refiner_pack_bf16")
    return out


def resample_table(in_size: int, out_size: int, bicubic: bool):
    bounds = np.empty(out_size * 2, np.int32)
    coefs = np.empty(out_size * MAX_TAPS, np.int32)
    ks = lib().sslam_resample_table_host(in_size, out_size, int(bicubic), bounds.ctypes.data, coefs.ctypes.data, coefs.size)
    if ks < 0:
        _check(ks, "resample_table")
    return bounds, coefs[: out_size * ks].copy(), ks


# ------------------------------------------------------------------------------------------------ device calls
def preprocess_u8(img, size, tab_h, tab_v, out=None):
    n, h, w, _ = img.shape
    assert img.dtype == torch.uint8
    if out is None:
        out = torch.empty((n, 3, size, size), dtype=torch.float32, device=img.device)
    (bh, ch, kh), (bv, cv, kv) = tab_h, tab_v
    _run("preprocess_u8", lib().sslam_preprocess_u8, (img, bh, ch, bv, cv, out,),
         _dp(img), n, h, w, size, _dp(bh), _dp(ch), kh, _dp(bv), _dp(cv), kv, _dp(out))
    return out


def preprocess_u8_patches(img, size, tab_h, tab_v, out=None):
    """A0 written as the ViT's patch-embedding operand: (n, (size/16)^2, 768) bf16 (include/sslam_hip.h).  Returns None where the
    tiled kernel does not cover the resampling ratio (callers then take preprocess_u8 + vit_forward)."""
    n, h, w, _ = img.shape
    assert img.dtype == torch.uint8 and size % 16 == 0
    if out is None:
        out = torch.empty((n, (size // 16) ** 2, 768), dtype=torch.bfloat16, device=img.device)
    (bh, ch, kh), (bv, cv, kv) = tab_h, tab_v
    dev = common_device(img, bh, ch, bv, cv, out)
    with torch.cuda.device(dev):
        rc = lib().sslam_preprocess_u8_patches(_dp(img), n, h, w, size, _dp(bh), _dp(ch), kh, _dp(bv), _dp(cv), kv, _dp(out),
                                               C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    if rc == E_UNSUPPORTED:
        return None
    _check(rc, "preprocess_u8_patches")
    return out


def bn_tokens(tokens, n_prefix, group, gamma, beta, run_mean, run_var, train, eps, out=None, want_stats=True, bf16_copy=False,
              out_bf16=None):
    """bf16_copy=True: returns (out, mean, var, out_bf16) - the bf16 copy is written by the same kernel pass (into out_bf16 if given)."""
    n, t, c = tokens.shape
    assert c == C_FEAT and tokens.dtype == torch.float32
    cells = t - n_prefix
    if out is None:
        out = torch.empty((n, cells, c), dtype=torch.float32, device=tokens.device)
    assert out.dtype == torch.float32 and out.numel() == n * cells * c and out.is_contiguous()
    mean = var = None
    if train and want_stats:
        mean = torch.empty((n // group, c), dtype=torch.float32, device=tokens.device)
        var = torch.empty_like(mean)
    if bf16_copy:
        out_bf = out_bf16 if out_bf16 is not None else torch.empty((n, cells, c), dtype=torch.bfloat16, device=tokens.device)
        assert out_bf.dtype == torch.bfloat16 and out_bf.numel() == n * cells * c and out_bf.is_contiguous()
        _run("bn_tokens_bf16copy", lib().sslam_bn_tokens_bf16copy, (tokens, gamma, beta, run_mean, run_var, out, out_bf, mean, var,),
         _dp(tokens), n, t, n_prefix, group, _dp(gamma), _dp(beta), _dp(run_mean),
                                              _dp(run_var), int(bool(train)), C.c_float(eps), _dp(out), _dp(out_bf), _dp(mean),
                                              _dp(var))
        return out, mean, var, out_bf
    _run("bn_tokens", lib().sslam_bn_tokens, (tokens, gamma, beta, run_mean, run_var, out, mean, var,),
         _dp(tokens), n, t, n_prefix, group, _dp(gamma), _dp(beta), _dp(run_mean), _dp(run_var),
                                 int(bool(train)), C.c_float(eps), _dp(out), _dp(mean), _dp(var))
    return out, mean, var


def selector_saliency(feat, w1p, b1, w2, b2, hs, out=None, workspace=None):
    """workspace: optional caller-owned scratch tensor (sslam_workspace_bytes); allocated here through torch if absent and
    the launch form needs one (few-frame calls only)."""
    n, g = feat.shape[0], feat.shape[1]
    if out is None:
        out = torch.empty((n, g, g), dtype=torch.float32, device=feat.device)
    ws, wsb = _scratch(workspace, int(lib().sslam_selector_saliency_workspace_bytes(n, g)), feat.device)
    _run("selector_saliency", lib().sslam_selector_saliency_ws, (feat, w1p, b1, w2, b2, out, ws),
         _dp(feat), n, g, _dp(w1p), _dp(b1), _dp(w2), _dp(b2), hs, _dp(out), _dp(ws), wsb)
    return out


def pack_conv3x3_bf16(w: np.ndarray) -> np.ndarray:
    """-> uint16 array holding the bf16 bit patterns (view it as torch.bfloat16 on the device)."""
    w = np.ascontiguousarray(w, np.float32)
    hs = w.shape[0]
    assert w.shape == (hs, C_FEAT, 3, 3)
    out = np.empty(9 * C_FEAT * hs, np.uint16)
    _check(lib().sslam_pack_conv3x3_bf16_host(w.ctypes.data, hs, out.ctypes.data), "pack_conv3x3_bf16")
    return out


def to_bf16(x, out=None):
    x = x.contiguous()
    if out is None:
        out = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    _run("f32_to_bf16", lib().sslam_f32_to_bf16, (x, out,),
         _dp(x), _dp(out), x.numel())
    return out


def selector_saliency_bf16(feat_bf16, w1p_bf16, b1, w2, b2, hs, out=None):
    n, g = feat_bf16.shape[0], feat_bf16.shape[1]
    if out is None:
        out = torch.empty((n, g, g), dtype=torch.float32, device=feat_bf16.device)
    _run("selector_saliency_bf16", lib().sslam_selector_saliency_bf16, (feat_bf16, w1p_bf16, b1, w2, b2, out,),
         _dp(feat_bf16), n, g, _dp(w1p_bf16), _dp(b1), _dp(w2), _dp(b2), hs, _dp(out))
    return out


def select_keypoints(sal, K, radius=2, pct=0.5, want_idx=True, want_pixel=True, out=None):
    """out: optional (kp, sc, idx, px, st) tensors to write into (a launch group's slice of the caller's buffers)."""
    n, g = sal.shape[0], sal.shape[1]
    dev = sal.device
    if out is not None:
        kp, sc, idx, px, st = out
    else:
        kp = torch.empty((n, K, 2), dtype=torch.float32, device=dev)
        sc = torch.empty((n, K), dtype=torch.float32, device=dev)
        idx = torch.empty((n, K), dtype=torch.int32, device=dev) if want_idx else None
        px = torch.empty((n, K, 2), dtype=torch.float32, device=dev) if want_pixel else None
        st = torch.empty((n,), dtype=torch.int32, device=dev)
    _run("select_keypoints", lib().sslam_select_keypoints, (sal, kp, sc, idx, px, st,),
         _dp(sal), n, g, K, radius, C.c_double(pct), _dp(kp), _dp(sc), _dp(idx), _dp(px),
                                        _dp(st))
    return kp, sc, idx, px, st


def gather(feat, kp, out=None):
    n, g = feat.shape[0], feat.shape[1]
    K = kp.shape[1]
    if out is None:
        out = torch.empty((n, K, C_FEAT), dtype=torch.float32, device=feat.device)
    _run("gather", lib().sslam_gather, (feat, kp, out,),
         _dp(feat), n, g, _dp(kp), K, _dp(out))
    return out


def refine(x, packed, n_blocks, out=None):
    rows = x.numel() // C_FEAT
    if out is None:
        out = torch.empty(x.shape[:-1] + (D_OUT,), dtype=torch.float32, device=x.device)
    _run("refine", lib().sslam_refine, (x, packed, out,),
         _dp(x), rows, _dp(packed), n_blocks, _dp(out))
    return out


def gather_refine(feat, kp, packed, n_blocks, out=None):
    n, g = feat.shape[0], feat.shape[1]
    K = kp.shape[1]
    if out is None:
        out = torch.empty((n, K, D_OUT), dtype=torch.float32, device=feat.device)
    _run("gather_refine", lib().sslam_gather_refine, (feat, kp, packed, out,),
         _dp(feat), n, g, _dp(kp), K, _dp(packed), n_blocks, _dp(out))
    return out


def gather_refine_bf16(feat, kp, packed_bf16, n_blocks, out=None):
    n, g = feat.shape[0], feat.shape[1]
    K = kp.shape[1]
    if out is None:
        out = torch.empty((n, K, D_OUT), dtype=torch.float32, device=feat.device)
    _run("gather_refine_bf16", lib().sslam_gather_refine_bf16, (feat, kp, packed_bf16, out,),
         _dp(feat), n, g, _dp(kp), K, _dp(packed_bf16), n_blocks, _dp(out))
    return out


def refine_bf16(x, packed_bf16, n_blocks, out=None):
    rows = x.shape[0]
    if out is None:
        out = torch.empty((rows, D_OUT), dtype=torch.float32, device=x.device)
    _run("refine_bf16", lib().sslam_refine_bf16, (x, packed_bf16, out,),
         _dp(x), rows, _dp(packed_bf16), n_blocks, _dp(out))
    return out


def keypoint_intensity(img, size, tab_h, tab_v, kp_pixel, out=None):
    n, h, w, _ = img.shape
    K = kp_pixel.shape[1]
    if out is None:
        out = torch.empty((n, K), dtype=torch.float32, device=img.device)
    (bh, ch, kh), (bv, cv, kv) = tab_h, tab_v
    _run("keypoint_intensity", lib().sslam_keypoint_intensity, (img, bh, ch, bv, cv, kp_pixel, out,),
         _dp(img), n, h, w, size, _dp(bh), _dp(ch), kh, _dp(bv), _dp(cv), kv, _dp(kp_pixel), K,
                                          _dp(out))
    return out


def sim_argmax(d1, stride1, n1, d2, stride2, n2, n_pairs, want_s21=False, want_second=False, workspace=None):
    """workspace: optional caller-owned scratch tensor (sslam_workspace_bytes); batched calls without one get a torch
    allocation of n_pairs * n2 * 8 bytes here - the library itself never allocates."""
    dev = d1.device
    nn12 = torch.empty((n_pairs, n1), dtype=torch.int32, device=dev)
    s12 = torch.empty((n_pairs, n1), dtype=torch.float32, device=dev)
    nn21 = torch.empty((n_pairs, n2), dtype=torch.int32, device=dev)
    s21 = torch.empty((n_pairs, n2), dtype=torch.float32, device=dev) if want_s21 else None
    sec = torch.empty((n_pairs, n1), dtype=torch.float32, device=dev) if want_second else None
    ws, wsb = _scratch(workspace, int(lib().sslam_sim_argmax_workspace_bytes(n2, n_pairs)), dev)
    _run("sim_argmax", lib().sslam_sim_argmax_ws, (nn12, s12, nn21, s21, sec, d1, d2, ws),
         C.c_void_p(d1.data_ptr()), stride1, n1, C.c_void_p(d2.data_ptr()), stride2, n2, n_pairs,
                                  _dp(nn12), _dp(s12), _dp(nn21), _dp(s21), _dp(sec), _dp(ws), wsb)
    return nn12, s12, nn21, s21, sec


def match_finalize(nn12, s12, nn21, n1, n2, n_pairs, sc1, ss1, sc2, ss2, in1, in2, w_desc, w_sal, t_sal, t_sim, t_int, out=None):
    """out: optional (matches, quality, count) tensors to write into."""
    dev = nn12.device
    if out is not None:
        matches, quality, count = out
    else:
        matches = torch.empty((n_pairs, n1, 2), dtype=torch.int64, device=dev)
        quality = torch.empty((n_pairs, n1), dtype=torch.float32, device=dev)
        count = torch.empty((n_pairs,), dtype=torch.int32, device=dev)
    f = C.c_float
    _run("match_finalize", lib().sslam_match_finalize, (nn12, s12, nn21, matches, quality, count, sc1, sc2, in1, in2,),
         _dp(nn12), _dp(s12), _dp(nn21), n1, n2, n_pairs, C.c_void_p(sc1.data_ptr()), ss1,
                                      C.c_void_p(sc2.data_ptr()), ss2,
                                      None if in1 is None else C.c_void_p(in1.data_ptr()),
                                      None if in2 is None else C.c_void_p(in2.data_ptr()),
                                      f(w_desc), f(w_sal), f(t_sal), f(t_sim), f(t_int), _dp(matches), _dp(quality),
                                      _dp(count))
    return matches, quality, count


def pack_vit_linear(w: np.ndarray) -> np.ndarray:
    """(n_out, k_in) fp32 nn.Linear weight -> uint16 array of bf16 bit patterns in the ViT GEMM's streaming order."""
    w = np.ascontiguousarray(w, np.float32)
    out = np.empty(w.size, np.uint16)
    _check(lib().sslam_vit_pack_linear_host(w.ctypes.data, w.shape[0], w.shape[1], out.ctypes.data), "vit_pack_linear")
    return out


def pack_vit_mlp(w_up: np.ndarray, w_down: np.ndarray, row_scale: np.ndarray | None) -> np.ndarray:
    """(1536, 384) up_proj + (384, 1536) down_proj [+ per-row scale of down_proj] -> the fused MLP kernel's weight stream."""
    w_up, w_down = np.ascontiguousarray(w_up, np.float32), np.ascontiguousarray(w_down, np.float32)
    assert w_up.shape == (1536, 384) and w_down.shape == (384, 1536)
    rs = None if row_scale is None else np.ascontiguousarray(row_scale, np.float32)
    out = np.empty(2 * 1536 * 384, np.uint16)
    _check(lib().sslam_vit_pack_mlp_host(w_up.ctypes.data, w_down.ctypes.data, None if rs is None else rs.ctypes.data, out.ctypes.data),
           "vit_pack_mlp")
    return out


def vit_workspace_bytes(n_frames: int, size: int) -> int:
    b = int(lib().sslam_vit_workspace_bytes(n_frames, size))
    if b < 0:
        _check(b, "vit_workspace_bytes")
    return b


def vit_forward_patches(patches, size: int, weights: VitWeights, workspace, out=None):
    """patches (n, (size/16)^2, 768) bf16 from preprocess_u8_patches -> tokens (n, 5 + (size/16)^2, 384) fp32."""
    n = patches.shape[0]
    t = 5 + (size // 16) ** 2
    if out is None:
        out = torch.empty((n, t, C_FEAT), dtype=torch.float32, device=patches.device)
    _run("vit_forward_patches", lib().sslam_vit_forward_patches, (patches, workspace, out,),
         _dp(patches), n, size, C.byref(weights), _dp(workspace), workspace.numel() * workspace.element_size(), _dp(out))
    return out


def vit_forward(images_chw, weights: VitWeights, workspace, out=None):
    n, _, size, _ = images_chw.shape
    t = 5 + (size // 16) ** 2
    if out is None:
        out = torch.empty((n, t, C_FEAT), dtype=torch.float32, device=images_chw.device)
    _run("vit_forward", lib().sslam_vit_forward, (images_chw, workspace, out,),
         _dp(images_chw), n, size, C.byref(weights), _dp(workspace), workspace.numel() * workspace.element_size(),
                                   _dp(out))
    return out


def pack_vit_f32_linear(w: np.ndarray) -> np.ndarray:
    """(n_out, k_in) fp32 nn.Linear weight -> the same values in the fragment order of the fp32 ViT's per-layer GEMM."""
    w = np.ascontiguousarray(w, np.float32)
    out = np.empty(w.size, np.float32)
    _check(lib().sslam_vit_f32_pack_linear_host(w.ctypes.data, w.shape[0], w.shape[1], out.ctypes.data), "vit_f32_pack_linear")
    return out


def vit_f32_workspace_bytes(n_frames: int, size: int) -> int:
    b = int(lib().sslam_vit_f32_workspace_bytes(n_frames, size))
    if b < 0:
        _check(b, "vit_f32_workspace_bytes")
    return b


ATTN_ONE_PASS, ATTN_KEY_SPLIT, ATTN_KEY_SPLIT_MAX_FRAMES = 0, 1, 8          # include/sslam_hip.h


def vit_forward_f32(images_chw, weights: VitWeightsF32, workspace, out=None, attention_form: int | None = None):
    """(n, 3, S, S) fp32 -> tokens (n, 5 + (S/16)^2, 384) fp32 by the fp32-operand HIP ViT (reference numerics for A1).
    attention_form: None - the library's choice by this launch's frame count (key split up to 8 frames); ATTN_ONE_PASS /
    ATTN_KEY_SPLIT - the caller's (a batch cut into several launches passes the form of the whole batch to each)."""
    n, _, size, _ = images_chw.shape
    t = 5 + (size // 16) ** 2
    if out is None:
        out = torch.empty((n, t, C_FEAT), dtype=torch.float32, device=images_chw.device)
    if attention_form is None:
        _run("vit_forward_f32", lib().sslam_vit_forward_f32, (images_chw, workspace, out,),
             _dp(images_chw), n, size, C.byref(weights), _dp(workspace), workspace.numel() * workspace.element_size(), _dp(out))
    else:
        _run("vit_forward_f32_form", lambda *a: lib().sslam_vit_forward_f32_form(*a[:-1], int(attention_form), a[-1]), (images_chw, workspace, out,),
             _dp(images_chw), n, size, C.byref(weights), _dp(workspace), workspace.numel() * workspace.element_size(), _dp(out))
    return out

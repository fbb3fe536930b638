#!/usr/bin/env python3
"""Single-frame latency of the hot path, the shape every reference caller has (B = 1: visualize_matches_sequence.py:306-307,
test/test_performance.py:89-131 - whose protocol this follows: warm-up, then per-stage timers with a device synchronisation
around each stage, mean over the repeats).  Prints one JSON object; `measure()` is also called by bench.py for its `latency`
block.  Stages: A0 preprocess, A1 HIP ViT (batch 1), A2..A9 extract from tokens, M1 one pair."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "semantic-slam-master_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
import torch


def _timed(fn, reps, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def measure(pipe, toks, imgs, vit_pipe=None, reps=100):
    """pipe: SequencePipeline; toks (>= 2, T, 384) / imgs (>= 2, H, W, 3) on the device.  Returns ms per call, device time
    (HIP events around `reps` back-to-back calls) and host wall time of one synchronous extract + match step."""
    from sslam_amd import lib
    cfg, s = pipe.cfg, pipe.selector
    t1, i1 = toks[:1], imgs[:1]
    out = {"protocol": "B = 1, 10 warm-up + %d timed calls per stage (test/test_performance.py:89-131)" % reps}
    feat = pipe.features(t1)
    sal = lib.selector_saliency(feat, s.w1p, s.b1, s.w2, s.b2, s.hidden)
    kp, sc, idx, px, st = lib.select_keypoints(sal, cfg.num_keypoints, cfg.nms_radius, cfg.min_score_percentile)
    ex0, ex1 = pipe.extract(toks[:1], imgs[:1]), pipe.extract(toks[1:2], imgs[1:2])
    d = torch.cat([ex0["descriptors"], ex1["descriptors"]])
    scs = torch.cat([ex0["scores"], ex1["scores"]])
    it = torch.cat([ex0["intensity"], ex1["intensity"]])
    stage = {
        "A0_preprocess": _timed(lambda: pipe.preprocess(i1), reps),
        "A2_bn_tokens": _timed(lambda: pipe.features(t1), reps),
        "A3_selector_saliency": _timed(lambda: lib.selector_saliency(feat, s.w1p, s.b1, s.w2, s.b2, s.hidden), reps),
        "A45_select_keypoints": _timed(lambda: lib.select_keypoints(sal, cfg.num_keypoints, cfg.nms_radius, cfg.min_score_percentile), reps),
        "A67_gather_refine": _timed(lambda: lib.gather_refine(feat, kp, pipe.refiner.packed, pipe.refiner.n_blocks), reps),
        "M1_match_one_pair": _timed(lambda: pipe.match(d, scs, it), reps),
    }
    out["stage_ms"] = {k: round(v, 4) for k, v in stage.items()}
    out["extract_1_frame_tokens_in_ms"] = round(_timed(lambda: pipe.extract(t1, i1), reps), 4)
    if vit_pipe is not None:
        x = vit_pipe.preprocess(i1)
        out["vit_batch1_ms"] = round(_timed(lambda: vit_pipe.vit_hip.forward_features(x), max(10, reps // 4)), 4)
        out["vit_batch8_ms"] = round(_timed(lambda: vit_pipe.vit_hip.forward_features(vit_pipe.preprocess(imgs[:8])), max(10, reps // 4)), 4) if imgs.shape[0] >= 8 else None
    # host view: one synchronous step = extract the new frame + match it against the previous one
    prev = ex0

    def step():
        ex = pipe.extract(toks[1:2], imgs[1:2])
        m = pipe.match(torch.cat([prev["descriptors"], ex["descriptors"]]), torch.cat([prev["scores"], ex["scores"]]),
                       torch.cat([prev["intensity"], ex["intensity"]]))
        return int(m["match_count"][0])          # the caller reads the result: a real synchronisation

    for _ in range(10):
        step()
    t0 = time.perf_counter()
    for _ in range(reps):
        step()
    out["step_wall_ms_tokens_in"] = round((time.perf_counter() - t0) / reps * 1e3, 4)
    # the online loop on static buffers (sslam_amd/online.py): ordinary launches against replay of the captured HIP graph
    from sslam_amd.online import FrameStepper
    online = {}
    for name, p_, tok_in in (("tokens_in", pipe, True),) + ((("vit_inside", vit_pipe, False),) if vit_pipe is not None else ()):
        for graph in (False, True):
            st = FrameStepper(p_, imgs.shape[1], imgs.shape[2], use_graph=graph, tokens_in=tok_in)

            def frame(i=[0]):
                i[0] ^= 1
                o = st.step(imgs[i[0]], toks[i[0]] if tok_in else None)
                return o

            for _ in range(10):
                frame()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                frame()
            torch.cuda.synchronize()
            online[f"{name}_{'graph' if graph else 'launches'}_ms"] = round((time.perf_counter() - t0) / reps * 1e3, 4)
    out["online_step_ms"] = online
    return out


if __name__ == "__main__":
    import synth
    from sslam_amd.pipeline import ExtractorConfig, SequencePipeline
    from sslam_amd.vit import DinoV3ViT
    torch.manual_seed(0)
    dev = torch.device("cuda", 0)
    vp = SequencePipeline(ExtractorConfig(), synth.selector_state(0), synth.refiner_state(0), device=dev, vit=DinoV3ViT().to(dev).eval())
    toks = torch.from_numpy(synth.token_sequence(8, 28)).to(dev)
    imgs = torch.from_numpy(synth.image_sequence(8)).to(dev)
    res = measure(vp, toks, imgs, vit_pipe=vp)
    from sslam_amd import lib
    with lib.knobs(SSLAM_CONV_LATENCY_ROWS=0):
        res["A3_throughput_form_ms"] = round(_timed(lambda: vp.extract(toks[:1], imgs[:1]), 50), 4)
    print(json.dumps(res, indent=1))

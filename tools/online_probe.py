import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "semantic-slam-master_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch, synth
from sslam_amd import lib
from sslam_amd.pipeline import ExtractorConfig, SequencePipeline
from sslam_amd.online import FrameStepper
from sslam_amd.vit import DinoV3ViT
dev = torch.device("cuda", 0)
torch.manual_seed(0)
n = 6
imgs = torch.from_numpy(synth.image_sequence(n)).to(dev)
toks = torch.from_numpy(synth.token_sequence(n, 28)).to(dev)
pipe = SequencePipeline(ExtractorConfig(), synth.selector_state(0), synth.refiner_state(0), device=dev)
want = pipe.run(imgs, tokens=toks)
def run_stepper(st, with_tokens):
    res = []
    for i in range(n):
        n0 = lib.launch_count()
        o = st.step(imgs[i], toks[i] if with_tokens else None)
        res.append(({k: (v.clone() if v is not None else None) for k, v in o.items()}, lib.launch_count() - n0))
    return res
for graph in (False, True):
    st = FrameStepper(pipe, 480, 640, use_graph=graph, tokens_in=True)
    res = run_stepper(st, True)
    ok = True
    for i, (o, nl) in enumerate(res):
        for k in ("idx", "descriptors", "intensity", "scores", "saliency"):
            ok &= torch.equal(o[k], want[k][i])
        if i:
            c = int(o["match_count"]); ok &= c == int(want["match_count"][i - 1]) and torch.equal(o["matches"][:c], want["matches"][i - 1][:c]) and torch.equal(o["quality"][:c], want["quality"][i - 1][:c])
    print("tokens-in graph", graph, "equal to batch run:", ok, "launches per step:", [nl for _, nl in res])
    def t():
        for i in range(n): st.step(imgs[i], toks[i])
    for _ in range(5): t()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): t()
    torch.cuda.synchronize(); print("   ms per step (async, back to back): %.4f" % ((time.perf_counter() - t0) / 50 / n * 1e3))
    t0 = time.perf_counter()
    for _ in range(50):
        for i in range(n):
            o = st.step(imgs[i], toks[i]); int(o["match_count"]) if o["match_count"] is not None else None
    print("   ms per step (caller reads the count each frame): %.4f" % ((time.perf_counter() - t0) / 50 / n * 1e3))
for prec in ("bf16", "fp32"):
    pv = SequencePipeline(ExtractorConfig(), synth.selector_state(0), synth.refiner_state(0), device=dev, vit=DinoV3ViT().to(dev).eval(), vit_precision=prec)
    wantv = pv.run(imgs)
    outs = {}
    for graph in (False, True):
        st = FrameStepper(pv, 480, 640, use_graph=graph)
        res = run_stepper(st, False)
        outs[graph] = res
        ok = all(torch.equal(o[k], wantv[k][i]) for i, (o, _) in enumerate(res) for k in ("idx", "descriptors", "intensity"))
        print(prec, "ViT inside, graph", graph, "equal to batch run:", ok, "launches per step:", [nl for _, nl in res])
        def t():
            for i in range(n): st.step(imgs[i])
        for _ in range(5): t()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(30): t()
        torch.cuda.synchronize(); print("   ms per step (async): %.4f" % ((time.perf_counter() - t0) / 30 / n * 1e3))
        t0 = time.perf_counter()
        for _ in range(30):
            for i in range(n):
                o = st.step(imgs[i]); int(o["match_count"]) if o["match_count"] is not None else None
        print("   ms per step (count read each frame): %.4f" % ((time.perf_counter() - t0) / 30 / n * 1e3))
    same = all(torch.equal(a[0][k], b[0][k]) for a, b in zip(outs[False], outs[True]) for k in ("idx", "descriptors", "intensity", "scores"))
    print("   graph == eager stepping:", same)

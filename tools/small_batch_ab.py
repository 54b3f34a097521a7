"""Small batches, one batch in flight (what a synchronous Segmenter.__call__ runs): the GEMM tile model with the lone-round rule (default) against the
round-6a model (SYLBER_OPT_GEMM_MODEL = 6), same box, alternating (development aid; profiles/r06_small_tiles.md).  Forward + boundary detection on resident
inputs, and the whole synchronous call on host tensors."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sylber_amd import Segmenter
from sylber_amd.weights import synthetic_state_dict

sd = synthetic_state_dict(0, num_layers=9)
S = Segmenter(model_ckpt=sd)
enc = S.speech_model
g = torch.Generator().manual_seed(0)
print("| clips x seconds | forward + segment ms: lone-round rule | round-6a model | gain | Segmenter.__call__ ms: rule | round-6a | gain |")
print("|---|---:|---:|---:|---:|---:|---:|")
for clips, sec in ((1, 10), (1, 3), (2, 10), (4, 10), (8, 10), (1, 60), (16, 10)):
    n = int(sec * 16000)
    host = [torch.randn(1, n, generator=g) for _ in range(clips)]
    dev = torch.cat(host, 0).cuda()
    res = {}
    for rep in range(2):
        for model in (0, 6):
            enc.set_option(12, model)
            for _ in range(3):
                h = enc.forward(dev, None); enc.segment(h, 2.6, 0.8)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(30):
                h = enc.forward(dev, None); enc.segment(h, 2.6, 0.8)
            torch.cuda.synchronize()
            fwd = (time.perf_counter() - t0) / 30 * 1e3
            for _ in range(3):
                S(wav=host, in_second=True)
            ts = []
            for _ in range(20):
                t0 = time.perf_counter()
                S(wav=host, in_second=True)
                ts.append((time.perf_counter() - t0) * 1e3)
            ts.sort()
            res.setdefault(model, []).append((fwd, ts[10]))
    f0, c0 = min(r[0] for r in res[0]), min(r[1] for r in res[0])
    f6, c6 = min(r[0] for r in res[6]), min(r[1] for r in res[6])
    print("| %d x %d s | %.3f | %.3f | %+.1f %% | %.3f | %.3f | %+.1f %% |" % (clips, sec, f0, f6, (f6 / f0 - 1) * 100, c0, c6, (c6 / c0 - 1) * 100), flush=True)
enc.set_option(12, 0)

"""Segmenter.stream on 32 x 10 s host batches: ms per batch, the GPU's compute-completion gaps and the host's phases per batch"""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sylber_amd import Segmenter
from sylber_amd.synth import noise_batch
from sylber_amd.weights import synthetic_state_dict
S = Segmenter(model_ckpt=synthetic_state_dict(0))
wavs = [w[None, :].clone() for w in noise_batch(32, 160000, seed=1000)]
for o in S.stream([wavs] * 8): pass
for n in (40, 40, 40):
    S._trace = []
    t0 = time.perf_counter()
    for o in S.stream([wavs] * n): pass
    tot = (time.perf_counter() - t0) / n * 1e3
    tr = S._trace; S._trace = None
    evs = [m[2] for m in tr if m[0] == "compute issued"]
    gaps = [evs[i].elapsed_time(evs[i + 1]) for i in range(len(evs) - 1)]
    def dur(a, b):
        out, last = [], {}
        for m in tr:
            if m[0] == b and a in last: out.append((m[1] - last[a]) * 1e3)
            last[m[0]] = m[1]
        return statistics.median(out) if out else float("nan")
    print("n %d: %.2f ms per batch | GPU gaps median %.2f max %.2f | host: pad+upload %.2f, issue %.2f, wait for results %.2f ms" % (
        n, tot, statistics.median(gaps), max(gaps), dur("compute issued", "padded"), dur("padded", "compute issued"), dur("finish enter", "results on the host")))

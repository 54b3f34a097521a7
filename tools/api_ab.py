"""Same-process A/B of the host-side hand-over of Segmenter.__call__ (development aid, GPU box): padding threads x hidden-state D2H overlap"""
import os, sys, time, statistics, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sylber_amd import Segmenter
from sylber_amd.synth import noise_batch
from sylber_amd.weights import synthetic_state_dict
sd = synthetic_state_dict(0)
wavs = [w[None, :].clone() for w in noise_batch(32, 160000, seed=1)]
variants = {(t, o): Segmenter(model_ckpt=sd, host_pad_threads=t, overlap_d2h=o) for t in (1, 2) for o in (False, True)}
for S in variants.values():
    for _ in range(3): S(wav=wavs)
res = {k: [] for k in variants}
for rep in range(6):
    for k, S in variants.items():
        for _ in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter(); out = S(wav=wavs); res[k].append((time.perf_counter() - t0) * 1e3); del out
for k, v in res.items():
    print("pad threads %d, D2H overlap %-5s: min %.2f  med %.2f  p90 %.2f ms" % (k[0], k[1], min(v), statistics.median(v), sorted(v)[int(len(v) * 0.9)]), flush=True)

"""host timeline of one Segmenter.__call__ (N x 10 s host tensors, N = argv[1] or 32): where the call's milliseconds go"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sylber_amd import Segmenter
from sylber_amd.synth import noise_batch
from sylber_amd.weights import synthetic_state_dict
S = Segmenter(model_ckpt=synthetic_state_dict(0))
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 32
wavs = [w[None, :].clone() for w in noise_batch(NB, 160000, seed=1000)]
for _ in range(3):
    S(wav=wavs)
runs, gruns = [], []
for _ in range(15):
    S._trace, S._gpu_trace = [], []
    out = S(wav=wavs)
    torch.cuda.synchronize()
    runs.append(S._trace)
    gruns.append(S._gpu_trace)
    del out
names = [n for n, _ in runs[0]]
for i in range(1, len(names)):
    d = [(r[i][1] - r[i - 1][1]) * 1e3 for r in runs]
    print("%-58s median %.2f  min %.2f  max %.2f ms" % (names[i], statistics.median(d), min(d), max(d)))
tot = [(r[-1][1] - r[0][1]) * 1e3 for r in runs]
print("%-58s median %.2f" % ("total", statistics.median(tot)))
print("GPU timeline (timing events on the streams involved, ms after the call's first event):")
for i in range(1, len(gruns[0])):
    d = [r[0][1].elapsed_time(r[i][1]) for r in gruns]
    print("  %-56s median %.2f  min %.2f  max %.2f ms" % (gruns[0][i][0], statistics.median(d), min(d), max(d)))

"""Segmenter.stream soak: N batches (alternating two different ragged batches), every yielded array compared with the synchronous
__call__'s bits on the host"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sylber_amd import Segmenter
from sylber_amd.synth import syllable_wave
from sylber_amd.weights import synthetic_state_dict
n = int(sys.argv[1]) if len(sys.argv) > 1 else 600
S = Segmenter(model_ckpt=synthetic_state_dict(0))
rng = np.random.default_rng(3)
A = [syllable_wave(int(rng.integers(60000, 160000)), 40 + i) for i in range(32)]
B = [syllable_wave(int(rng.integers(20000, 90000)), 90 + i) for i in range(20)]
ref = {0: S(wav=A, in_second=False), 1: S(wav=B, in_second=False)}
ref = {k: [{kk: np.array(vv) for kk, vv in d.items()} for d in v] for k, v in ref.items()}
bad = 0
t0 = time.time()
for j, out in enumerate(S.stream((A if i % 2 == 0 else B for i in range(n)), in_second=False)):
    for g, e in zip(out, ref[j % 2]):
        if not (np.array_equal(g["hidden_states"], e["hidden_states"]) and np.array_equal(g["segments"], e["segments"])
                and np.array_equal(g["segment_features"], e["segment_features"], equal_nan=True)):
            bad += 1
            break
print("%d streamed batches in %.1f s, batches with any differing bit: %d" % (n, time.time() - t0, bad))

"""Segmenter.__call__ on a 32 x 10 s host batch with the call cut into 0 / 2 / 3 / 4 pipelined parts (Segmenter(call_split=n)), same box, alternating
(development aid; profiles/r06_call_split.md).  Prints ms per call (median / min of 20) per setting and checks that every setting returns the same bits."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sylber_amd import Segmenter
from sylber_amd.weights import synthetic_state_dict

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
SEC = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
OUTS = tuple(sys.argv[3].split(",")) if len(sys.argv) > 3 else ("segments", "segment_features", "hidden_states")
sd = synthetic_state_dict(0, num_layers=9)
g = torch.Generator().manual_seed(0)
wavs = [torch.randn(1, int(SEC * 16000), generator=g) for _ in range(B)]
segs = {n: Segmenter(model_ckpt=sd, call_split=n, outputs=OUTS) for n in (0, 2, 3, 4)}
ref = None
for n, S in segs.items():
    out = S(wav=wavs, in_second=False)
    out = S(wav=wavs, in_second=False)
    if ref is None:
        ref = out
    else:
        for a, b in zip(ref, out):
            for k in a:
                assert np.array_equal(a[k], b[k]), (n, k)
    del out
print("B = %d x %.0f s, outputs %s" % (B, SEC, ",".join(OUTS)))
for rep in range(2):
    for n, S in segs.items():
        ts = []
        for _ in range(20):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = S(wav=wavs, in_second=True)
            ts.append((time.perf_counter() - t0) * 1e3)
            del out
        ts.sort()
        print("call_split=%d  median %.2f ms  min %.2f  max %.2f" % (n, ts[10], ts[0], ts[-1]), flush=True)

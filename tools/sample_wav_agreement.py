"""Boundary agreement of every precision mode with the REFERENCE's fp32 tables on the committed sample clip
(tests/golden/e2e.npz: PCM of samples/sample.wav + the reference's own output) -> the floors of tests/test_gpu_e2e.py."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from sylber_amd import Segmenter  # noqa: E402
from sylber_amd.weights import synthetic_state_dict  # noqa: E402

g = np.load(os.path.join(ROOT, "tests", "golden", "e2e.npz"))
wav = torch.from_numpy(g["sample_pcm"].astype(np.float32) / 32768.0)[None, :]
wav = (wav - wav.mean()) / wav.std()
sd = synthetic_state_dict(0)
ref = g["sample_segments"]
rb = set(ref.reshape(-1).tolist())
print("| mode | hidden rel-RMS vs reference | boundaries of the reference found | table identical |")
print("|---|---:|---:|---|")
for prec in sys.argv[1:] or ["bf16", "fp16", "fp8", "fp32"]:
    S = Segmenter(model_ckpt=sd, precision=prec)
    o = S(wav=wav, in_second=False)
    h = o["hidden_states"].astype(np.float64)
    rel = np.sqrt(((h - g["sample_hidden"]) ** 2).mean() / (g["sample_hidden"].astype(np.float64) ** 2).mean())
    gb = set(o["segments"].reshape(-1).tolist())
    same = o["segments"].shape == ref.shape and np.array_equal(o["segments"], ref)
    print("| %s | %.2e | %d / %d = %.3f | %s |" % (prec, rel, len(rb & gb), len(rb), len(rb & gb) / len(rb), same), flush=True)
    del S

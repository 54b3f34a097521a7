"""addressing beyond 4 GB: a 48 x 30 s batch (conv0 output 4.7 GB, workspace ~9 GB) must reproduce, row for row and bit for bit,
what the same clips give in batches of 8 (utterances are independent; every GEMM tile is bit-identical across tile shapes)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sylber_amd import HubertEncoderHIP
from sylber_amd.synth import noise_batch
from sylber_amd.weights import synthetic_state_dict
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
B, N = int(os.environ.get("B", 48)), int(os.environ.get("N", 480000))
e = HubertEncoderHIP(synthetic_state_dict(0), precision=prec)
x = noise_batch(B, N, seed=5).cuda()
lens = [N - 1000 * (i % 7) for i in range(B)]
big = e.forward(x, lens)
torch.cuda.synchronize()
print(prec, "workspace GB", round(e.workspace_bytes() / 2 ** 30, 2), "finite", bool(torch.isfinite(big).all()))
bad = 0
for i in range(0, B, 8):
    small = e.forward(x[i:i + 8].contiguous(), lens[i:i + 8])
    if not torch.equal(small, big[i:i + 8]):
        bad += 1
        d = (small - big[i:i + 8]).abs().max().item()
        print("  rows", i, "differ, max abs", d)
seg_b, n_b, f_b = e.segment(big, 2.6, 0.8)
print("mismatching groups:", bad, "| segments per clip:", float(n_b.float().mean()))

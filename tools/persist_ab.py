"""same-box A/B inside the real forward: the 256x256 GEMM as one tile per workgroup (option -1) vs persistent with
cross-tile operand prefetch (default), per-kernel HIP-event times (development aid)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sylber_amd import HubertEncoderHIP
from sylber_amd.synth import noise_batch
from sylber_amd.weights import synthetic_state_dict
sd = synthetic_state_dict(0)
x = noise_batch(32, 160000, 1).cuda()
E = {0: HubertEncoderHIP(sd), -1: HubertEncoderHIP(sd)}
E[-1].set_option(3, -1)
assert torch.equal(E[0].forward(x, None), E[-1].forward(x, None))
keys = ["gemm_conv1", "gemm_conv2", "gemm_conv3", "gemm_conv4", "gemm_ffn1"]
for rep in range(3):
    for mode in (0, -1):
        e = E[mode]
        for _ in range(2): e.forward(x, None)
        e.set_profiling(True)
        for _ in range(8): e.forward(x, None)
        torch.cuda.synchronize()
        p = e.get_profile(); e.set_profiling(False)
        print("rep %d %-26s" % (rep, "persistent+prefetch" if mode == 0 else "one tile per workgroup"),
              " ".join("%s %.4f" % (k[5:], p[k] / 8) for k in keys), " sum %.4f ms" % (sum(p[k] for k in keys) / 8))

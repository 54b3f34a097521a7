"""same-box A/B inside the real forward of per-handle GEMM options (development aid): option 3 (GEMM_PERSISTENT) values"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sylber_amd import HubertEncoderHIP
from sylber_amd.synth import noise_batch
from sylber_amd.weights import synthetic_state_dict
sd = synthetic_state_dict(0)
x = noise_batch(32, 160000, 1).cuda()
vals = [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["0", "-1"])]
E = {v: HubertEncoderHIP(sd) for v in vals}
for v, e in E.items(): e.set_option(3, v)
ref = E[vals[0]].forward(x, None)
for v in vals[1:]: assert torch.equal(ref, E[v].forward(x, None))
keys = ["gemm_conv1", "gemm_conv2", "gemm_conv3", "gemm_qkv", "gemm_out", "gemm_ffn1", "gemm_ffn2"]
for rep in range(3):
    for v in vals:
        e = E[v]
        for _ in range(2): e.forward(x, None)
        e.set_profiling(True)
        for _ in range(8): e.forward(x, None)
        torch.cuda.synchronize()
        p = e.get_profile(); e.set_profiling(False)
        print("rep %d persist=%2d " % (rep, v), " ".join("%s %.4f" % (k[5:], p[k] / 8) for k in keys), " sum %.4f ms" % (sum(p[k] for k in keys) / 8))

import sys, os
sys.path.insert(0, os.getcwd())
import torch
from sylber_amd import HubertEncoderHIP
from sylber_amd.synth import noise_batch
from sylber_amd.weights import synthetic_state_dict
sd = synthetic_state_dict(0)
for prec in ("split16", "bf16"):
    enc = HubertEncoderHIP(sd, precision=prec)
    x = noise_batch(32, 160000, seed=3).cuda()
    for st in (1, 2, 3, 4, 7, 0):
        a = enc.forward(x, None, stop_stage=st).clone()
        a2 = enc.forward(x, None, stop_stage=st).clone()
        b = enc.forward(x[8:12].contiguous(), None, stop_stage=st)
        b2 = enc.forward(x[8:12].contiguous(), None, stop_stage=st)
        d = (a[8:12] - b).abs()
        print(prec, "stage", st, "B32 repro", bool(torch.equal(a, a2)), "B4 repro", bool(torch.equal(b, b2)), "max diff rows 8:12", float(d.max()),
              "n diff", int((d > 0).sum()), flush=True)

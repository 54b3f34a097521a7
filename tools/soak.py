"""soak: N pipelined steps (two handles, probed streams, boundary detection on side streams, as bench.py runs them) with EVERY
step's hidden states, segment tables, counts and pooled features compared bit for bit against the first step's -- rare races
(LDS-DMA hazards, stream ordering, workspace aliasing) would show up as a mismatch.   python tools/soak.py [steps] [precision] [clips = 32]
(24 clips: FFN1 / conv5 on the 192-row tile 46; 4 clips: the eight-wave small tiles)"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sylber_amd import HubertEncoderHIP
from sylber_amd.streams import concurrent_streams
from sylber_amd.synth import syllable_wave
from sylber_amd.weights import synthetic_state_dict
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
sd = synthetic_state_dict(0)
B, N = (int(sys.argv[3]) if len(sys.argv) > 3 else 32), 160000
x = torch.cat([syllable_wave(N, 500 + i) for i in range(B)], 0).cuda()
lens = [N - 3000 * (i % 5) for i in range(B)]
encs = [HubertEncoderHIP(sd, precision=prec) for _ in range(2)]
st = concurrent_streams(4, "cuda:0")
T = encs[0].num_frames(N)
ref_h = encs[0].forward(x, lens).clone(); ref = [t.clone() for t in encs[0].segment(ref_h, 2.6, 0.8)]
torch.cuda.synchronize()
nset = 4
bufs = [(torch.empty(B, T, 768, device="cuda"), (torch.empty(B, T, 2, dtype=torch.int64, device="cuda"), torch.empty(B, dtype=torch.int32, device="cuda"), torch.empty(B, T, 768, device="cuda"))) for _ in range(nset)]
done = [None] * nset
bad = [torch.zeros((), dtype=torch.int64, device="cuda") for _ in range(2)]
nref = ref[1].to(torch.int64)
mask = (torch.arange(T, device="cuda")[None, :] < nref[:, None])
t0 = time.time()
for i in range(steps):
    k, ks = i % 2, i % nset
    hidden, out = bufs[ks]
    with torch.cuda.stream(st[k]):
        if done[ks] is not None: st[k].wait_event(done[ks])
        encs[k].forward(x, lens, out=hidden)
        ready = torch.cuda.Event(); ready.record(st[k])
    with torch.cuda.stream(st[2 + k]):
        st[2 + k].wait_event(ready)
        encs[k].segment(hidden, 2.6, 0.8, out=out)
        mism = (hidden != ref_h).any().to(torch.int64) + (out[1] != ref[1]).any().to(torch.int64) \
             + ((out[0] != ref[0]) & mask[:, :, None]).any().to(torch.int64) + ((out[2] != ref[2]) & mask[:, :, None] & ~(out[2].isnan() & ref[2].isnan())).any().to(torch.int64)
        bad[k].add_(mism)  # (device-side comparison, one counter per side stream, one host sync at the end)
        ev = torch.cuda.Event(); ev.record(st[2 + k]); done[ks] = ev
torch.cuda.synchronize()
print("%s: %d pipelined steps in %.1f s, mismatching steps (upper bound): %d" % (prec, steps, time.time() - t0, int(bad[0] + bad[1])))

"""same-box A/B: ShardedSegmenter.run_stream with boundary detection + gather on a side stream per engine vs on the engine stream
(one rank, no collective: the exchange degenerates to hand-over, so this isolates the stream structure)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sylber_amd import HubertEncoderHIP
from sylber_amd.dist import ShardedSegmenter
from sylber_amd.weights import synthetic_state_dict

sd = synthetic_state_dict(0)
encs = [HubertEncoderHIP(sd) for _ in range(2)]
x = torch.randn(32, 160000, device="cuda")
for rep in range(3):
    for side in (True, False):
        sh = ShardedSegmenter(encs, segment_on_side_stream=side)
        for n in (6, 40):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for out in sh.run_stream([x] * n, None, max_segments=192):
                pass
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n * 1e3
        print("side stream" if side else "engine stream", "%.3f ms per step" % dt)

#!/bin/bash
# segment_kernel: where the 0.3 ms go.  "build" here (hipcc cross-compiles patched copies), "run" on the GPU box.
cd "$(dirname "$0")"
SRC=../../sylber_amd/csrc
OUT=seg_phase_bin
mkdir -p $OUT
variant() {  # name, sed script
    sed -e "$2" $SRC/segment.hip > $OUT/segment_$1.hip
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -fno-vectorize -ffp-contract=off -I$SRC -DSEG_SRC="\"$OUT/segment_$1.hip\"" -o $OUT/seg_$1 seg_phase.hip 2>&1 | grep -E "error"
}
if [ "$1" = build ]; then
    variant full 's/@@@//'
    variant p0 's|    // ---- phase 1: greedy scan|    if (T > 0) { if (tid == 0) nseg_out[b] = 0; return; }\n    // ---- phase 1: greedy scan|'
    variant p01 's|    // ---- phase 2: boundary refinement|    if (T > 0) { if (tid == 0) nseg_out[b] = nseg; return; }\n    // ---- phase 2: boundary refinement|'
    variant p012 's|    // ---- compaction + mean-pool|    if (T > 0) { if (tid == 0) nseg_out[b] = nseg; return; }\n    // ---- compaction + mean-pool|'
else
    python - <<'PY'
import os, sys
sys.path.insert(0, os.path.abspath("../.."))
import torch
from sylber_amd import HubertEncoderHIP
from sylber_amd.synth import syllable_wave
from sylber_amd.weights import synthetic_state_dict
e = HubertEncoderHIP(synthetic_state_dict(0))
x = torch.cat([syllable_wave(160000, seed=9000 + i) for i in range(32)], 0).float().cuda()
h = e.forward(x, None)
h.cpu().numpy().tofile("seg_phase_bin/hidden.bin")
print("hidden", tuple(h.shape))
PY
    for v in full p0 p01 p012; do printf "%-6s" $v; $OUT/seg_$v $OUT/hidden.bin 32 499; done
fi

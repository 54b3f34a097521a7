"""256-clip segment-agreement table of every precision mode against this library's fp32 parity mode (development aid; the floors of
tests/test_gpu_e2e.py::test_agreement_table_floors come from here)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sylber_amd.segmenter import HubertEncoderHIP
from sylber_amd.agreement import segment_agreement
from sylber_amd.weights import synthetic_state_dict
sd = synthetic_state_dict(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
truth = HubertEncoderHIP(sd, precision="fp32")
for prec in ("bf16", "fp16", "fp8", "split16"):
    e = HubertEncoderHIP(sd, precision=prec)
    r = segment_agreement(sd, e, n, truth=truth)
    print(prec, json.dumps({k: r[k] for k in ("clips", "tables_identical", "boundaries_fp32", "boundaries_found", "boundary_recall", "boundary_precision", "hidden_rel_rms_vs_fp32")}), flush=True)
    del e

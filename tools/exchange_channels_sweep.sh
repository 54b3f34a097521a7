#!/bin/bash
# The N > 1 step on a one-rank RCCL communicator (bench.py --exchange-selftest), swept over RCCL's channel count: how many CUs the
# collectives' kernels take from the 256 persistent GEMM workgroups they share the chip with.  Runs ON THE GPU BOX:
#   gpurun -- 'bash tools/exchange_channels_sweep.sh > gpurun_out/exchange_channels.md'
# One-rank scatter / gather move the root's own block through RCCL's copy kernels (device-to-device, zero remote bytes): what this
# measures is the kernels' footprint beside the compute, not xGMI.
echo "| NCCL_MIN/MAX_NCHANNELS | exchange ms/step | resident ms/step (same process) | overhead | results left sharded ms/step |"
echo "|---|---:|---:|---:|---:|"
for ch in default 1 2 4 8 16 32; do
  if [ "$ch" = default ]; then unset NCCL_MIN_NCHANNELS NCCL_MAX_NCHANNELS; else export NCCL_MIN_NCHANNELS=$ch NCCL_MAX_NCHANNELS=$ch; fi
  python bench.py --exchange-selftest --no-api --no-cpu-baseline --no-other-configs --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read())
r = d.get('resident_shards') or {}
n = d.get('results_left_sharded') or {}
print('| $ch | %.3f | %.3f | %+.1f %% | %s |' % (d['ms_per_step'], r.get('ms_per_step', float('nan')), 100 * (d['ms_per_step'] / r['ms_per_step'] - 1) if r.get('ms_per_step') else float('nan'), n.get('ms_per_step', n.get('error', '-'))))
"
done

for rep in 1 2; do for v in "" g1 g2; do
  if [ -z "$v" ]; then L=$(pwd)/sylber_amd/libsylber_hip.so; else L=$(pwd)/sylber_amd/libsylber_hip_$v.so; fi
  python tools/with_lib.py $L bench.py --no-api --no-cpu-baseline --no-other-configs --steps 20 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_forward']; print('variant [$v] step', d['ms_per_step'], 'conv0', k['conv0_gn_gelu'], 'conv1-5', round(sum(k['gemm_conv%d'%i] for i in range(1,6)),4), 'ffn1', k['gemm_ffn1'], 'posconv', k['posconv'])"
done; done

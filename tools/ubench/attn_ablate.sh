#!/bin/bash
# Builds patched copies of csrc/attention.hip (one ablation each) and times them on the GPU box (development aid).
# Run HERE with "build" (hipcc cross-compiles), then on the box with "run".
cd "$(dirname "$0")"
SRC=../../sylber_amd/csrc
OUT=attn_abl_bin
mkdir -p $OUT
variant() {  # name, sed script
    sed -e "$2" $SRC/attention.hip > $OUT/attention_$1.hip
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -fno-vectorize -I$SRC -DATTN_SRC="\"$OUT/attention_$1.hip\"" -o $OUT/attn_$1 attn_ablate.hip 2>&1 | grep -E "error" 
}
if [ "$1" = build ]; then
    variant base 's/@@@//'
    variant noexp 's/__builtin_amdgcn_exp2f(fmaf(\(.*\), LOG2E, -mb))/fmaf(\1, LOG2E, -mb)/'
    variant nodmawait 's/asm volatile("s_waitcnt vmcnt(0)" ::: "memory");/asm volatile("" ::: "memory");/'
    variant nobarrier 's/^        __syncthreads();$//'
    variant noshfl 's/mx = fmaxf(mx, __shfl_xor(mx, 32, 64));//'
    variant nopv 's/oacc\[qs\]\[ds\] = H16<FMT>::mfma(vf, pf\[qs\]\[j\], oacc\[qs\]\[ds\]);/oacc[qs][ds][0] += (float)pf[qs][j][0] + (float)vf[0];/'
    variant noqk 's/sacc\[qs\] = H16<FMT>::mfma(kf, qf\[qs\]\[ks\], sacc\[qs\]);/sacc[qs][ks] += (float)kf[0] * (float)qf[qs][ks][0];/'
else
    for v in base noexp nodmawait nobarrier noshfl nopv noqk; do
        for cfg in "32 499" "8 2999"; do printf "%-10s B,T=%-8s" $v "$cfg"; timeout 60 $OUT/attn_$v $cfg 30; done
    done
fi

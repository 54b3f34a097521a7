#!/usr/bin/env python3
"""Generate sylber_amd/csrc/gemm_asm_loop.inc: the hand-scheduled K loop of gemma_bf16_kernel (csrc/gemm_asm.hip).

The loop is emitted as ONE inline-asm statement so that the instruction order is exactly the one written here
(hipcc's scheduler moves LDS reads and LDS-DMA issue around freely otherwise; profiles/r03_gemm_variants_ab.md §6).

Geometry (fixed): 256x256 tile, 4 waves (one per SIMD), each wave 128x128 = 4x4 MFMA 32x32x16 tiles = 256 accumulator
registers; K step 32 (64-byte LDS rows), 4-slot ring of 32 KiB steps; wave w stages pieces w + 4 i (i < 4: X rows,
i >= 4: W rows) of every step with `buffer_load_dwordx4 ... lds`.

One K step s (fragment set P = s & 1 holds step s; set Q = P ^ 1 receives step s+1):
    s_waitcnt lgkmcnt(8)                      kk0 fragments of step s landed (the 8 kk1 reads may still fly)
    MFMA 0..15  (kk0)                         + the 8 LDS-DMA pieces of step s+3 -> slot (s+3)&3, one per two MFMAs
    s_waitcnt lgkmcnt(0)                      kk1 fragments of step s
    s_waitcnt vmcnt(16|8|0) ; s_barrier       my pieces of step s+1 landed (younger: steps s+2, s+3); behind the barrier
                                              everyone's have, and everyone is done reading slot (s)&3's fragments
    MFMA 16..31 (kk1)                         + ds_read_b128 x16 of step s+1 (kk0 first) -> set Q
Slot (s+3)&3 = (s-1)&3 was last read during step s-2 and those reads were waited for before barrier(s-1): free.
"""
import os

FM = FN = 4
SLOT = 32768
XT = 256 * 64


def mfma(P, kk, fm, fn):
    return f'MF " %[c{fm}{fn}], %[w{P}{kk}{fn}], %[x{P}{kk}{fm}], %[c{fm}{fn}]\\n"'


def q(s):
    return f'"{s}\\n"'


def reads(Q, slot):
    """the 16 fragment reads of one step from ring slot `slot` into set Q, kk0 first"""
    out = []
    hi = "h" if slot >= 2 else ""
    base = (slot & 1) * SLOT
    for kk in range(2):
        for f in range(FM):
            out.append(q(f"ds_read_b128 %[x{Q}{kk}{f}], %[ax{kk}{hi}] offset:{base + f * 2048}"))
        for f in range(FN):
            out.append(q(f"ds_read_b128 %[w{Q}{kk}{f}], %[aw{kk}{hi}] offset:{base + f * 2048}"))
    return out


def step(sm, dma, vmcnt, nxt):
    """K step with (s mod 4) = sm. dma: issue step s+3; vmcnt: None = no barrier; nxt: read step s+1's fragments"""
    P, Q = sm & 1, (sm & 1) ^ 1
    slot_r, slot_d = (sm + 1) & 3, (sm + 3) & 3
    L = [q(f"; ---- step {sm}: set {P}, dma {int(dma)}, vmcnt {vmcnt}"), q("s_waitcnt lgkmcnt(8)")]
    m = [mfma(P, kk, fm, fn) for kk in range(2) for fm in range(FM) for fn in range(FN)]
    rd = reads(Q, slot_r) if nxt else []
    for i in range(32):
        if i == 16:
            L.append(q("s_waitcnt lgkmcnt(0)"))
        L.append(m[i])
        if dma and i < 16:
            piece = i // 2
            if i % 2 == 0:
                L.append(q(f"s_add_u32 m0, %[lbase], {slot_d * SLOT + piece * 4096}"))
            else:
                rs = "rx" if piece < 4 else "rw"
                L.append(q(f"buffer_load_dwordx4 %[vo{piece}], %[{rs}], %[koff] offen lds"))
        if i == 15:
            if dma:
                L.append(q("s_add_u32 %[koff], %[koff], 64"))
            if vmcnt is not None:
                L.append(q(f"s_waitcnt vmcnt({vmcnt})"))
                L.append(q("s_barrier"))
        if i >= 16 and rd:
            # kk0 reads two per MFMA (16..19), kk1 reads one per MFMA (20..27)
            n = 2 if i < 20 else 1
            for _ in range(n):
                if rd:
                    L.append(rd.pop(0))
    assert not rd
    return L


def main():
    L = []
    L.append(q("; ---- fragments of step 0 (ring slot 0)"))
    L += reads(0, 0)
    # main loop: groups of 4 steps, all with DMA
    L.append(q("s_cmp_eq_u32 %[nloop], 0"))
    L.append(q("s_cbranch_scc1 L_gemma_tail_%="))
    L.append(q("L_gemma_loop_%=:"))
    for sm in range(4):
        L += step(sm, True, 16, True)
    L.append(q("s_sub_u32 %[nloop], %[nloop], 1"))
    L.append(q("s_cmp_lg_u32 %[nloop], 0"))
    L.append(q("s_cbranch_scc1 L_gemma_loop_%="))
    L.append(q("L_gemma_tail_%=:"))
    L += step(0, True, 16, True)
    L += step(1, False, 8, True)
    L += step(2, False, 0, True)
    L += step(3, False, None, False)
    L.append(q("s_nop 15"))
    L.append(q("s_nop 15"))

    outs = []
    for fm in range(FM):
        for fn in range(FN):
            outs.append(f'[c{fm}{fn}] "+a"(acc[{fm}][{fn}])')
    for P in range(2):
        for kk in range(2):
            for f in range(4):
                outs.append(f'[x{P}{kk}{f}] "=&v"(fx[{P}][{kk}][{f}])')
                outs.append(f'[w{P}{kk}{f}] "=&v"(fw[{P}][{kk}][{f}])')
    outs.append('[koff] "+s"(koff)')
    outs.append('[nloop] "+s"(nloop)')
    ins = []
    for kk in range(2):
        ins.append(f'[ax{kk}] "v"(ax[{kk}])')
        ins.append(f'[ax{kk}h] "v"(axh[{kk}])')
        ins.append(f'[aw{kk}] "v"(aw[{kk}])')
        ins.append(f'[aw{kk}h] "v"(awh[{kk}])')
    for i in range(8):
        ins.append(f'[vo{i}] "v"(voff[{i}])')
    ins += ['[rx] "s"(rx)', '[rw] "s"(rw)', '[lbase] "s"(lbase)']

    here = os.path.dirname(os.path.abspath(__file__))
    dst = os.path.join(here, "..", "sylber_amd", "csrc", "gemm_asm_loop.inc")
    with open(dst, "w") as f:
        f.write("// GENERATED by tools/gen_gemm_asm.py -- do not edit; the schedule is documented there.\n")
        f.write("// Expects: MF (mnemonic string literal), acc[4][4], fx[2][2][4], fw[2][2][4], ax/axh/aw/awh[2], voff[8], rx, rw,\n")
        f.write("// lbase, koff, nloop in scope.\n")
        f.write("asm volatile(\n")
        for l in L:
            f.write("    " + l + "\n")
        f.write("    : " + ",\n      ".join(outs) + "\n")
        f.write("    : " + ",\n      ".join(ins) + "\n")
        f.write('    : "scc", "memory");   // m0 is written too: a reserved register the compiler re-materialises before each of its own uses\n')
    print("wrote", os.path.normpath(dst), len(L), "lines")


if __name__ == "__main__":
    main()

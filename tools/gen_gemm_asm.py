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

OUTDIR = os.environ.get("GEN_GEMM_ASM_OUT") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "sylber_amd", "build", "gen_exp")

FM = FN = 4
SLOT = 32768
XT = 256 * 64


def mfma(P, kk, fm, fn):
    return f'MF " %[c{fm}{fn}], %[w{P}{kk}{fn}], %[x{P}{kk}{fm}], %[c{fm}{fn}]\\n"'


def q(s):
    return f'"{s}\\n"'


def write_asm(f, L, outs, ins, clobbers='"scc", "memory"'):
    """One asm statement that leaves M0 as it found it.  The loops re-point M0 for every LDS-DMA instruction; M0 is a RESERVED register
    (hipcc refuses it as a clobber: "may lead to undefined behaviour"), so the statement saves it in a scalar of its own and puts it back
    (s_nop: an M0 write needs a wait state before the compiler's next LDS-DMA / movrel) -- the surrounding code may keep a live M0."""
    f.write("{ int m0_keep_;\n")
    f.write("asm volatile(\n")
    f.write("    " + q("s_mov_b32 %[m0k], m0") + "\n")
    for l in L:
        f.write("    " + l + "\n")
    f.write("    " + q("s_mov_b32 m0, %[m0k]") + "\n")
    f.write("    " + q("s_nop 0") + "\n")
    f.write("    : " + ",\n      ".join(list(outs) + ['[m0k] "=&s"(m0_keep_)']) + "\n")
    f.write("    : " + ",\n      ".join(ins) + "\n")
    f.write("    : " + clobbers + "); }\n")


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


# knock-out switches for timing experiments (results are wrong with any of them off): see VARIANTS
OPT = {"dma": True, "barrier": True, "reads": True, "vmwait": True, "cpol": "", "spread": False, "freeze": False, "ndma": 8, "dmaop": "buffer_load_dwordx4"}


def step(sm, dma, vmcnt, nxt):
    """K step with (s mod 4) = sm. dma: issue step s+3; vmcnt: None = no barrier; nxt: read step s+1's fragments"""
    dma = dma and OPT["dma"]
    nxt = nxt and OPT["reads"]
    P, Q = sm & 1, (sm & 1) ^ 1
    slot_r, slot_d = (sm + 1) & 3, (sm + 3) & 3
    L = [q(f"; ---- step {sm}: set {P}, dma {int(dma)}, vmcnt {vmcnt}"), q("s_waitcnt lgkmcnt(8)")]
    m = [mfma(P, kk, fm, fn) for kk in range(2) for fm in range(FM) for fn in range(FN)]
    rd = reads(Q, slot_r) if nxt else []
    # after[i] = instructions that follow MFMA i
    after = [[] for _ in range(32)]
    if dma:
        for piece in range(OPT["ndma"]):
            rs = "rx" if piece < 4 else "rw"
            i0 = 4 * piece if OPT["spread"] else 2 * piece          # spread: one piece per four MFMAs over the whole step
            after[i0].append(q(f"s_add_u32 m0, %[lbase], {slot_d * SLOT + piece * 4096}"))
            after[i0 + 1].append(q(f"{OPT['dmaop']} %[vo{piece}], %[{rs}], %[koff] offen{OPT['cpol']} lds"))
        if not OPT["freeze"]:
            after[31 if OPT["spread"] else 15].append(q("s_add_u32 %[koff], %[koff], 64"))
    if vmcnt is not None:
        if OPT["vmwait"]:
            v = vmcnt - 4 if (OPT["spread"] and dma) else vmcnt     # spread: only 4 pieces of step s+3 precede the barrier
            after[15].append(q(f"s_waitcnt vmcnt({v}) lgkmcnt(0)"))
        if OPT["barrier"]:
            after[15].append(q("s_barrier"))
    # fragment reads of step s+1: kk0 two per MFMA (16..19), kk1 one per MFMA (20..27)
    for i in range(16, 32):
        n = 2 if i < 20 else 1
        for _ in range(n):
            if rd:
                after[i].append(rd.pop(0))
    for i in range(32):
        L.append(m[i])
        L += after[i]
    assert not rd
    return L


VARIANTS = {
    0: {},                                                      # the shipping schedule
    1: {"dma": False},
    2: {"barrier": False},
    3: {"reads": False},
    4: {"dma": False, "barrier": False, "reads": False, "vmwait": False},
    5: {"vmwait": False},
    6: {"cpol": " sc1"},
    7: {"cpol": " nt"},
    8: {"cpol": " sc0 sc1"},
    9: {"spread": True},
    10: {"spread": True, "cpol": " sc1"},
    11: {"freeze": True},                                       # every step re-requests the same K slice (cache-hot source)
    12: {"reads": False, "barrier": False, "vmwait": False},    # MFMA + DMA only
    13: {"dma": False, "barrier": False, "vmwait": False},      # MFMA + fragment reads only
    14: {"reads": False, "barrier": False, "vmwait": False, "ndma": 4},                          # MFMA + half the DMA instructions
    15: {"reads": False, "barrier": False, "vmwait": False, "dmaop": "buffer_load_dword"},     # MFMA + 8 DMA of 4 B per lane
    16: {"reads": False, "barrier": False, "vmwait": False, "spread": True},                     # MFMA + DMA spread over the step
    17: {"reads": False, "barrier": False, "vmwait": False},    # as 12; the kernel requests whole 128-byte lines (8 rows x 128 B per piece)
    18: {},                                                     # as 0 with whole-line requests
}


def emit(var):
    OPT.update({"dma": True, "barrier": True, "reads": True, "vmwait": True, "cpol": "", "spread": False, "freeze": False, "ndma": 8, "dmaop": "buffer_load_dwordx4"})
    OPT.update(VARIANTS[var])
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
    dst = os.path.join(OUTDIR, "gemm_asm_loop.inc" if var == 0 else f"gemm_asm_loop_v{var}.inc")
    with open(dst, "w") as f:
        f.write("// GENERATED by tools/gen_gemm_asm.py -- do not edit; the schedule is documented there.\n")
        f.write("// Expects: MF (mnemonic string literal), acc[4][4], fx[2][2][4], fw[2][2][4], ax/axh/aw/awh[2], voff[8], rx, rw,\n")
        f.write("// lbase, koff, nloop in scope.\n")
        write_asm(f, L, outs, ins)
    print("wrote", os.path.normpath(dst), len(L), "lines")




# ======================================================================================================================
# K64 layout (tile id 80): 128-byte LDS rows (K step 64), TWO 64-KiB slots.  One LDS-DMA instruction then requests 8 whole
# 128-byte lines instead of 16 half lines -- measured with the knock-out variants above, the L1 handles requests per line
# and the half-line form made the DMA, not the MFMAs, the longest stream of the loop (profiles/r03_gemm_asm_loop.md).
#
# One K step j (slot j & 1) = four slices kk of 16 MFMAs; fragment set kk & 1 (8 fragments) holds slice kk:
#   slice 0..2:  lgkmcnt(0) | 16 MFMA + 8 ds_read of slice kk+1 -> other set   (+ the tail pieces of DMA(j+1) -> other slot)
#   slice 3:     lgkmcnt(0) | MFMA 0,1 | vmcnt(0) ; s_barrier | MFMA 2.. + 8 ds_read of (j+1, 0) from the other slot
#                                                              + the head pieces of DMA(j+2) -> THIS slot
# Behind the barrier of (j, 3) every wave has read its last fragments of slot j & 1 (waited at the head of the slice) and
# every wave's pieces of step j+1 have landed.
K64 = {"nhead": 7, "head_stride": 2, "tail_stride": 2, "cpol": "", "fn": 4, "nw": 4, "bn": 256}


def k64_reads(setq, slot, kk):
    out = []
    h = "h" if slot else ""
    for f in range(4):
        out.append(q(f"ds_read_b128 %[x{setq}{f}], %[ax{kk}{h}] offset:{f * 4096}"))
    for f in range(K64["fn"]):
        out.append(q(f"ds_read_b128 %[w{setq}{f}], %[aw{kk}{h}] offset:{f * 4096}"))
    return out


def k64_dma(piece, slot):
    nw = K64["nw"]
    rs = "rx" if piece < 32 // nw else "rw"              # wave w stages pieces w + nw i; the first 32 pieces are X rows
    stage = (256 + K64["bn"]) * 128
    return (q(f"s_add_u32 m0, %[lbase], {slot * stage + piece * nw * 1024}"),
            q(f"buffer_load_dwordx4 %[vo{piece}], %[{rs}], %[koff] offen{K64['cpol']} lds"))


def k64_step(par, tail, head, nxt=True, first=False):
    FN_ = K64["fn"]
    n_mf = 4 * FN_                       # MFMAs per slice
    n_rd = 4 + FN_                       # fragment reads per slice
    n_dma = (256 + K64["bn"]) // 8 // K64["nw"]           # pieces per wave and step
    L = [q(f"; ---- K64 step parity {par}: tail {int(tail)} head {int(head)} next {int(nxt)}")]
    nh = min(K64["nhead"], (n_mf - 3 + K64["head_stride"] - 1) // K64["head_stride"])
    tail_pieces = list(range(nh, n_dma)) if tail else []
    for kk in range(4):
        S = kk & 1
        after = [[] for _ in range(n_mf)]
        pre = [[] for _ in range(n_mf)]
        if kk < 3:
            rd = k64_reads(S ^ 1, par, kk + 1)
            for i in range(n_rd):
                after[i].append(rd[i])
            # tail pieces of DMA(j+1) -> the other slot, from MFMA 1 of slice 0 on
            i = 1 if kk == 0 else 0
            while tail_pieces and i < n_mf and kk < 2:
                p = tail_pieces.pop(0)
                a, b = k64_dma(p, par ^ 1)
                pre[i].append(a)
                after[i].append(b)
                if not tail_pieces:
                    after[i].append(q("s_add_u32 %[koff], %[koff], 128"))
                i += K64["tail_stride"]
        else:
            after[1].append(q("s_waitcnt vmcnt(0)"))
            after[1].append(q("s_barrier"))
            if nxt:
                rd = k64_reads(S ^ 1, par ^ 1, 0)
                for i in range(n_rd):
                    after[2 + i].append(rd[i])
            if head:
                i = 3
                for p in range(nh):
                    assert i <= n_mf - 1
                    a, b = k64_dma(p, par)
                    pre[i].append(a)
                    after[i].append(b)
                    i += K64["head_stride"]
        L.append(q("s_waitcnt lgkmcnt(0)"))
        n = 0
        for fm in range(4):
            for fn in range(FN_):
                L += pre[n]
                srcc = "0" if (first and kk == 0) else f"%[c{fm}{fn}]"     # the tile's first slice starts the accumulators
                L.append(f'MF " %[c{fm}{fn}], %[w{S}{fn}], %[x{S}{fm}], {srcc}\\n"')
                L += after[n]
                n += 1
    assert not tail_pieces
    return L


def emit_k64(var, opts, fn=4, nw=4):
    bn = 64 * fn if nw == 4 else 128 * fn                # 4 waves: 2 x 2 grid; 8 waves: 2 x 4 grid
    K64.update({"nhead": 7, "head_stride": 2, "tail_stride": 2, "cpol": "", "fn": fn, "nw": nw, "bn": bn})
    K64.update(opts)
    L = [q("; ---- fragments of (step 0, slice 0)")]
    L += k64_reads(0, 0, 0)
    L += k64_step(0, False, True, first=True)           # j = 0: DMA(1) came with the prologue
    L.append(q("s_cmp_eq_u32 %[nloop], 0"))
    L.append(q("s_cbranch_scc1 L_gemmb_tail_%="))
    L.append(q("L_gemmb_loop_%=:"))
    L += k64_step(1, True, True)
    L += k64_step(0, True, True)
    L.append(q("s_sub_u32 %[nloop], %[nloop], 1"))
    L.append(q("s_cmp_lg_u32 %[nloop], 0"))
    L.append(q("s_cbranch_scc1 L_gemmb_loop_%="))
    L.append(q("L_gemmb_tail_%=:"))
    L += k64_step(1, True, True)                        # j = nj - 3
    L += k64_step(0, True, False)                       # j = nj - 2
    L += k64_step(1, False, False, nxt=False)           # j = nj - 1
    L.append(q("s_barrier"))                             # every wave is done reading operand tiles (fragments waited above)
    L.append(q("s_nop 15"))
    outs = [f'[c{fm}{f}] "=a"(acc[{fm}][{f}])' for fm in range(4) for f in range(fn)]
    for S in range(2):
        for f in range(4):
            outs.append(f'[x{S}{f}] "=&v"(fx[{S}][{f}])')
        for f in range(fn):
            outs.append(f'[w{S}{f}] "=&v"(fw[{S}][{f}])')
    outs += ['[koff] "+s"(koff)', '[nloop] "+s"(nloop)']
    ins = []
    for kk in range(4):
        ins += [f'[ax{kk}] "v"(ax[{kk}])', f'[ax{kk}h] "v"(axh[{kk}])', f'[aw{kk}] "v"(aw[{kk}])', f'[aw{kk}h] "v"(awh[{kk}])']
    ins += [f'[vo{i}] "v"(voff[{i}])' for i in range((256 + bn) // 8 // nw)]
    ins += ['[rx] "s"(rx)', '[rw] "s"(rw)', '[lbase] "s"(lbase)']
    here = os.path.dirname(os.path.abspath(__file__))
    name = "gemm_asm_k64" + ("" if nw == 4 else f"_w{nw}") + ("" if fn == (4 if nw == 4 else 2) else f"_n{fn}") + ("" if var == 0 else f"_v{var}") + ".inc"
    dst = os.path.join(OUTDIR, name)
    with open(dst, "w") as f:
        f.write("// GENERATED by tools/gen_gemm_asm.py (K64 layout) -- do not edit; the schedule is documented there.\n")
        write_asm(f, L, outs, ins)
    print("wrote", os.path.normpath(dst), len(L), "lines")


# ======================================================================================================================
# X3 ring (tile ids 85 / 91 / 97): the K64 loop with THREE slots for the X operand (the activations, which the forward reads
# cold: written by the previous kernel, at best in the MALL) and two for W (the weights, L2-resident).  X(j+2) is requested
# during step j and has two K steps to land, W(j+2) one, as before.  LDS: X ring 3 x 32 KiB at 0, W ring 2 x BN x 128 B behind
# it (256x256 tile: exactly the 160 KiB of a CU).  The X slot rotates through registers (xr: read-slot byte offset, the four X
# fragment addresses advanced by VALU adds once per step); the W slot stays a compile-time parity of the 2x unrolled loop.
#   step j, slices 0-1:  rest of W(j+1) -> W slot (j+1)&1, then X(j+2) -> X slot (j+2)%3 (= the slot step j-1 read), koff += 128
#   step j, slice 3:     lgkmcnt(0) | MFMA 0,1 | vmcnt(#X pieces) ; s_barrier | rotate xr, advance X addresses |
#                        ds_read (j+1, 0) | first pieces of W(j+2) -> W slot j&1
# VMEM operations retire in order, so at the barrier "all but the youngest #X pieces" = every piece of step j+1.
X3 = {}


def x3_reads(setq, wslot, kk):
    out = []
    h = "h" if wslot else ""
    for f in range(X3["fm"]):
        out.append(q(f"ds_read_b128 %[x{setq}{f}], %[ax{kk}] offset:{f * 4096}"))
    for f in range(X3["fn"]):
        out.append(q(f"ds_read_b128 %[w{setq}{f}], %[aw{kk}{h}] offset:{f * 4096}"))
    return out


def x3_dma_w(i, wslot):
    nw, nxp = X3["nw"], 8 * X3["fm"] // X3["nw"]
    return (q(f"s_add_u32 m0, %[lbase], {3 * X3['xt'] + wslot * X3['bn'] * 128 + (i - nxp) * nw * 1024}"),
            q(f"buffer_load_dwordx4 %[vo{i}], %[rw], %[koff] offen lds"))


def x3_dma_x(i):
    # X runs one K step ahead of W: its byte offset is kofx = koff + 128 (an instruction offset would move the LDS address too)
    return (q(f"s_add_u32 m0, %[xwl], {i * X3['nw'] * 1024}"),
            q(f"buffer_load_dwordx4 %[vo{i}], %[rx], %[kofx] offen lds"))


def x3_tap_update():
    """conv 3-tap chunk-major K order (X3["tap"]): the X byte offset of K step s is 128 (s / 3) + (0, 2048, 1024)[s % 3] -- for each
    64-channel chunk: tap 0, tap 2, tap 1 -- so that the rows tap 2 of output row m shares with tap 0 of row m + 1 are re-read ONE
    step later (L2-hot) instead of 8-16 steps later (profiles/r04_conv_fetch_account.md: the 1.5x over-fetch of the conv GEMMs).
    kofx is carried (not derived from koff), ph = step mod 3 of the step kofx points at."""
    return [q("s_cmp_eq_u32 %[ph], 0"),
            q("s_cselect_b32 %[dk], %[c2048], %[cm1024]"),
            q("s_cmp_eq_u32 %[ph], 2"),
            q("s_cselect_b32 %[dk], %[cm896], %[dk]"),
            q("s_add_u32 %[kofx], %[kofx], %[dk]"),
            q("s_add_u32 %[ph], %[ph], 1"),
            q("s_cmp_eq_u32 %[ph], 3"),
            q("s_cselect_b32 %[ph], 0, %[ph]")]


def x3_step(par, tail_w, x_next, head, vm, nxt=True, first=False, ploads=()):
    """ploads: residual-prefetch loads of this step (asm lines): issued behind the step's LDS-DMA pieces, in the free MFMA gaps of
    slices 1 and 2, and left outstanding by the step's wait (vm is raised by their number: VMEM retires in order)"""
    FN_ = X3["fn"]
    ploads = list(ploads)
    vm = vm + len(ploads)
    nw = X3["nw"]
    FM_ = X3["fm"]
    nxp, nwp = 8 * FM_ // nw, X3["bn"] // 8 // nw
    n_mf, n_rd = FM_ * FN_, FM_ + FN_
    XT = X3["xt"]
    L = [q(f"; ---- X3 step parity {par}: tail_w {int(tail_w)} x_next {int(x_next)} head {int(head)} vmcnt {vm}")]
    nh = min(nwp - 1, (n_mf - 3 + 1) // 2)                   # W pieces issued in slice 3 (positions 3, 5, ...)
    queue = []                                               # (m0 line, dma line) of slices 0-1, W first then X
    if tail_w:
        queue += [x3_dma_w(i, par ^ 1) for i in range(nxp + nh, nxp + nwp)]
    if x_next:
        xq = [x3_dma_x(i) for i in range(nxp)]
        if not X3.get("tap"):
            xq[0] = (q("s_add_u32 %[kofx], %[koff], 128") + "\n    " + xq[0][0], xq[0][1])
        queue += xq
    bump = tail_w or x_next
    npieces_01 = len(queue)
    for kk in range(4):
        S = kk & 1
        after = [[] for _ in range(n_mf)]
        pre = [[] for _ in range(n_mf)]
        if kk < 3:
            rd = x3_reads(S ^ 1, par, kk + 1)
            for i in range(n_rd):
                after[i].append(rd[i])
            i = 1 if kk == 0 else 0
            # one piece per two MFMAs where the two slices have room for that, else one per MFMA (wave tile 128x64 on four waves)
            stride = 2 if (n_mf - 1 + 1) // 2 + (n_mf + 1) // 2 >= npieces_01 else 1
            while queue and i < n_mf and kk < 2:
                a, b = queue.pop(0)
                pre[i].append(a)
                after[i].append(b)
                if not queue and bump:
                    after[i].append(q("s_add_u32 %[koff], %[koff], 128"))
                    if x_next and X3.get("tap"):
                        after[i] += x3_tap_update()
                i += stride
            if kk >= 1 and ploads:
                # behind the last DMA piece (slice 1) / behind the fragment reads (slice 2), one load per two MFMAs
                j = max(i, n_rd) if kk == 1 else n_rd
                assert not queue
                while ploads and j < n_mf:
                    after[j].append(ploads.pop(0))
                    j += 2
        else:
            after[1].append(q(f"s_waitcnt vmcnt({vm})"))
            after[1].append(q("s_barrier"))
            if nxt:
                # X(j+3) goes where step j was read; then rotate the read slot and advance the four X fragment addresses
                after[1] += [q("s_add_u32 %[xwl], %[lbase], %[xr]"),
                             q(f"s_add_u32 %[xr], %[xr], 0x{XT:x}"),
                             q(f"s_cmp_eq_u32 %[xr], 0x{3 * XT:x}"),
                             q(f"s_cselect_b32 %[dlt], %[cneg], 0x{XT:x}"),
                             q("s_cselect_b32 %[xr], 0, %[xr]")]
                after[1] += [q(f"v_add_u32_e32 %[ax{k}], %[dlt], %[ax{k}]") for k in range(4)]
                rd = x3_reads(S ^ 1, par ^ 1, 0)
                for i in range(n_rd):
                    after[min(2 + i, n_mf - 1)].append(rd[i])
            if head:
                i = 3
                for p in range(nh):
                    assert i <= n_mf - 1
                    a, b = x3_dma_w(nxp + p, par)
                    pre[i].append(a)
                    after[i].append(b)
                    i += 2
        L.append(q("s_waitcnt lgkmcnt(0)"))
        n = 0
        for fm in range(FM_):
            for fn in range(FN_):
                L += pre[n]
                srcc = "0" if (first and kk == 0) else f"%[c{fm}{fn}]"
                L.append(f'MF " %[c{fm}{fn}], %[w{S}{fn}], %[x{S}{fm}], {srcc}\\n"')
                L += after[n]
                n += 1
    assert not queue and not ploads
    return L


def emit_x3(fn=4, nw=4, pre_e=0, pre_cols=0, tap=False, fm=4):
    """pre_e = E > 0: the residual-prefetch form for the fp32-residual epilogue (EPI_F32_RESLN, tile 91): the loop's last 2 E
    steps are peeled, and the peeled steps plus steps nj-3 and nj-2 carry the 16 FN loads of the wave's residual tile
    (buffer_load_dwordx4, MFMA C layout: lane = row, 4 columns) into registers that stay live until the epilogue: the 50 MB
    read of the residual stream rides under the K loop instead of standing, chip-wide, behind it (tools/resln_cost.py: that
    read costs 9 us with hot and 22 us with cold operands, of a 36 / 58 us out-projection).  Loads retire in order, so each
    step's wait leaves that step's loads outstanding and the next step's wait collects them: they have one K step to arrive,
    like the X pieces.  The statement ends with vmcnt(0): the compiler does not know these registers are load results."""
    # fm = X fragments (32 rows) per wave: 4 = the 256-row tiles; 3 = their 192-row siblings (round 6: a second tile HEIGHT, so that a launch
    # whose 256-row tiles leave a partial round of the 256 persistent workgroups can be cut into 3/4-size tiles instead: X slot 24 KiB)
    bn = 64 * fn if nw == 4 else 128 * fn
    X3.update({"fn": fn, "nw": nw, "bn": bn, "tap": tap, "fm": fm, "xt": 64 * fm * 128})
    nxp = 8 * fm // nw
    assert not (pre_e and fm != 4)
    L = [q("; ---- fragments of (step 0, slice 0)")]
    L += x3_reads(0, 0, 0)
    L += x3_step(0, False, False, True, nxp, first=True)     # j = 0: W(1), X(1), X(2) came with the prologue
    L.append(q("s_cmp_eq_u32 %[nloop], 0"))
    L.append(q("s_cbranch_scc1 L_gemmx_tail_%="))
    L.append(q("L_gemmx_loop_%=:"))
    L += x3_step(1, True, True, True, nxp)
    L += x3_step(0, True, True, True, nxp)
    L.append(q("s_sub_u32 %[nloop], %[nloop], 1"))
    L.append(q("s_cmp_lg_u32 %[nloop], 0"))
    L.append(q("s_cbranch_scc1 L_gemmx_loop_%="))
    L.append(q("L_gemmx_tail_%=:"))
    pre_names = []
    if pre_e:
        assert nw == 4
        # consumption order of the epilogue: fragment column fn outermost, then row block fm, then 4-column run g
        loads = []
        pre_cols = pre_cols or fn                            # fragment columns prefetched (the epilogue loads the others itself)
        for f in range(pre_cols):
            for m in range(4):
                for g in range(4):
                    nm = f"rr{m}{f}{g}"
                    pre_names.append(nm)
                    loads.append(q(f"buffer_load_dwordx4 %[{nm}], %[vres], %[rres], %[sres{m}] offen offset:{f * 128 + g * 32}"))
        nsteps = 2 * pre_e + 2
        per = [len(loads) // nsteps + (1 if i < len(loads) % nsteps else 0) for i in range(nsteps)]
        assert max(per) <= 6
        chunks, k = [], 0
        for n in per:
            chunks.append(loads[k:k + n]); k += n
        for e in range(pre_e):                               # peeled full steps (parities 1, 0, ...)
            L += x3_step(1, True, True, True, nxp, ploads=chunks[2 * e])
            L += x3_step(0, True, True, True, nxp, ploads=chunks[2 * e + 1])
        L += x3_step(1, True, True, True, nxp, ploads=chunks[-2])
        L += x3_step(0, True, False, False, 0, ploads=chunks[-1])
    else:
        L += x3_step(1, True, True, True, nxp)                   # j = nj - 3: X(nj - 1), first pieces of W(nj - 1)
        L += x3_step(0, True, False, False, 0)                   # j = nj - 2: rest of W(nj - 1)
    L += x3_step(1, False, False, False, 0, nxt=False)       # j = nj - 1
    if pre_e:
        L.append(q("s_waitcnt vmcnt(0)"))
    L.append(q("s_barrier"))
    L.append(q("s_nop 15"))
    outs = [f'[c{m}{f}] "=a"(acc[{m}][{f}])' for m in range(fm) for f in range(fn)]
    # residual registers: the LAST 16 runs (consumed last) in the AGPRs the 128x96 wave tile leaves free, the rest in VGPRs (the fewer registers the
    # kernel holds, the more of the OTHER stream's LayerNorm / conv0 waves fit beside it on the SIMD: bench.py runs two batches in flight)
    for i, nm in enumerate(pre_names):
        m, f, g = int(nm[2]), int(nm[3]), int(nm[4])
        outs.append(f'[{nm}] "=&{"a" if i >= len(pre_names) - 16 else "v"}"(rr[{m}][{f}][{g}])')
    for S in range(2):
        for f in range(fm):
            outs.append(f'[x{S}{f}] "=&v"(fx[{S}][{f}])')
        for f in range(fn):
            outs.append(f'[w{S}{f}] "=&v"(fw[{S}][{f}])')
    # in/out operands are early-clobber too: an input-only operand that happens to hold the same VALUE as one of them (the
    # residual block offset 0 and the ring position xr = 0, found the hard way) would otherwise be given the same register
    outs += [f'[ax{k}] "+&v"(axc[{k}])' for k in range(4)]
    outs += ['[koff] "+&s"(koff)', '[nloop] "+&s"(nloop)', '[xr] "+&s"(xr)', '[xwl] "=&s"(xwl)', '[dlt] "=&s"(dlt)']
    outs += ['[kofx] "+&s"(kofx)', '[ph] "+&s"(ph)', '[dk] "=&s"(dk)'] if tap else ['[kofx] "=&s"(kofx)']
    ins = []
    for kk in range(4):
        ins += [f'[aw{kk}] "v"(aw[{kk}])', f'[aw{kk}h] "v"(awh[{kk}])']
    ins += [f'[vo{i}] "v"(voff[{i}])' for i in range((64 * fm + bn) // 8 // nw)]
    ins += ['[rx] "s"(rx)', '[rw] "s"(rw)', '[lbase] "s"(lbase)', '[cneg] "s"(cneg)']
    if tap:
        ins += ['[c2048] "s"(c2048)', '[cm1024] "s"(cm1024)', '[cm896] "s"(cm896)']
    if pre_e:
        ins += ['[vres] "v"(vres)', '[rres] "s"(rres)'] + [f'[sres{m}] "s"(sres[{m}])' for m in range(4)]
    here = os.path.dirname(os.path.abspath(__file__))
    name = "gemm_asm_x3" + ("" if fm == 4 else f"_m{fm}") + ("" if nw == 4 else f"_w{nw}") + ("" if fn == (4 if nw == 4 else 2) else f"_n{fn}") + (f"_p{pre_e}" + (f"c{pre_cols}" if pre_cols and pre_cols != fn else "") if pre_e else "") + ("_t" if tap else "") + ".inc"
    dst = os.path.join(OUTDIR, name)
    with open(dst, "w") as f:
        f.write("// GENERATED by tools/gen_gemm_asm.py (X3 ring) -- do not edit; the schedule is documented there.\n")
        write_asm(f, L, outs, ins)
    print("wrote", os.path.normpath(dst), len(L), "lines")


# ======================================================================================================================
# Y3 (round 6, tile id 47): the X3 ring loop of tile 97 on v_mfma_f32_16x16x32_bf16 -- the instruction shape the vendor library issues on this
# part.  profiles/r06_mfma_shape.md: under the package power cap a register-only stream of 16x16x32 sustains 8.5 % more FLOP/s than 32x32x16 (a
# quarter of the accumulator traffic per instruction, half per FLOP).  Same LDS image, same LDS-DMA pieces, same ring, same barrier count as X3; what
# changes is how a wave cuts its 128 x (32 FN) tile into instructions and therefore which bytes a fragment register holds:
#   * a fragment = 16 rows x 32 k: lane l reads row l % 16, 16-byte chunk (4 h + l / 16) ^ swizzle(row) of K-step slice h (two slices of 32 k per
#     step where X3 has four of 16 k); the wave holds GX = 2 FM X fragments (its tokens) and GW = 2 FN W fragments per slice
#   * a slice = GX x GW MFMAs of 16 cycles, X-fragment-major: X fragment i is dead behind its GW MFMAs and is REFILLED IN PLACE with fragment i of the
#     next slice (one ds_read_b128 per GW MFMAs, a register ring instead of X3's two fragment sets); W fragments are double-buffered by slice parity
#   * in-order LDS returns make every wait a count: the generator keeps the queue of outstanding reads and emits lgkmcnt(#younger) before the
#     first MFMA that needs a fragment (asserted identical at the loop's back edge)
#   * slice 1 of step j: behind X fragment GB's MFMAs -- every read of step j has been issued in slice 0 -- vmcnt + lgkmcnt(0) + s_barrier, then
#     the burst of step j+1's first fragments (X 0..GB, all W) and the first pieces of W(j+2), as in X3
# The fp32 chain of an output element adds 32-k blocks where the 32x32x16 loops add 16-k blocks: results are NOT bit-identical to the other
# tiles, so this tile is never picked by the cost model -- it is the measured answer to "what does the instruction shape buy in the real loop".
Y3 = {}


class ReadQueue:
    """outstanding ds_reads in issue order; LDS operations of a wave return in order"""
    def __init__(self):
        self.q = []

    def issue(self, tag):
        assert tag not in self.q, tag
        self.q.append(tag)
        assert len(self.q) <= 15, "lgkmcnt is a 4-bit counter"

    def need(self, *tags):
        """asm lines that make every one of `tags` available (nothing when earlier waits already covered them)"""
        idx = [self.q.index(t) for t in tags if t in self.q]
        if not idx:
            return []
        i = max(idx)
        n = len(self.q) - i - 1
        self.q = self.q[i + 1:]
        return [q(f"s_waitcnt lgkmcnt({n})")]

    def drain(self):
        self.q = []


def y3_read_x(rq, i, h):
    rq.issue(("x", i))
    return q(f"ds_read_b128 %[x{i}], %[ax{h}] offset:{i * 2048}")


def y3_read_w(rq, P, j, wslot):
    rq.issue(("w", P, j))
    return q(f"ds_read_b128 %[w{P}{j}], %[aw{P}{'h' if wslot else ''}] offset:{j * 2048}")


def y3_dma_w(i, wslot):
    nw, nxp = Y3["nw"], 8 * Y3["fm"] // Y3["nw"]
    return (q(f"s_add_u32 m0, %[lbase], {3 * Y3['xt'] + wslot * Y3['bn'] * 128 + (i - nxp) * nw * 1024}"),
            q(f"buffer_load_dwordx4 %[vo{i}], %[rw], %[koff] offen lds"))


def y3_dma_x(i):
    return (q(f"s_add_u32 m0, %[xwl], {i * Y3['nw'] * 1024}"),
            q(f"buffer_load_dwordx4 %[vo{i}], %[rx], %[kofx] offen lds"))


def y3_step(rq, par, tail_w, x_next, head, vm, nxt=True, first=False):
    """one K step (64 k) of parity par (= its W slot): slice 0 carries the rest of W(j+1) and X(j+2), slice 1 the barrier, the first
    fragments of step j+1 and the first pieces of W(j+2)"""
    GX, GW, nw, GB = 2 * Y3["fm"], 2 * Y3["fn"], Y3["nw"], Y3["gb"]
    nxp, nwp = 8 * Y3["fm"] // nw, Y3["bn"] // 8 // nw
    n_mf = GX * GW
    XT = Y3["xt"]
    L = [q(f"; ---- Y3 step parity {par}: tail_w {int(tail_w)} x_next {int(x_next)} head {int(head)} vmcnt {vm}")]
    nh = min(nwp - Y3.get("wtail", 1), (n_mf - (GB + 1) * GW - Y3.get("hpos", 2) + 3) // 4)   # W pieces issued behind the barrier (one per four MFMAs); the rest in the next step's slice 0
    queue = []
    if tail_w:
        queue += [y3_dma_w(i, par ^ 1) for i in range(nxp + nh, nxp + nwp)]
    if x_next:
        xq = [y3_dma_x(i) for i in range(nxp)]
        if not Y3.get("tap"):
            xq[0] = (q("s_add_u32 %[kofx], %[koff], 128") + "\n    " + xq[0][0], xq[0][1])
        queue += xq
    bump = tail_w or x_next
    for h in range(2):
        P = h
        pre = [[] for _ in range(n_mf)]
        after = [[] for _ in range(n_mf)]
        if h == 0:
            # W fragments of slice 1 (set 1; this step's W slot): one per MFMA from MFMA 1 on; X fragment i of slice 1 behind group i
            for j in range(GW):
                after[1 + j].append(("rw", 1, j, par))
            for i in range(GX):
                after[i * GW + GW - 1].append(("rx", i, 1))
            pos = Y3["dpos"]
            stride = Y3["dstride"] if pos + Y3["dstride"] * (len(queue) - 1) < n_mf else 2
            while queue:
                assert pos < n_mf
                a, b = queue.pop(0)
                pre[pos].append(a)
                after[pos].append(b)
                if not queue and bump:
                    after[pos].append(q("s_add_u32 %[koff], %[koff], 128"))
                    if x_next and Y3.get("tap"):
                        after[pos] += x3_tap_update()
                pos += stride
        else:
            bpos = (GB + 1) * GW - 1                          # behind X fragment GB's last MFMA
            after[bpos].append(("barrier",))
            if nxt:
                after[bpos] += [q("s_add_u32 %[xwl], %[lbase], %[xr]"),
                                q(f"s_add_u32 %[xr], %[xr], 0x{XT:x}"),
                                q(f"s_cmp_eq_u32 %[xr], 0x{3 * XT:x}"),
                                q(f"s_cselect_b32 %[dlt], %[cneg], 0x{XT:x}"),
                                q("s_cselect_b32 %[xr], 0, %[xr]")]
                after[bpos] += [q(f"v_add_u32_e32 %[ax{k}], %[dlt], %[ax{k}]") for k in range(2)]
                burst = [("rx", 0, 0)] + [("rw", 0, j, par ^ 1) for j in range(GW)] + [("rx", i, 0) for i in range(1, GB + 1)]
                for n, r in enumerate(burst):
                    after[min(bpos + n, n_mf - 1)].append(r)
                for i in range(GB + 1, GX):
                    after[i * GW + GW - 1].append(("rx", i, 0))
            if head:
                pos = bpos + Y3["hpos"]
                for p_ in range(nh):
                    assert pos < n_mf
                    a, b = y3_dma_w(nxp + p_, par)
                    pre[pos].append(a)
                    after[pos].append(b)
                    pos += 4
        n = 0
        for i in range(GX):
            L += rq.need(("x", i), *([("w", P, j) for j in range(GW)] if i == 0 else []))
            for j in range(GW):
                L += pre[n]
                srcc = "0" if (first and h == 0) else f"%[c{i}{j}]"
                L.append(f'MF " %[c{i}{j}], %[w{P}{j}], %[x{i}], {srcc}\\n"')
                for item in after[n]:
                    if isinstance(item, tuple):
                        if item[0] == "barrier":
                            L.append(q(f"s_waitcnt vmcnt({vm}) lgkmcnt(0)"))
                            L.append(q("s_barrier"))
                            rq.drain()
                        elif item[0] == "rx":
                            L.append(y3_read_x(rq, item[1], item[2]))
                        else:
                            L.append(y3_read_w(rq, item[1], item[2], item[3]))
                    else:
                        L.append(item)
                n += 1
    assert not queue
    return L


def emit_y3(fn=2, nw=8, tap=False, fm=4, gb=3):
    # schedule knobs for same-box A/B builds (SYLBER_BUILD_VARIANT; defaults = the shipping schedule): Y3_GB = X fragment behind whose MFMAs the
    # barrier of slice 1 sits, Y3_DSTRIDE / Y3_DPOS = spacing / first position of slice 0's LDS-DMA pieces, Y3_HPOS = first W piece behind the barrier
    gb = int(os.environ.get("Y3_GB", gb))
    Y3["dstride"] = int(os.environ.get("Y3_DSTRIDE", 4))
    Y3["dpos"] = int(os.environ.get("Y3_DPOS", 2))
    Y3["hpos"] = int(os.environ.get("Y3_HPOS", 2))
    Y3["wtail"] = int(os.environ.get("Y3_WTAIL", 1))          # W pieces of step j+1 left for slice 0 of step j (0: all of them behind the barrier of step j-1)
    bn = 64 * fn if nw == 4 else 128 * fn
    Y3.update({"fn": fn, "nw": nw, "bn": bn, "tap": tap, "fm": fm, "xt": 64 * fm * 128, "gb": gb})
    GX, GW = 2 * fm, 2 * fn
    nxp = 8 * fm // nw
    rq = ReadQueue()
    L = [q("; ---- fragments of (step 0, slice 0)")]
    L.append(y3_read_x(rq, 0, 0))
    for j in range(GW):
        L.append(y3_read_w(rq, 0, j, 0))
    for i in range(1, GX):
        L.append(y3_read_x(rq, i, 0))
    L += y3_step(rq, 0, False, False, True, nxp, first=True)     # j = 0: W(1), X(1), X(2) came with the prologue
    entry = list(rq.q)
    L.append(q("s_cmp_eq_u32 %[nloop], 0"))
    L.append(q("s_cbranch_scc1 L_gemmy_tail_%="))
    L.append(q("L_gemmy_loop_%=:"))
    L += y3_step(rq, 1, True, True, True, nxp)
    L += y3_step(rq, 0, True, True, True, nxp)
    assert rq.q == entry, "the read queue must be the same at the loop's back edge"
    L.append(q("s_sub_u32 %[nloop], %[nloop], 1"))
    L.append(q("s_cmp_lg_u32 %[nloop], 0"))
    L.append(q("s_cbranch_scc1 L_gemmy_loop_%="))
    L.append(q("L_gemmy_tail_%=:"))
    L += y3_step(rq, 1, True, True, True, nxp)                   # j = nj - 3
    L += y3_step(rq, 0, True, False, False, 0)                   # j = nj - 2
    L += y3_step(rq, 1, False, False, False, 0, nxt=False)       # j = nj - 1
    assert not rq.q
    L.append(q("s_barrier"))
    L.append(q("s_nop 15"))
    outs = [f'[c{i}{j}] "=a"(acc[{i}][{j}])' for i in range(GX) for j in range(GW)]
    outs += [f'[x{i}] "=&v"(fx[{i}])' for i in range(GX)]
    outs += [f'[w{P}{j}] "=&v"(fw[{P}][{j}])' for P in range(2) for j in range(GW)]
    outs += [f'[ax{k}] "+&v"(axc[{k}])' for k in range(2)]
    outs += ['[koff] "+&s"(koff)', '[nloop] "+&s"(nloop)', '[xr] "+&s"(xr)', '[xwl] "=&s"(xwl)', '[dlt] "=&s"(dlt)']
    outs += ['[kofx] "+&s"(kofx)', '[ph] "+&s"(ph)', '[dk] "=&s"(dk)'] if tap else ['[kofx] "=&s"(kofx)']
    ins = []
    for P in range(2):
        ins += [f'[aw{P}] "v"(aw[{P}])', f'[aw{P}h] "v"(awh[{P}])']
    ins += [f'[vo{i}] "v"(voff[{i}])' for i in range((64 * fm + bn) // 8 // nw)]
    ins += ['[rx] "s"(rx)', '[rw] "s"(rw)', '[lbase] "s"(lbase)', '[cneg] "s"(cneg)']
    if tap:
        ins += ['[c2048] "s"(c2048)', '[cm1024] "s"(cm1024)', '[cm896] "s"(cm896)']
    name = "gemm_asm_y3" + ("" if fm == 4 else f"_m{fm}") + ("" if nw == 4 else f"_w{nw}") + ("" if fn == (4 if nw == 4 else 2) else f"_n{fn}") + ("_t" if tap else "") + ".inc"
    dst = os.path.join(OUTDIR, name)
    with open(dst, "w") as f:
        f.write("// GENERATED by tools/gen_gemm_asm.py (Y3: the X3 ring on 16x16x32 MFMAs) -- do not edit; the schedule is documented there.\n")
        write_asm(f, L, outs, ins)
    print("wrote", os.path.normpath(dst), len(L), "lines")


# ======================================================================================================================
# MXFP8 form of the X3 loop (BASELINE configs[4], gemm_asm_f8.hip): the SAME byte geometry -- 256 rows x 128-byte LDS rows per
# operand and K step, three X slots, two W slots, one barrier per step, the same LDS-DMA pieces and the same fragment addresses
# (chunk (2 kk + h) ^ swizzle of a row) -- but a row is 128 e4m3 values, contracted by v_mfma_scale_f32_32x32x64_f8f6f4: a K step is
# TWO slices of 4 x FN MFMAs of 64 cycles (the bf16 loop: four slices of 32-cycle MFMAs), and the operand of slice s of a fragment
# is what the bf16 loop reads for kk = 2 s and kk = 2 s + 1: two ds_read_b128 into the halves of ONE 8-register tuple.  An asm
# operand cannot be addressed by halves, so the fragment registers are literal (v128 .. v255, declared clobbered).
# Block scales (E8M0, one byte per row and 32 k; memory layout [K / 64][rows][2]): lane (r, h) needs, per fragment and slice, the
# byte of (its row, K block 2 s + h).  Per step and fragment two 16-bit loads (slice 2 j and 2 j + 1: buffer_load_ushort +
# byte of (its row, K block 2 s + h).  Per step, fragment and slice ONE buffer_load_ubyte (lane offset = 2 row + h), requested one
# step ahead at the HEAD of the step -- older than the step's X pieces, so the step's wait covers them (VMEM retires in order); the
# MFMA selects byte 0.  Scale registers are double-buffered by step parity.
F8 = {}


def f8_xreg(S, fm):
    b = 128 + S * 32 + fm * 8
    return b


def f8_wreg(S, fn):
    b = 192 + S * 8 * F8["fn"] + fn * 8
    return b


def f8_reads(S, wslot, s):
    """the fragment reads of slice s (of the step whose W slot is wslot) into set S: 8 registers per fragment"""
    out = []
    h = "h" if wslot else ""
    for half, kk in enumerate((2 * s, 2 * s + 1)):
        for f in range(4):
            b = f8_xreg(S, f) + 4 * half
            out.append(q(f"ds_read_b128 v[{b}:{b + 3}], %[ax{kk}] offset:{f * 4096}"))
        for f in range(F8["fn"]):
            b = f8_wreg(S, f) + 4 * half
            out.append(q(f"ds_read_b128 v[{b}:{b + 3}], %[aw{kk}{h}] offset:{f * 4096}"))
    return out


def f8_scale_loads(Q):
    """requests of the next step's scales into register set Q (plus the scalar offsets of its two slices): ONE BYTE per lane,
    fragment and slice (the lane's own K block: its byte offset carries + h), zero-extended, so the MFMA selects byte 0.
    (A 16-bit pair per slice merged with buffer_load_short_d16_hi does not work on this part: with SRAM ECC a D16 load
    clears the other half of its destination.)"""
    L = [q("s_add_u32 %[ksx1], %[ksx], %[xstep]"), q("s_add_u32 %[ksw1], %[ksw], %[wstep]")]
    for f in range(4):
        L.append(q(f"buffer_load_ubyte %[xs{Q}0{f}], %[vsx], %[rxs], %[ksx] offen offset:{f * 64}"))
        L.append(q(f"buffer_load_ubyte %[xs{Q}1{f}], %[vsx], %[rxs], %[ksx1] offen offset:{f * 64}"))
    for f in range(F8["fn"]):
        L.append(q(f"buffer_load_ubyte %[ws{Q}0{f}], %[vsw], %[rws], %[ksw] offen offset:{f * 64}"))
        L.append(q(f"buffer_load_ubyte %[ws{Q}1{f}], %[vsw], %[rws], %[ksw1] offen offset:{f * 64}"))
    L += [q("s_add_u32 %[ksx], %[ksx1], %[xstep]"), q("s_add_u32 %[ksw], %[ksw1], %[wstep]")]
    return L


def f8_scale_piece(Q):
    """LDS-staged scales (256x192 tile: 16 KiB of LDS are free): ONE LDS-DMA piece per wave and step -- wave 0 / 2 the X scales of
    the next step (256 rows x 2 slices x 2 bytes = 1 KiB: lanes 0-31 slice a, lanes 32-63 slice b), wave 1 / 3 the W scales (same
    image; the two waves of a pair write identical bytes) -- instead of 2 (4 + FN) byte loads per wave: the texture path works per
    request, and 14 more requests per step beside 14 LDS-DMA pieces cost a third of the loop's rate."""
    return [q(f"s_add_u32 m0, %[lsc], {Q * 2048}"),
            q("s_nop 0"),
            q("buffer_load_dwordx4 %[vsc], %[rsc], %[ksc] offen lds"),
            q("s_add_u32 %[ksc], %[ksc], %[scstep]")]


def f8_scale_reads(Q):
    L = []
    for s_ in range(2):
        for f in range(4):
            L.append(q(f"ds_read_u8 %[xs{Q}{s_}{f}], %[axs] offset:{Q * 2048 + s_ * 512 + f * 64}"))
        for f in range(F8["fn"]):
            L.append(q(f"ds_read_u8 %[ws{Q}{s_}{f}], %[aws] offset:{Q * 2048 + 1024 + s_ * 512 + f * 64}"))
    return L


def f8_dma_w(i, wslot):
    nxp = 8
    return (q(f"s_add_u32 m0, %[lbase], {98304 + wslot * F8['bn'] * 128 + (i - nxp) * 4096}"),
            q(f"buffer_load_dwordx4 %[vo{i}], %[rw], %[koff] offen lds"))


def f8_dma_x(i):
    return (q(f"s_add_u32 m0, %[xwl], {i * 4096}"),
            q(f"buffer_load_dwordx4 %[vo{i}], %[rx], %[kofx] offen lds"))


def f8_step(par, tail_w, x_next, head, vm, s_next, nxt=True, first=False):
    """one K step (128 e4m3 per row): parity par = W slot = scale set; s_next: request the scales of the next step"""
    FN_ = F8["fn"]
    nxp, nwp = 8, F8["bn"] // 32
    n_mf, n_rd = 4 * FN_, 2 * (4 + FN_)
    L = [q(f"; ---- F8 step parity {par}: tail_w {int(tail_w)} x_next {int(x_next)} head {int(head)} vmcnt {vm} scales_next {int(s_next)}")]
    nh = min(nwp - 1, (n_mf - 3 + 1) // 2)
    queue = []
    if tail_w:
        queue += [f8_dma_w(i, par ^ 1) for i in range(nxp + nh, nxp + nwp)]
    if x_next:
        xq = [f8_dma_x(i) for i in range(nxp)]
        xq[0] = (q("s_add_u32 %[kofx], %[koff], 128") + "\n    " + xq[0][0], xq[0][1])
        queue += xq
    bump = tail_w or x_next
    for s_ in range(2):
        S = s_
        after = [[] for _ in range(n_mf)]
        pre = [[] for _ in range(n_mf)]
        if s_ == 0:
            rd = f8_reads(1, par, 1)                           # slice 1 of this step -> set 1
            for i, r in enumerate(rd):
                after[min(i, n_mf - 1) if FN_ == 4 else i * n_mf // len(rd)].append(r)
            if s_next and F8["lds_scales"]:
                after[0] += f8_scale_piece(par ^ 1)            # older than this step's operand pieces: the step's wait covers it
            elif s_next:
                sl = f8_scale_loads(par ^ 1)
                # all of them behind MFMA 0 and 1: they must be OLDER than this step's LDS-DMA pieces
                after[0] += sl[:len(sl) // 2]
                after[1] += sl[len(sl) // 2:]
            i = 2
            while queue and i < n_mf:
                a, b = queue.pop(0)
                pre[i].append(a)
                after[i].append(b)
                if not queue and bump:
                    after[i].append(q("s_add_u32 %[koff], %[koff], 128"))
                i += 1 if len(queue) + 1 > (n_mf - i + 1) // 2 else 2
        else:
            after[1].append(q(f"s_waitcnt vmcnt({vm})"))
            after[1].append(q("s_barrier"))
            if nxt:
                after[1] += [q("s_add_u32 %[xwl], %[lbase], %[xr]"),
                             q("s_add_u32 %[xr], %[xr], 0x8000"),
                             q("s_cmp_eq_u32 %[xr], 0x18000"),
                             q("s_cselect_b32 %[dlt], %[cneg], 0x8000"),
                             q("s_cselect_b32 %[xr], 0, %[xr]")]
                after[1] += [q(f"v_add_u32_e32 %[ax{k}], %[dlt], %[ax{k}]") for k in range(4)]
                rd = f8_reads(0, par ^ 1, 0)                   # slice 0 of the next step -> set 0
                if s_next and F8["lds_scales"]:
                    rd = rd + f8_scale_reads(par ^ 1)           # and its scales (landed: the wait + barrier above)
                for i, r in enumerate(rd):
                    after[min(2 + i * (n_mf - 2) // len(rd), n_mf - 1)].append(r)
            if head:
                i = 3
                for p_ in range(nh):
                    assert i <= n_mf - 1
                    a, b = f8_dma_w(nxp + p_, par)
                    pre[i].append(a)
                    after[i].append(b)
                    i += 2
        assert not (s_ == 0 and queue), "slice 0 could not place every LDS-DMA piece"
        L.append(q("s_waitcnt lgkmcnt(0)"))
        n = 0
        for fm in range(4):
            for fn in range(FN_):
                L += pre[n]
                srcc = "0" if (first and s_ == 0) else f"%[c{fm}{fn}]"
                wb, xb = f8_wreg(S, fn), f8_xreg(S, fm)
                L.append(q(f"v_mfma_scale_f32_32x32x64_f8f6f4 %[c{fm}{fn}], v[{wb}:{wb + 7}], v[{xb}:{xb + 7}], {srcc}, %[ws{par}{s_}{fn}], %[xs{par}{s_}{fm}] op_sel_hi:[0,0,0]"))
                L += after[n]
                n += 1
    return L


def emit_f8(fn=4, lds_scales=False):
    F8.update({"fn": fn, "bn": 64 * fn, "lds_scales": lds_scales})
    nxp = 8
    L = [q("; ---- fragments of (step 0, slice 0)")]
    L += f8_reads(0, 0, 0)
    # j = 0: W(1), X(1), X(2), scales(0) came with the prologue.  Its wait is vmcnt(0): the step issues no X pieces, so its
    # youngest VMEM operations are the scale loads of step 1, which the shifts behind the barrier consume
    L += f8_step(0, False, False, True, 0, True, first=True)
    L.append(q("s_cmp_eq_u32 %[nloop], 0"))
    L.append(q("s_cbranch_scc1 L_gemmf_tail_%="))
    L.append(q("L_gemmf_loop_%=:"))
    L += f8_step(1, True, True, True, nxp, True)
    L += f8_step(0, True, True, True, nxp, True)
    L.append(q("s_sub_u32 %[nloop], %[nloop], 1"))
    L.append(q("s_cmp_lg_u32 %[nloop], 0"))
    L.append(q("s_cbranch_scc1 L_gemmf_loop_%="))
    L.append(q("L_gemmf_tail_%=:"))
    L += f8_step(1, True, True, True, nxp, True)                     # j = nj - 3
    L += f8_step(0, True, False, False, 0, True)                     # j = nj - 2 (scales of the last step)
    L += f8_step(1, False, False, False, 0, False, nxt=False)        # j = nj - 1
    L.append(q("s_barrier"))
    L.append(q("s_nop 15"))
    outs = [f'[c{m}{f}] "=a"(acc[{m}][{f}])' for m in range(4) for f in range(fn)]
    for s_ in range(2):
        outs += [f'[xs0{s_}{f}] "+&v"(xs0[{s_}][{f}])' for f in range(4)] + [f'[ws0{s_}{f}] "+&v"(ws0[{s_}][{f}])' for f in range(fn)]
    for s_ in range(2):
        outs += [f'[xs1{s_}{f}] "=&v"(xs1[{s_}][{f}])' for f in range(4)] + [f'[ws1{s_}{f}] "=&v"(ws1[{s_}][{f}])' for f in range(fn)]
    outs += [f'[ax{k}] "+&v"(axc[{k}])' for k in range(4)]
    outs += ['[koff] "+&s"(koff)', '[nloop] "+&s"(nloop)', '[xr] "+&s"(xr)', '[xwl] "=&s"(xwl)', '[dlt] "=&s"(dlt)', '[kofx] "=&s"(kofx)']
    if not F8["lds_scales"]:
        outs += ['[ksx] "+&s"(ksx)', '[ksw] "+&s"(ksw)', '[ksx1] "=&s"(ksx1)', '[ksw1] "=&s"(ksw1)']
    ins = []
    for kk in range(4):
        ins += [f'[aw{kk}] "v"(aw[{kk}])', f'[aw{kk}h] "v"(awh[{kk}])']
    ins += [f'[vo{i}] "v"(voff[{i}])' for i in range(8 + 2 * fn)]
    ins += ['[rx] "s"(rx)', '[rw] "s"(rw)', '[lbase] "s"(lbase)', '[cneg] "s"(cneg)']
    if F8["lds_scales"]:
        outs += ['[ksc] "+&s"(ksc)']
        ins += ['[vsc] "v"(vsc)', '[rsc] "s"(rsc)', '[scstep] "s"(scstep)', '[lsc] "s"(lsc)', '[axs] "v"(axs)', '[aws] "v"(aws)']
    else:
        ins += ['[vsx] "v"(vsx)', '[vsw] "v"(vsw)', '[rxs] "s"(rxs)', '[rws] "s"(rws)', '[xstep] "s"(xstep)', '[wstep] "s"(wstep)']
    clob = ", ".join(f'"v{r}"' for r in range(128, 192 + 16 * fn))
    name = "gemm_asm_f8" + ("" if fn == 4 else f"_n{fn}") + ("s" if lds_scales else "") + ".inc"
    dst = os.path.join(OUTDIR, name)
    with open(dst, "w") as f:
        f.write("// GENERATED by tools/gen_gemm_asm.py (MXFP8 X3 loop) -- do not edit; the schedule is documented there.\n")
        write_asm(f, L, outs, ins, '"scc", "memory", ' + clob)
    print("wrote", os.path.normpath(dst), len(L), "lines")


K64_VARIANTS = {
    0: {},
    1: {"nhead": 7, "tail_stride": 1},
    2: {"nhead": 5, "head_stride": 3, "tail_stride": 2},      # 11 tail pieces: 8 in slice 0, 3 in slice 1
    3: {"cpol": " sc1"},
}

def emit_product():
    """the loops compiled into libsylber_hip.so (committed under sylber_amd/csrc/)"""
    emit(0)
    emit_k64(0, {})
    emit_k64(0, {}, fn=3)
    emit_k64(0, {}, fn=2, nw=8)
    emit_x3(4, 4)
    emit_x3(3, 4)
    emit_x3(2, 8)
    emit_x3(2, 8, tap=True)          # the 3-tap conv layers: chunk-major K order (tap 0, tap 2, tap 1 per 64-channel chunk)
    emit_x3(2, 4)                    # 256x128 tile on four waves (wave tile 128x64): the loop of the ping-pong kernel's groups
    emit_x3(2, 4, tap=True)
    emit_x3(3, 4, fm=3)              # 192x192 (tile 51): tile 91's loop on three row fragments per wave
    emit_x3(2, 8, fm=3)              # 192x256 on eight waves (tile 57): tile 97's
    for cols in (1, 2, 3):
        emit_x3(3, 4, pre_e=4, pre_cols=cols)          # K = 768 (12 steps): nothing left in the loop
        emit_x3(3, 4, pre_e=7, pre_cols=cols)          # K >= 1152 (FFN2, K = 3072: 48 steps, the last 17 unrolled)
    emit_y3(2, 8)                    # tile 47: tile 97's geometry on 16x16x32 MFMAs (forced only: its own fp32 grouping)
    emit_y3(2, 8, tap=True)
    emit_y3(2, 8, fm=3)              # tile 46: its 192-row sibling
    emit_f8(4)
    emit_f8(3)
    emit_f8(3, lds_scales=True)     # scales through LDS (one DMA piece per wave and step): the long-K launches


def emit_experiments():
    """knock-out / schedule variants, timing only (results wrong by construction).  Never committed: build.py generates
    them into sylber_amd/build/gen_exp/ for a SYLBER_EXPERIMENTS=1 build (library name libsylber_hip_exp.so)"""
    for v in sorted(VARIANTS):
        if v:
            emit(v)
    for v in sorted(K64_VARIANTS):
        if v:
            emit_k64(v, K64_VARIANTS[v])


if __name__ == "__main__":
    import sys
    what = sys.argv[1] if len(sys.argv) > 1 else "product"
    if what == "product":
        os.makedirs(OUTDIR, exist_ok=True)
        emit_product()
    elif what == "experiments":
        if not os.environ.get("GEN_GEMM_ASM_OUT"):
            OUTDIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "sylber_amd", "build", "gen_exp")
        os.makedirs(OUTDIR, exist_ok=True)
        emit_experiments()
    elif what == "hashes":
        # rewrite tools/gemm_asm_hashes.json from the generator as it stands (tests/test_gemm_asm_gen.py checks against it)
        import hashlib, json, tempfile
        with tempfile.TemporaryDirectory() as tmp:
            OUTDIR = tmp
            emit_product()
            h = {f: hashlib.sha256(open(os.path.join(tmp, f), "rb").read()).hexdigest() for f in sorted(os.listdir(tmp))}
        dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gemm_asm_hashes.json")
        json.dump(h, open(dst, "w"), indent=1, sort_keys=True)
        print("wrote", dst, len(h), "entries")
    else:
        raise SystemExit("usage: gen_gemm_asm.py [product|experiments|hashes]   (GEN_GEMM_ASM_OUT overrides the directory)")

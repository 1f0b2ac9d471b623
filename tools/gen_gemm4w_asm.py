#!/usr/bin/env python3
"""Generator of the lab kernels `gemm4w_asm` / `gemm8w_asm` (round 6): hand-scheduled bf16 GEMM MAIN LOOPS for
gfx950 with REGISTER STAGING (global_load_dwordx4 -> VGPR -> ds_write_b128) and every wait counted.

  NW = 4  four waves per CU, one per SIMD, 128 x 128 wave tiles, 256 accumulator registers in AGPRs — the design DESIGN.md named and
          never built.  hipcc cannot hold it (it selects the VGPR form of the MFMA, shuffles through v_accvgpr_read/write and spills
          500-700 registers: tools/gemm4w_lab.hip keeps that attempt as VAR 0 / 1 for the record).
  NW = 8  the production geometry (eight waves, two per SIMD, 128 x 64 wave tiles, 128 accumulators in VGPRs) with this file's
          schedule instead of the compiler's and register staging instead of LDS-DMA: separates what the 4-wave TILE buys from what
          the hand-placed pipeline buys (the production kernel's epilogues run at full VALU rate only with two waves per SIMD).

Registers, NW = 4 (NW = 8 in brackets):
  a[0:255] (v[0:127])   acc[a][b], a = 16-column block 0..7 (0..3), b = 16-row block 0..7 of the wave tile
  S[16] (S[8])          staging pieces (8 rows x 128 B) of the wave's share of the operand tiles
  wf[2][8] (wf[2][4])   weight fragments of the two k32 halves of a K-step;  xf[4] activation fragments, read two clusters ahead
  per-piece global byte offsets, LDS addresses (stage = bit 16, toggled by v_xor)

One K-step (64 deep) = 16 clusters (one activation fragment x the wave's weight fragments); per MFMA gap at most one piece's
(s_waitcnt vmcnt, ds_write, global_load) and the fragment reads; lgkmcnt by simulation of the in-order LDS queue.  One s_barrier per
K-step after cluster 13; clusters 14 / 15 compute from registers and read the next step's first fragments from the other stage.  The
first K-step of a tile accumulates onto the inline constant 0.  Tiles come from a host-built table (the production walk).

  python3 tools/gen_gemm4w_asm.py tools/bin/gemm4w.s 4 && python3 tools/gen_gemm4w_asm.py tools/bin/gemm8w.s 8
  clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c X.s -o X.o && ld.lld -shared X.o -o X.co     (both from /opt/rocm/lib/llvm/bin)
"""
import sys

L = []
NW = 4
NPRE = 14           # clusters in front of the barrier
WD = 0              # lab (NW = 8, timing only — results are wrong): 1 = no weight traffic at all (weight waves stage nothing, no weight
                    # fragment reads), 2 = weight fragments by direct contiguous global loads into the fragment registers


def e(s):
    L.append("  " + s)


def lab(s):
    L.append(s + ":")


def cmt(s):
    L.append("  ; " + s)


class Cfg:
    def __init__(self, nw):
        self.nw = nw
        self.na = 8 if nw == 4 else 4            # weight fragments (16-column blocks) per wave
        self.np = 16 if nw == 4 else 8           # staging pieces per wave and K-step
        if nw == 4:
            self.S0, self.WF, self.XF, self.GOFF, self.WA = 0, (64, 96), 128, 144, (160, 161)
            self.RW0, self.RW1, self.RX0, self.RX1 = 162, 163, 164, 165
            self.VT, self.VY = 166, 176
            self.VTID, self.VWAVE, self.VLANE, self.VL15, self.VLQ, self.VLROW, self.VLC, self.VSWZ = 180, 181, 182, 183, 184, 185, 186, 187
            self.DBG = 188
        else:
            self.S0, self.WF, self.XF, self.GOFF, self.WA = 128, (160, 176), 192, 208, (216, 217)
            self.RW0, self.RW1, self.RX0, self.RX1 = 218, 219, 220, 221
            self.VT, self.VY = 222, 232
            self.VTID, self.VWAVE, self.VLANE, self.VL15, self.VLQ, self.VLROW, self.VLC, self.VSWZ = 236, 237, 238, 239, 240, 241, 242, 243
            self.DBG = 244
        self.name = f"gemm{nw}w_asm"


C = None


def vr(base, n=4):
    return f"v[{base}:{base + n - 1}]"


def ar(a, b):
    i = 4 * (8 * a + b)
    return f"a[{i}:{i + 3}]" if NW == 4 else f"v[{i}:{i + 3}]"


def piece_cluster(p):
    return (p * (NPRE - 1)) // C.np


class LdsQueue:
    """In-order LDS queue of one wave: tags of the ds ops issued and not yet known complete."""

    def __init__(self, carried):
        self.q = list(carried)

    def issue(self, tag):
        self.q.append(tag)

    def need(self, tag):
        if tag not in self.q:
            return
        idx = self.q.index(tag)
        after = len(self.q) - 1 - idx
        e(f"s_waitcnt lgkmcnt({min(after, 15)})")
        self.q = self.q[idx + 1:] if after <= 15 else []

    def drain(self):
        self.q = []


def post_reads(tagw, tagx):
    """Issue order of the reads behind the barrier (and of the prologue): what a step finds in flight when it starts."""
    h = C.na // 2
    if WD:
        return [(tagx, 0), (tagx, 1)]
    return [(tagx, 0)] + [(tagw, a) for a in range(h)] + [(tagx, 1)] + [(tagw, a) for a in range(h, C.na)]


def kstep(first):
    """One K-step.  `first`: the tile's first step — its s2 = 0 clusters accumulate onto 0.
    Activation fragments are read TWO clusters ahead into slot (cluster % 4): the reads of clusters 14 / 15 are issued in 12 / 13
    (in front of the barrier), clusters 14 / 15 read the NEXT step's clusters 0 / 1 from the other stage."""
    q = LdsQueue(post_reads("w0", "x"))
    na, h = C.na, C.na // 2
    if WD == 2:
        e("s_waitcnt vmcnt(3)")                      # the weight fragments requested in the step before (3 staging requests are younger)
    for c in range(16):
        s2, b = c >> 3, c & 7
        slot = c & 3
        cmt(f"cluster {c}")
        mem = []            # groups of (kind, text, tag) in issue order, one group per MFMA gap
        cn = c + 2          # the activation fragment two clusters ahead first: it is the read with the least slack
        if cn < 16:
            mem.append([("lds", f"ds_read_b128 {vr(C.XF + 4 * (cn & 3))}, v{C.RX1 if cn >> 3 else C.RX0} offset:{(cn & 7) * 2048}", ("x", cn))])
        else:
            mem.append([("lds", f"ds_read_b128 {vr(C.XF + 4 * (cn & 3))}, v{C.RX0} offset:{(cn & 7) * 2048}", ("nx", cn & 7))])
        if c < na and not WD:
            mem.append([("lds", f"ds_read_b128 {vr(C.WF[1] + 4 * c)}, v{C.RW1} offset:{c * 2048}", ("w1", c))])
        if c >= 14 and not WD:
            for a in range(h * (c - 14), h * (c - 14) + h):
                mem.append([("lds", f"ds_read_b128 {vr(C.WF[0] + 4 * a)}, v{C.RW0} offset:{a * 2048}", ("nw", a))])
        for p in range(C.np):
            if piece_cluster(p) == c:
                if WD:                               # the weight waves stage nothing: their staging instructions run with EXEC = 0
                    mem.append([("alu", "s_mov_b64 exec, s[48:49]", None),
                                ("wait", f"s_waitcnt vmcnt({C.np - 1 + (8 if WD == 2 else 0)})", None),
                                ("lds", f"ds_write_b128 v{C.WA[p & 1]}, {vr(C.S0 + 4 * p)} offset:{p * 1024}", ("s", p)),
                                ("vm", f"global_load_dwordx4 {vr(C.S0 + 4 * p)}, v{C.GOFF + p}, s[20:21]", None),
                                ("alu", "s_mov_b64 exec, -1", None)])
                    continue
                mem.append([("wait", f"s_waitcnt vmcnt({C.np - 1})", None),
                            ("lds", f"ds_write_b128 v{C.WA[p & 1]}, {vr(C.S0 + 4 * p)} offset:{p * 1024}", ("s", p)),
                            ("vm", f"global_load_dwordx4 {vr(C.S0 + 4 * p)}, v{C.GOFF + p}, s[20:21]", None)])
        if WD == 2 and c < 8:
            # the NEXT step's weight fragment (k32 half c >> 2, block c & 3) straight from global memory: 1 KiB contiguous per wave and
            # instruction (a fragment-ordered packed copy of the weights would be read like this); timing only — it lands in a live register
            mem.append([("vm", f"global_load_dwordx4 {vr(C.WF[c >> 2] + 4 * (c & 3))}, v{248 + (c >> 2)}, s[50:51] offset:{(c & 3) * 1024}", None)])
            if c == 7:
                mem.append([("alu", "s_add_u32 s50, s50, 0x8000", None), ("alu", "s_addc_u32 s51, s51, 0", None)])
        # address / stream bookkeeping placed after the last use of each register (VALU / SALU fillers in MFMA gaps)
        if c == 8:
            mem.append([("alu", f"v_xor_b32 v{C.RX0}, 0x10000, v{C.RX0}", None), ("alu", f"v_xor_b32 v{C.RW1}, 0x10000, v{C.RW1}", None)])
        if c == 13:
            mem.append([("alu", f"v_xor_b32 v{C.WA[0]}, 0x10000, v{C.WA[0]}", None), ("alu", f"v_xor_b32 v{C.WA[1]}, 0x10000, v{C.WA[1]}", None)])
            mem.append([("alu", "s_add_u32 s20, s20, 128", None), ("alu", "s_addc_u32 s21, s21, 0", None)])
        gaps = [[] for _ in range(na)]
        for gi, g in enumerate(mem):
            gaps[min(gi, na - 1) if len(mem) <= na else (gi * na) // len(mem)].extend(g)
        for a in range(na):
            q.need(("w0" if s2 == 0 else "w1", a))
            if a == 0:
                q.need(("x", c))
            srcc = "0" if (first and s2 == 0) else ar(a, b)
            e(f"v_mfma_f32_16x16x32_bf16 {ar(a, b)}, {vr(C.WF[s2] + 4 * a)}, {vr(C.XF + 4 * slot)}, {srcc}")
            for kind, text, tag in gaps[a]:
                e(text)
                if kind == "lds":
                    q.issue(tag)
        if c == NPRE - 1:
            e("s_waitcnt lgkmcnt(0)")
            e("s_barrier")
            q.drain()
        if c == 14:
            e(f"v_xor_b32 v{C.RX1}, 0x10000, v{C.RX1}")
    e(f"v_xor_b32 v{C.RW0}, 0x10000, v{C.RW0}")
    # sanity: what is in flight at the end of a step is exactly what the next step assumes
    assert q.q == post_reads("nw", "nx"), q.q


def calc_base(mt, nt):
    """s[20:21] = global base (bytes) of this wave's operand share in tile (mt, nt) at K-step 0.
    s27 = operand (0 weights, 1 activations), s28 = share index (row block of 128 (NW 4) / 64 (NW 8) rows)."""
    e("s_cmp_eq_u32 s27, 0")
    e(f"s_cselect_b32 s36, s{nt}, s{mt}")          # op == 0: weights (column tile), else activations (row panel)
    e("s_lshl_b32 s36, s36, 8")
    e(f"s_lshl_b32 s37, s28, {7 if NW == 4 else 6}")
    e("s_add_u32 s36, s36, s37")
    e("s_mul_hi_u32 s37, s36, s17")
    e("s_mul_i32 s36, s36, s17")
    e("s_cmp_eq_u32 s27, 0")
    e("s_cselect_b32 s38, s6, s4")
    e("s_cselect_b32 s39, s7, s5")
    e("s_add_u32 s20, s38, s36")
    e("s_addc_u32 s21, s39, s37")


uid = [0]


def advance_load_stream(bump):
    """After a K-step's loads have been issued: next K-step of the load stream (bump: s[20:21] += 128 not yet done in the step)."""
    uid[0] += 1
    n = uid[0]
    if bump:
        e("s_add_u32 s20, s20, 128")
        e("s_addc_u32 s21, s21, 0")
    e("s_add_u32 s22, s22, 1")
    e("s_cmp_lt_u32 s22, s16")
    e(f"s_cbranch_scc1 .Lsame_tile_{n}")
    e("s_mov_b32 s22, 0")                           # the load stream enters its next tile (past the last one it re-reads the last: legal, unused)
    e("s_add_u32 s23, s23, 1")
    e("s_sub_u32 s36, s18, 1")
    e("s_min_u32 s23, s23, s36")
    e("s_lshl_b32 s36, s23, 3")
    e("s_add_u32 s36, s36, 8")
    e("s_load_dwordx2 s[34:35], s[10:11], s36")
    e("s_waitcnt lgkmcnt(0)")
    calc_base(34, 35)
    lab(f".Lsame_tile_{n}")


def main(out, nw, wd=0):
    global NW, C, WD
    NW = nw
    WD = wd
    C = Cfg(nw)
    na, h, npc = C.na, C.na // 2, C.np
    name = C.name
    L.append('.amdgcn_target "amdgcn-amd-amdhsa--gfx950"')
    L.append(".text")
    L.append(f".globl {name}")
    L.append(".p2align 8")
    L.append(f".type {name},@function")
    lab(name)
    cmt("kernarg: X 0, W 8, Y 16, table 24, K 32, N 36, stride 40 (int32 per table row), store 44")
    e("s_load_dwordx8 s[4:11], s[0:1], 0x0")
    e("s_load_dwordx4 s[12:15], s[0:1], 0x20")
    e("s_waitcnt lgkmcnt(0)")
    e("s_mul_i32 s36, s2, s14")
    e("s_lshl_b32 s36, s36, 2")
    e("s_add_u32 s10, s10, s36")
    e("s_addc_u32 s11, s11, 0")
    e("s_load_dword s18, s[10:11], 0x0")             # tiles of this workgroup
    e("s_load_dwordx2 s[32:33], s[10:11], 0x8")      # its first tile (mt, nt)
    e("s_lshr_b32 s16, s12, 6")                      # nk
    e("s_lshl_b32 s17, s12, 1")                      # bytes per operand row
    e(f"v_mov_b32 v{C.VTID}, v0")
    e(f"v_lshrrev_b32 v{C.VWAVE}, 6, v{C.VTID}")
    e("s_nop 4")                                     # gfx940+: a VALU write of a VGPR needs a wait state before v_readlane / v_readfirstlane reads it
    e(f"v_readfirstlane_b32 s26, v{C.VWAVE}")         # (found the hard way: the wave index came back as the register's previous content)
    e(f"v_and_b32 v{C.VLANE}, 63, v{C.VTID}")
    e(f"v_and_b32 v{C.VL15}, 15, v{C.VLANE}")
    e(f"v_lshrrev_b32 v{C.VLQ}, 4, v{C.VLANE}")
    e(f"v_lshrrev_b32 v{C.VLROW}, 3, v{C.VLANE}")
    e(f"v_and_b32 v{C.VLC}, 7, v{C.VLANE}")
    e(f"v_lshrrev_b32 v{C.VSWZ}, 1, v{C.VL15}")
    e(f"v_and_b32 v{C.VSWZ}, 7, v{C.VSWZ}")
    e("s_nop 4")
    if nw == 4:
        e("s_lshr_b32 s27, s26, 1")                  # staging operand = wm = wave >> 1
        e("s_and_b32 s28, s26, 1")                   # staging share = wn = wave & 1
    else:
        e("s_lshr_b32 s27, s26, 2")                  # staging operand = wave >> 2
        e("s_and_b32 s28, s26, 3")                   # staging share (64 rows) = wave & 3
    if WD:
        e("s_cmp_eq_u32 s27, 1")                     # staging EXEC: all lanes in the activation waves, none in the weight waves
        e("s_cselect_b64 s[48:49], -1, 0")
        e("s_mov_b32 s50, s6")
        e("s_mov_b32 s51, s7")
        e(f"v_lshlrev_b32 v248, 4, v{C.VLANE}")
        e("s_lshl_b32 s36, s26, 13")
        e("v_add_u32 v248, s36, v248")
        e("v_add_u32 v249, 0x1000, v248")
    e("s_mov_b32 s29, s28")                          # wn: the wave tile's column block
    e("s_mov_b32 s30, s27")                          # wm: its 128-row block
    e(f"v_mov_b32 v{C.DBG}, s26")                    # debug copies (dumped by the store-bit-31 path)
    e(f"v_mov_b32 v{C.DBG + 1}, s27")
    e(f"v_mov_b32 v{C.DBG + 2}, s28")
    e(f"v_mov_b32 v{C.DBG + 3}, v0")
    wcol = 14 if nw == 4 else 13                     # log2(bytes of the wave tile's column block in the weight image: 128 / 64 rows)
    cmt("fragment read addresses: R?s2 = (cols w? + l15) * 128 + (((4 s2 + lq) ^ swz) << 4) [+ 32768 for activations]")
    for s2, rw, rx in ((0, C.RW0, C.RX0), (1, C.RW1, C.RX1)):
        e(f"v_add_u32 v{C.VT}, {4 * s2}, v{C.VLQ}")
        e(f"v_xor_b32 v{C.VT}, v{C.VT}, v{C.VSWZ}")
        e(f"v_lshlrev_b32 v{C.VT}, 4, v{C.VT}")
        e(f"v_lshl_add_u32 v{C.VT}, v{C.VL15}, 7, v{C.VT}")      # + l15 * 128
        e(f"s_lshl_b32 s36, s29, {wcol}")
        e(f"v_add_u32 v{rw}, s36, v{C.VT}")
        e("s_lshl_b32 s36, s30, 14")                            # 128 rows of the activation image
        e("s_add_u32 s36, s36, 0x8000")
        e(f"v_add_u32 v{rx}, s36, v{C.VT}")
    cmt("staging write addresses: op * 32768 + (share rows + lrow) * 128 + ((lc ^ ((4 par + (lrow >> 1)) & 7)) << 4)")
    for par in (0, 1):
        e(f"v_lshrrev_b32 v{C.VT}, 1, v{C.VLROW}")
        e(f"v_add_u32 v{C.VT}, {4 * par}, v{C.VT}")
        e(f"v_and_b32 v{C.VT}, 7, v{C.VT}")
        e(f"v_xor_b32 v{C.VT}, v{C.VT}, v{C.VLC}")
        e(f"v_lshlrev_b32 v{C.VT}, 4, v{C.VT}")
        e(f"v_lshl_add_u32 v{C.VT}, v{C.VLROW}, 7, v{C.VT}")
        e("s_lshl_b32 s36, s27, 15")
        e(f"s_lshl_b32 s37, s28, {14 if nw == 4 else 13}")
        e("s_add_u32 s36, s36, s37")
        e(f"v_add_u32 v{C.WA[par]}, s36, v{C.VT}")
    cmt("global byte offsets of the pieces: (8 p + lrow) * rowbytes + lc * 16")
    e(f"v_mul_lo_u32 v{C.VT}, v{C.VLROW}, s17")
    e(f"v_lshl_add_u32 v{C.VT}, v{C.VLC}, 4, v{C.VT}")
    for p in range(npc):
        e(f"s_mul_i32 s36, s17, {8 * p}")
        e(f"v_add_u32 v{C.GOFF + p}, s36, v{C.VT}")
    cmt("store: lane offset ((128 wm + l15) * N + cols wn + 4 lq) * 4")
    e(f"v_mov_b32 v{C.VY}, s30")
    e(f"v_lshl_add_u32 v{C.VY}, v{C.VY}, 7, v{C.VL15}")
    e(f"v_mul_lo_u32 v{C.VY}, v{C.VY}, s13")
    e(f"s_lshl_b32 s36, s29, {7 if nw == 4 else 6}")
    e(f"v_add_u32 v{C.VY}, s36, v{C.VY}")
    e(f"v_lshl_add_u32 v{C.VY}, v{C.VLQ}, 2, v{C.VY}")
    e(f"v_lshlrev_b32 v{C.VY}, 2, v{C.VY}")
    e("s_waitcnt lgkmcnt(0)")
    e("s_cmp_eq_u32 s18, 0")
    e("s_cbranch_scc1 .Lexit")
    cmt("prologue: K-step 0 into stage 0, K-step 1 requested, first fragments in registers")
    e("s_mov_b32 s22, 0")
    e("s_mov_b32 s23, 0")
    e("s_mov_b32 s24, 0")
    e("s_mov_b32 s25, 0")
    calc_base(32, 33)
    for p in range(npc):
        e(f"global_load_dwordx4 {vr(C.S0 + 4 * p)}, v{C.GOFF + p}, s[20:21]")
    advance_load_stream(True)
    e("s_waitcnt vmcnt(0)")
    for p in range(npc):
        e(f"ds_write_b128 v{C.WA[p & 1]}, {vr(C.S0 + 4 * p)} offset:{p * 1024}")
    for p in range(npc):
        e(f"global_load_dwordx4 {vr(C.S0 + 4 * p)}, v{C.GOFF + p}, s[20:21]")
    advance_load_stream(True)
    e("s_waitcnt lgkmcnt(0)")
    e("s_barrier")
    e(f"ds_read_b128 {vr(C.XF)}, v{C.RX0}")
    for a in range(h):
        e(f"ds_read_b128 {vr(C.WF[0] + 4 * a)}, v{C.RW0} offset:{a * 2048}")
    e(f"ds_read_b128 {vr(C.XF + 4)}, v{C.RX0} offset:2048")
    for a in range(h, na):
        e(f"ds_read_b128 {vr(C.WF[0] + 4 * a)}, v{C.RW0} offset:{a * 2048}")
    for r in (C.WA[0], C.WA[1], C.RW0):
        e(f"v_xor_b32 v{r}, 0x10000, v{r}")
    lab(".Ltile")
    kstep(True)
    advance_load_stream(False)
    e("s_mov_b32 s24, 1")
    lab(".Lk")
    kstep(False)
    advance_load_stream(False)
    e("s_add_u32 s24, s24, 1")
    e("s_cmp_lt_u32 s24, s16")
    e("s_cbranch_scc1 .Lk")
    cmt("tile done: the lab's epilogue (check mode: f32 rows to Y; timing mode: nothing)")
    e("s_cmp_eq_u32 s15, 0")
    e("s_cbranch_scc1 .Lnostore")
    e("s_nop 15")
    e("s_nop 15")
    e("s_mul_i32 s36, s32, s13")                     # mt * N
    e("s_lshl_b32 s36, s36, 8")                      # * 256 (elements; < 2^31 for the lab's sizes: 65536 x 3072)
    e("s_lshl_b32 s37, s33, 8")
    e("s_add_u32 s36, s36, s37")
    e("s_mov_b32 s37, 0")
    e("s_lshl_b64 s[36:37], s[36:37], 2")
    e("s_add_u32 s40, s8, s36")
    e("s_addc_u32 s41, s9, s37")
    e("s_lshl_b32 s38, s13, 6")                      # 16 rows * N * 4 bytes
    T = C.VT
    cmt("debug dump (store bit 31): per thread {VY, s40, s41, N | s26 s27 s28 v0 at start | s26 s27 s28 mt now} to Y + (min(wg, 255) * 512 + tid) * 64")
    e("s_bitcmp1_b32 s15, 31")
    e("s_cbranch_scc0 .Lnodump")
    e("s_min_u32 s44, s2, 255")
    e("s_lshl_b32 s44, s44, 15")
    e(f"v_and_b32 v{T + 4}, 511, v{C.VTID}")
    e(f"v_lshlrev_b32 v{T + 4}, 6, v{T + 4}")
    e(f"v_add_u32 v{T + 4}, s44, v{T + 4}")
    e(f"v_mov_b32 v{T + 1}, s9")
    e(f"v_add_co_u32 v{T + 2}, vcc, s8, v{T + 4}")
    e(f"v_addc_co_u32 v{T + 3}, vcc, 0, v{T + 1}, vcc")
    e(f"v_mov_b32 v{T + 6}, v{C.VY}")
    e(f"v_mov_b32 v{T + 7}, s40")
    e(f"v_mov_b32 v{T + 8}, s41")
    e(f"v_mov_b32 v{T + 9}, s13")
    e(f"global_store_dwordx4 v[{T + 2}:{T + 3}], v[{T + 6}:{T + 9}], off")
    e("s_nop 4")
    e(f"global_store_dwordx4 v[{T + 2}:{T + 3}], v[{C.DBG}:{C.DBG + 3}], off offset:16")
    e(f"v_mov_b32 v{C.DBG + 4}, s26")
    e(f"v_mov_b32 v{C.DBG + 5}, s27")
    e(f"v_mov_b32 v{C.DBG + 6}, s28")
    e(f"v_mov_b32 v{C.DBG + 7}, s32")
    e("s_nop 4")
    e(f"global_store_dwordx4 v[{T + 2}:{T + 3}], v[{C.DBG + 4}:{C.DBG + 7}], off offset:32")
    e("s_nop 4")
    e("s_waitcnt vmcnt(0)")
    e("s_branch .Lexit")
    lab(".Lnodump")
    cmt("guard: kernarg `store` = rows M of Y; every lane's 64-bit address is held inside [Y, Y + M N 4) before its stores")
    span = 64 * (na - 1) + 16
    e("s_mul_i32 s42, s15, s13")                     # M * N elements (< 2^32 for the lab's sizes)
    e("s_mov_b32 s43, 0")
    e("s_lshl_b64 s[42:43], s[42:43], 2")
    e("s_add_u32 s42, s8, s42")
    e("s_addc_u32 s43, s9, s43")
    e(f"s_sub_u32 s42, s42, {span - 1}")
    e("s_subb_u32 s43, s43, 0")
    for b_ in range(8):
        e(f"v_mov_b32 v{T + 1}, s41")
        e(f"v_add_co_u32 v{T + 2}, vcc, s40, v{C.VY}")
        e(f"v_addc_co_u32 v{T + 3}, vcc, 0, v{T + 1}, vcc")
        e(f"v_cmp_lt_u64 vcc, v[{T + 2}:{T + 3}], s[42:43]")
        e("s_and_saveexec_b64 s[46:47], vcc")
        e(f"v_cmp_ge_u64 vcc, v[{T + 2}:{T + 3}], s[8:9]")
        e("s_and_b64 exec, exec, vcc")
        for a_ in range(na):
            e(f"global_store_dwordx4 v[{T + 2}:{T + 3}], {ar(a_, b_)}, off offset:{64 * a_}")
        e("s_mov_b64 exec, s[46:47]")
        e("s_add_u32 s40, s40, s38")
        e("s_addc_u32 s41, s41, 0")
    e("s_waitcnt vmcnt(0)")
    lab(".Lnostore")
    e("s_add_u32 s25, s25, 1")
    e("s_cmp_ge_u32 s25, s18")
    e("s_cbranch_scc1 .Lexit")
    e("s_lshl_b32 s36, s25, 3")
    e("s_add_u32 s36, s36, 8")
    e("s_load_dwordx2 s[32:33], s[10:11], s36")      # (mt, nt) of the tile now starting (store only)
    e("s_waitcnt lgkmcnt(0)")
    e("s_branch .Ltile")
    lab(".Lexit")
    e("s_waitcnt vmcnt(0) lgkmcnt(0)")
    e("s_endpgm")
    L.append(".Lfunc_end:")
    L.append(f".size {name}, .Lfunc_end-{name}")
    nvgpr = 512 if nw == 4 else 256
    L.append(f"""
.rodata
.p2align 6
.amdhsa_kernel {name}
  .amdhsa_group_segment_fixed_size 131072
  .amdhsa_private_segment_fixed_size 0
  .amdhsa_kernarg_size 48
  .amdhsa_user_sgpr_count 2
  .amdhsa_user_sgpr_kernarg_segment_ptr 1
  .amdhsa_system_sgpr_workgroup_id_x 1
  .amdhsa_system_vgpr_workitem_id 0
  .amdhsa_next_free_vgpr {nvgpr}
  .amdhsa_next_free_sgpr 96
  .amdhsa_accum_offset 256
  .amdhsa_reserve_vcc 1
  .amdhsa_float_denorm_mode_32 3
  .amdhsa_float_denorm_mode_16_64 3
  .amdhsa_dx10_clamp 1
  .amdhsa_ieee_mode 1
.end_amdhsa_kernel

.amdgpu_metadata
---
amdhsa.kernels:
  - .agpr_count:     {256 if nw == 4 else 0}
    .args:
      - {{.address_space: global, .offset: 0, .size: 8, .value_kind: global_buffer}}
      - {{.address_space: global, .offset: 8, .size: 8, .value_kind: global_buffer}}
      - {{.address_space: global, .offset: 16, .size: 8, .value_kind: global_buffer}}
      - {{.address_space: global, .offset: 24, .size: 8, .value_kind: global_buffer}}
      - {{.offset: 32, .size: 4, .value_kind: by_value}}
      - {{.offset: 36, .size: 4, .value_kind: by_value}}
      - {{.offset: 40, .size: 4, .value_kind: by_value}}
      - {{.offset: 44, .size: 4, .value_kind: by_value}}
    .group_segment_fixed_size: 131072
    .kernarg_segment_align: 8
    .kernarg_segment_size: 48
    .max_flat_workgroup_size: {64 * nw}
    .name:           {name}
    .private_segment_fixed_size: 0
    .sgpr_count:     96
    .sgpr_spill_count: 0
    .symbol:         {name}.kd
    .uniform_work_group_size: 1
    .uses_dynamic_stack: false
    .vgpr_count:     {nvgpr}
    .vgpr_spill_count: 0
    .wavefront_size: 64
amdhsa.target:   amdgcn-amd-amdhsa--gfx950
amdhsa.version:
  - 1
  - 2
...
.end_amdgpu_metadata
""")
    with open(out, "w") as f:
        f.write("\n".join(L) + "\n")
    n_mfma = sum("v_mfma" in x for x in L)
    print(f"{out}: {len(L)} lines, {n_mfma} MFMAs, NW = {nw}")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "tools/bin/gemm4w.s", int(sys.argv[2]) if len(sys.argv) > 2 else 4, int(sys.argv[3]) if len(sys.argv) > 3 else 0)

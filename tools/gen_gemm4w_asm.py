#!/usr/bin/env python3
"""Generator of the lab kernel `gemm4w_asm` (round 6): the 4-wave / 128x128-wave-tile bf16 GEMM MAIN LOOP for
gfx950 as hand-scheduled assembly.  hipcc cannot hold this design (256 accumulator registers + 144 operand / staging registers:
it selects the VGPR form of the MFMA, shuffles through v_accvgpr_read/write and spills 500-700 registers — tools/gemm4w_lab.hip keeps
that attempt as VAR 0 / 1 for the record), so the instruction stream is written out here with explicit registers:

  a[0:255]     acc[a][b]  (a = 16-column block 0..7, b = 16-row block 0..7 of the wave's 128 x 128 tile), AGPR form of the MFMA
  v[0:63]      S[16]      staging: 16 pieces (8 rows x 128 B) of the wave's operand half, global_load_dwordx4 -> ds_write_b128
  v[64:127]    wf[2][8]   weight fragments of the two k32 halves of a K-step
  v[128:143]   xf[4]      activation fragments (slots 0 / 1 alternate, 2 / 3 for the two clusters behind the barrier)
  v[144:159]   per-piece global byte offsets;  v[160:165] LDS addresses (stage bit 16 toggled by v_xor)

One K-step (64 deep) = 16 clusters of 8 MFMAs (one activation fragment x 8 weight fragments); per cluster at most one ds_write +
global_load pair per MFMA gap and <= 3 ds_read_b128, every wait counted (lgkmcnt by simulation of the in-order LDS queue, vmcnt(15):
the 16 loads of a step land one step later).  One s_barrier per K-step after cluster 13; clusters 14 / 15 compute from registers and
read the next step's first fragments from the other stage.  The first K-step of a tile accumulates onto the inline constant 0, so
no accumulator is ever zeroed.  Tiles come from a host-built table (the production walk: persistent grid, XCD remap, row-panel-major).

  python3 tools/gen_gemm4w_asm.py tools/bin/gemm4w.s
  /opt/rocm/lib/llvm/bin/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c tools/bin/gemm4w.s -o tools/bin/gemm4w.o
  /opt/rocm/lib/llvm/bin/ld.lld -shared tools/bin/gemm4w.o -o tools/bin/gemm4w.co
"""
import sys

L = []


def e(s):
    L.append("  " + s)


def lab(s):
    L.append(s + ":")


def cmt(s):
    L.append("  ; " + s)


S0, WF, XF, GOFF, WA = 0, (64, 96), 128, 144, (160, 161)
RW0, RW1, RX0, RX1 = 162, 163, 164, 165
VT = 166            # temporaries 166..179
VY = 176            # store: lane offset into Y
VTID, VWAVE, VLANE, VL15, VLQ, VLROW, VLC, VSWZ = 180, 181, 182, 183, 184, 185, 186, 187
NPRE = 14           # clusters in front of the barrier


def vr(base, n=4):
    return f"v[{base}:{base + n - 1}]"


def ar(a, b):
    i = 4 * (8 * a + b)
    return f"a[{i}:{i + 3}]"


def piece_cluster(p):
    return (p * (NPRE - 1)) // 16


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


# issue order of the reads behind the barrier (and of the prologue): what a step finds in flight when it starts
POST_READS = [("x", 0)] + [("w", 0, a) for a in range(4)] + [("x", 1)] + [("w", 0, a) for a in range(4, 8)]


def kstep(first):
    """One K-step.  `first`: the tile's first step — its s2 = 0 clusters accumulate onto 0.
    Activation fragments are read TWO clusters ahead into slot (cluster % 4): the reads of clusters 14 / 15 are issued in 12 / 13
    (in front of the barrier), clusters 14 / 15 read the NEXT step's clusters 0 / 1 from the other stage."""
    q = LdsQueue(POST_READS)
    for c in range(16):
        s2, b = c >> 3, c & 7
        slot = c & 3
        cmt(f"cluster {c}")
        mem = []            # (kind, text, tag) in issue order
        cn = c + 2          # the activation fragment two clusters ahead first: it is the read with the least slack
        if cn < 16:
            mem.append(("lds", f"ds_read_b128 {vr(XF + 4 * (cn & 3))}, v{RX1 if cn >> 3 else RX0} offset:{(cn & 7) * 2048}", ("x", cn)))
        else:
            mem.append(("lds", f"ds_read_b128 {vr(XF + 4 * (cn & 3))}, v{RX0} offset:{(cn & 7) * 2048}", ("nx", cn & 7)))
        if c < 8:
            mem.append(("lds", f"ds_read_b128 {vr(WF[1] + 4 * c)}, v{RW1} offset:{c * 2048}", ("w", 1, c)))
        if c >= 14:
            for a in range(4 * (c - 14), 4 * (c - 14) + 4):
                mem.append(("lds", f"ds_read_b128 {vr(WF[0] + 4 * a)}, v{RW0} offset:{a * 2048}", ("nw", a)))
        for p in range(16):
            if piece_cluster(p) == c:
                mem.append(("wait", "s_waitcnt vmcnt(15)", None))
                mem.append(("lds", f"ds_write_b128 v{WA[p & 1]}, {vr(S0 + 4 * p)} offset:{p * 1024}", ("s", p)))
                mem.append(("vm", f"global_load_dwordx4 {vr(S0 + 4 * p)}, v{GOFF + p}, s[20:21]", None))
        # address / stream bookkeeping placed after the last use of each register (VALU / SALU fillers in MFMA gaps)
        if c == 8:
            mem.append(("alu", f"v_xor_b32 v{RX0}, 0x10000, v{RX0}", None))
            mem.append(("alu", f"v_xor_b32 v{RW1}, 0x10000, v{RW1}", None))
        if c == 13:
            mem.append(("alu", f"v_xor_b32 v{WA[0]}, 0x10000, v{WA[0]}", None))
            mem.append(("alu", f"v_xor_b32 v{WA[1]}, 0x10000, v{WA[1]}", None))
            mem.append(("alu", "s_add_u32 s20, s20, 128", None))
            mem.append(("alu", "s_addc_u32 s21, s21, 0", None))
        # one group per MFMA gap: a piece's (wait, write, load) stays together
        groups = []
        i = 0
        while i < len(mem):
            take = 3 if mem[i][0] == "wait" else 1
            groups.append(mem[i:i + take])
            i += take
        gaps = [[] for _ in range(8)]
        for gi, g in enumerate(groups):
            gaps[min(gi, 7) if len(groups) <= 8 else (gi * 8) // len(groups)].extend(g)
        for a in range(8):
            q.need(("w", s2, a))
            if a == 0:
                q.need(("x", c))
            srcc = "0" if (first and s2 == 0) else ar(a, b)
            e(f"v_mfma_f32_16x16x32_bf16 {ar(a, b)}, {vr(WF[s2] + 4 * a)}, {vr(XF + 4 * slot)}, {srcc}")
            for kind, text, tag in gaps[a]:
                e(text)
                if kind == "lds":
                    q.issue(tag)
        if c == NPRE - 1:
            e("s_waitcnt lgkmcnt(0)")
            e("s_barrier")
            q.drain()
        if c == 14:
            e(f"v_xor_b32 v{RX1}, 0x10000, v{RX1}")
    e(f"v_xor_b32 v{RW0}, 0x10000, v{RW0}")
    # sanity: what is in flight at the end of a step is exactly what the next step assumes
    want = [("nx", 0)] + [("nw", a) for a in range(4)] + [("nx", 1)] + [("nw", a) for a in range(4, 8)]
    assert q.q == want, q.q


def calc_base(mt, nt):
    """s[20:21] = global base (bytes) of this wave's operand half in tile (mt, nt) at K-step 0."""
    e("s_cmp_eq_u32 s27, 0")
    e(f"s_cselect_b32 s36, s{nt}, s{mt}")          # op == 0: weights (column tile), else activations (row panel)
    e("s_lshl_b32 s36, s36, 8")
    e("s_lshl_b32 s37, s28, 7")
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


def main(out):
    L.append('.amdgcn_target "amdgcn-amd-amdhsa--gfx950"')
    L.append(".text")
    L.append(".globl gemm4w_asm")
    L.append(".p2align 8")
    L.append(".type gemm4w_asm,@function")
    lab("gemm4w_asm")
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
    e(f"v_mov_b32 v{VTID}, v0")
    e(f"v_lshrrev_b32 v{VWAVE}, 6, v{VTID}")
    e("s_nop 4")                                     # gfx940+: a VALU write of a VGPR needs a wait state before v_readlane / v_readfirstlane reads it
    e(f"v_readfirstlane_b32 s26, v{VWAVE}")           # (found the hard way: the wave index came back as the register's previous content)
    e(f"v_and_b32 v{VLANE}, 63, v{VTID}")
    e(f"v_and_b32 v{VL15}, 15, v{VLANE}")
    e(f"v_lshrrev_b32 v{VLQ}, 4, v{VLANE}")
    e(f"v_lshrrev_b32 v{VLROW}, 3, v{VLANE}")
    e(f"v_and_b32 v{VLC}, 7, v{VLANE}")
    e(f"v_lshrrev_b32 v{VSWZ}, 1, v{VL15}")
    e(f"v_and_b32 v{VSWZ}, 7, v{VSWZ}")
    e("s_nop 4")
    e("s_lshr_b32 s27, s26, 1")                      # op = wm = wave >> 1
    e("s_and_b32 s28, s26, 1")                       # hf = wn = wave & 1
    e("v_mov_b32 v188, s26")                         # debug copies (dumped by the store-bit-31 path)
    e("v_mov_b32 v189, s27")
    e("v_mov_b32 v190, s28")
    e("v_mov_b32 v191, v0")
    cmt("fragment read addresses: R?s2 = (128 w? + l15) * 128 + (((4 s2 + lq) ^ swz) << 4) [+ 32768 for activations]")
    for s2, rw, rx in ((0, RW0, RX0), (1, RW1, RX1)):
        e(f"v_add_u32 v{VT}, {4 * s2}, v{VLQ}")
        e(f"v_xor_b32 v{VT}, v{VT}, v{VSWZ}")
        e(f"v_lshlrev_b32 v{VT}, 4, v{VT}")
        e(f"v_lshl_add_u32 v{VT}, v{VL15}, 7, v{VT}")          # + l15 * 128
        e("s_lshl_b32 s36, s28, 14")                            # 128 wn * 128
        e(f"v_add_u32 v{rw}, s36, v{VT}")
        e("s_lshl_b32 s36, s27, 14")
        e("s_add_u32 s36, s36, 0x8000")
        e(f"v_add_u32 v{rx}, s36, v{VT}")
    cmt("staging write addresses: op * 32768 + (128 hf + lrow) * 128 + ((lc ^ ((4 par + (lrow >> 1)) & 7)) << 4)")
    for par in (0, 1):
        e(f"v_lshrrev_b32 v{VT}, 1, v{VLROW}")
        e(f"v_add_u32 v{VT}, {4 * par}, v{VT}")
        e(f"v_and_b32 v{VT}, 7, v{VT}")
        e(f"v_xor_b32 v{VT}, v{VT}, v{VLC}")
        e(f"v_lshlrev_b32 v{VT}, 4, v{VT}")
        e(f"v_lshl_add_u32 v{VT}, v{VLROW}, 7, v{VT}")
        e("s_lshl_b32 s36, s27, 15")
        e("s_lshl_b32 s37, s28, 14")
        e("s_add_u32 s36, s36, s37")
        e(f"v_add_u32 v{WA[par]}, s36, v{VT}")
    cmt("global byte offsets of the 16 pieces: (8 p + lrow) * rowbytes + lc * 16")
    e(f"v_mul_lo_u32 v{VT}, v{VLROW}, s17")
    e(f"v_lshl_add_u32 v{VT}, v{VLC}, 4, v{VT}")
    for p in range(16):
        e(f"s_mul_i32 s36, s17, {8 * p}")
        e(f"v_add_u32 v{GOFF + p}, s36, v{VT}")
    cmt("store: lane offset ((128 wm + l15) * N + 128 wn + 4 lq) * 4")
    e(f"v_mov_b32 v{VY}, s27")
    e(f"v_lshl_add_u32 v{VY}, v{VY}, 7, v{VL15}")
    e(f"v_mul_lo_u32 v{VY}, v{VY}, s13")
    e("s_lshl_b32 s36, s28, 7")
    e(f"v_add_u32 v{VY}, s36, v{VY}")
    e(f"v_lshl_add_u32 v{VY}, v{VLQ}, 2, v{VY}")
    e(f"v_lshlrev_b32 v{VY}, 2, v{VY}")
    e("s_waitcnt lgkmcnt(0)")
    e("s_cmp_eq_u32 s18, 0")
    e("s_cbranch_scc1 .Lexit")
    cmt("prologue: K-step 0 into stage 0, K-step 1 requested, first fragments in registers")
    e("s_mov_b32 s22, 0")
    e("s_mov_b32 s23, 0")
    e("s_mov_b32 s24, 0")
    e("s_mov_b32 s25, 0")
    calc_base(32, 33)
    for p in range(16):
        e(f"global_load_dwordx4 {vr(S0 + 4 * p)}, v{GOFF + p}, s[20:21]")
    advance_load_stream(True)
    e("s_waitcnt vmcnt(0)")
    for p in range(16):
        e(f"ds_write_b128 v{WA[p & 1]}, {vr(S0 + 4 * p)} offset:{p * 1024}")
    for p in range(16):
        e(f"global_load_dwordx4 {vr(S0 + 4 * p)}, v{GOFF + p}, s[20:21]")
    advance_load_stream(True)
    e("s_waitcnt lgkmcnt(0)")
    e("s_barrier")
    e(f"ds_read_b128 {vr(XF)}, v{RX0}")
    for a in range(4):
        e(f"ds_read_b128 {vr(WF[0] + 4 * a)}, v{RW0} offset:{a * 2048}")
    e(f"ds_read_b128 {vr(XF + 4)}, v{RX0} offset:2048")
    for a in range(4, 8):
        e(f"ds_read_b128 {vr(WF[0] + 4 * a)}, v{RW0} offset:{a * 2048}")
    for r in (WA[0], WA[1], RW0):
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
    cmt("debug dump (store bit 31): per thread {v176, s40, s41, N, s32, s33, s18, s27|s28<<8} to Y + (min(wg, 255) * 256 + (tid & 255)) * 32")
    e("s_bitcmp1_b32 s15, 31")
    e("s_cbranch_scc0 .Lnodump")
    e("s_min_u32 s44, s2, 255")
    e("s_lshl_b32 s44, s44, 14")
    e(f"v_and_b32 v{VT + 4}, 255, v{VTID}")
    e(f"v_lshlrev_b32 v{VT + 4}, 6, v{VT + 4}")
    e(f"v_add_u32 v{VT + 4}, s44, v{VT + 4}")
    e(f"v_mov_b32 v{VT + 1}, s9")
    e(f"v_add_co_u32 v{VT + 2}, vcc, s8, v{VT + 4}")
    e(f"v_addc_co_u32 v{VT + 3}, vcc, 0, v{VT + 1}, vcc")
    e(f"v_mov_b32 v{VT + 6}, v{VY}")
    e(f"v_mov_b32 v{VT + 7}, s40")
    e(f"v_mov_b32 v{VT + 8}, s41")
    e(f"v_mov_b32 v{VT + 9}, s13")
    e(f"global_store_dwordx4 v[{VT + 2}:{VT + 3}], v[{VT + 6}:{VT + 9}], off")
    e("s_nop 4")
    e(f"global_store_dwordx4 v[{VT + 2}:{VT + 3}], v[188:191], off offset:16")
    e("v_mov_b32 v192, s26")
    e("v_mov_b32 v193, s27")
    e("v_mov_b32 v194, s28")
    e("v_mov_b32 v195, s32")
    e("s_nop 4")
    e(f"global_store_dwordx4 v[{VT + 2}:{VT + 3}], v[192:195], off offset:32")
    e("s_nop 4")
    e("s_waitcnt vmcnt(0)")
    e("s_branch .Lexit")
    lab(".Lnodump")
    cmt("guard: kernarg `store` = rows M of Y; every lane's 64-bit address is held to [Y, Y + M N 4 - 464] before its 8 stores")
    e("s_mul_i32 s42, s15, s13")                     # M * N elements (< 2^32 for the lab's sizes)
    e("s_mov_b32 s43, 0")
    e("s_lshl_b64 s[42:43], s[42:43], 2")
    e("s_add_u32 s42, s8, s42")
    e("s_addc_u32 s43, s9, s43")
    e("s_sub_u32 s42, s42, 463")
    e("s_subb_u32 s43, s43, 0")
    for b_ in range(8):
        e(f"v_mov_b32 v{VT + 1}, s41")
        e(f"v_add_co_u32 v{VT + 2}, vcc, s40, v{VY}")
        e(f"v_addc_co_u32 v{VT + 3}, vcc, 0, v{VT + 1}, vcc")
        e(f"v_cmp_lt_u64 vcc, v[{VT + 2}:{VT + 3}], s[42:43]")
        e("s_and_saveexec_b64 s[46:47], vcc")
        e(f"v_cmp_ge_u64 vcc, v[{VT + 2}:{VT + 3}], s[8:9]")
        e("s_and_b64 exec, exec, vcc")
        for a_ in range(8):
            e(f"global_store_dwordx4 v[{VT + 2}:{VT + 3}], {ar(a_, b_)}, off offset:{64 * a_}")
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
    L.append(".size gemm4w_asm, .Lfunc_end-gemm4w_asm")
    L.append("""
.rodata
.p2align 6
.amdhsa_kernel gemm4w_asm
  .amdhsa_group_segment_fixed_size 131072
  .amdhsa_private_segment_fixed_size 0
  .amdhsa_kernarg_size 48
  .amdhsa_user_sgpr_count 2
  .amdhsa_user_sgpr_kernarg_segment_ptr 1
  .amdhsa_system_sgpr_workgroup_id_x 1
  .amdhsa_system_vgpr_workitem_id 0
  .amdhsa_next_free_vgpr 512
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
  - .agpr_count:     256
    .args:
      - {.address_space: global, .offset: 0, .size: 8, .value_kind: global_buffer}
      - {.address_space: global, .offset: 8, .size: 8, .value_kind: global_buffer}
      - {.address_space: global, .offset: 16, .size: 8, .value_kind: global_buffer}
      - {.address_space: global, .offset: 24, .size: 8, .value_kind: global_buffer}
      - {.offset: 32, .size: 4, .value_kind: by_value}
      - {.offset: 36, .size: 4, .value_kind: by_value}
      - {.offset: 40, .size: 4, .value_kind: by_value}
      - {.offset: 44, .size: 4, .value_kind: by_value}
    .group_segment_fixed_size: 131072
    .kernarg_segment_align: 8
    .kernarg_segment_size: 48
    .max_flat_workgroup_size: 256
    .name:           gemm4w_asm
    .private_segment_fixed_size: 0
    .sgpr_count:     96
    .sgpr_spill_count: 0
    .symbol:         gemm4w_asm.kd
    .uniform_work_group_size: 1
    .uses_dynamic_stack: false
    .vgpr_count:     512
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
    print(f"{out}: {len(L)} lines, {n_mfma} MFMAs")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "tools/bin/gemm4w.s")

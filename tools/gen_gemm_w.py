#!/usr/bin/env python3
"""Generator of manner_amd/csrc/gemm_w8_asm.inc and gemm_w4_asm.inc — the hand-scheduled main loops of the 16-bit GEMM (round 6).

Two geometries of ONE schedule (register staging: global_load_dwordx4 -> VGPR -> ds_write_b128, every wait counted, one barrier per
K-step, operands requested two K-steps ahead across tiles):
  NW = 8 (production since round 6): the 8-wave geometry of gemm_tn_x16_kernel — 128 x 64 wave tiles, two waves per SIMD, the 128
         accumulator registers of a wave in v[0:127] — so that the epilogues keep running at the full VALU issue rate of two waves
         per SIMD.  Lab A/B of the main loops on one box (tools/gemm4w_lab.hip, profiles/r6_final/lab_w4_w8_time.txt): Q|K|V 195 ->
         152 us, out-projection 67 -> 52.5, FFN1 243 -> 201, FFN2 240 -> 216 against the compiler-scheduled LDS-DMA main loop.
  NW = 4 the design DESIGN.md had named since round 2 — described next; its main loop is as fast as NW = 8's, but
         behind the production epilogues it LOSES what it gained: one wave per SIMD issues vector instructions at half the rate of two.

The production 8-wave kernel (gemm_tn_x16_kernel: wave tile 128 x 64, LDS-DMA staging) reads 12 operand fragments per 32 MFMAs; this
main loop runs FOUR waves per CU, one per SIMD, each owning 128 x 128 of the 256 x 256 tile with its 256 accumulator registers in AGPRs:
16 fragment reads per 64 MFMAs (-33 % LDS fragment traffic), a 4-wave barrier per K-step, register staging (global_load_dwordx4 ->
VGPR -> ds_write_b128) instead of LDS-DMA, every wait counted.  Same-box lab A/B against the production main loop (tools/gemm4w_lab.hip,
profiles/r6_final/lab4w_time.txt): 16-20 % less time on the K = 768 shapes, 5-9 % on FFN2.  hipcc cannot hold the design (it selects
the VGPR form of the MFMA, shuffles through v_accvgpr_read/write and spills 500-700 registers), so the K-loop of ONE TILE is emitted
here as one inline-asm block with explicit registers; tile walk, prologue and the epilogues stay C++ (gemm.hip: gemm_tn_w4_kernel).

Registers of the block (clobbered): a[0:255] acc[a][b] (a = 16-column block 0..7, b = 16-row block 0..7 of the wave tile), v[0:63]
S[16] staging pieces, v[64:127] wf[2][8] weight fragments of the two k32 halves, v[128:143] xf[4] activation fragments.
Operands: g (+v) the lane's global byte offset of piece 0 (advances 128 B per K-step, net change over a tile: 0; piece p adds the
wave-uniform c<p> = p * 8 * rowbytes through one address temporary), wa0 / wa1 (+v) staging
write addresses, rw0 / rw1 / rx0 / rx1 (+v) fragment read addresses (LDS stage = bit 16, toggled by v_xor), base / nbase (s, 64-bit)
this tile's / the next tile's operand base of the wave, rowb (s) bytes per operand row, cnt (+s) K-steps of the middle loop (nk - 3).

State at block entry (= at exit, for the next tile): K-step 0 of the tile in stage P, K-step 1 in the other stage Q, both visible
(a barrier has passed), S free, nothing in flight; rw0, rw1, rx0, rx1 -> P; wa0, wa1 -> Q; g = lane offset + 2 * 128.
K-step s: 16 clusters of 8 MFMAs (one activation fragment x 8 weight fragments).  Clusters 0..12 carry, one per MFMA gap, the 16
pieces' (wait vmcnt(15), ds_write S[p] -> other stage, global_load S[p] <- K-step s + 2) and the fragment reads (activation fragment
two clusters ahead, weight fragment of the second k32 half); s_barrier after cluster 13; clusters 14 / 15 compute from registers and
read the next step's first fragments from the other stage.  First step of a tile: accumulates onto the inline constant 0 (no
accumulator is ever zeroed) and writes nothing (its operands were staged by the previous tile's last steps).  The second-to-last
step's loads fetch the NEXT tile's K-step 0 (written to LDS by the last step); the next tile's K-step 1 goes by LDS-DMA, issued behind
the last barrier of the tile, straight into the stage the tile has finished with — no register holds it, so it lands while the
epilogue (C++) runs, and the next tile's first barrier is preceded by the counted wait that covers it.

  python3 tools/gen_gemm_w4.py            (rewrites manner_amd/csrc/gemm_w4_asm.inc; the file is committed)
"""
import os
import sys

S0, WF, XF = 0, (64, 96), 128
NPRE = 14
L = []
NW = 4
NA = 8            # weight fragments per wave (16-column blocks)
NP = 16           # staging pieces per wave and K-step
VTMP = 144        # the block's address temporary


def e(s):
    L.append(s)


def vr(base, n=4):
    return f"v[{base}:{base + n - 1}]"


def ar(a, b):
    i = 4 * (8 * a + b)
    return f"a[{i}:{i + 3}]" if NW == 4 else f"v[{i}:{i + 3}]"


def piece_cluster(p):
    return (p * (NPRE - 1)) // NP


class LdsQueue:
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


def post_reads(tw, tx):
    h = NA // 2
    return [(tx, 0)] + [(tw, a) for a in range(h)] + [(tx, 1)] + [(tw, a) for a in range(h, NA)]


def entry_reads():
    e("s_waitcnt lgkmcnt(0)")
    e(f"ds_read_b128 {vr(XF)}, %[rx0]")
    for a in range(NA // 2):
        e(f"ds_read_b128 {vr(WF[0] + 4 * a)}, %[rw0] offset:{a * 2048}")
    e(f"ds_read_b128 {vr(XF + 4)}, %[rx0] offset:2048")
    for a in range(NA // 2, NA):
        e(f"ds_read_b128 {vr(WF[0] + 4 * a)}, %[rw0] offset:{a * 2048}")
    e("v_xor_b32 %[rw0], 0x10000, %[rw0]")


def addr_of(p):
    """Global byte offset of piece p: %[g] (+ the wave-uniform p * 8 * rowbytes in %[c<p>]) through the block's one address temporary."""
    if p == 0:
        return [], "%[g]"
    return [("alu", f"v_add_u32 v{VTMP}, %[c{p}], %[g]", None)], f"v{VTMP}"


def kstep(first, last, base, stage="dma", early=False):
    q = LdsQueue(post_reads("w0", "x"))
    h = NA // 2
    if early and stage == "regs":
        for p in range(NP):                          # the next tile's K-step 1 (g stands at its K-step 0: + 128 bytes) -> sa
            pre, ad = addr_of(p)
            for _, text, _ in pre:
                e(text)
            e(f"global_load_dwordx4 %[sa{p}], {ad}, %[{base}] offset:128")
    for c in range(16):
        s2, b = c >> 3, c & 7
        slot = c & 3
        mem = []
        cn = c + 2
        if cn < 16:
            mem.append(("lds", f"ds_read_b128 {vr(XF + 4 * (cn & 3))}, %[rx{cn >> 3}] offset:{(cn & 7) * 2048}", ("x", cn)))
        elif not last:
            mem.append(("lds", f"ds_read_b128 {vr(XF + 4 * (cn & 3))}, %[rx0] offset:{(cn & 7) * 2048}", ("nx", cn & 7)))
        if c < NA:
            mem.append(("lds", f"ds_read_b128 {vr(WF[1] + 4 * c)}, %[rw1] offset:{c * 2048}", ("w1", c)))
        if c >= 14 and not last:
            for a in range(h * (c - 14), h * (c - 14) + h):
                mem.append(("lds", f"ds_read_b128 {vr(WF[0] + 4 * a)}, %[rw0] offset:{a * 2048}", ("nw", a)))
        for p in range(NP):
            if piece_cluster(p) == c:
                grp = []
                if stage == "regs":
                    # the next tile's K-step 1 rides through the epilogue in the operands sa0.. (registers of the COMPILER's choosing,
                    # live across the epilogue).  They are free from the tile's first step on, so the request goes out EARLY: at the head
                    # of the second-to-last step (`early`), two K-steps before the epilogue, which therefore never waits behind them
                    # (vector-memory operations complete in order: a request issued late stalls the epilogue's own first loads — 3.5 k
                    # cycles per tile by the in-kernel stamps).  (AGPRs would be free of charge — VMEM loads and DS writes address them
                    # directly — but hipcc halves the VGPR budget of a 256-register kernel to 128 the moment it sees one.)
                    if first:                              # sa landed long ago (the last step's counted waits on younger loads cover it)
                        grp.append(("lds", f"ds_write_b128 %[wa{p & 1}], %[sa{p}] offset:{p * 1024}", ("s", p)))
                    else:
                        # `early` step: the NP requests into sa are in the queue too (younger than S[p]'s load of the step before)
                        grp.append(("wait", f"s_waitcnt vmcnt({NP - 1 - p if last else (2 * NP - 1 if early else NP - 1)})", None))
                        grp.append(("lds", f"ds_write_b128 %[wa{p & 1}], {vr(S0 + 4 * p)} offset:{p * 1024}", ("s", p)))
                    if not last:
                        pre, ad = addr_of(p)
                        grp.extend(pre)
                        grp.append(("vm", f"global_load_dwordx4 {vr(S0 + 4 * p)}, {ad}, %[{base}]", None))
                    if p == NP - 1:
                        grp.append(("alu", "v_add_u32 %[g], 0x80, %[g]", None))
                    mem.append(("group", grp, None))
                    continue
                if stage == "regs_unused":
                    # the next tile's K-step 1 rides through the epilogue in the operands sa0.. (registers of the COMPILER's choosing,
                    # live across the epilogue): requested by the last step right behind each piece's write (as every step requests two
                    # ahead), written to LDS by the next tile's first step.  (AGPRs would be free of charge — VMEM loads and DS writes
                    # address them directly — but hipcc halves the VGPR budget of a 256-register kernel to 128 the moment it sees one.)
                    grp.append(("wait", f"s_waitcnt vmcnt({NP - 1})", None))
                    src = f"%[sa{p}]" if first else vr(S0 + 4 * p)
                    grp.append(("lds", f"ds_write_b128 %[wa{p & 1}], {src} offset:{p * 1024}", ("s", p)))
                    dst = f"%[sa{p}]" if last else vr(S0 + 4 * p)
                    pre, ad = addr_of(p)
                    grp.extend(pre)
                    grp.append(("vm", f"global_load_dwordx4 {dst}, {ad}, %[{base}]", None))
                    if p == NP - 1:
                        grp.append(("alu", "v_add_u32 %[g], 0x80, %[g]", None))
                    mem.append(("group", grp, None))
                    continue
                if not first:
                    # (the last step requests nothing into S: its loads in flight shrink with every piece written)
                    grp.append(("wait", f"s_waitcnt vmcnt({NP - 1 - p if last else NP - 1})", None))
                    grp.append(("lds", f"ds_write_b128 %[wa{p & 1}], {vr(S0 + 4 * p)} offset:{p * 1024}", ("s", p)))
                if not last:
                    pre, ad = addr_of(p)
                    grp.extend(pre)
                    grp.append(("vm", f"global_load_dwordx4 {vr(S0 + 4 * p)}, {ad}, %[{base}]", None))
                    if p == NP - 1:
                        grp.append(("alu", "v_add_u32 %[g], 0x80, %[g]", None))
                mem.append(("group", grp, None))
        if last and c >= 14 and stage == "dma":
            # behind the last barrier of the tile the stage it has finished with is free: the NEXT tile's K-step 1 goes there by LDS-DMA
            # (no register holds it, so it lands while the epilogue runs: the one place where the DMA's issue cost buys something).
            # Linear 1 KiB destination per piece (M0), the row swizzle of the LDS image applied to the SOURCE chunk (d0 / d1).
            for p in range((NP // 2) * (c - 14), (NP // 2) * (c - 14) + NP // 2):
                grp = [("alu", f"v_add_u32 v{S0 + p}, %[d{p & 1}], %[g]", None)]
                if p:
                    grp.append(("alu", f"v_add_u32 v{S0 + p}, %[c{p}], v{S0 + p}", None))
                grp += [("alu", f"s_add_u32 m0, %[cnt], {p * 1024}", None),
                        ("alu", "s_nop 0", None),
                        ("vm", f"global_load_lds_dwordx4 v{S0 + p}, %[{base}]", None)]
                if p == NP - 1:
                    grp.append(("alu", "v_add_u32 %[g], 0x80, %[g]", None))
                mem.append(("group", grp, None))
        if c == 8:
            mem.append(("alu", "v_xor_b32 %[rx0], 0x10000, %[rx0]", None))
            mem.append(("alu", "v_xor_b32 %[rw1], 0x10000, %[rw1]", None))
        if c == 13:
            mem.append(("alu", "v_xor_b32 %[wa0], 0x10000, %[wa0]", None))
            mem.append(("alu", "v_xor_b32 %[wa1], 0x10000, %[wa1]", None))
        groups = [m[1] if m[0] == "group" else [m] for m in mem]
        gaps = [[] for _ in range(NA)]
        for gi, g in enumerate(groups):
            gaps[min(gi, NA - 1) if len(groups) <= NA else (gi * NA) // len(groups)].extend(g)
        for a in range(NA):
            q.need(("w0" if s2 == 0 else "w1", a))
            if a == 0:
                q.need(("x", c))
            srcc = "0" if (first and s2 == 0) else ar(a, b)
            e(f"@MFMA@ {ar(a, b)}, {vr(WF[s2] + 4 * a)}, {vr(XF + 4 * slot)}, {srcc}")
            for kind, text, tag in gaps[a]:
                e(text)
                if kind == "lds":
                    q.issue(tag)
        if c == NPRE - 1:
            # first step of a tile: the K-step 1 operands came by LDS-DMA behind the previous tile's last barrier — everything older than
            # this step's own NP loads has to have landed before the barrier that publishes it
            e(f"s_waitcnt vmcnt({NP}) lgkmcnt(0)" if (first and stage == "dma") else "s_waitcnt lgkmcnt(0)")
            e("s_barrier")
            q.drain()
            if last and stage == "dma":
                e("v_readfirstlane_b32 %[cnt], %[wa0]")      # lane 0's staging address = the wave's piece 0 in the freed stage (cnt is dead: reused)
        if c == 14:
            e("v_xor_b32 %[rx1], 0x10000, %[rx1]")
    if not last:
        e("v_xor_b32 %[rw0], 0x10000, %[rw0]")
        assert q.q == post_reads("nw", "nx"), q.q
    else:
        assert q.q == [], q.q
        e("s_nop 7")                                  # the last matrix instruction's result registers are read by the epilogue next
        e("s_nop 7")


def tile_block(stage="dma"):
    if stage == "dma":
        e("s_mov_b32 %[m0s], m0")                    # M0 addresses the LDS-DMA pieces: saved and restored (it is not ours to clobber)
    entry_reads()
    kstep(True, False, "base", stage)
    e("s_cmp_eq_u32 %[cnt], 0")
    e("s_cbranch_scc1 .Lw4_switch_%=")
    e(".Lw4_mid_%=:")
    # (a per-tile L2 touch of the next tile's first K-steps — one 4-byte load per row, two passes before the load stream enters the next
    # tile — was built and measured: Q|K|V 202 -> 206 us, FFN1 307 -> 313: the tile boundary is not an HBM-latency problem.  Removed.)
    kstep(False, False, "base", stage)
    e("s_sub_u32 %[cnt], %[cnt], 1")
    e("s_cmp_lg_u32 %[cnt], 0")
    e("s_cbranch_scc1 .Lw4_mid_%=")
    e(".Lw4_switch_%=:")
    e("v_subrev_u32 %[g], %[rowb], %[g]")            # the load stream enters the next tile: K-offset back to 0
    kstep(False, False, "nbase", stage, early=True)
    kstep(False, True, "nbase", stage)
    if stage == "dma":
        e("s_mov_b32 m0, %[m0s]")


def main():
    global NW, NA, NP, S0, WF, XF, VTMP
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for nw in (8, 4):
        NW, NA, NP = nw, (8 if nw == 4 else 4), (16 if nw == 4 else 8)
        # NW = 4: a[0:255] acc | v[0:63] S, v[64:127] wf, v[128:143] xf.   NW = 8: v[0:127] acc | v[128:159] S, v[160:191] wf, v[192:207] xf
        S0, WF, XF = (0, (64, 96), 128) if nw == 4 else (128, (160, 176), 192)
        VTMP = 144 if nw == 4 else 208
        del L[:]
        out = os.path.join(root, "manner_amd", "csrc", f"gemm_w{nw}_asm.inc")
        tile_block()
        body = L[:]
        body_a = None
        if nw == 8:
            del L[:]
            tile_block("regs")
            body_a = L[:]
        n_mfma = sum(x.startswith("@MFMA@") for x in body)
        H = []
        H.append("// GENERATED by tools/gen_gemm_w.py — do not edit; the design is described there and in gemm.hip.")
        H.append(f"// One tile's K-loop of the {nw}-wave 16-bit GEMM: {len(body)} instructions, {n_mfma} matrix instructions in 4 step bodies.")
        H.append(f"#define MANNER_W{nw}_TILE_ASM(MFMA) \\")
        for ln in body:
            if ln.startswith("@MFMA@"):
                H.append(f'  MFMA "{ln[len("@MFMA@"):]}\\n" \\')
            else:
                H.append(f'  "{ln}\\n" \\')
        H.append('  ""')
        H.append("")
        if body_a is not None:
            H.append("// the same K-loop with the next tile's K-step 1 carried through the epilogue in 32 registers (operands sa0..sa7) instead of by LDS-DMA")
            H.append(f"#define MANNER_W{nw}_TILE_ASM_REGS(MFMA) \\")
            for ln in body_a:
                if ln.startswith("@MFMA@"):
                    H.append(f'  MFMA "{ln[len("@MFMA@"):]}\\n" \\')
                else:
                    H.append(f'  "{ln}\\n" \\')
            H.append('  ""')
            H.append("")
            H.append("#define MANNER_W8_REGS_OPERANDS(sa) \\")
            H.append("  " + ", ".join(f'[sa{j}] "+v"(sa[{j}])' for j in range(8)))
            H.append("")
        if nw == 4:
            regs = [f'"v{i}"' for i in range(145)] + [f'"a{i}"' for i in range(256)]
        else:
            regs = [f'"v{i}"' for i in range(128, 209)]      # the accumulators v[0:127] are OUTPUTS of the block (8 x 16 registers)
        H.append(f"#define MANNER_W{nw}_CLOBBERS \"memory\", \"scc\", \\")
        for i in range(0, len(regs), 16):
            H.append("  " + ", ".join(regs[i:i + 16]) + (", \\" if i + 16 < len(regs) else ""))
        H.append("")
        if nw == 4:
            for h in (0, 1):
                H.append(f"// accumulators of the wave tile's column half {h} (a = {4 * h}..{4 * h + 3}) -> acc[a - {4 * h}][b]")
                H.append(f"#define MANNER_W4_READ_HALF{h}(acc) \\")
                lines = []
                for a in range(4 * h, 4 * h + 4):
                    for b in range(8):
                        i = 4 * (8 * a + b)
                        lines.append(f'  asm volatile("" : "={{a[{i}:{i + 3}]}}"(acc[{a - 4 * h}][{b}]));')
                H.append(" \\\n".join(lines))
                H.append("")
        else:
            H.append("// the block's accumulator outputs: o[j] = v[16 j : 16 j + 15] = acc[j / 2][4 (j & 1) .. + 3]")
            H.append("#define MANNER_W8_ACC_OUTPUTS(o) \\")
            H.append("  " + ", ".join(f'"=&{{v[{16 * j}:{16 * j + 15}]}}"(o[{j}])' for j in range(8)))
            H.append("")
        with open(out, "w") as f:
            f.write("\n".join(H) + "\n")
        print(f"{out}: {len(body)} instructions, {n_mfma} MFMAs")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Generator of manner_amd/csrc/gemm_p4_asm.inc — the K-loop of `gemm_tn_p4_kernel` (round 6): the PAIRED 4-wave form of the 16-bit GEMM.

Why: by the in-kernel stamps (profiles/r6_final/stamps_w8_x16.txt) a 256 x 256 tile of the 8-wave kernel spends 15 - 27 % of its time in
the epilogue, where the matrix pipe idles, and its K-loop keeps neither the matrix pipe (72 %) nor the LDS (70 %) busy because all
eight waves of the one workgroup a CU holds stop at the same barriers.  Here a CU holds TWO independent workgroups of four waves (one
per SIMD each, so still two waves per SIMD: the epilogues keep their VALU issue rate), each walking its own 256 x 128 tiles: one
workgroup's barriers, LDS write phase and epilogue are the other's matrix time.

Geometry of one workgroup: tile 256 rows x 128 columns, wave tile 128 x 64 (wave w: columns 64 (w & 1).., rows 128 (w >> 1)..), the
128 accumulators of a wave in v[0:127] as in the 8-wave kernel (same fragments, same matrix instruction per output element in the same
order over K: the same bits).  LDS: ONE stage of a K-step's operands — weights 128 rows x 128 B at 0, activations 256 rows x 128 B at
16 KiB — plus a 4 KiB epilogue slab per wave: 64 KiB, two workgroups per CU.  Register staging: a wave's share of a K-step is 12 pieces
of 8 rows x 128 B (weight rows 32 w.., activation rows 64 w..), requested one K-step before they are written.

K-step: clusters 0..13 (4 matrix instructions each) read their fragments from the stage; lgkmcnt(0), s_barrier (every wave has read
the stage); clusters 14 / 15 compute from registers while the 12 pieces of the NEXT step are written over the stage and the step after
that is requested into the same registers; lgkmcnt(0), s_barrier (the stage is published); the next step's first fragments.  Nothing is
in flight across the epilogue: the tile's last step writes the next tile's K-step 0 (requested by the step before it), the next tile's
first step requests its K-step 1 at once.

Registers of the block (clobbered): v[128:175] S[12], v[176:207] wf[2][4], v[208:223] xf[4], v224 address temporary; outputs v[0:127].
Operands: g (+v) lane byte offset of piece 0 at the load stream's K-step; ww0/ww1, xw0/xw1 (v) piece write addresses by row-block parity;
rw0/rw1, rx0/rx1 (v) fragment read addresses of the two k32 halves; basew/basex, nbasew/nbasex (s, 64-bit) the wave's operand shares of
this / the next tile; rowb (s), c1..c7 (s) = p * 8 * rowb; cnt (+s) middle steps (nk - 3).

  python3 tools/gen_gemm_p4.py            (rewrites manner_amd/csrc/gemm_p4_asm.inc; the file is committed)
"""
import os

NA, NPW, NPX = 4, 4, 8
NP = NPW + NPX
S0, WF, XF, VTMP = 128, (176, 192), 208, 224
NPRE = 14
L = []


def e(s):
    L.append(s)


def vr(base, n=4):
    return f"v[{base}:{base + n - 1}]"


def ar(a, b):
    i = 4 * (8 * a + b)
    return f"v[{i}:{i + 3}]"


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


def entry_read_list():
    h = NA // 2
    return [("x", 0)] + [("w0", a) for a in range(h)] + [("x", 1)] + [("w0", a) for a in range(h, NA)]


def entry_reads(wait=False):
    if wait:
        e("s_waitcnt lgkmcnt(0)")                   # (scalar loads return out of order: nothing of the compiler's may be in flight)
    for kind, i in entry_read_list():
        if kind == "x":
            e(f"ds_read_b128 {vr(XF + 4 * i)}, %[rx0] offset:{i * 2048}")
        else:
            e(f"ds_read_b128 {vr(WF[0] + 4 * i)}, %[rw0] offset:{i * 2048}")


def piece(p):
    """(write address operand, LDS offset, global base suffix, piece index within its operand)"""
    if p < NPW:
        return f"%[ww{p & 1}]", p * 1024, "w", p
    j = p - NPW
    return f"%[xw{j & 1}]", j * 1024, "x", j


def load_group(p, base):
    _, _, op, j = piece(p)
    grp = []
    ad = "%[g]"
    if j:
        grp.append(("alu", f"v_add_u32 v{VTMP}, %[c{j}], %[g]"))
        ad = f"v{VTMP}"
    grp.append(("vm", f"global_load_dwordx4 {vr(S0 + 4 * p)}, {ad}, %[{base}{op}]"))
    if p == NP - 1:
        grp.append(("alu", "v_add_u32 %[g], 0x80, %[g]"))
    return grp


def pstep(first, last, base):
    q = LdsQueue(entry_read_list())
    for c in range(16):
        s2, b = c >> 3, c & 7
        slot = c & 3
        groups = []
        cn = c + 2
        if cn < 16:
            groups.append([("lds", f"ds_read_b128 {vr(XF + 4 * (cn & 3))}, %[rx{cn >> 3}] offset:{(cn & 7) * 2048}", ("x", cn))])
        if c < NA:
            groups.append([("lds", f"ds_read_b128 {vr(WF[1] + 4 * c)}, %[rw1] offset:{c * 2048}", ("w1", c))])
        if first and c < 6:
            # the tile's K-step 1: requested at once (nothing rode through the epilogue), written behind this step's first barrier
            for p in (2 * c, 2 * c + 1):
                groups.append([(k, t, None) for k, t in load_group(p, base)])
        if c >= 14:
            for p in range((NP // 2) * (c - 14), (NP // 2) * (c - 14) + NP // 2):
                ad, off, _, _ = piece(p)
                grp = [("wait", f"s_waitcnt vmcnt({NP - 1 - p if last else NP - 1})", None),
                       ("lds", f"ds_write_b128 {ad}, {vr(S0 + 4 * p)} offset:{off}", ("s", p))]
                if not last:
                    grp += [(k, t, None) for k, t in load_group(p, base)]
                groups.append(grp)
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
            e("s_waitcnt lgkmcnt(0)")               # every fragment of this step is in registers
            e("s_barrier")                          # ... in every wave's: the stage may be overwritten
            q.q = []
    e("s_waitcnt lgkmcnt(0)")
    e("s_barrier")                                  # the stage holds the next K-step
    if last:
        e("s_nop 7")                                # the last matrix instruction's result registers are read by the epilogue next
        e("s_nop 7")
    else:
        entry_reads()


def tile_block():
    entry_reads(True)
    pstep(True, False, "base")
    e("s_cmp_eq_u32 %[cnt], 0")
    e("s_cbranch_scc1 .Lp4_switch_%=")
    e(".Lp4_mid_%=:")
    pstep(False, False, "base")
    e("s_sub_u32 %[cnt], %[cnt], 1")
    e("s_cmp_lg_u32 %[cnt], 0")
    e("s_cbranch_scc1 .Lp4_mid_%=")
    e(".Lp4_switch_%=:")
    e("v_subrev_u32 %[g], %[rowb], %[g]")            # the load stream enters the next tile: K-offset back to 0
    pstep(False, False, "nbase")
    pstep(False, True, "nbase")


def main():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = os.path.join(root, "manner_amd", "csrc", "gemm_p4_asm.inc")
    del L[:]
    tile_block()
    n_mfma = sum(x.startswith("@MFMA@") for x in L)
    H = ["// GENERATED by tools/gen_gemm_p4.py — do not edit; the design is described there and in gemm.hip.",
         f"// One 256 x 128 tile's K-loop of the paired 4-wave 16-bit GEMM: {len(L)} instructions, {n_mfma} matrix instructions in 4 step bodies.",
         "#define MANNER_P4_TILE_ASM(MFMA) \\"]
    for ln in L:
        if ln.startswith("@MFMA@"):
            H.append(f'  MFMA "{ln[len("@MFMA@"):]}\\n" \\')
        else:
            H.append(f'  "{ln}\\n" \\')
    H.append('  ""')
    H.append("")
    regs = [f'"v{i}"' for i in range(S0, VTMP + 1)]
    H.append('#define MANNER_P4_CLOBBERS "memory", "scc", \\')
    for i in range(0, len(regs), 16):
        H.append("  " + ", ".join(regs[i:i + 16]) + (", \\" if i + 16 < len(regs) else ""))
    H.append("")
    H.append("// the block's accumulator outputs: o[j] = v[16 j : 16 j + 15] = acc[j / 2][4 (j & 1) .. + 3]")
    H.append("#define MANNER_P4_ACC_OUTPUTS(o) \\")
    H.append("  " + ", ".join(f'"=&{{v[{16 * j}:{16 * j + 15}]}}"(o[{j}])' for j in range(8)))
    H.append("")
    with open(out, "w") as f:
        f.write("\n".join(H) + "\n")
    print(f"{out}: {len(L)} instructions, {n_mfma} MFMAs")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Writes gnn-lm_amd/csrc/gemm_sched_loop.inc: the hand-placed main loop (one inline-asm statement) of the scheduled
f32 GEMM kernel (csrc/gemm_f32_sched.hip).  128x128x32 stages, 4 waves of 64x64, v_mfma_f32_32x32x2_f32.

Per stage and wave: 64 MFMAs, 16 ds_read_b128 (fragments of k-group g + 1 under the MFMAs of group g), 8 global loads of the
NEXT stage into staging registers (early in the stage), 8 ds_write_b128 of them into the other LDS buffer (late, behind
counted vmcnt waits), one barrier.  Everything that is not an MFMA costs ~4-5 issue cycles of the SIMD next to the f32
MFMA stream (tools/probes/mfma_mix.hip), so the loop is written down instruction by instruction instead of left to the
scheduler.

Operands of the asm statement (named):
  c00, c01, c10, c11   the wave's four 32x32 accumulators (f32x16, "+v")
  cnt  remaining stage pairs ("+s");  koff  byte offset of the NEXT stage's k inside a row ("+s": the loads' soffset)
  sa, sw   buffer descriptors of the A / W panel (four SGPRs each);  inc  bytes per stage (128)
  oa0..3 / ow0..3   per-thread byte offsets of the four A / W rows it stages ("v")
  lw   LDS write address of the thread (row t/8, swizzled 16-B slot)
  ra0..3 / rb0..3   LDS read addresses of the A / W fragments of k-groups 0..3
Fixed registers (clobbers): v160-v191 staging, v192-v207 A fragments (2 sets x 2 tiles x 4), v208-v223 W fragments."""
import os

BASE = int(os.environ.get("GNNLM_SCHED_VBASE", 160))        # first of the 64 fixed registers
ST = lambda q: f"v[{BASE + 4 * q}:{BASE + 3 + 4 * q}]"
FA = lambda s, i, k=None: (f"v[{BASE + 32 + (2 * s + i) * 4}:{BASE + 35 + (2 * s + i) * 4}]" if k is None else f"v{BASE + 32 + (2 * s + i) * 4 + k}")
FB = lambda s, j, k=None: (f"v[{BASE + 48 + (2 * s + j) * 4}:{BASE + 51 + (2 * s + j) * 4}]" if k is None else f"v{BASE + 48 + (2 * s + j) * 4 + k}")
ACC = {(0, 0): "%[c00]", (0, 1): "%[c01]", (1, 0): "%[c10]", (1, 1): "%[c11]"}
BUF = 32768


def frag_reads(g, buf, s):
    """the four fragment reads of k-group g from LDS buffer `buf` into register set s"""
    o = buf * BUF
    return [f"ds_read_b128 {FA(s, 0)}, %[ra{g}] offset:{o}",
            f"ds_read_b128 {FB(s, 0)}, %[rb{g}] offset:{o}",
            f"ds_read_b128 {FA(s, 1)}, %[ra{g}] offset:{o + 4096}",
            f"ds_read_b128 {FB(s, 1)}, %[rb{g}] offset:{o + 4096}"]


def stage(b, last_of_pair):
    L = []
    loads = [f"buffer_load_dwordx4 {ST(q)}, %[oa{q}], %[sa], %[koff] offen" for q in range(4)] + \
            [f"buffer_load_dwordx4 {ST(4 + q)}, %[ow{q}], %[sw], %[koff] offen" for q in range(4)]
    wbase = (1 - b) * BUF
    writes = [f"ds_write_b128 %[lw], {ST(q)} offset:{wbase + 4096 * q}" for q in range(4)] + \
             [f"ds_write_b128 %[lw], {ST(4 + q)} offset:{wbase + 16384 + 4096 * q}" for q in range(4)]
    n = 0
    for g in range(4):
        s = g & 1
        L.append("s_waitcnt lgkmcnt(0)")                                   # fragments of group g
        nxt = frag_reads(g + 1, b, 1 - s) if g < 3 else []
        for kk in range(4):
            for (i, j) in ((0, 0), (0, 1), (1, 1), (1, 0)):
                x, y = (FB(s, j, kk), FA(s, i, kk)) if SWAP else (FA(s, i, kk), FB(s, j, kk))     # SWAP: transposed accumulators (LSE epilogue)
                L.append(f"v_mfma_f32_32x32x2_f32 {ACC[(i, j)]}, {x}, {y}, {ACC[(i, j)]}")
                m = n % 16
                if g == 0 and m < 8:
                    L.append(loads[m])                                      # the next stage's rows, early
                if g == 0 and m == 9:                                       # k advances while stages remain (the last stage re-reads its own rows)
                    if not last_of_pair:                                     # stage nk - 2 loaded the last stage: stay on it
                        L += ["s_cmp_lg_u32 %[cnt], 1", "s_cselect_b32 vcc_lo, %[inc], 0", "s_add_u32 %[koff], %[koff], vcc_lo"]
                    else:
                        L += ["s_add_u32 %[koff], %[koff], %[inc]"]
                if nxt and m in (2, 5, 8, 11):
                    L.append(nxt[(m - 2) // 3])
                if g == 3 and m % 2 == 1:
                    q = m // 2
                    L.append(f"s_waitcnt vmcnt({7 - q})")
                    L.append(writes[q])
                n += 1
    L += ["s_waitcnt lgkmcnt(0)", "s_barrier"]
    L += frag_reads(0, 1 - b, 0)                                            # group 0 of the next stage
    return L


SWAP = False


OUT_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gnn-lm_amd", "csrc")


def main():
    global SWAP, OUT_DIR
    import sys
    if len(sys.argv) > 1:                      # tools/gen_gemm_sched.py <dir>: write there (tests compare with csrc/)
        OUT_DIR = sys.argv[1]
    for SWAP, fname in ((False, "gemm_sched_loop.inc"), (True, "gemm_sched_loop_t.inc")):
        emit(fname)
    out = os.path.join(OUT_DIR, "gemm_sched_clobbers.inc")
    with open(out, "w") as f:
        f.write("// generated by tools/gen_gemm_sched.py -- do not edit: the fixed registers of gemm_sched_loop*.inc\n")
        regs = [f'"v{BASE + i}"' for i in range(64)]
        for i in range(0, 64, 16):
            f.write(", ".join(regs[i:i + 16]) + ("," if i < 48 else "") + "\n")


def emit(fname):
    L = frag_reads(0, 0, 0)
    L.append("1:")
    L += stage(0, False)
    L += stage(1, True)
    L += ["s_sub_u32 %[cnt], %[cnt], 1", "s_cmp_lg_u32 %[cnt], 0", "s_cbranch_scc1 1b",
          "s_waitcnt lgkmcnt(0)", "s_nop 15", "s_nop 7"]
    txt = "\n".join(L)
    out = os.path.join(OUT_DIR, fname)
    with open(out, "w") as f:
        f.write("// generated by tools/gen_gemm_sched.py -- do not edit\n")
        for l in txt.split("\n"):
            f.write('"' + l + '\\n"\n')
    print(out, len(L), "lines")


if __name__ == "__main__":
    main()

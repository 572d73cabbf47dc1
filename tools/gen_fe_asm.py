#!/usr/bin/env python3
"""Writes lightning_amd/csrc/fe_asm.inc: fe_mul / fe_sqr and their fused forms (fe.h, LAMD_FE_CHAINS) as ONE inline-asm
statement each -- the whole operation, tail included.  Same arithmetic, same order as the C macro; the point of the single
statement is that hipcc's hazard recogniser, which assumes a dst-forwarding hazard after every opaque inline-asm statement,
cannot put s_nop between the dependent v_mad_u64_u32, and that the instruction order is ours.

Schedule (round 4; fe.h explains the arithmetic):
  column 8     in the LO accumulator; HI = LO >> 29 carries on, LO = [h8 : 0] is kept as a 64-bit addend
  high chain   columns 9..15 in HI -> h[1..7] (29 bits each); column 16 stays whole in HI: V = [Vlo : Vhi]
  fold         h[1] += Vhi << 11;  P8 = 8*R0*Vhi + h8 + 256*Vlo (in LO);  h[1] += P8.hi << 3;  P8.lo moves to a temporary
  low chain    columns 0..7 in LO with R0*h[k+1] (column 7: R0*Vlo) and 256*h[k] -> r[0..7], LO = carry of column 7
  end          LO += P8.lo;  r8 = LO & (2^24-1);  e2 = LO >> 24;  r0 += 977*e2;  r1 += 8*e2
The two 64-bit accumulators are pinned to fixed VGPR pairs because an inline-asm operand cannot name the halves of a 64-bit
register operand (the halves feed v_and_b32 / v_alignbit_b32 / a multiply-add).  r[k] (1 <= k <= 7) is written into the
register that held h[k], dead by then.
Operands: %0 r[0] | %1 temporary (P8.lo, then e2) | %2..%8 h[1..7] = r[1..7] | %9 r[8] | %10 HI (pinned) | %11 LO (pinned) |
%12..%20 a[0..8] | %21..%29 b[0..8] (fe_sqr: the doubled limbs d[0..8]) | %30 R0 | %31 256 | %32 977 | %33 8*R0 |
%34.. second product / addend limbs.

--ilp: the LATENCY schedule (lightning_amd/csrc/fe_asm_ilp.inc, used by the translation unit of k_small_verify only).  The schedule above is ONE dependent
chain of 139 instructions: right for a SIMD that holds three waves (its issue port is busy 94 % of the time), wrong for the latency path, where a wave is
alone on its SIMD for most of its life and a dependent instruction issues every ~8.7 cycles instead of every ~4.3.  Here the pure products of the low
columns 0..7 go into EIGHT accumulators of their own (L0..L7, pinned to v[24:39]), emitted between the instructions of the high chain (diagonally, so that
neighbours are independent); the folds R0*h[k+1] / 256*h[k] join L_k as soon as the high chain has cut those limbs; what is left at the end is a carry
pass of three instructions per column (v_lshl_add_u64 carry + L_k, mask, shift).  Same column sums, same cuts, the same result bit for bit; 146 instructions
instead of 139, about 90 of them on the critical path.  Eight more pinned outputs shift the input operands by eight (%20.. a, %29.. b, ...).
usage: tools/gen_fe_asm.py > lightning_amd/csrc/fe_asm.inc;  tools/gen_fe_asm.py --ilp > lightning_amd/csrc/fe_asm_ilp.inc"""
import sys

ILP = "--ilp" in sys.argv
# --signed: the pure products (limb x limb) as v_mad_i64_i32 -- the same 64-bit result for factors below 2^31 (the additions wrap mod 2^64 either way) and
# ~6 % cheaper to issue (profiles/r02_fe_bench_variants.txt); the folds whose factor is a raw 32-bit accumulator half, and the addend's "times 1", stay
# unsigned.  Contract: every LIMB factor below 2^31 -- magnitude <= 3 for a multiplication's operands, <= 1 for a squaring's (its doubled limbs).
SIGNED = "--signed" in sys.argv
SCARRY = "--sgpr-carry" in sys.argv      # the (unused) carry-out into s[20:21] instead of vcc
OFF = 8 if ILP else 0                # the latency schedule has eight more (pinned) outputs in front of the inputs
HI, LO = 20, 22                      # v[20:21], v[22:23]
LK = lambda k: 24 + 2 * k            # v[24:25] .. v[38:39]: L0..L7 (latency schedule)
R0r = "%0"
H = lambda j: "%%%d" % (1 + j)       # h[j], j = 0..8; r[k] = H(k) for k >= 1
A = lambda i: "%%%d" % (12 + OFF + i)
B = lambda i: "%%%d" % (21 + OFF + i)
SR0, S256, S977, S8R0 = ("%%%d" % (30 + OFF), "%%%d" % (31 + OFF), "%%%d" % (32 + OFF), "%%%d" % (33 + OFF))
C2 = lambda i: "%%%d" % (34 + OFF + i)     # second product's left operand (LAMD_FE_MUL2_ASM) / the addend limbs (..._ADD_ASM)
D2 = lambda i: "%%%d" % (43 + OFF + i)     # second product's right operand
M29, M24 = "0x1fffffff", "0xffffff"
hi, lo = "v[%d:%d]" % (HI, HI + 1), "v[%d:%d]" % (LO, LO + 1)
hil, hih, lol, loh = "v%d" % HI, "v%d" % (HI + 1), "v%d" % LO, "v%d" % (LO + 1)

EXTRA = {"prod2": False, "addend": False}   # set per emitted macro


def mad(acc, x, y, addend=None, limbs=False):
    return "%s %s, %s, %s, %s, %s" % ("v_mad_i64_i32" if SIGNED and limbs else "v_mad_u64_u32", acc, "s[20:21]" if SCARRY else "vcc", x, y, acc if addend is None else addend)


def column(acc, k, square, first):
    out = []
    for i in range(9):
        j = k - i
        if j < 0 or j > 8 or (square and i > j):
            continue
        if square:
            x, y = (A(i), A(j)) if i == j else (B(i), A(j))      # B = doubled limbs
        else:
            x, y = A(i), B(j)
        out.append(mad(acc, x, y, "0" if first and not out else None, limbs=True))
    if EXTRA["prod2"]:      # a second full product c*d accumulated into the same columns: one reduction for a*b + c*d
        for i in range(9):
            j = k - i
            if 0 <= j <= 8:
                out.append(mad(acc, C2(i), D2(j), limbs=True))
    if EXTRA["addend"] and k <= 8:   # + e: limb k of a lazy field element joins column k (times the inline constant 1)
        out.append(mad(acc, C2(k), "1"))
    return out


def body(square):
    T0, R8 = H(0), H(8)
    ins = column(lo, 8, square, True)                # column 8; its carry heads the high chain, its low 29 bits wait in LO
    ins += ["v_lshrrev_b64 %s, 29, %s" % (hi, lo), "v_and_b32 %s, %s, %s" % (lol, M29, lol), "v_mov_b32 %s, 0" % loh]
    for j in range(1, 8):                            # high chain: columns 9..15
        ins += column(hi, 8 + j, square, False)
        ins += ["v_and_b32 %s, %s, %s" % (H(j), M29, hil), "v_lshrrev_b64 %s, 29, %s" % (hi, hi)]
    ins += column(hi, 16, square, False)             # column 16 stays whole: V = Vlo + 2^32 * Vhi
    ins.append("v_lshl_add_u32 %s, %s, 11, %s" % (H(1), hih, H(1)))         # 2^8 * (8 * Vhi) * 2^(29*9) joins h9
    ins += [mad(lo, hih, S8R0), mad(lo, hil, S256)]                         # P8 = R0 * 8 * Vhi + 2^8 * Vlo + h8
    ins.append("v_lshl_add_u32 %s, %s, 3, %s" % (H(1), loh, H(1)))          # P8.hi * 2^32 * 2^232 = 8 * P8.hi * 2^261
    ins.append("v_mov_b32 %s, %s" % (T0, lol))
    for k in range(8):                               # low chain: columns 0..7
        ins += column(lo, k, square, k == 0)
        ins.append(mad(lo, H(k + 1) if k < 7 else hil, SR0))
        if k > 0:
            ins.append(mad(lo, H(k), S256))
        ins += ["v_and_b32 %s, %s, %s" % (R0r if k == 0 else H(k), M29, lol), "v_lshrrev_b64 %s, 29, %s" % (lo, lo)]
    ins.append(mad(lo, T0, "1"))                     # column 8 = carry of column 7 + P8.lo
    ins.append("v_and_b32 %s, %s, %s" % (R8, M24, lol))
    ins.append("v_alignbit_b32 %s, %s, %s, 24" % (T0, loh, lol))            # e2 < 2^12
    ins.append("v_mad_u32_u24 %s, %s, %s, %s" % (R0r, T0, S977, R0r))
    ins.append("v_lshl_add_u32 %s, %s, 3, %s" % (H(1), T0, H(1)))
    return ins


def body_ilp(square):
    T0, R8 = H(0), H(8)
    L = lambda k: "v[%d:%d]" % (LK(k), LK(k) + 1)
    # stream A: column 8 and the high chain, one dependent chain.  avail[j] = number of A instructions after which h[j] is final
    a = column(lo, 8, square, True)
    a += ["v_lshrrev_b64 %s, 29, %s" % (hi, lo), "v_and_b32 %s, %s, %s" % (lol, M29, lol), "v_mov_b32 %s, 0" % loh]
    avail = {}
    for j in range(1, 8):
        a += column(hi, 8 + j, square, False)
        a += ["v_and_b32 %s, %s, %s" % (H(j), M29, hil), "v_lshrrev_b64 %s, 29, %s" % (hi, hi)]
        avail[j] = len(a)
    a += column(hi, 16, square, False)
    avail["vlo"] = len(a)
    a.append("v_lshl_add_u32 %s, %s, 11, %s" % (H(1), hih, H(1)))
    a += [mad(lo, hih, S8R0), mad(lo, hil, S256)]
    a.append("v_lshl_add_u32 %s, %s, 3, %s" % (H(1), loh, H(1)))
    a.append("v_mov_b32 %s, %s" % (T0, lol))
    avail[1] = len(a)                                # h[1] is final only now
    # stream B: the pure products of columns 0..7, each into its own accumulator, in diagonal order
    cols = [column(L(k), k, square, True) for k in range(8)]
    fill = []
    while any(cols):
        for c in cols:
            if c:
                fill.append((0, c.pop(0)))
    # stream C: the folds, as soon as their limbs exist
    for k in list(range(2, 8)) + [0, 1]:
        src = H(k + 1) if k < 7 else hil
        need = avail[k + 1] if k < 7 else avail["vlo"]
        fill.append((need, mad(L(k), src, SR0)))
        if k > 0:
            fill.append((max(need, avail[k]), mad(L(k), H(k), S256)))
    ins = []
    for n, x in enumerate(a, 1):
        ins.append(x)
        for q, (need, y) in enumerate(fill):
            if need <= n:
                ins.append(y)
                del fill[q]
                break
    # (what is left of the folds: a fold of column k only waits for other folds of column k)
    rest = [y for _, y in fill]
    ins += rest
    # stream D: the carry pass
    ins += ["v_and_b32 %s, %s, v%d" % (R0r, M29, LK(0)), "v_lshrrev_b64 %s, 29, %s" % (lo, L(0))]
    for k in range(1, 8):
        ins += ["v_lshl_add_u64 %s, %s, 0, %s" % (lo, lo, L(k)), "v_and_b32 %s, %s, %s" % (H(k), M29, lol), "v_lshrrev_b64 %s, 29, %s" % (lo, lo)]
    ins.append(mad(lo, T0, "1"))
    ins.append("v_and_b32 %s, %s, %s" % (R8, M24, lol))
    ins.append("v_alignbit_b32 %s, %s, %s, 24" % (T0, loh, lol))
    ins.append("v_mad_u32_u24 %s, %s, %s, %s" % (R0r, T0, S977, R0r))
    ins.append("v_lshl_add_u32 %s, %s, 3, %s" % (H(1), T0, H(1)))
    return ins


def emit(name, square, prod2=False, addend=False):
    EXTRA["prod2"], EXTRA["addend"] = prod2, addend
    ins = body_ilp(square) if ILP else body(square)
    nm = sum(1 for x in ins if x.startswith("v_mad_u64"))
    print("// %s: %d v_mad_u64_u32 + %d other instructions" % (name, nm, len(ins) - nm))
    print("#define %s \\" % name)
    for x in ins:
        print('  "%s\\n\\t" \\' % x)
    print('  ""')
    print()


print("// GENERATED by tools/gen_fe_asm.py -- do not edit.  See that file for the operand map.")
print("#define LAMD_FE_ASM_HI \"{v[%d:%d]}\"" % (HI, HI + 1))
print("#define LAMD_FE_ASM_LO \"{v[%d:%d]}\"" % (LO, LO + 1))
if SCARRY:
    print("#define LAMD_FE_ASM_CLOBBER \"vcc\", \"s20\", \"s21\"")
if ILP:
    print("// the latency schedule's own accumulators: eight more pinned outputs (fe.h appends them to LAMD_FE_ASM_DECL / LAMD_FE_ASM_OUT)")
    print("#define LAMD_FE_ASM_EXTRA_DECL u64 " + ", ".join("l%d" % k for k in range(8)) + ";")
    print("#define LAMD_FE_ASM_EXTRA_OUT , " + ", ".join('"=&{v[%d:%d]}"(l%d)' % (LK(k), LK(k) + 1, k) for k in range(8)))
emit("LAMD_FE_MUL_ASM", False)
emit("LAMD_FE_SQR_ASM", True)
emit("LAMD_FE_MULADD_ASM", False, addend=True)   # a*b + e      (%34.. = e[0..8])
emit("LAMD_FE_SQRADD_ASM", True, addend=True)    # a^2 + e
emit("LAMD_FE_MUL2_ASM", False, prod2=True)      # a*b + c*d    (%34.. = c[0..8], %43.. = d[0..8])

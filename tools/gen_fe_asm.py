#!/usr/bin/env python3
"""Writes lightning_amd/csrc/fe_asm.inc: the two multiply-add chains of fe_mul / fe_sqr (fe.h, LAMD_FE_COLUMNS) as ONE
inline-asm statement each.  Same arithmetic, same order as the C macro; the point of the single statement is that hipcc's
hazard recogniser, which assumes a dst-forwarding hazard after every opaque inline-asm statement, cannot put s_nop between
the 98 dependent v_mad_u64_u32 any more, and that the instruction order is ours.

The two 64-bit accumulators are pinned to fixed VGPR pairs because an inline-asm operand cannot name the halves of a 64-bit
register operand (the low half feeds v_and_b32).  The high chain (columns 9..16, 36 multiply-adds) and the low chain
(columns 0..8 + the folds, 61) are independent except that low column k needs h[k]; the high chain runs one column ahead.
Operands: %0..%7 r[0..7] | %8..%10 three rotating h registers | %11 HI (pinned) | %12 LO (pinned) | %13..%21 a[0..8] |
%22..%30 b[0..8] (fe_sqr: the doubled limbs d[0..8]) | %31 R0 | %32 256.
usage: tools/gen_fe_asm.py > lightning_amd/csrc/fe_asm.inc"""
import sys
HI, LO = 20, 22                      # v[20:21], v[22:23]
# --interleave alternates the two chains inside a step.  Measured on MI355X (three waves per SIMD hide the dependent
# latency anyway): same throughput, but the plain order is 5 % faster for a single latency-bound wave (484-row batch
# 1.14 ms vs 1.21 ms), so the plain order is what ships.
INTERLEAVE = "--interleave" in sys.argv
# --carry=sgpr: the (dead) carry-out of the two chains goes to two SGPR pairs instead of VCC (two extra outputs %13, %14; inputs move up by two).
#   Measured (tools/fe_bench.hip): v_mad_u64_u32 with a VCC carry-out saturates at 2.9e13 lane-ops/s whatever the occupancy, with an SGPR pair
#   it reaches 3.2e13 at 4 waves per SIMD and 3.7e13 at 8.
# --mask=sgpr: the 29-bit mask comes from an SGPR input (last operand) instead of a 32-bit literal (4-byte shorter encoding)
# --shift=alignbit: the 64-bit right shift as v_alignbit_b32 + v_lshrrev_b32
CARRY_SGPR = "--carry=sgpr" in sys.argv
MASK_SGPR = "--mask=sgpr" in sys.argv
SHIFT_ALIGN = "--shift=alignbit" in sys.argv
NX = 2 if CARRY_SGPR else 0          # extra outputs in front of the inputs
R = lambda k: "%%%d" % k             # r[k]
T = lambda k: "%%%d" % (8 + k % 3)   # h[k] lives in rotating temp k mod 3
A = lambda i: "%%%d" % (13 + NX + i)
B = lambda i: "%%%d" % (22 + NX + i)
SR0, S256 = "%%%d" % (31 + NX), "%%%d" % (32 + NX)
C2 = lambda i: "%%%d" % (33 + NX + i)   # second product's left operand (LAMD_FE_MUL2_ASM) / the addend limbs (..._ADD_ASM)
D2 = lambda i: "%%%d" % (42 + NX + i)   # second product's right operand
assert not MASK_SGPR or True
MASK = "%%%d" % (33 + NX) if MASK_SGPR else "0x1fffffff"   # (--mask=sgpr is only generated for the plain multiply / square)
CY = {True: "%13" if CARRY_SGPR else "vcc", False: "%14" if CARRY_SGPR else "vcc"}   # keyed by "is the high chain"
hi, lo = "v[%d:%d]" % (HI, HI + 1), "v[%d:%d]" % (LO, LO + 1)
hil, lol = "v%d" % HI, "v%d" % LO


EXTRA = {"prod2": False, "addend": False}   # set per emitted macro


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
        out.append("v_mad_u64_u32 %s, %s, %s, %s, %s" % (acc, CY[acc == hi], x, y, "0" if first and not out else acc))
    if EXTRA["prod2"]:      # a second full product c*d accumulated into the same columns: one reduction for a*b + c*d
        for i in range(9):
            j = k - i
            if 0 <= j <= 8:
                out.append("v_mad_u64_u32 %s, %s, %s, %s, %s" % (acc, CY[acc == hi], C2(i), D2(j), acc))
    if EXTRA["addend"] and k <= 8:   # + e: limb k of a lazy field element joins column k (times the inline constant 1)
        out.append("v_mad_u64_u32 %s, %s, %s, 1, %s" % (acc, CY[acc == hi], C2(k), acc))
    return out


def extract(dst, acc):
    """dst = acc & (2^29 - 1); acc >>= 29"""
    accl, acch = ("v%d" % HI, "v%d" % (HI + 1)) if acc == hi else ("v%d" % LO, "v%d" % (LO + 1))
    out = ["v_and_b32 %s, %s, %s" % (dst, MASK, accl)]
    if SHIFT_ALIGN:
        out += ["v_alignbit_b32 %s, %s, %s, 29" % (accl, acch, accl), "v_lshrrev_b32 %s, 29, %s" % (acch, acch)]
    else:
        out.append("v_lshrrev_b64 %s, 29, %s" % (acc, acc))
    return out


def body(square):
    # per step: (high-chain instructions, low-chain instructions); step 0 is the head (column 9 alone)
    steps = [(column(hi, 9, square, True) + extract(T(0), hi), [])]
    for k in range(8):
        h = []
        if k < 7:
            h = column(hi, 10 + k, square, False)
            h += extract(T(k + 1), hi)
        l = column(lo, k, square, k == 0)
        l.append("v_mad_u64_u32 %s, %s, %s, %s, %s" % (lo, CY[False], T(k), SR0, lo))
        if k > 0:
            l.append("v_mad_u64_u32 %s, %s, %s, %s, %s" % (lo, CY[False], T(k - 1), S256, lo))
        l += extract(R(k), lo)
        steps.append((h, l))
    steps.append(([], column(lo, 8, square, False) + ["v_mad_u64_u32 %s, %s, %s, %s, %s" % (lo, CY[False], T(7), S256, lo)]))
    ins = []
    for h, l in steps:
        # within a step the two lists are independent (h writes T(k+1), l reads T(k), T(k-1)): alternate them
        i = j = 0
        while i < len(h) or j < len(l):
            # keep the remaining lengths proportional so neither list is left alone at the end of the step
            take_h = i < len(h) and (not INTERLEAVE or j >= len(l) or (len(h) - i) * len(l) >= (len(l) - j) * len(h))
            if take_h:
                ins.append(h[i]); i += 1
            else:
                ins.append(l[j]); j += 1
    return ins


def emit(name, square, prod2=False, addend=False):
    EXTRA["prod2"], EXTRA["addend"] = prod2, addend
    ins = body(square)
    nm = sum(1 for x in ins if x.startswith("v_mad"))
    print("// %s: %d v_mad_u64_u32 + %d other instructions" % (name, nm, len(ins) - nm))
    print("#define %s \\" % name)
    for x in ins:
        print('  "%s\\n\\t" \\' % x)
    print('  ""')
    print()


print("// GENERATED by tools/gen_fe_asm.py -- do not edit.  See that file for the operand map.")
print("#define LAMD_FE_ASM_HI \"{v[%d:%d]}\"" % (HI, HI + 1))
print("#define LAMD_FE_ASM_LO \"{v[%d:%d]}\"" % (LO, LO + 1))
print("// extra operands of the asm statements in fe.h (cy0, cy1: dead carry-outs; the 29-bit mask)")
print("#define LAMD_FE_ASM_EXTRA_OUT %s" % (', "=&s"(cy0), "=&s"(cy1)' if CARRY_SGPR else ""))
print("#define LAMD_FE_ASM_EXTRA_IN %s" % (', "s"(FE_M29)' if MASK_SGPR else ""))
print("#define LAMD_FE_ASM_CLOBBER %s" % ("" if CARRY_SGPR else '"vcc"'))
emit("LAMD_FE_MUL_ASM", False)
emit("LAMD_FE_SQR_ASM", True)
if not MASK_SGPR:
    emit("LAMD_FE_MULADD_ASM", False, addend=True)   # a*b + e      (%33.. = e[0..8])
    emit("LAMD_FE_SQRADD_ASM", True, addend=True)    # a^2 + e
    emit("LAMD_FE_MUL2_ASM", False, prod2=True)      # a*b + c*d    (%33.. = c[0..8], %42.. = d[0..8])

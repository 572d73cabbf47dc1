// secp256k1 base field, p = 2^256 - 2^32 - 977, for one signature per wavefront lane.
//
// Representation: 9 limbs of 29 bits in u32 registers, LAZY: limbs may exceed 29 bits.
// A value of "magnitude" m has limbs[0..7] <= m*LIM29 and limb[8] <= m*LIM24 (m <= 7 so
// nothing overflows a u32; LIM29 = 2^29 + 2^21: a product leaves its last wrap-around on limb 0
// unpropagated, see LAMD_FE_CHAINS).  Why this shape (measured, profiles/r01_microbench_valu_rates.txt):
// on gfx950 v_mad_u64_u32 runs at the same half rate as v_add_co/v_addc (and carry chains
// additionally pay the VALU-writes-VCC -> VALU-reads-VCC 2-wait-state hazard), so the cheap
// resource is the 64-bit accumulate inside the multiplier, not the adder.  29-bit limbs let a
// whole product column (<= 9 partial products of <= 2^58..2^61) accumulate in one 64-bit
// register with no carry handling at all: 81 mads per multiply (45 per square), additions are
// 9 independent v_add_u32, and carries are resolved once per multiply.
//
// Replaces (together with scalar.h/group.h): the field layer of libsecp256k1 that
// bitcoin/signature.c:188,425 reaches through secp256k1_ecdsa_verify / schnorrsig_verify.
#pragma once
#include "lamd_common.h"
#if defined(__HIP_DEVICE_COMPILE__) && !defined(LAMD_FE_NO_ASM_BLOCK)
#define LAMD_FE_ASM_BLOCK 1
#if defined(LAMD_FE_ASM_INC)  // tools/fe_bench.hip times alternative schedules of the generator
#include LAMD_FE_ASM_INC
#else
#include "fe_asm.inc"
#endif
#endif

namespace lamd {

struct fe {
  u32 n[9];
#if defined(LAMD_CHECK_MAG)
  int mag;
#endif
};

constexpr u32 FE_M29 = 0x1FFFFFFFu;
constexpr u32 FE_M24 = 0x00FFFFFFu;
constexpr u32 FE_LIM29 = (1u << 29) + (1u << 21);
constexpr u32 FE_LIM24 = (1u << 24) + (1u << 13);
// p in 29-bit limbs
constexpr u32 FE_P0 = 0x1FFFFC2Fu, FE_P1 = 0x1FFFFFF7u, FE_PM = 0x1FFFFFFFu, FE_P8 = 0x00FFFFFFu;
// 2^261 mod p = 2^37 + 31264 = 31264 + 256 * 2^29
constexpr u32 FE_R0 = 31264u;
constexpr int FE_R1_SHIFT = 8;

#if defined(LAMD_CHECK_MAG)
#define FE_SETMAG(r, m) ((r).mag = (m))
#define FE_MAG(a) ((a).mag)
LAMD_HD void fe_verify(const fe &a) {
  LAMD_ASSERT(a.mag >= 0 && a.mag <= 7);
  for (int i = 0; i < 8; i++) LAMD_ASSERT((u64)a.n[i] <= (u64)a.mag * FE_LIM29);
  LAMD_ASSERT((u64)a.n[8] <= (u64)a.mag * FE_LIM24);
}
#else
#define FE_SETMAG(r, m) ((void)0)
#define FE_MAG(a) 0
LAMD_HD void fe_verify(const fe &) {}
#endif

LAMD_HD fe fe_zero() {
  fe r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.n[i] = 0;
  FE_SETMAG(r, 0);
  return r;
}
LAMD_HD fe fe_set_int(u32 v) {  // v < 2^29
  fe r = fe_zero();
  r.n[0] = v;
  FE_SETMAG(r, 1);
  return r;
}

// r = a + b (lazy)
LAMD_HD fe fe_add(const fe &a, const fe &b) {
  fe r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.n[i] = a.n[i] + b.n[i];
  FE_SETMAG(r, FE_MAG(a) + FE_MAG(b));
  fe_verify(r);
  return r;
}
// r = -a for a of magnitude <= m: (m+1)*p - a, limb-wise non-negative; magnitude m+1
LAMD_HD fe fe_neg(const fe &a, int m) {
  LAMD_ASSERT(FE_MAG(a) <= m && m + 1 <= 7);
  fe r;
  const u32 k = (u32)(m + 1);
  r.n[0] = k * FE_P0 - a.n[0];
  r.n[1] = k * FE_P1 - a.n[1];
#pragma unroll
  for (int i = 2; i < 8; i++) r.n[i] = k * FE_PM - a.n[i];
  r.n[8] = k * FE_P8 - a.n[8];
  FE_SETMAG(r, m + 1);
  fe_verify(r);
  return r;
}
// r = a - b, b of magnitude <= mb; magnitude mag(a) + mb + 1
LAMD_HD fe fe_sub(const fe &a, const fe &b, int mb) { return fe_add(a, fe_neg(b, mb)); }

LAMD_HD fe fe_mul_int(const fe &a, u32 k) {
  fe r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.n[i] = a.n[i] * k;
  FE_SETMAG(r, FE_MAG(a) * (int)k);
  fe_verify(r);
  return r;
}

// Magnitude -> 1 with fully parallel (non-rippling) carries: every limb keeps its low bits and
// receives its neighbour's overflow (<= 7), the 2^256 overflow of the top limb folds back
// through 2^256 = 2^32 + 977 (mod p).  ~30 independent VALU ops, no carry chain.
LAMD_HD fe fe_norm_weak(const fe &a) {
  fe r;
  const u32 e = a.n[8] >> 24;
  r.n[0] = (a.n[0] & FE_M29) + e * 977u;
  r.n[1] = (a.n[1] & FE_M29) + (a.n[0] >> 29) + (e << 3);
#pragma unroll
  for (int i = 2; i < 8; i++) r.n[i] = (a.n[i] & FE_M29) + (a.n[i - 1] >> 29);
  r.n[8] = (a.n[8] & FE_M24) + (a.n[7] >> 29);
  FE_SETMAG(r, 1);
  fe_verify(r);
  return r;
}

// Exact carry propagation: limbs[0..7] < 2^29, limb[8] <= 2^24 (bit 24 may be set once);
// value < 2^256 + 2^41.  Input any magnitude <= 7.
LAMD_HD fe fe_carry(const fe &a) {
  fe r;
  u32 e = a.n[8] >> 24;
  u32 t = a.n[0] + e * 977u;  // < 2^32
  r.n[0] = t & FE_M29;
  u32 c = t >> 29;
  t = a.n[1] + (e << 3) + c;
  r.n[1] = t & FE_M29;
  c = t >> 29;
#pragma unroll
  for (int i = 2; i < 8; i++) {
    t = a.n[i] + c;
    r.n[i] = t & FE_M29;
    c = t >> 29;
  }
  r.n[8] = (a.n[8] & FE_M24) + c;
  FE_SETMAG(r, 1);
  return r;
}

// canonical representative in [0, p)
LAMD_HD fe fe_normalize(const fe &a) {
  fe r = fe_carry(fe_carry(a));  // second pass clears a possible bit 24; now value < 2^256
  // t = r + (2^32 + 977); if that reaches 2^256 then r >= p and the answer is t - 2^256
  fe t;
  u32 v = r.n[0] + 977u;
  t.n[0] = v & FE_M29;
  u32 c = v >> 29;
  v = r.n[1] + 8u + c;
  t.n[1] = v & FE_M29;
  c = v >> 29;
#pragma unroll
  for (int i = 2; i < 8; i++) {
    v = r.n[i] + c;
    t.n[i] = v & FE_M29;
    c = v >> 29;
  }
  v = r.n[8] + c;
  t.n[8] = v & FE_M24;
  const bool ge = (v >> 24) != 0;
#pragma unroll
  for (int i = 0; i < 9; i++) r.n[i] = ge ? t.n[i] : r.n[i];
  FE_SETMAG(r, 1);
  return r;
}

// does a (any magnitude <= 7) represent 0 mod p?
LAMD_HD bool fe_is_zero(const fe &a) {
  const fe r = fe_carry(a);  // value in [0, 2^256 + 2^41): zero iff 0 or p
  u32 z0 = r.n[0] | r.n[1] | r.n[8];
  u32 z1 = (r.n[0] ^ FE_P0) | (r.n[1] ^ FE_P1) | (r.n[8] ^ FE_P8);
#pragma unroll
  for (int i = 2; i < 8; i++) {
    z0 |= r.n[i];
    z1 |= r.n[i] ^ FE_PM;
  }
  return (z0 == 0) | (z1 == 0);
}
LAMD_HD bool fe_equal(const fe &a, const fe &b, int mb) { return fe_is_zero(fe_sub(a, b, mb)); }
LAMD_HD bool fe_is_odd_canonical(const fe &a) { return a.n[0] & 1; }  // a already fe_normalize()d

LAMD_HD fe fe_select(bool take_a, const fe &a, const fe &b) {
  fe r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.n[i] = take_a ? a.n[i] : b.n[i];
  FE_SETMAG(r, FE_MAG(a) > FE_MAG(b) ? FE_MAG(a) : FE_MAG(b));
  return r;
}

// acc += a*b as ONE v_mad_u64_u32 whose 64-bit addend is the running accumulator.  Written as inline asm on
// the device because hipcc re-associates a C sum of products into "independent column sum, then a separate
// 64-bit add of the carry" (v_lshl_add_u64, another half-rate issue slot per column); the asm pins the chain.
// (No builtin exists for this instruction; the carry-out SGPR pair is a dead output, exactly as in hipcc's own
// code.)  Host builds use the plain C expression.
// CH: which of fe_mul's two interleaved chains the multiply-add belongs to.  The chains discard their carry-out into
// different scalar registers (an SGPR pair / VCC): with one shared register hipcc's hazard recogniser sees every
// inline-asm statement re-define a register the previous one defined and, assuming the worst about opaque asm (a
// dst-forwarding hazard that v_mad_u64_u32 does not have), puts an s_nop between every two of them.
template <int CH = 0>
LAMD_HD void fe_mac(u64 &acc, u32 a, u32 b) {
#if defined(__HIP_DEVICE_COMPILE__)
  if (CH == 0) {
    u64 cy;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(acc), "=s"(cy) : "v"(a), "v"(b));
  } else {
    asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b) : "vcc");
  }
#else
  acc += (u64)a * b;
#endif
}
// same with a small compile-time constant multiplier held in an SGPR
LAMD_HD void fe_mac_k(u64 &acc, u32 a, u32 k) {
#if defined(__HIP_DEVICE_COMPILE__)
  u64 cy;
  asm("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(acc), "=s"(cy) : "v"(a), "s"(k));
#else
  acc += (u64)a * k;
#endif
}

// Multiplication / squaring with RUNNING carries: a column's 64-bit sum, shifted right by 29, is the addend of
// the next column's first multiply-add, so a carry costs one 64-bit shift and one mask per column and no
// additions.  Two chains, one after the other (round 4: the wrap-around of the top is known BEFORE the low columns run,
// so nothing has to be carried through finished limbs afterwards):
//   column 8    first; its low 29 bits h8 are set aside, its overflow (the largest of all: ~2^33) is the high chain's first carry
//   high chain  columns 9..15 -> h[1..7] (29 bits each; h[j] has weight 2^(29(8+j))); column 16 is not cut: V = Vlo + 2^32 * Vhi
//               (Vlo a full 32-bit word, Vhi < 2^19 -- the two registers of the accumulator as they are)
//   fold        2^261 = R0 + 2^8 * 2^29 (mod p): h[j] joins column j-1 as R0*h[j] and column j as 2^8*h[j]; Vlo does the same
//               from position 16.  Vhi * 2^32 * 2^(29*16) = (8*Vhi) * 2^(29*17) would join column 8 as R0*8*Vhi and column 9 as
//               2^8*8*Vhi -- the latter simply makes h[1] (weight 2^261) larger: h[1] += Vhi << 11.  What column 8 holds before
//               the low chain reaches it, P8 = 8*R0*Vhi + 2^8*Vlo + h8 (< 2^41), is split at bit 32: P8.hi * 2^32 * 2^232 =
//               8*P8.hi * 2^261 also joins h[1] (now < 2^31, still a 32-bit factor); P8.lo waits for the end
//   low chain   columns 0..7 with the folds -> r[0..7]; its last carry c7 (< 2^35)
//   end         column 8 = c7 + P8.lo (< 2^36): r[8] = its low 24 bits, the rest e2 (<= 2064) is a multiple of
//               2^256 = 977 + 8 * 2^29: r[0] += 977*e2 (< 2^21: stays on the limb, hence FE_LIM29), r[1] += 8*e2
// PROD(k, acc, ch) must add the partial products of column k into acc with fe_mac<ch>().
// On the device the whole operation is one hand-ordered asm statement (LAMD_FE_ASM_BLOCK, generated into fe_asm.inc by
// tools/gen_fe_asm.py from exactly this schedule); the C form below is what the host build checks and what
// -DLAMD_FE_NO_ASM_BLOCK falls back to.
// Column budget (magnitude products summing to M <= 7, L = FE_LIM29): a column is at most 8 full products + one with a
// 24-bit limb < 8.04 * M * L^2 < 56.8 * 2^58, the folds (< 2^47) and carries (< 2^35) are noise: < 2^64.
#define LAMD_FE_CHAINS(PROD)                                                                        \
  u32 h[8];                                                                                         \
  fe r;                                                                                             \
  u64 hi = 0, lo = 0;                                                                               \
  PROD(8, hi, 1);                                                                                   \
  const u32 h8 = (u32)hi & FE_M29;                                                                  \
  hi >>= 29;                                                                                        \
  _Pragma("unroll") for (int j = 1; j < 8; j++) {                                                   \
    PROD(8 + j, hi, 1);                                                                             \
    h[j] = (u32)hi & FE_M29;                                                                        \
    hi >>= 29;                                                                                      \
  }                                                                                                 \
  PROD(16, hi, 1);                                                                                  \
  const u32 vlo = (u32)hi, vhi = (u32)(hi >> 32);                                                   \
  LAMD_ASSERT(vhi < (1u << 19));                          /* column 16 is one product of 24-bit limbs */ \
  h[1] += vhi << (3 + FE_R1_SHIFT);                                                                 \
  u64 p8 = (u64)vhi * (8u * FE_R0) + h8;                                                            \
  p8 += (u64)vlo << FE_R1_SHIFT;                                                                    \
  LAMD_ASSERT((p8 >> 41) == 0);                                                                     \
  h[1] += (u32)(p8 >> 32) << 3;                                                                     \
  _Pragma("unroll") for (int k = 0; k < 8; k++) {                                                   \
    PROD(k, lo, 0);                                                                                 \
    fe_mac_k(lo, k < 7 ? h[k < 7 ? k + 1 : 1] : vlo, FE_R0);                                        \
    if (k > 0) fe_mac_k(lo, h[k], 1u << FE_R1_SHIFT);                                               \
    r.n[k] = (u32)lo & FE_M29;                                                                      \
    lo >>= 29;                                                                                      \
  }                                                                                                 \
  lo += (u32)p8;                                                                                    \
  r.n[8] = (u32)lo & FE_M24;                                                                        \
  const u32 e2 = (u32)(lo >> 24);                                                                   \
  LAMD_ASSERT((lo >> 24) <= 2064);                                                                  \
  r.n[0] += e2 * 977u;                                                                              \
  r.n[1] += e2 << 3;                                                                                \
  FE_SETMAG(r, 1);                                                                                  \
  fe_verify(r);                                                                                     \
  return r;

// the asm statements' operand lists (tools/gen_fe_asm.py: operand map) and what follows them
// (the latency schedule, fe_asm_ilp.inc, brings eight more pinned accumulators: LAMD_FE_ASM_EXTRA_DECL / _OUT)
#if !defined(LAMD_FE_ASM_CLOBBER)   // (a generated schedule that parks the multiply-adds' unused carry-out in an SGPR pair names the pair here)
#define LAMD_FE_ASM_CLOBBER "vcc"
#endif
#if !defined(LAMD_FE_ASM_EXTRA_DECL)
#define LAMD_FE_ASM_EXTRA_DECL
#define LAMD_FE_ASM_EXTRA_OUT
#endif
#define LAMD_FE_ASM_DECL                                                                            \
  fe r;                                                                                             \
  u64 hi, lo;                                                                                       \
  u32 e2;                                                                                           \
  LAMD_FE_ASM_EXTRA_DECL
#define LAMD_FE_ASM_OUT                                                                             \
  "=&v"(r.n[0]), "=&v"(e2), "=&v"(r.n[1]), "=&v"(r.n[2]), "=&v"(r.n[3]), "=&v"(r.n[4]), "=&v"(r.n[5]), "=&v"(r.n[6]),       \
      "=&v"(r.n[7]), "=&v"(r.n[8]), "=&" LAMD_FE_ASM_HI(hi), "=&" LAMD_FE_ASM_LO(lo) LAMD_FE_ASM_EXTRA_OUT
#define LAMD_FE_ASM_K "s"(FE_R0), "s"(1u << FE_R1_SHIFT), "s"(977u), "s"(8u * FE_R0)
#define LAMD_FE_ASM_DONE                                                                            \
  (void)e2;                                                                                         \
  FE_SETMAG(r, 1);                                                                                  \
  return r;

template <int CH>
LAMD_HD void fe_mul_col(const fe &a, const fe &b, int k, u64 &acc) {
#pragma unroll
  for (int i = 0; i < 9; i++) {
    const int j = k - i;
    if (j < 0 || j > 8) continue;
    fe_mac<CH>(acc, a.n[i], b.n[j]);
  }
}
template <int CH>
LAMD_HD void fe_sqr_col(const fe &a, const u32 d[9], int k, u64 &acc) {
#pragma unroll
  for (int i = 0; i < 9; i++) {
    const int j = k - i;
    if (j < 0 || j > 8 || i > j) continue;
    fe_mac<CH>(acc, (i == j) ? a.n[i] : d[i], a.n[j]);
  }
}

#define LAMD_FE_ASM_IO9(x) "v"(x.n[0]), "v"(x.n[1]), "v"(x.n[2]), "v"(x.n[3]), "v"(x.n[4]), "v"(x.n[5]), "v"(x.n[6]), "v"(x.n[7]), "v"(x.n[8])

// r = a*b; requires mag(a)*mag(b) <= 7
LAMD_HD fe fe_mul(const fe &a, const fe &b) {
  LAMD_ASSERT(FE_MAG(a) * FE_MAG(b) <= 7);
#if defined(LAMD_FE_ASM_BLOCK)
  // the whole multiplication as ONE asm statement (fe_asm.inc): no compiler-inserted s_nop between the dependent multiply-adds
  LAMD_FE_ASM_DECL
  asm(LAMD_FE_MUL_ASM : LAMD_FE_ASM_OUT : LAMD_FE_ASM_IO9(a), LAMD_FE_ASM_IO9(b), LAMD_FE_ASM_K : LAMD_FE_ASM_CLOBBER);
  LAMD_FE_ASM_DONE
#else
#define LAMD_P(k, acc, ch) fe_mul_col<ch>(a, b, k, acc)
  LAMD_FE_CHAINS(LAMD_P)
#undef LAMD_P
#endif
}

// r = a^2; requires mag(a) <= 2
LAMD_HD fe fe_sqr(const fe &a) {
  LAMD_ASSERT(FE_MAG(a) <= 2);
  fe d;
#pragma unroll
  for (int i = 0; i < 9; i++) d.n[i] = a.n[i] << 1;
#if defined(LAMD_FE_ASM_BLOCK)
  LAMD_FE_ASM_DECL
  asm(LAMD_FE_SQR_ASM : LAMD_FE_ASM_OUT : LAMD_FE_ASM_IO9(a), LAMD_FE_ASM_IO9(d), LAMD_FE_ASM_K : LAMD_FE_ASM_CLOBBER);
  LAMD_FE_ASM_DONE
#else
#define LAMD_P(k, acc, ch) fe_sqr_col<ch>(a, d.n, k, acc)
  LAMD_FE_CHAINS(LAMD_P)
#undef LAMD_P
#endif
}

// ---- fused forms: ONE reduction (fold, carries, wrap-around) for a product plus something else.  The group law is full of
// "product minus value" and "product minus product" (U2 - X1, S2 - Y1, R^2 - H^3 - 2V, R*(V - X3) - Y1*H^3): as separate
// operations each costs a lazy negation, an addition and a fe_norm_weak (~48 instructions) or a whole second reduction
// (~75); here the extra terms join the column accumulators of the multiply -- limb k of the addend as one more multiply-add
// (times 1) in column k, a second product as 81 more multiply-adds -- and the result comes out carried (magnitude 1).
// Column budget as for fe_mul: mag(a)*mag(b) [+ mag(c)*mag(d)] <= 7; an addend of magnitude <= 7 is noise against that.
// r = a*b + e
LAMD_HD fe fe_mul_add(const fe &a, const fe &b, const fe &ad) {
  LAMD_ASSERT(FE_MAG(a) * FE_MAG(b) <= 7 && FE_MAG(ad) <= 7);
#if defined(LAMD_FE_ASM_BLOCK)
  LAMD_FE_ASM_DECL
  asm(LAMD_FE_MULADD_ASM : LAMD_FE_ASM_OUT : LAMD_FE_ASM_IO9(a), LAMD_FE_ASM_IO9(b), LAMD_FE_ASM_K, LAMD_FE_ASM_IO9(ad) : LAMD_FE_ASM_CLOBBER);
  LAMD_FE_ASM_DONE
#else
#define LAMD_P(k, acc, ch) do { fe_mul_col<ch>(a, b, k, acc); if ((k) < 9) acc += ad.n[(k) < 9 ? (k) : 0]; } while (0)
  LAMD_FE_CHAINS(LAMD_P)
#undef LAMD_P
#endif
}
// r = a^2 + e; mag(a) <= 2
LAMD_HD fe fe_sqr_add(const fe &a, const fe &ad) {
  LAMD_ASSERT(FE_MAG(a) <= 2 && FE_MAG(ad) <= 7);
  fe d;
#pragma unroll
  for (int i = 0; i < 9; i++) d.n[i] = a.n[i] << 1;
#if defined(LAMD_FE_ASM_BLOCK)
  LAMD_FE_ASM_DECL
  asm(LAMD_FE_SQRADD_ASM : LAMD_FE_ASM_OUT : LAMD_FE_ASM_IO9(a), LAMD_FE_ASM_IO9(d), LAMD_FE_ASM_K, LAMD_FE_ASM_IO9(ad) : LAMD_FE_ASM_CLOBBER);
  LAMD_FE_ASM_DONE
#else
#define LAMD_P(k, acc, ch) do { fe_sqr_col<ch>(a, d.n, k, acc); if ((k) < 9) acc += ad.n[(k) < 9 ? (k) : 0]; } while (0)
  LAMD_FE_CHAINS(LAMD_P)
#undef LAMD_P
#endif
}
// r = a*b + c*d
LAMD_HD fe fe_mul2(const fe &a, const fe &b, const fe &c, const fe &d) {
  LAMD_ASSERT(FE_MAG(a) * FE_MAG(b) + FE_MAG(c) * FE_MAG(d) <= 7);
#if defined(LAMD_FE_ASM_BLOCK)
  LAMD_FE_ASM_DECL
  asm(LAMD_FE_MUL2_ASM : LAMD_FE_ASM_OUT : LAMD_FE_ASM_IO9(a), LAMD_FE_ASM_IO9(b), LAMD_FE_ASM_K, LAMD_FE_ASM_IO9(c), LAMD_FE_ASM_IO9(d) : LAMD_FE_ASM_CLOBBER);
  LAMD_FE_ASM_DONE
#else
#define LAMD_P(k, acc, ch) do { fe_mul_col<ch>(a, b, k, acc); fe_mul_col<ch>(c, d, k, acc); } while (0)
  LAMD_FE_CHAINS(LAMD_P)
#undef LAMD_P
#endif
}

LAMD_HD fe fe_sqr_n(fe a, int n) {
#pragma unroll 1
  for (int i = 0; i < n; i++) a = fe_sqr(a);
  return a;
}

// a^(2^223 - 1) and a^(2^22-1), a^(2^2-1): shared prefix of the inversion and square-root chains
struct fe_chain { fe x2, x22, x223; };
LAMD_HD fe_chain fe_pow_chain(const fe &a) {  // a magnitude 1
  fe_chain o;
  const fe x2 = fe_mul(fe_sqr(a), a);
  const fe x3 = fe_mul(fe_sqr(x2), a);
  const fe x6 = fe_mul(fe_sqr_n(x3, 3), x3);
  const fe x9 = fe_mul(fe_sqr_n(x6, 3), x3);
  const fe x11 = fe_mul(fe_sqr_n(x9, 2), x2);
  const fe x22 = fe_mul(fe_sqr_n(x11, 11), x11);
  const fe x44 = fe_mul(fe_sqr_n(x22, 22), x22);
  const fe x88 = fe_mul(fe_sqr_n(x44, 44), x44);
  const fe x176 = fe_mul(fe_sqr_n(x88, 88), x88);
  const fe x220 = fe_mul(fe_sqr_n(x176, 44), x44);
  o.x223 = fe_mul(fe_sqr_n(x220, 3), x3);
  o.x2 = x2;
  o.x22 = x22;
  return o;
}
// a^(p-2): 255 squarings + 15 multiplications
LAMD_HD fe fe_inv(const fe &a) {
  const fe_chain ch = fe_pow_chain(a);
  fe t = fe_mul(fe_sqr_n(ch.x223, 23), ch.x22);
  t = fe_mul(fe_sqr_n(t, 5), a);
  t = fe_mul(fe_sqr_n(t, 3), ch.x2);
  t = fe_mul(fe_sqr_n(t, 2), a);
  return t;
}
// candidate square root a^((p+1)/4) (p = 3 mod 4); caller checks r^2 == a
LAMD_HD fe fe_sqrt_candidate(const fe &a) {
  const fe_chain ch = fe_pow_chain(a);
  fe t = fe_mul(fe_sqr_n(ch.x223, 23), ch.x22);
  t = fe_mul(fe_sqr_n(t, 6), ch.x2);
  return fe_sqr_n(t, 2);
}

// ---- conversions.  w[0..7]: 256-bit value as little-endian 32-bit words.
LAMD_HD fe fe_from_words(const u32 w[8]) {
  fe r;
  r.n[0] = w[0] & FE_M29;
  r.n[1] = ((w[0] >> 29) | (w[1] << 3)) & FE_M29;
  r.n[2] = ((w[1] >> 26) | (w[2] << 6)) & FE_M29;
  r.n[3] = ((w[2] >> 23) | (w[3] << 9)) & FE_M29;
  r.n[4] = ((w[3] >> 20) | (w[4] << 12)) & FE_M29;
  r.n[5] = ((w[4] >> 17) | (w[5] << 15)) & FE_M29;
  r.n[6] = ((w[5] >> 14) | (w[6] << 18)) & FE_M29;
  r.n[7] = ((w[6] >> 11) | (w[7] << 21)) & FE_M29;
  r.n[8] = w[7] >> 8;
  FE_SETMAG(r, 1);
  return r;
}
// a must be canonical (fe_normalize) or at least exactly carried with value < 2^256
LAMD_HD void fe_to_words(u32 w[8], const fe &a) {
  w[0] = a.n[0] | (a.n[1] << 29);
  w[1] = (a.n[1] >> 3) | (a.n[2] << 26);
  w[2] = (a.n[2] >> 6) | (a.n[3] << 23);
  w[3] = (a.n[3] >> 9) | (a.n[4] << 20);
  w[4] = (a.n[4] >> 12) | (a.n[5] << 17);
  w[5] = (a.n[5] >> 15) | (a.n[6] << 14);
  w[6] = (a.n[6] >> 18) | (a.n[7] << 11);
  w[7] = (a.n[7] >> 21) | (a.n[8] << 8);
}
// is the 256-bit integer in w >= p ?  (p = FFFFFFFF x6 | FFFFFFFE | FFFFFC2F)
LAMD_HD bool words_ge_p(const u32 w[8]) {
  const u32 all1 = w[7] & w[6] & w[5] & w[4] & w[3] & w[2];
  const bool low_ge = (w[1] == 0xFFFFFFFFu) | ((w[1] == 0xFFFFFFFEu) & (w[0] >= 0xFFFFFC2Fu));
  return (all1 == 0xFFFFFFFFu) & low_ge;
}

}  // namespace lamd

// secp256k1 group law (y^2 = x^3 + 7) on top of fe.h, shaped for one signature per lane:
// branch-free main paths, magnitudes tracked so that every fe_mul/fe_sqr operand stays inside
// the 64-bit column budget, and the (adversarially reachable) degenerate cases of an addition
// -- P + P, P + (-P), infinity -- taken on a cold, wave-rarely-entered path.
//
// Semantics replaced: libsecp256k1's group/ecmult layer as reached from
// bitcoin/signature.c:188 (secp256k1_ecdsa_verify) and :425 (secp256k1_schnorrsig_verify).
#pragma once
#include "fe.h"

namespace lamd {

struct ge {  // affine, magnitude-1 coordinates
  fe x, y;
};
struct gej {  // Jacobian: x, y magnitude 1, z magnitude <= 2
  fe x, y, z;
  bool inf;
};

LAMD_HD gej gej_infinity() {
  gej r;
  r.x = fe_zero(); r.y = fe_zero(); r.z = fe_zero();
  FE_SETMAG(r.x, 1); FE_SETMAG(r.y, 1); FE_SETMAG(r.z, 1);
  r.inf = true;
  return r;
}
LAMD_HD gej gej_from_ge(const ge &a) {
  gej r;
  r.x = a.x; r.y = a.y; r.z = fe_set_int(1); r.inf = false;
  return r;
}

// y^2 == x^3 + 7 ?
LAMD_HD bool ge_on_curve(const ge &a) {
  const fe y2 = fe_sqr(a.y);
  const fe x3 = fe_mul(fe_sqr(a.x), a.x);
  return fe_equal(fe_add(x3, fe_set_int(7)), y2, 1);
}

// 3M + 4S.  secp256k1 has no point of order 2, so y = 0 cannot occur for a curve point.
LAMD_HD gej gej_double(const gej &a) {
  gej r;
  const fe yy = fe_sqr(a.y);
  const fe u = fe_mul_int(yy, 2);                 // 2Y^2        (2)
  const fe uu = fe_sqr(u);                         // 4Y^4        (1)
  const fe w = fe_mul(a.x, u);                     // 2XY^2       (1)
  const fe m = fe_norm_weak(fe_mul_int(fe_sqr(a.x), 3));  // 3X^2  (1)
  r.x = fe_sqr_add(m, fe_neg(fe_mul_int(w, 4), 4));                           // M^2 - 8XY^2: one reduction
  const fe t = fe_add(fe_mul_int(w, 2), fe_neg(r.x, 1));                      // 4XY^2 - X3  (4)
  r.y = fe_mul_add(m, t, fe_neg(fe_mul_int(uu, 2), 2));                       // M*t - 8Y^4: one reduction
  r.z = fe_mul_int(fe_mul(a.y, a.z), 2);                                      // 2YZ         (2)
  r.inf = a.inf;
  return r;
}

// r = a + b for Jacobian a (not infinity) and affine b, assuming a != +-b; 8M + 3S.
// *degenerate = (H == 0): a = +-b and r is garbage -- then *rr_out == 0 tells P + P from P + (-P) (tested by the
// caller on its cold path only); *h_out = H with Z3 = Z1*H.
LAMD_HD gej gej_add_ge_core(const gej &a, const ge &b, bool *degenerate, fe *h_out, fe *rr_out) {
  gej r;
  const fe zz = fe_sqr(a.z);
  const fe h = fe_mul_add(b.x, zz, fe_neg(a.x, 1));                  // U2 - X1, exactly carried
  const fe rr = fe_mul_add(b.y, fe_mul(a.z, zz), fe_neg(a.y, 1));    // S2 - Y1
  *degenerate = fe_is_zero(h);
  const fe hh = fe_sqr(h);
  const fe hhh = fe_mul(h, hh);
  const fe v = fe_mul(a.x, hh);
  r.x = fe_sqr_add(rr, fe_neg(fe_add(hhh, fe_mul_int(v, 2)), 3));   // R^2 - H^3 - 2V
  const fe t = fe_add(v, fe_neg(r.x, 1));  // (3)
  r.y = fe_mul2(rr, t, a.y, fe_neg(hhh, 1));                         // R*(V - X3) - Y1*H^3: two products, one reduction
  r.z = fe_mul(a.z, h);
  r.inf = false;
  *h_out = h;
  *rr_out = rr;
  return r;
}

// The same formula with nothing around it: no zero test of H, no infinity / skip handling.  For callers that detect
// every degenerate event afterwards: H == 0 (a = +-b) makes Z3 = Z1*H zero, and a zero Z stays zero through every later
// doubling (Z3 = 2*Y*Z) and addition (Z3 = Z*H), so one fe_is_zero(Z) at the end of a chain of these catches them all
// (p is prime: Z1*H = 0 only if a factor is).  b.y may have magnitude 2 (ge_neg_if_lazy).
LAMD_HD gej gej_add_ge_fast(const gej &a, const ge &b) {
  gej r;
  const fe zz = fe_sqr(a.z);
  const fe h = fe_mul_add(b.x, zz, fe_neg(a.x, 1));
  const fe rr = fe_mul_add(b.y, fe_mul(a.z, zz), fe_neg(a.y, 1));
  const fe hh = fe_sqr(h);
  const fe hhh = fe_mul(h, hh);
  const fe v = fe_mul(a.x, hh);
  r.x = fe_sqr_add(rr, fe_neg(fe_add(hhh, fe_mul_int(v, 2)), 3));
  const fe t = fe_add(v, fe_neg(r.x, 1));  // (3)
  r.y = fe_mul2(rr, t, a.y, fe_neg(hhh, 1));
  r.z = fe_mul(a.z, h);
  r.inf = false;
  return r;
}

// XYZZ coordinates (x = X/ZZ, y = Y/ZZZ with ZZ^3 = ZZZ^2) for a RUN of bare mixed additions with no doubling in between -- the eleven windows
// of u1*G at the end of a verification.  The Jacobian addition above recomputes Z^2 and Z^3 from Z in every step; here they are carried:
// ZZ3 = ZZ1*H^2, ZZZ3 = ZZZ1*H^3 (two multiplications where the Jacobian form has Z^2, Z*Z^2 and Z1*H: a squaring less per addition), and the
// acceptance test r*Z^2 == X has its Z^2 for free.  Entering the run costs what the first Jacobian addition pays anyway (Z^2, Z^3).  A doubling
// needs Z itself (Z3 = 2*Y*Z), which is why the comb columns stay Jacobian.  Degenerate events: H = 0 makes ZZ3 = 0, and a zero ZZ stays zero
// through every later addition of the run -- the caller's one test moves from Z to ZZ.
struct gexz {
  fe x, y, zz, zzz;  // all magnitude 1
};
LAMD_HD gexz gexz_from_gej(const gej &a) {
  gexz r;
  r.x = a.x;
  r.y = a.y;
  r.zz = fe_sqr(a.z);
  r.zzz = fe_mul(a.z, r.zz);
  return r;
}
LAMD_HD gexz gexz_add_ge_fast(const gexz &a, const ge &b) {
  gexz r;
  const fe h = fe_mul_add(b.x, a.zz, fe_neg(a.x, 1));
  const fe rr = fe_mul_add(b.y, a.zzz, fe_neg(a.y, 1));
  const fe hh = fe_sqr(h);
  const fe hhh = fe_mul(h, hh);
  const fe v = fe_mul(a.x, hh);
  r.x = fe_sqr_add(rr, fe_neg(fe_add(hhh, fe_mul_int(v, 2)), 3));
  const fe t = fe_add(v, fe_neg(r.x, 1));  // (3)
  r.y = fe_mul2(rr, t, a.y, fe_neg(hhh, 1));
  r.zz = fe_mul(a.zz, hh);
  r.zzz = fe_mul(a.zzz, hhh);
  return r;
}

LAMD_HD gej gej_select(bool take_a, const gej &a, const gej &b) {
  gej r;
  r.x = fe_select(take_a, a.x, b.x);
  r.y = fe_select(take_a, a.y, b.y);
  r.z = fe_select(take_a, a.z, b.z);
  r.inf = take_a ? a.inf : b.inf;
  return r;
}

// Complete mixed addition with a lane predicate: r = skip ? a : a + b.  The degenerate outcomes
// are handled under a divergent branch that is never taken on honest inputs.
LAMD_HD gej gej_add_ge(const gej &a, const ge &b, bool skip) {
  bool degenerate;
  fe h, rr;
  gej r = gej_add_ge_core(a, b, &degenerate, &h, &rr);
  if (__builtin_expect(!skip && !a.inf && degenerate, 0)) {
    if (fe_is_zero(rr)) {
      r = gej_double(gej_from_ge(b));
    } else {
      r = gej_infinity();
    }
  }
  const gej bj = gej_from_ge(b);
  r = gej_select(a.inf, bj, r);  // infinity + b = b
  return gej_select(skip, a, r);
}

// Complete Jacobian + Jacobian addition (12M + 4S), branching on the special cases: for the few places that MERGE partial sums
// (the latency path splits one verification into tasks run by different waves, verify_core.h "task split").  Inputs as the other
// gej functions leave them (x, y magnitude 1, z magnitude <= 2); any point may be infinity, equal or opposite to the other.
LAMD_HD gej gej_add_var(const gej &a, const gej &b) {
  if (a.inf) return b;
  if (b.inf) return a;
  const fe az = fe_norm_weak(a.z), bz = fe_norm_weak(b.z);
  const fe z12 = fe_sqr(az), z22 = fe_sqr(bz);
  const fe u1 = fe_mul(a.x, z22), u2 = fe_mul(b.x, z12);
  const fe s1 = fe_mul(a.y, fe_mul(z22, bz)), s2 = fe_mul(b.y, fe_mul(z12, az));
  const fe h = fe_norm_weak(fe_add(u2, fe_neg(u1, 1)));
  const fe rr = fe_norm_weak(fe_add(s2, fe_neg(s1, 1)));
  if (fe_is_zero(h)) {
    if (fe_is_zero(rr)) return gej_double(a);
    return gej_infinity();
  }
  const fe hh = fe_sqr(h);
  const fe hhh = fe_mul(h, hh);
  const fe v = fe_mul(u1, hh);
  gej r;
  r.x = fe_norm_weak(fe_add(fe_add(fe_sqr(rr), fe_neg(hhh, 1)), fe_neg(fe_mul_int(v, 2), 2)));  // R^2 - H^3 - 2V  (1 + 2 + 3)
  const fe t = fe_norm_weak(fe_add(v, fe_neg(r.x, 1)));
  r.y = fe_norm_weak(fe_add(fe_mul(rr, t), fe_neg(fe_mul(s1, hhh), 1)));                          // R*(V - X3) - S1*H^3
  r.z = fe_mul(fe_mul(az, bz), h);
  r.inf = false;
  return r;
}

LAMD_HD ge ge_neg_if(const ge &a, bool neg) {
  ge r;
  r.x = a.x;
  r.y = fe_select(neg, fe_norm_weak(fe_neg(a.y, 1)), a.y);
  return r;
}

// Conditional negation without a normalisation pass: y' = neg ? 2p - y : y, limb-wise as (y ^ m) + (m & (2*p_i + 1))
// (two's complement: ~y + 2p_i + 1 = 2p_i - y, and y <= 2p_i for a magnitude-1 y) -- three simple ops per limb.  The result has magnitude 2.
LAMD_HD ge ge_neg_if_lazy(const ge &a, bool neg) {
  LAMD_ASSERT(FE_MAG(a.y) <= 1);
  ge r;
  r.x = a.x;
  const u32 m = neg ? 0xFFFFFFFFu : 0u;
  r.y.n[0] = (a.y.n[0] ^ m) + (m & (2u * FE_P0 + 1u));
  r.y.n[1] = (a.y.n[1] ^ m) + (m & (2u * FE_P1 + 1u));
#pragma unroll
  for (int i = 2; i < 8; i++) r.y.n[i] = (a.y.n[i] ^ m) + (m & (2u * FE_PM + 1u));
  r.y.n[8] = (a.y.n[8] ^ m) + (m & (2u * FE_P8 + 1u));
  FE_SETMAG(r.y, 2);
  fe_verify(r.y);
  return r;
}

// ---- 64-byte packed affine points (canonical 8x32-bit little-endian words per coordinate)
LAMD_HD ge ge_from_words(const u32 xw[8], const u32 yw[8]) {
  ge r;
  r.x = fe_from_words(xw);
  r.y = fe_from_words(yw);
  return r;
}

// secp256k1 generator
#define LAMD_GX {0x16F81798u, 0x59F2815Bu, 0x2DCE28D9u, 0x029BFCDBu, 0xCE870B07u, 0x55A06295u, 0xF9DCBBACu, 0x79BE667Eu}
#define LAMD_GY {0xFB10D4B8u, 0x9C47D08Fu, 0xA6855419u, 0xFD17B448u, 0x0E1108A8u, 0x5DA4FBFCu, 0x26A3C465u, 0x483ADA77u}
// beta: cube root of unity mod p with lambda*(x, y) = (beta*x, y)
#define LAMD_BETA {0x719501EEu, 0xC1396C28u, 0x12F58995u, 0x9CF04975u, 0xAC3434E9u, 0x6E64479Eu, 0x657C0710u, 0x7AE96A2Bu}

}  // namespace lamd

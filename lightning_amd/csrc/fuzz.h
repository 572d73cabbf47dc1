// Randomised fuzz of the field and group primitives at their magnitude limits (diagnostic: lamd_fuzz_field runs it on the
// device and in the library's host pass; tests/devmath_host.cpp runs it with the magnitude assertions on).
#pragma once
#include "group.h"

namespace lamd {

LAMD_HD u64 fz_next(u64 &s) {
  s += 0x9E3779B97F4A7C15ULL;
  u64 x = s;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
  return x ^ (x >> 31);
}
// limbs uniform in [0, m * bound], 3 in 16 draws with EVERY limb at its bound, 1 in 16 zero
LAMD_HD fe fz_fe(u64 &s, u32 m) {
  fe r;
  const u32 mode = (u32)fz_next(s) & 15u;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    const u32 lim = m * (i < 8 ? FE_LIM29 : FE_LIM24);
    const u64 v = fz_next(s) >> 32;
    r.n[i] = mode < 3u ? lim : (mode == 3u ? 0u : (u32)((v * ((u64)lim + 1)) >> 32));
  }
  FE_SETMAG(r, (int)m);
  return r;
}
LAMD_HD void fz_mix(u64 &h, const fe &a) {
#pragma unroll
  for (int i = 0; i < 9; i++) h = (h ^ a.n[i]) * 0x100000001B3ULL;
}
constexpr int FZ_OPS_PER_ITER = 2 + 4 + 7 + 11 + 11 + 11, FZ_OPS_TAIL = 270;  // mul + sqr | fused forms | double | add core | complete add | bare add; one inversion per lane
LAMD_HD u64 fuzz_lane(u64 seed, u64 lane, int iters) {
  u64 s = seed ^ (lane * 0xD1342543DE82EF95ULL), h = 0xCBF29CE484222325ULL;
  const u32 pa[8] = {1, 1, 7, 2, 3, 1, 2, 4}, pb[8] = {1, 7, 1, 3, 2, 4, 2, 1};
  gej P;
  P.x = fz_fe(s, 1); P.y = fz_fe(s, 1); P.z = fz_fe(s, 2); P.inf = false;
  fe keep = fz_fe(s, 1);
#pragma unroll 1
  for (int it = 0; it < iters; it++) {
    const u32 k = (u32)fz_next(s);
    const fe a = fz_fe(s, pa[k & 7]), b = fz_fe(s, pb[k & 7]);
    const fe m1 = fe_mul(a, b);
    const fe c = fz_fe(s, 1 + ((k >> 3) & 1));
    const fe s1 = fe_sqr(c);
    fz_mix(h, m1); fz_mix(h, s1);
    // the fused forms at their budget: a*b + e (7 + a magnitude-7 addend), c^2 + e, a*b + c*d with the magnitude products adding up to 7
    const u32 pc[4] = {3, 2, 1, 6}, pd[4] = {2, 3, 6, 1}, pe[4] = {1, 1, 1, 1}, pf[4] = {1, 1, 1, 1};
    const u32 q = (k >> 9) & 3;
    const fe ad = fz_fe(s, 1 + ((k >> 11) % 7));
    const fe f1 = fe_mul_add(a, b, ad), f2 = fe_sqr_add(c, ad);
    const fe f3 = fe_mul2(fz_fe(s, pc[q]), fz_fe(s, pd[q]), fz_fe(s, pe[q]), fz_fe(s, pf[q]));   // 6 + 1
    const fe f4 = fe_mul2(fz_fe(s, 2), fz_fe(s, 2), fz_fe(s, 3), fz_fe(s, 1));                   // 4 + 3
    fz_mix(h, f1); fz_mix(h, f2); fz_mix(h, f3); fz_mix(h, f4);
    const fe t = fe_add(fe_neg(m1, 1), fe_mul_int(s1, 3));  // magnitude 5
    const fe nw = fe_norm_weak(t);
    fz_mix(h, nw); fz_mix(h, fe_carry(t)); fz_mix(h, fe_normalize(t));
    h = (h ^ (u64)fe_is_zero(t) ^ ((u64)fe_is_zero(fe_sub(nw, nw, 1)) << 1)) * 0x100000001B3ULL;  // nw - nw = 2p: zero
    P = gej_double(P);
    fz_mix(h, P.x); fz_mix(h, P.y); fz_mix(h, P.z);
    ge Q;
    Q.x = m1; Q.y = s1;
    bool degenerate;
    fe hh, rr;
    P = gej_add_ge_core(P, Q, &degenerate, &hh, &rr);
    fz_mix(h, P.x); fz_mix(h, P.y); fz_mix(h, P.z); fz_mix(h, hh); fz_mix(h, rr);
    Q.x = nw; Q.y = keep;
    P = gej_add_ge(P, ge_neg_if(Q, (k >> 4) & 1), ((k >> 5) & 7) == 0);
    fz_mix(h, P.x); fz_mix(h, P.y); fz_mix(h, P.z);
    h = (h ^ (u64)degenerate ^ ((u64)P.inf << 1)) * 0x100000001B3ULL;
    P.inf = false;
    Q.x = f1; Q.y = fe_add(f2, f3);   // the table-driven kernels' form: no tests, a magnitude-2 y
    P = gej_add_ge_fast(P, Q);
    fz_mix(h, P.x); fz_mix(h, P.y); fz_mix(h, P.z);
    P.z = fe_norm_weak(P.z);
    keep = fe_select((k >> 8) & 1, m1, s1);
  }
  fz_mix(h, fe_normalize(fe_inv(fe_norm_weak(P.x))));
  return h;
}

}  // namespace lamd

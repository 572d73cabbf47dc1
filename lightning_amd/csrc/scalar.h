// Scalars modulo the secp256k1 group order n, 8 x 32-bit saturated limbs (little-endian words).
// Only the per-signature preparation uses these (s^-1, u1 = z*s^-1, u2 = r*s^-1, the BIP-340
// challenge reduction and the GLV split) -- a few percent of a verification once the inversion
// is batched (Montgomery's trick across the signatures one thread owns), so this code favours
// being obviously right over being fast.
//
// Semantics replaced: libsecp256k1's scalar layer as reached from bitcoin/signature.c:188,425.
#pragma once
#include "lamd_common.h"

namespace lamd {

struct sc { u32 w[8]; };  // value in [0, n)

// n = FFFFFFFF FFFFFFFF FFFFFFFF FFFFFFFE BAAEDCE6 AF48A03B BFD25E8C D0364141
#define LAMD_SC_N {0xD0364141u, 0xBFD25E8Cu, 0xAF48A03Bu, 0xBAAEDCE6u, 0xFFFFFFFEu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}
// (n-1)/2
#define LAMD_SC_HALF {0x681B20A0u, 0xDFE92F46u, 0x57A4501Du, 0x5D576E73u, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0x7FFFFFFFu}
// 2^256 - n (129 bits)
#define LAMD_SC_NC {0x2FC9BEBFu, 0x402DA173u, 0x50B75FC4u, 0x45512319u, 1u}

LAMD_HD bool sc_is_zero(const sc &a) {
  u32 z = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) z |= a.w[i];
  return z == 0;
}
// a >= b on raw 256-bit words
LAMD_HD bool words_ge(const u32 a[8], const u32 b[8]) {
  bool ge = true;  // equal so far
#pragma unroll
  for (int i = 0; i < 8; i++) ge = (a[i] > b[i]) | ((a[i] == b[i]) & ge);  // low word first, high word decides last
  return ge;
}
LAMD_HD bool words_ge_n(const u32 a[8]) {
  const u32 n[8] = LAMD_SC_N;
  return words_ge(a, n);
}
// s > (n-1)/2 ?  (BOLT #2 / BIP-62 low-S rule enforced by secp256k1_ecdsa_verify)
LAMD_HD bool sc_is_high(const sc &a) {
  const u32 h[8] = LAMD_SC_HALF;
  return !words_ge(h, a.w);
}
// r = a - n (mod 2^256); only meaningful when a >= n
LAMD_HD void words_sub_n(u32 r[8], const u32 a[8]) {
  const u32 nc[5] = LAMD_SC_NC;  // a - n = a + (2^256 - n) mod 2^256
  u64 c = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    c += (u64)a[i] + (i < 5 ? nc[i] : 0u);
    r[i] = (u32)c;
    c >>= 32;
  }
}
// reduce a 256-bit integer once (it is < 2n because n > 2^255)
LAMD_HD sc sc_from_words(const u32 a[8], bool *overflow) {
  sc r;
  u32 t[8];
  words_sub_n(t, a);
  const bool ge = words_ge_n(a);
#pragma unroll
  for (int i = 0; i < 8; i++) r.w[i] = ge ? t[i] : a[i];
  if (overflow) *overflow = ge;
  return r;
}
LAMD_HD sc sc_neg(const sc &a) {
  const u32 n[8] = LAMD_SC_N;
  sc r;
  u64 borrow = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const u64 t = (u64)n[i] - a.w[i] - borrow;
    r.w[i] = (u32)t;
    borrow = (t >> 32) & 1;
  }
  const bool z = sc_is_zero(a);
#pragma unroll
  for (int i = 0; i < 8; i++) r.w[i] = z ? 0u : r.w[i];
  return r;
}

// out[0..HL+5] = lo[0..7] + hi[0..HL-1] * (2^256 - n); the caller knows how many words can be non-zero
template <int HL>
LAMD_HD void sc_fold(u32 *out, const u32 *lo, const u32 *hi) {
  const u32 nc[5] = LAMD_SC_NC;
  constexpr int OL = (HL + 5 > 8 ? HL + 5 : 8) + 1;
#pragma unroll
  for (int i = 0; i < OL; i++) out[i] = i < 8 ? lo[i] : 0u;
#pragma unroll
  for (int i = 0; i < HL; i++) {
    u32 carry = 0;
#pragma unroll
    for (int j = 0; j < 5; j++) {
      const u64 t = (u64)hi[i] * nc[j] + out[i + j] + carry;
      out[i + j] = (u32)t;
      carry = (u32)(t >> 32);
    }
    // ripple the row's carry
#pragma unroll
    for (int j = i + 5; j < OL; j++) {
      const u64 t = (u64)out[j] + carry;
      out[j] = (u32)t;
      carry = (u32)(t >> 32);
    }
  }
}

// 512-bit t -> t mod n
LAMD_HD sc sc_reduce512(const u32 t[16]) {
  u32 a[14];           // lo + hi*NC: < 2^256 + 2^385 -> 13 words (+1 spare)
  sc_fold<8>(a, t, t + 8);
  u32 b[11];           // a[0..7] + a[8..12]*NC: < 2^256 + 2^(130+129) -> 9 words
  sc_fold<5>(b, a, a + 8);
  u32 c[9];            // b[0..7] + b[8]*NC with b[8] < 2^4: < 2^256 + 2^133 -> bit 256 at most
  sc_fold<1>(c, b, b + 8);
  // c[8] in {0,1}: subtract n once if c >= n (c < 2n since c < 2^256 + 2^133 < 2n)
  u32 d[8];
  words_sub_n(d, c);
  const bool ge = (c[8] != 0) | words_ge_n(c);
  sc r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.w[i] = ge ? d[i] : c[i];
  return r;
}

LAMD_HD sc sc_mul(const sc &a, const sc &b) {
  u32 t[16];
#pragma unroll
  for (int i = 0; i < 16; i++) t[i] = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    u32 carry = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const u64 v = (u64)a.w[i] * b.w[j] + t[i + j] + carry;
      t[i + j] = (u32)v;
      carry = (u32)(v >> 32);
    }
    t[i + 8] = carry;
  }
  return sc_reduce512(t);
}
LAMD_HD sc sc_sqr_n(sc a, int n) {
#pragma unroll 1
  for (int i = 0; i < n; i++) a = sc_mul(a, a);
  return a;
}

// a^(n-2).  n-2 = (2^127 - 1) << 129 | 0 << 128 | 0xBAAEDCE6AF48A03BBFD25E8CD036413F
LAMD_HD sc sc_inv(const sc &a) {
  const sc x2 = sc_mul(sc_sqr_n(a, 1), a);
  const sc x4 = sc_mul(sc_sqr_n(x2, 2), x2);
  const sc x8 = sc_mul(sc_sqr_n(x4, 4), x4);
  const sc x16 = sc_mul(sc_sqr_n(x8, 8), x8);
  const sc x32 = sc_mul(sc_sqr_n(x16, 16), x16);
  const sc x64 = sc_mul(sc_sqr_n(x32, 32), x32);
  sc t = sc_mul(sc_sqr_n(x64, 32), x32);
  t = sc_mul(sc_sqr_n(t, 16), x16);
  t = sc_mul(sc_sqr_n(t, 8), x8);
  t = sc_mul(sc_sqr_n(t, 4), x4);
  t = sc_mul(sc_sqr_n(t, 2), x2);
  t = sc_mul(sc_sqr_n(t, 1), a);  // a^(2^127 - 1)
  t = sc_sqr_n(t, 1);             // bit 128 of n-2 is 0
  const u32 low[4] = {0xD036413Fu, 0xBFD25E8Cu, 0xAF48A03Bu, 0xBAAEDCE6u};
#pragma unroll 1
  for (int i = 127; i >= 0; i--) {
    t = sc_mul(t, t);
    if ((low[i >> 5] >> (i & 31)) & 1) t = sc_mul(t, a);
  }
  return t;
}

// ---- The multiplicative work of the ECDSA preparation (prefix products, the shared inversion, u1 = z/s, u2 = r/s) in 9 limbs
// of 29 bits, the representation fe.h uses for the field: a product column (<= 9 partial products of 2^58) accumulates in one
// 64-bit register without carry handling, and 2^261 = 32 * (2^256 - n) (mod n) is a 134-bit constant, so the high half folds back
// with 9x5, 5x5 and 1x5 limb products.  Values are residues below 2^261 with exactly carried limbs (not reduced below n until
// sc29_to_sc); 161 multiply-adds per multiplication against 64 + 70 + the 32-bit carry chains of sc_mul above.
struct sc29 { u32 n[9]; };
constexpr u32 SC29_M = 0x1FFFFFFFu;
#define LAMD_SC29_C {0x1937D7E0u, 0x0DA1732Fu, 0x1AFE2201u, 0x08C6542Du, 0x00028AA2u}  /* 2^261 mod n = 32 * LAMD_SC_NC, 29-bit limbs */

LAMD_HD sc29 sc29_from_words(const u32 w[8]) {  // any 256-bit value
  sc29 r;
  r.n[0] = w[0] & SC29_M;
  r.n[1] = ((w[0] >> 29) | (w[1] << 3)) & SC29_M;
  r.n[2] = ((w[1] >> 26) | (w[2] << 6)) & SC29_M;
  r.n[3] = ((w[2] >> 23) | (w[3] << 9)) & SC29_M;
  r.n[4] = ((w[3] >> 20) | (w[4] << 12)) & SC29_M;
  r.n[5] = ((w[4] >> 17) | (w[5] << 15)) & SC29_M;
  r.n[6] = ((w[5] >> 14) | (w[6] << 18)) & SC29_M;
  r.n[7] = ((w[6] >> 11) | (w[7] << 21)) & SC29_M;
  r.n[8] = w[7] >> 8;
  return r;
}
LAMD_HD sc29 sc29_from_sc(const sc &a) { return sc29_from_words(a.w); }
LAMD_HD sc29 sc29_one() {
  sc29 r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.n[i] = i == 0;
  return r;
}
// canonical representative in [0, n)
LAMD_HD sc sc29_to_sc(const sc29 &a) {
  // 261 bits as nine 32-bit words (the ninth holds bits 256..260)
  u32 w[9];
  w[0] = a.n[0] | (a.n[1] << 29);
  w[1] = (a.n[1] >> 3) | (a.n[2] << 26);
  w[2] = (a.n[2] >> 6) | (a.n[3] << 23);
  w[3] = (a.n[3] >> 9) | (a.n[4] << 20);
  w[4] = (a.n[4] >> 12) | (a.n[5] << 17);
  w[5] = (a.n[5] >> 15) | (a.n[6] << 14);
  w[6] = (a.n[6] >> 18) | (a.n[7] << 11);
  w[7] = (a.n[7] >> 21) | (a.n[8] << 8);
  w[8] = a.n[8] >> 24;
  const u32 nc[5] = LAMD_SC_NC;
  // v = low256 + q * (2^256 - n), twice: q <= 31 first, then the carry bit of that addition
  u32 q = w[8];
#pragma unroll
  for (int round = 0; round < 2; round++) {
    u64 c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      c += (u64)w[i] + (i < 5 ? (u64)q * nc[i] : 0u);
      w[i] = (u32)c;
      c >>= 32;
    }
    q = (u32)c;
  }
  return sc_from_words(w, nullptr);  // < 2^256 < 2n: one conditional subtraction
}
// exact carries of `len` 64-bit columns into 29-bit limbs; the last limb takes what is left (caller knows it is small)
template <int LEN>
LAMD_HD void sc29_carry(u32 *out, const u64 *col) {
  u64 cy = 0;
#pragma unroll
  for (int k = 0; k < LEN; k++) {
    cy += col[k];
    out[k] = (u32)cy & SC29_M;
    cy >>= 29;
  }
  out[LEN] = (u32)cy;
}
LAMD_HD sc29 sc29_mul(const sc29 &a, const sc29 &b) {
  const u32 C[5] = LAMD_SC29_C;
  u64 col[17];
#pragma unroll
  for (int k = 0; k < 17; k++) {
    u64 acc = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
      const int j = k - i;
      if (j < 0 || j > 8) continue;
      acc += (u64)a.n[i] * b.n[j];
    }
    col[k] = acc;
  }
  u32 t[18];
  sc29_carry<17>(t, col);  // t[17] < 2^30
  // fold 1: t[9..17] * 2^261 = t[9..17] * C -> columns 0..12
  u64 d[13];
#pragma unroll
  for (int k = 0; k < 13; k++) {
    u64 acc = k < 9 ? t[k] : 0;
#pragma unroll
    for (int j = 0; j < 5; j++) {
      const int i = k - j;
      if (i < 0 || i > 8) continue;
      acc += (u64)t[9 + i] * C[j];
    }
    d[k] = acc;
  }
  u32 r1[14];
  sc29_carry<13>(r1, d);
  // fold 2: r1[9..13] (5 limbs) * C -> columns 0..8
  u64 e[9];
#pragma unroll
  for (int k = 0; k < 9; k++) {
    u64 acc = r1[k];
#pragma unroll
    for (int j = 0; j < 5; j++) {
      const int i = k - j;
      if (i < 0 || i > 4) continue;
      acc += (u64)r1[9 + i] * C[j];
    }
    e[k] = acc;
  }
  u32 r2[10];
  sc29_carry<9>(r2, e);
  // folds 3 and 4: the limb above 2^261 is < 2^9, then 0 or 1; after the fourth the value is below 2^261
#pragma unroll
  for (int round = 0; round < 2; round++) {
    u64 f[9];
#pragma unroll
    for (int k = 0; k < 9; k++) f[k] = (u64)r2[k] + (k < 5 ? (u64)r2[9] * C[k] : 0u);
    sc29_carry<9>(r2, f);
  }
  LAMD_ASSERT(r2[9] == 0);
  sc29 r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.n[i] = r2[i];
  return r;
}
LAMD_HD sc29 sc29_sqr_n(sc29 a, int n) {
#pragma unroll 1
  for (int i = 0; i < n; i++) a = sc29_mul(a, a);
  return a;
}
// a^(n-2), the chain of sc_inv
LAMD_HD sc29 sc29_inv(const sc29 &a) {
  const sc29 x2 = sc29_mul(sc29_sqr_n(a, 1), a);
  const sc29 x4 = sc29_mul(sc29_sqr_n(x2, 2), x2);
  const sc29 x8 = sc29_mul(sc29_sqr_n(x4, 4), x4);
  const sc29 x16 = sc29_mul(sc29_sqr_n(x8, 8), x8);
  const sc29 x32 = sc29_mul(sc29_sqr_n(x16, 16), x16);
  const sc29 x64 = sc29_mul(sc29_sqr_n(x32, 32), x32);
  sc29 t = sc29_mul(sc29_sqr_n(x64, 32), x32);
  t = sc29_mul(sc29_sqr_n(t, 16), x16);
  t = sc29_mul(sc29_sqr_n(t, 8), x8);
  t = sc29_mul(sc29_sqr_n(t, 4), x4);
  t = sc29_mul(sc29_sqr_n(t, 2), x2);
  t = sc29_mul(sc29_sqr_n(t, 1), a);  // a^(2^127 - 1)
  t = sc29_sqr_n(t, 1);               // bit 128 of n-2 is 0
  const u32 low[4] = {0xD036413Fu, 0xBFD25E8Cu, 0xAF48A03Bu, 0xBAAEDCE6u};
#pragma unroll 1
  for (int i = 127; i >= 0; i--) {
    t = sc29_mul(t, t);
    if ((low[i >> 5] >> (i & 31)) & 1) t = sc29_mul(t, a);
  }
  return t;
}

// ---- Modular inversion by division steps (Bernstein-Yang "safegcd", the variable-time form: signatures are public).
// (delta, f, g) -> g odd and delta > 0: (1 - delta, g, (g - f) / 2); else (1 + delta, f, (g + (g & 1) * f) / 2), starting from
// (1, n, a).  Thirty steps at a time only look at the low 32 bits and yield a 2x2 integer matrix t with
// 2^30 * (f', g') = t * (f, g); the 256-bit values are then updated once per batch in signed 30-bit limbs, and so is the pair
// (d, e) with d * a = f, e * a = g (mod n), whose division by 2^30 is made exact by adding a multiple of n.  When g reaches 0,
// f = +-1 and a^-1 = +-d.  <= 19 batches on every input tried (bound: 724 steps = 25 batches); ~10^4 instructions against ~1.2 * 10^5
// for the exponentiation a^(n-2) -- the scalar inversion was what a small batch waited for (0.3 ms of its 0.55).
struct s30 { int32_t v[9]; };  // limbs 0..7 in [0, 2^30), limb 8 signed
#define LAMD_S30_N {0x10364141, 0x3F497A33, 0x348A03BB, 0x2BB739AB, 0x3FFFFEBA, 0x3FFFFFFF, 0x3FFFFFFF, 0x3FFFFFFF, 0xFFFF}
constexpr u32 S30_NINV = 0x2A774EC1u;  // n^-1 mod 2^30
struct s30_mat { int32_t u, v, q, r; };
LAMD_HD void s30_update_fg(s30 &f, s30 &g, const s30_mat &t) {
  const int64_t M = (1 << 30) - 1;
  int64_t cf = (int64_t)t.u * f.v[0] + (int64_t)t.v * g.v[0], cg = (int64_t)t.q * f.v[0] + (int64_t)t.r * g.v[0];
  LAMD_ASSERT((cf & M) == 0 && (cg & M) == 0);
  cf >>= 30; cg >>= 30;
#pragma unroll
  for (int i = 1; i < 9; i++) {
    cf += (int64_t)t.u * f.v[i] + (int64_t)t.v * g.v[i];
    cg += (int64_t)t.q * f.v[i] + (int64_t)t.r * g.v[i];
    f.v[i - 1] = (int32_t)(cf & M); cf >>= 30;
    g.v[i - 1] = (int32_t)(cg & M); cg >>= 30;
  }
  f.v[8] = (int32_t)cf;
  g.v[8] = (int32_t)cg;
}
LAMD_HD void s30_update_de(s30 &d, s30 &e, const s30_mat &t, const int32_t nl[9], u32 ninv) {
  const int64_t M = (1 << 30) - 1;
  int64_t cd = (int64_t)t.u * d.v[0] + (int64_t)t.v * e.v[0], ce = (int64_t)t.q * d.v[0] + (int64_t)t.r * e.v[0];
  const int64_t md = (int64_t)((0u - (u32)(cd & M)) * ninv & (u32)M), me = (int64_t)((0u - (u32)(ce & M)) * ninv & (u32)M);
  cd += md * nl[0]; ce += me * nl[0];
  LAMD_ASSERT((cd & M) == 0 && (ce & M) == 0);
  cd >>= 30; ce >>= 30;
#pragma unroll
  for (int i = 1; i < 9; i++) {
    cd += (int64_t)t.u * d.v[i] + (int64_t)t.v * e.v[i] + md * nl[i];
    ce += (int64_t)t.q * d.v[i] + (int64_t)t.r * e.v[i] + me * nl[i];
    d.v[i - 1] = (int32_t)(cd & M); cd >>= 30;
    e.v[i - 1] = (int32_t)(ce & M); ce >>= 30;
  }
  d.v[8] = (int32_t)cd;
  e.v[8] = (int32_t)ce;
}
// x with x * a = 1 (mod m) for an odd 256-bit modulus m (30-bit limbs ml, minv = m^-1 mod 2^30) and a in [0, m): a two's-complement
// integer in (-26 m, 26 m), nine 32-bit words (a = 0 gives 0).  The caller adds a multiple of m and reduces.
LAMD_HD void s30_inverse(u32 out[9], const u32 a[8], const int32_t ml[9], u32 minv) {
  s30 f, g, d, e;
#pragma unroll
  for (int i = 0; i < 9; i++) { f.v[i] = ml[i]; d.v[i] = 0; e.v[i] = i == 0; }
  g.v[0] = (int32_t)(a[0] & 0x3FFFFFFFu);
  g.v[1] = (int32_t)(((a[0] >> 30) | (a[1] << 2)) & 0x3FFFFFFFu);
  g.v[2] = (int32_t)(((a[1] >> 28) | (a[2] << 4)) & 0x3FFFFFFFu);
  g.v[3] = (int32_t)(((a[2] >> 26) | (a[3] << 6)) & 0x3FFFFFFFu);
  g.v[4] = (int32_t)(((a[3] >> 24) | (a[4] << 8)) & 0x3FFFFFFFu);
  g.v[5] = (int32_t)(((a[4] >> 22) | (a[5] << 10)) & 0x3FFFFFFFu);
  g.v[6] = (int32_t)(((a[5] >> 20) | (a[6] << 12)) & 0x3FFFFFFFu);
  g.v[7] = (int32_t)(((a[6] >> 18) | (a[7] << 14)) & 0x3FFFFFFFu);
  g.v[8] = (int32_t)(a[7] >> 16);
  int32_t delta = 1;
#pragma unroll 1
  for (int batch = 0; batch < 25; batch++) {
    int32_t nz = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) nz |= g.v[i];
    if (nz == 0) break;
    u32 fl = (u32)f.v[0] | ((u32)f.v[1] << 30), gl = (u32)g.v[0] | ((u32)g.v[1] << 30);
    s30_mat t = {1, 0, 0, 1};
#pragma unroll 1
    for (int i = 0; i < 30; i++) {
      const bool odd = gl & 1u;
      u32 x = fl;
      int32_t y = t.u, z = t.v;
      if (odd && delta > 0) {
        fl = gl; t.u = t.q; t.v = t.r;
        x = 0u - x; y = -y; z = -z;
        delta = -delta;
      }
      if (odd) { gl += x; t.q += y; t.r += z; }
      gl >>= 1;
      t.u *= 2; t.v *= 2;
      delta += 1;
    }
    s30_update_fg(f, g, t);
    s30_update_de(d, e, t, ml, minv);
  }
  // f = +-1: the inverse is +-d
  const bool neg = f.v[8] < 0;
  int64_t c = 0;
  u32 limb[9];
#pragma unroll
  for (int i = 0; i < 9; i++) {
    c += neg ? -(int64_t)d.v[i] : (int64_t)d.v[i];
    limb[i] = (u32)(c & 0x3FFFFFFF);
    c >>= 30;
  }
  // limb[0..8] (30 bits each) + c * 2^270 is the two's-complement value; c is 0 or -1
  out[0] = limb[0] | (limb[1] << 30);
  out[1] = (limb[1] >> 2) | (limb[2] << 28);
  out[2] = (limb[2] >> 4) | (limb[3] << 26);
  out[3] = (limb[3] >> 6) | (limb[4] << 24);
  out[4] = (limb[4] >> 8) | (limb[5] << 22);
  out[5] = (limb[5] >> 10) | (limb[6] << 20);
  out[6] = (limb[6] >> 12) | (limb[7] << 18);
  out[7] = (limb[7] >> 14) | (limb[8] << 16);
  out[8] = (limb[8] >> 16) | ((u32)c << 14);   // sign-extended top word
}
// a in [0, n) -> a^-1 mod n (0 -> 0)
LAMD_HD sc sc_inv_var(const sc &a) {
  const int32_t nl[9] = LAMD_S30_N;
  u32 w[9];
  s30_inverse(w, a.w, nl, S30_NINV);
  const u32 nw[8] = LAMD_SC_N;
  // + 32 n  (n << 5), modulo 2^288: the sum is in [0, 64 n)
  u64 cy = 0;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    const u32 ni = i < 8 ? ((nw[i] << 5) | (i ? nw[i - 1] >> 27 : 0u)) : (nw[7] >> 27);
    cy += (u64)w[i] + ni;
    w[i] = (u32)cy;
    cy >>= 32;
  }
  const u32 nc[5] = LAMD_SC_NC;
  u32 q = w[8];  // < 64
#pragma unroll
  for (int round = 0; round < 2; round++) {
    u64 k = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      k += (u64)w[i] + (i < 5 ? (u64)q * nc[i] : 0u);
      w[i] = (u32)k;
      k >>= 32;
    }
    q = (u32)k;
  }
  return sc_from_words(w, nullptr);
}

// ---- GLV endomorphism split: k = k1 + k2*lambda (mod n) with |k1|, |k2| < 2^128.
// Lattice basis (a1, b1), (a2, b2) of {(x, y): x + y*lambda = 0 mod n}; g1 = round(2^384*b2/n),
// g2 = round(2^384*(-b1)/n); c1 = round(k*g1 / 2^384), c2 = round(k*g2 / 2^384);
// k1 = k - c1*a1 - c2*a2, k2 = c1*(-b1) - c2*b2 as exact (small) integers.
// Constants re-derived and the 128-bit bound checked in tests (tests/test_devmath_host.py).
struct glv_half {
  u32 mag[4];  // |k_i| + 0x8888...8 (low 128 bits): window j, digit = nibble_j - 8 in [-8, 7]
  u32 top;     // carry out of that addition: digit 32 in {0, 1}
  u32 neg;     // 1 if k_i < 0
};

LAMD_HD void mul_words_lo(u32 *out, int outl, const u32 *a, int al, const u32 *b, int bl) {
  for (int i = 0; i < outl; i++) out[i] = 0;
  for (int i = 0; i < al; i++) {
    u32 carry = 0;
    for (int j = 0; j < bl && i + j < outl; j++) {
      const u64 t = (u64)a[i] * b[j] + out[i + j] + carry;
      out[i + j] = (u32)t;
      carry = (u32)(t >> 32);
    }
    if (i + bl < outl) out[i + bl] = carry;
  }
}

// round(k * g / 2^384) for a 256-bit g: top 128 bits of the 512-bit product, rounded
LAMD_HD void glv_mulshift384(u32 c[4], const sc &k, const u32 g[8]) {
  u32 t[16];
#pragma unroll
  for (int i = 0; i < 16; i++) t[i] = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    u32 carry = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const u64 v = (u64)k.w[i] * g[j] + t[i + j] + carry;
      t[i + j] = (u32)v;
      carry = (u32)(v >> 32);
    }
    t[i + 8] = carry;
  }
  // + 2^383 then >> 384: add bit 31 of word 11 into words 12..15
  u64 cy = (t[11] >> 31) & 1;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    cy += t[12 + i];
    c[i] = (u32)cy;
    cy >>= 32;
  }
}

LAMD_HD glv_half glv_finish(const u32 v[6]) {
  // v: signed 192-bit two's complement, |v| < 2^128
  glv_half h;
  const u32 neg = v[5] >> 31;
  u32 m[4];
  u64 c = neg;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    c += (u64)(neg ? ~v[i] : v[i]);
    m[i] = (u32)c;
    c >>= 32;
  }
  c = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    c += (u64)m[i] + 0x88888888u;
    h.mag[i] = (u32)c;
    c >>= 32;
  }
  h.top = (u32)c;
  h.neg = neg;
  return h;
}

LAMD_HD void glv_split(glv_half *h1, glv_half *h2, const sc &k) {
  const u32 g1[8] = {0x45DBB031u, 0xE893209Au, 0x71E8CA7Fu, 0x3DAA8A14u, 0x9284EB15u, 0xE86C90E4u, 0xA7D46BCDu, 0x3086D221u};
  const u32 g2[8] = {0x8AC47F71u, 0x1571B4AEu, 0x9DF506C6u, 0x221208ACu, 0x0ABFE4C4u, 0x6F547FA9u, 0x010E8828u, 0xE4437ED6u};
  const u32 a1[4] = {0x9284EB15u, 0xE86C90E4u, 0xA7D46BCDu, 0x3086D221u};          // a1 = b2
  const u32 mb1[4] = {0x0ABFE4C3u, 0x6F547FA9u, 0x010E8828u, 0xE4437ED6u};         // -b1
  const u32 a2[5] = {0x9D44CFD8u, 0x57C1108Du, 0xA8E2F3F6u, 0x14CA50F7u, 1u};
  u32 c1[4], c2[4];
  glv_mulshift384(c1, k, g1);
  glv_mulshift384(c2, k, g2);
  u32 p1[6], p2[6], q1[6], q2[6];
  mul_words_lo(p1, 6, c1, 4, a1, 4);   // c1*a1
  mul_words_lo(p2, 6, c2, 4, a2, 5);   // c2*a2
  mul_words_lo(q1, 6, c1, 4, mb1, 4);  // c1*(-b1)
  mul_words_lo(q2, 6, c2, 4, a1, 4);   // c2*b2
  u32 k1[6], k2[6];
  // k1 = k - p1 - p2 (mod 2^192), k2 = q1 - q2 (mod 2^192)
  u64 b = 0;
#pragma unroll
  for (int i = 0; i < 6; i++) {
    const u64 t = (u64)k.w[i] - p1[i] - b;
    k1[i] = (u32)t;
    b = (t >> 32) & 1;
  }
  b = 0;
#pragma unroll
  for (int i = 0; i < 6; i++) {
    const u64 t = (u64)k1[i] - p2[i] - b;
    k1[i] = (u32)t;
    b = (t >> 32) & 1;
  }
  b = 0;
#pragma unroll
  for (int i = 0; i < 6; i++) {
    const u64 t = (u64)q1[i] - q2[i] - b;
    k2[i] = (u32)t;
    b = (t >> 32) & 1;
  }
  *h1 = glv_finish(k1);
  *h2 = glv_finish(k2);
}

}  // namespace lamd

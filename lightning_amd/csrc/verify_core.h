// Per-signature verification logic, written once as host+device inline functions so that the
// kernels in kernels.hip and the CPU test harness (tests/devmath_host.cpp) run the same code.
//
// Pipeline (one signature per lane everywhere):
//   prep     ECDSA : parse r,s (range, low-S), z = hash mod n, s^-1 by Montgomery batch inversion
//                    over the signatures a thread owns, u1 = z/s, u2 = r/s, GLV-split u2
//            BIP340: r < p, s < n, e = H_challenge(r||pk||m) mod n, u1 = s, GLV-split (-e)
//   keys     SEC1 33/65-byte or x-only 32-byte public key -> validated affine point
//   ecmult   R = u1*G + u2*Q: per-lane 8-entry table of Q (shared-Z / isomorphic-curve trick so
//            the ladder only does mixed additions), 33 signed 4-bit windows x 2 half-scalars,
//            then 16 lookups in the 16-bit-window table of G; final x (and y-parity) check
//
// Reference semantics: secp256k1_ecdsa_verify / secp256k1_schnorrsig_verify /
// secp256k1_ec_pubkey_parse / secp256k1_xonly_pubkey_parse as called from
// bitcoin/signature.c:188,422,425, common/node_id.c:24, bitcoin/pubkey.c:19.
#pragma once
#include "group.h"
#include "scalar.h"
#include "sha256.h"

namespace lamd {

enum { MODE_ECDSA = 0, MODE_SCHNORR = 1 };

// ---- per-signature record produced by prep, consumed by ecmult (80 bytes, 16-byte aligned)
struct prep_rec {
  u32 u1[8];   // scalar for G (little-endian words)
  u32 k1[4];   // |k1| + 0x88..8  (see glv_half)
  u32 k2[4];
  u32 flags;   // PREP_*
  u32 pad[3];
};
enum { PREP_VALID = 1, PREP_K1NEG = 2, PREP_K2NEG = 4, PREP_K1TOP = 8, PREP_K2TOP = 16 };

// scratch slot owned by one ecmult lane (u32 words)
constexpr int SLOT_WORDS = 256;        // 1 KiB
constexpr int SLOT_ENTRY_WORDS = 24;   // x[8] | beta*x[8] | y[8]
constexpr int SLOT_H_OFF = 8 * SLOT_ENTRY_WORDS;  // 6 x 8 words of H_2..H_7

// Static table of G: window w, digit d -> d * 2^(BITS*w) * G as 64-byte affine words (d = 0 unused).
// 16-bit windows (64 MiB, lives in HBM / Infinity Cache) make u1*G sixteen mixed additions.
// The CPU test harness builds the same code with 8-bit windows to keep its table small.
#ifndef LAMD_GTABLE_WINDOW_BITS
#define LAMD_GTABLE_WINDOW_BITS 16
#endif
constexpr int GTABLE_WINDOW_BITS = LAMD_GTABLE_WINDOW_BITS;
constexpr int GTABLE_WINDOWS = 256 / GTABLE_WINDOW_BITS;
constexpr size_t GTABLE_ENTRIES = (size_t)GTABLE_WINDOWS << GTABLE_WINDOW_BITS;
constexpr size_t GTABLE_BYTES = GTABLE_ENTRIES * 64;

// ---- byte loads: 32 big-endian bytes -> 8 little-endian words (w[0] least significant)
LAMD_HD u32 load_be32(const u8 *p) {
  return ((u32)p[0] << 24) | ((u32)p[1] << 16) | ((u32)p[2] << 8) | (u32)p[3];
}
LAMD_HD void load_words_be(u32 w[8], const u8 *p) {
  if ((((uintptr_t)p) & 3) == 0) {
    const u32 *q = (const u32 *)p;
#pragma unroll
    for (int i = 0; i < 8; i++) w[7 - i] = __builtin_bswap32(q[i]);
  } else {
#pragma unroll
    for (int i = 0; i < 8; i++) w[7 - i] = load_be32(p + 4 * i);
  }
}

// ---- public keys.  len: 33 / 65 (SEC1) or 32 (BIP-340 x-only, lifted to even y).
// Writes canonical affine words; returns validity.
LAMD_HD bool parse_pubkey(const u8 *p, int len, u32 qx[8], u32 qy[8]) {
  bool ok = true;
  u32 prefix = 2;
  const u8 *xp = p;
  if (len != 32) {
    prefix = p[0];
    xp = p + 1;
  }
  load_words_be(qx, xp);
  ok &= !words_ge_p(qx);
  const fe x = fe_from_words(qx);
  const fe rhs = fe_add(fe_mul(fe_sqr(x), x), fe_set_int(7));  // x^3 + 7  (2)
  if (len == 65) {
    load_words_be(qy, p + 33);
    ok &= !words_ge_p(qy);
    ok &= (prefix == 4) | (prefix == 6) | (prefix == 7);
    ok &= (prefix == 4) | ((qy[0] & 1) == (prefix & 1));  // hybrid: parity byte must match
    const fe y = fe_from_words(qy);
    ok &= fe_equal(rhs, fe_sqr(y), 1);
  } else {
    ok &= (prefix == 2) | (prefix == 3);
    const fe rn = fe_norm_weak(rhs);
    fe y = fe_sqrt_candidate(rn);
    ok &= fe_equal(rn, fe_sqr(y), 1);
    y = fe_normalize(y);
    const bool flip = (y.n[0] & 1) != (prefix & 1);
    y = fe_select(flip, fe_normalize(fe_neg(y, 1)), y);
    fe_to_words(qy, y);
  }
  return ok;
}

// ---- ECDSA preparation for the signatures i = first, first+stride, ... < n owned by one thread
// recs[i].u1 doubles as the prefix-product store between the two passes.
LAMD_HD void ecdsa_load_rs(const u8 *sig64, sc *r, sc *s, bool *ok) {
  u32 rw[8], sw[8];
  load_words_be(rw, sig64);
  load_words_be(sw, sig64 + 32);
  bool v = !words_ge_n(rw) & !words_ge_n(sw);  // secp256k1_ecdsa_signature_parse_compact
#pragma unroll
  for (int i = 0; i < 8; i++) { r->w[i] = rw[i]; s->w[i] = sw[i]; }
  v &= !sc_is_zero(*r) & !sc_is_zero(*s);      // secp256k1_ecdsa_verify: r, s in [1, n-1]
  v &= !sc_is_high(*s);                        // ... and low-S
  *ok = v;
}

LAMD_HD void ecdsa_prep_thread(size_t first, size_t stride, size_t n, const u8 *hash32, const u8 *sig64,
                               prep_rec *recs) {
  sc acc;
#pragma unroll
  for (int i = 0; i < 8; i++) acc.w[i] = (i == 0);
  size_t last = first;
  bool any = false;
#pragma unroll 1
  for (size_t i = first; i < n; i += stride) {
    sc r, s;
    bool ok;
    ecdsa_load_rs(sig64 + 64 * i, &r, &s, &ok);
#pragma unroll
    for (int k = 0; k < 8; k++) recs[i].u1[k] = acc.w[k];  // product of the valid s before i
    if (ok) acc = sc_mul(acc, s);
    last = i;
    any = true;
  }
  if (!any) return;
  sc inv = sc_inv(acc);
#pragma unroll 1
  for (size_t i = last;; i -= stride) {
    sc r, s, prefix;
    bool ok;
    ecdsa_load_rs(sig64 + 64 * i, &r, &s, &ok);
#pragma unroll
    for (int k = 0; k < 8; k++) prefix.w[k] = recs[i].u1[k];
    prep_rec out;
#pragma unroll
    for (int k = 0; k < 8; k++) out.u1[k] = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) { out.k1[k] = 0x88888888u; out.k2[k] = 0x88888888u; }
    out.flags = 0;
    out.pad[0] = out.pad[1] = out.pad[2] = 0;
    if (ok) {
      const sc w = sc_mul(inv, prefix);  // s_i^-1
      inv = sc_mul(inv, s);
      u32 zw[8];
      load_words_be(zw, hash32 + 32 * i);
      const sc z = sc_from_words(zw, nullptr);
      const sc u1 = sc_mul(z, w);
      const sc u2 = sc_mul(r, w);
      glv_half h1, h2;
      glv_split(&h1, &h2, u2);
#pragma unroll
      for (int k = 0; k < 8; k++) out.u1[k] = u1.w[k];
#pragma unroll
      for (int k = 0; k < 4; k++) { out.k1[k] = h1.mag[k]; out.k2[k] = h2.mag[k]; }
      out.flags = PREP_VALID | (h1.neg ? PREP_K1NEG : 0) | (h2.neg ? PREP_K2NEG : 0) | (h1.top ? PREP_K1TOP : 0) |
                  (h2.top ? PREP_K2TOP : 0);
    }
    recs[i] = out;
    if (i == first) break;
  }
}

// ---- BIP-340 preparation, one signature
LAMD_HD void schnorr_prep_one(const u8 *msg32, const u8 *pk32, const u8 *sig64, prep_rec *rec) {
  u32 rw[8], sw[8];
  load_words_be(rw, sig64);
  load_words_be(sw, sig64 + 32);
  bool ok = !words_ge_p(rw) & !words_ge_n(sw);
  u32 rb[8], pb[8], mb[8], eh[8];
#pragma unroll
  for (int i = 0; i < 8; i++) rb[i] = rw[7 - i];
  u32 t[8];
  load_words_be(t, pk32);
#pragma unroll
  for (int i = 0; i < 8; i++) pb[i] = t[7 - i];
  load_words_be(t, msg32);
#pragma unroll
  for (int i = 0; i < 8; i++) mb[i] = t[7 - i];
  bip340_challenge(eh, rb, pb, mb);
  u32 ew[8];
#pragma unroll
  for (int i = 0; i < 8; i++) ew[i] = eh[7 - i];
  const sc e = sc_from_words(ew, nullptr);
  const sc ne = sc_neg(e);
  glv_half h1, h2;
  glv_split(&h1, &h2, ne);
  prep_rec out;
#pragma unroll
  for (int k = 0; k < 8; k++) out.u1[k] = sw[k];
#pragma unroll
  for (int k = 0; k < 4; k++) { out.k1[k] = h1.mag[k]; out.k2[k] = h2.mag[k]; }
  out.flags = (ok ? PREP_VALID : 0) | (h1.neg ? PREP_K1NEG : 0) | (h2.neg ? PREP_K2NEG : 0) | (h1.top ? PREP_K1TOP : 0) |
              (h2.top ? PREP_K2TOP : 0);
  out.pad[0] = out.pad[1] = out.pad[2] = 0;
  *rec = out;
}

// ---- table slot helpers
LAMD_HD void slot_store_fe(u32 *dst, const fe &a) {
  u32 w[8];
  fe_to_words(w, fe_normalize(a));
#pragma unroll
  for (int i = 0; i < 8; i++) dst[i] = w[i];
}
LAMD_HD fe slot_load_raw(const u32 *src) {
  fe r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.n[i] = src[i];
  FE_SETMAG(r, 1);
  return r;
}
LAMD_HD fe slot_load_fe(const u32 *src) {
  u32 w[8];
#pragma unroll
  for (int i = 0; i < 8; i++) w[i] = src[i];
  return fe_from_words(w);
}

// Build a table of 1Q..NE*Q brought to one shared Z (returned): entry e = (e+1)*Q as an affine point of the
// isomorphic curve y^2 = x^3 + 7*Zg^6, stored as x | beta*x | y (SLOT_ENTRY_WORDS each); the NE-2 values H of the
// addition chain are parked in hbuf ((NE-2)*8 words) until the rescale pass.
template <int NE>
LAMD_HD fe build_multiples(u32 *slot, u32 *hbuf, const ge &q) {
  gej p = gej_from_ge(q);
  slot_store_fe(slot + 0, p.x);
  slot_store_fe(slot + 16, p.y);
  p = gej_double(p);
  slot_store_fe(slot + SLOT_ENTRY_WORDS + 0, p.x);
  slot_store_fe(slot + SLOT_ENTRY_WORDS + 16, p.y);
#pragma unroll 1
  for (int i = 2; i < NE; i++) {  // entry i = (i+1)Q = entry(i-1) + Q
    bool degenerate;
    fe h, rr;
    p = gej_add_ge_core(p, q, &degenerate, &h, &rr);  // (i)Q = +-Q is impossible for i in 2..NE: no degenerate case
    slot_store_fe(slot + i * SLOT_ENTRY_WORDS + 0, p.x);
    slot_store_fe(slot + i * SLOT_ENTRY_WORDS + 16, p.y);
    slot_store_fe(hbuf + (i - 2) * 8, h);
  }
  const fe zg = fe_norm_weak(p.z);
  const u32 betaw[8] = LAMD_BETA;
  const fe beta = fe_from_words(betaw);
  // the last entry already has Z = Zg
  {
    const fe x = slot_load_fe(slot + (NE - 1) * SLOT_ENTRY_WORDS);
    slot_store_fe(slot + (NE - 1) * SLOT_ENTRY_WORDS + 8, fe_mul(x, beta));
  }
  fe rho = fe_set_int(1);
#pragma unroll 1
  for (int i = NE - 2; i >= 0; i--) {
    // rho = Zg / Z_entry(i): entry i+1 = entry i + Q had Z_{i+1} = Z_i * H (H stored at index i-1 for i >= 1),
    // and entry 1 = 2Q has Z = Z_2, entry 0 = Q has Z = 1 so its ratio is Zg itself.
    if (i >= 1) rho = fe_mul(rho, slot_load_fe(hbuf + (i - 1) * 8));
    else rho = zg;
    const fe r2 = fe_sqr(rho);
    const fe r3 = fe_mul(r2, rho);
    const fe x = fe_mul(slot_load_fe(slot + i * SLOT_ENTRY_WORDS + 0), r2);
    const fe y = fe_mul(slot_load_fe(slot + i * SLOT_ENTRY_WORDS + 16), r3);
    slot_store_fe(slot + i * SLOT_ENTRY_WORDS + 0, x);
    slot_store_fe(slot + i * SLOT_ENTRY_WORDS + 8, fe_mul(x, beta));
    slot_store_fe(slot + i * SLOT_ENTRY_WORDS + 16, y);
  }
  return zg;
}
// the per-signature ladder's table: 1Q..8Q inside the lane's 1 KiB slot
LAMD_HD fe build_q_table(u32 *slot, const ge &q) { return build_multiples<8>(slot, slot + SLOT_H_OFF, q); }

LAMD_HD int glv_digit(const u32 mag[4], u32 top, int i) {
  // window i of the biased magnitude; i == 32 is the carry bit
  if (i == 32) return (int)top;
  return (int)((mag[i >> 3] >> ((i & 7) * 4)) & 15u) - 8;
}

// R = u1*G + (k1 + k2*lambda)*Q for a prepared record; returns R (Jacobian on the real curve)
LAMD_HD gej ecmult_lane(const prep_rec &rec, const ge &q, u32 *slot, const u32 *gtable) {
  const fe zg = build_q_table(slot, q);
  const bool n1 = rec.flags & PREP_K1NEG, n2 = rec.flags & PREP_K2NEG;
  const u32 t1 = (rec.flags & PREP_K1TOP) ? 1u : 0u, t2 = (rec.flags & PREP_K2TOP) ? 1u : 0u;
  gej acc = gej_infinity();
#pragma unroll 1
  for (int i = 32; i >= 0; i--) {
    if (i != 32) {
#pragma unroll 1
      for (int j = 0; j < 4; j++) acc = gej_double(acc);
    }
#pragma unroll 1
    for (int half = 0; half < 2; half++) {
      int d = half ? glv_digit(rec.k2, t2, i) : glv_digit(rec.k1, t1, i);
      if (half ? n2 : n1) d = -d;
      const bool skip = d == 0;
      const int a = d < 0 ? -d : d;
      const u32 *e = slot + (skip ? 0 : a - 1) * SLOT_ENTRY_WORDS;
      ge pt;
      pt.x = slot_load_fe(e + (half ? 8 : 0));
      pt.y = slot_load_fe(e + 16);
      pt = ge_neg_if(pt, d < 0);
      acc = gej_add_ge(acc, pt, skip);
    }
  }
  // back from the isomorphic curve: (X, Y, Z) -> (X, Y, Z*Zg)
  acc.z = fe_mul(acc.z, zg);
  // + u1*G from the 16-bit-window table
#pragma unroll 1
  for (int w = 0; w < GTABLE_WINDOWS; w++) {
    const u32 d = (rec.u1[(w * GTABLE_WINDOW_BITS) >> 5] >> ((w * GTABLE_WINDOW_BITS) & 31)) & ((1u << GTABLE_WINDOW_BITS) - 1u);
    const bool skip = d == 0;
    const u32 *e = gtable + (((size_t)w << GTABLE_WINDOW_BITS) + d) * 16;
    ge pt;
    pt.x = slot_load_fe(e);
    pt.y = slot_load_fe(e + 8);
    acc = gej_add_ge(acc, pt, skip);
  }
  return acc;
}

// One entry of the static G table: out = d * B (B = 2^(BITS*w) * G given as affine words), d >= 1.
LAMD_HD void gtable_compute_entry(u32 out[16], const u32 base[16], u32 d) {
  const ge b = ge_from_words(base, base + 8);
  gej acc = gej_infinity();
#pragma unroll 1
  for (int bit = GTABLE_WINDOW_BITS - 1; bit >= 0; bit--) {
    acc = gej_double(acc);
    acc = gej_add_ge(acc, b, ((d >> bit) & 1u) == 0);
  }
  const fe zi = fe_inv(fe_norm_weak(acc.z));
  const fe zi2 = fe_sqr(zi);
  fe_to_words(out, fe_normalize(fe_mul(acc.x, zi2)));
  fe_to_words(out + 8, fe_normalize(fe_mul(acc.y, fe_mul(zi2, zi))));
}

// ================================================================================================
// Keyed path: when a batch re-uses public keys (gossip node ids, the 483 HTLC signatures of one
// commitment_signed share remote_htlckey -- channeld/channeld.c:2224-2225), each distinct key gets ONE
// table in HBM, shared by all of its signatures.  The 32 nibbles of a GLV half-scalar are cut into
// NPOS = 32/S chunks of S nibbles; the table holds d * 16^(c*S) * Q for chunk c = 0..NPOS (the last one only
// serves the recoding carry digit) and |digit| d = 1..8 as affine points (x | beta*x | y) of one per-key isomorphic
// curve (shared Z, no inversion needed to build it).  Evaluation is
// a comb: for j = S-1..0 { acc *= 16 (4 doublings, skipped first); add nibble c*S+j of every chunk }, i.e.
// 4(S-1) doublings and 66 mixed additions per verification instead of 132 + 66:
//   S = 1 (33 positions, 25 KiB/key): no doublings at all -- for heavily re-used keys (>= ~64 signatures/key)
//   S = 8 ( 5 positions,  4 KiB/key): 28 doublings, a 7x cheaper table -- pays from ~6 signatures per key
// Window width W = 4: signed nibbles of the bias-recoded half-scalar as stored in prep_rec (digits -8..7, 8 table entries
// per position, 32 nibbles + the recoding carry = 33 digits).  W = 5: the kernel re-biases |k| with 16 per 5-bit group
// (digits -16..15, 16 entries per position); |k| < 2^128 and the bias < 0.52 * 2^130 never carry out of 26 groups, so
// there is no carry digit: 26 digits dense, padded to 28 (two structurally zero digits) for the 4-position comb.
constexpr int kt_ne(int W) { return 1 << (W - 1); }                                   // table entries per position
constexpr int kt_ndigits(int W, int S) { return W == 4 ? 32 : (S == 1 ? 26 : 28); }   // digits handled by the chunks
constexpr int kt_npos(int W, int S) { return kt_ndigits(W, S) / S + (W == 4 ? 1 : 0); }  // + W = 4's carry position
constexpr int kt_words(int W, int S) { return kt_npos(W, S) * kt_ne(W) * SLOT_ENTRY_WORDS; }
constexpr int kt_stride(int W, int S) { return kt_words(W, S) + 16; }  // + Zc (8 words), 16-byte aligned
// scratch per key and position: base X, Y (18), Z_pos (9), prefix product / unify ratio (9), the H values of the chain
constexpr int kt_pos_scratch(int W) { return 36 + (kt_ne(W) - 2) * 8; }
constexpr int kt_scratch_words(int W, int S) { return kt_npos(W, S) * kt_pos_scratch(W); }
constexpr int KT_ZC_OFF = 0;  // word offset of Zc behind the entries

LAMD_HD void store_raw(u32 *dst, const fe &a) {
#pragma unroll
  for (int i = 0; i < 9; i++) dst[i] = a.n[i];
}

// Building one key's table, in four stages so that stages 2 and 4 can run one thread per (key, position):
//   1 kt_bases      per key       B_c = 2^(W*S*c) * Q, Jacobian, a chain of W*S doublings per position
//   2 kt_multiples  per position  1B..NE*B with (X, Y) of the Jacobian base taken as an affine point of the base's
//                                 isomorphic curve; Z_pos = (shared Z of the multiples) * Z_base
//   3 kt_prefix     per key       Zc = prod Z_pos and, per position, the product of the OTHER positions' Z
//   4 kt_rescale    per position  entry *= ratio^2 / ratio^3: every entry becomes an affine point of the ONE curve
//                                 y^2 = x^3 + 7*Zc^6 -- no inversion anywhere; the table-driven ecmult multiplies Zc back
//                                 into its accumulator's Z before the G additions
template <int W, int S>
LAMD_HD void kt_bases(u32 *scratch, const ge &q) {
  constexpr int NP = kt_npos(W, S), PS = kt_pos_scratch(W);
  gej b = gej_from_ge(q);
#pragma unroll 1
  for (int pos = 0; pos < NP; pos++) {
    if (pos) {
#pragma unroll 1
      for (int j = 0; j < W * S; j++) b = gej_double(b);
    }
    store_raw(scratch + pos * PS + 0, b.x);
    store_raw(scratch + pos * PS + 9, b.y);
    store_raw(scratch + pos * PS + 18, fe_norm_weak(b.z));
  }
}
template <int W, int S>
LAMD_HD void kt_multiples(u32 *tab, u32 *scratch, int pos) {
  constexpr int NE = kt_ne(W), PS = kt_pos_scratch(W);
  u32 *sp = scratch + pos * PS;
  ge base;
  base.x = slot_load_raw(sp + 0);
  base.y = slot_load_raw(sp + 9);
  const fe zg = build_multiples<NE>(tab + pos * NE * SLOT_ENTRY_WORDS, sp + 36, base);
  store_raw(sp + 18, fe_mul(zg, slot_load_raw(sp + 18)));
}
template <int W, int S>
LAMD_HD void kt_prefix(u32 *tab, u32 *scratch) {
  constexpr int NP = kt_npos(W, S), PS = kt_pos_scratch(W);
  fe acc = fe_set_int(1);
#pragma unroll 1
  for (int pos = 0; pos < NP; pos++) {
    store_raw(scratch + pos * PS + 27, acc);  // prefix
    acc = fe_mul(acc, slot_load_raw(scratch + pos * PS + 18));
  }
  slot_store_fe(tab + kt_words(W, S) + KT_ZC_OFF, acc);  // Zc
  fe suffix = fe_set_int(1);
#pragma unroll 1
  for (int pos = NP - 1; pos >= 0; pos--) {
    const fe ratio = fe_mul(suffix, slot_load_raw(scratch + pos * PS + 27));
    suffix = fe_mul(suffix, slot_load_raw(scratch + pos * PS + 18));
    store_raw(scratch + pos * PS + 27, ratio);
  }
}
template <int W, int S>
LAMD_HD void kt_rescale(u32 *tab, const u32 *scratch, int pos) {
  constexpr int NE = kt_ne(W), PS = kt_pos_scratch(W);
  const u32 betaw[8] = LAMD_BETA;
  const fe beta = fe_from_words(betaw);
  const fe ratio = slot_load_raw(scratch + pos * PS + 27);
  const fe r2 = fe_sqr(ratio);
  const fe r3 = fe_mul(r2, ratio);
#pragma unroll 1
  for (int e = 0; e < NE; e++) {
    u32 *ent = tab + (pos * NE + e) * SLOT_ENTRY_WORDS;
    const fe x = fe_mul(slot_load_fe(ent + 0), r2);
    const fe y = fe_mul(slot_load_fe(ent + 16), r3);
    slot_store_fe(ent + 0, x);
    slot_store_fe(ent + 8, fe_mul(x, beta));
    slot_store_fe(ent + 16, y);
  }
}
// sequential composition (CPU test harness; the engine launches the stages as separate kernels)
template <int W, int S>
LAMD_HD void keytable_build(u32 *tab, u32 *scratch, const ge &q) {
  kt_bases<W, S>(scratch, q);
  for (int pos = 0; pos < kt_npos(W, S); pos++) kt_multiples<W, S>(tab, scratch, pos);
  kt_prefix<W, S>(tab, scratch);
  for (int pos = 0; pos < kt_npos(W, S); pos++) kt_rescale<W, S>(tab, scratch, pos);
}

template <int NE>
LAMD_HD gej gej_add_table_digit(const gej &acc, const u32 *tab, int pos, int d, bool lambda_half) {
  const bool skip = d == 0;
  const int a = d < 0 ? -d : d;
  const u32 *e = tab + (pos * NE + (skip ? 0 : a - 1)) * SLOT_ENTRY_WORDS;
  ge pt;
  pt.x = slot_load_fe(e + (lambda_half ? 8 : 0));
  pt.y = slot_load_fe(e + 16);
  pt = ge_neg_if(pt, d < 0);
  return gej_add_ge(acc, pt, skip);
}

// 5-bit recoding of one GLV half from its prep_rec form: |k| = mag + top*2^128 - 0x88..8, then + 16 per 5-bit group
struct glv5 { u32 v[5]; };
LAMD_HD glv5 glv5_from_rec(const u32 mag[4], u32 top) {
  // 16 * sum_{i<28} 32^i as 32-bit words (140 bits)
  const u32 bias5[5] = {0x21084210u, 0x08421084u, 0x42108421u, 0x10842108u, 0x00000842u};
  glv5 r;
  u64 c = 0;
  u32 t[5];
  // t = mag + top*2^128 - 0x88888888 x4
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const u64 d = (u64)mag[i] - 0x88888888u - c;
    t[i] = (u32)d;
    c = (d >> 32) & 1;
  }
  t[4] = top - (u32)c;
  c = 0;
#pragma unroll
  for (int i = 0; i < 5; i++) {
    c += (u64)t[i] + bias5[i];
    r.v[i] = (u32)c;
    c >>= 32;
  }
  return r;
}
LAMD_HD int glv5_digit(const glv5 &g, int i) {
  const int bit = 5 * i, w = bit >> 5, sh = bit & 31;
  const u64 win = (u64)g.v[w] | ((u64)(w + 1 < 5 ? g.v[w + 1] : 0u) << 32);
  return (int)((win >> sh) & 31u) - 16;
}

// R = u1*G + (k1 + k2*lambda)*Q from the key's table
template <int W, int S>
LAMD_HD gej ecmult_lane_keyed(const prep_rec &rec, const u32 *tab, const u32 *gtable) {
  constexpr int ND = kt_ndigits(W, S), NC = ND / S, NE = kt_ne(W);
  const bool n1 = rec.flags & PREP_K1NEG, n2 = rec.flags & PREP_K2NEG;
  const u32 t1 = (rec.flags & PREP_K1TOP) ? 1u : 0u, t2 = (rec.flags & PREP_K2TOP) ? 1u : 0u;
  glv5 g1, g2;
  if (W == 5) {
    g1 = glv5_from_rec(rec.k1, t1);
    g2 = glv5_from_rec(rec.k2, t2);
  }
  gej acc = gej_infinity();
#pragma unroll 1
  for (int j = S - 1; j >= 0; j--) {
    if (j != S - 1) {
#pragma unroll 1
      for (int k = 0; k < W; k++) acc = gej_double(acc);
    }
    // W = 4: at j == 0 the extra position carries nibble 32 (the recoding carry)
#pragma unroll 1
    for (int c = 0; c < NC + ((W == 4 && j == 0) ? 1 : 0); c++) {
      const int i = c * S + j;
      if (W == 5 && i >= 26) continue;  // padding digits of the 28-digit comb are structurally zero
#pragma unroll 1
      for (int half = 0; half < 2; half++) {
        int d;
        if (W == 4) d = half ? glv_digit(rec.k2, t2, i) : glv_digit(rec.k1, t1, i);
        else d = half ? glv5_digit(g2, i) : glv5_digit(g1, i);
        if (half ? n2 : n1) d = -d;
        acc = gej_add_table_digit<NE>(acc, tab, c, d, half != 0);
      }
    }
  }
  // back from the table's isomorphic curve: (X, Y, Z) -> (X, Y, Z*Zc)
  acc.z = fe_mul(acc.z, slot_load_fe(tab + kt_words(W, S) + KT_ZC_OFF));
#pragma unroll 1
  for (int w = 0; w < GTABLE_WINDOWS; w++) {
    const u32 d = (rec.u1[(w * GTABLE_WINDOW_BITS) >> 5] >> ((w * GTABLE_WINDOW_BITS) & 31)) & ((1u << GTABLE_WINDOW_BITS) - 1u);
    const bool skip = d == 0;
    const u32 *e = gtable + (((size_t)w << GTABLE_WINDOW_BITS) + d) * 16;
    ge pt;
    pt.x = slot_load_fe(e);
    pt.y = slot_load_fe(e + 8);
    acc = gej_add_ge(acc, pt, skip);
  }
  return acc;
}

// p - n (129 bits): r + n < p  <=>  r < p - n
#define LAMD_P_MINUS_N {0x2FC9BAEEu, 0x402DA172u, 0x50B75FC4u, 0x45512319u, 1u, 0u, 0u, 0u}

// ECDSA acceptance: R != inf and x(R) mod n == r, tested without inversion
LAMD_HD bool ecdsa_final(const gej &R, const u32 rw[8]) {
  if (R.inf) return false;
  const fe z2 = fe_sqr(R.z);
  const fe rf = fe_from_words(rw);  // r < n < p
  bool ok = fe_equal(fe_mul(rf, z2), R.x, 1);
  const u32 pmn[8] = LAMD_P_MINUS_N;
  if (!words_ge(rw, pmn)) {  // r + n < p: x(R) may also be r + n
    const u32 nw[8] = LAMD_SC_N;
    u32 t[8];
    u64 c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      c += (u64)rw[i] + nw[i];
      t[i] = (u32)c;
      c >>= 32;
    }
    ok |= fe_equal(fe_mul(fe_from_words(t), z2), R.x, 1);
  }
  return ok;
}

// BIP-340 acceptance, split so that the one field inversion it needs (y = Y/Z^3 for the parity test) can be shared:
//   stage 1 (per lane, in the ecmult kernel): R != inf and x(R) == r tested as r*Z^2 == X; survivors park Y and Z
//            (raw limbs) in their table slot and report SCHNORR_PENDING
//   stage 2 (k_schnorr_final): a thread owns signatures i, i+T, ... and inverts all their pending Z with ONE
//            inversion (Montgomery's trick), then tests the parity of Y * Z^-3
constexpr u8 SCHNORR_PENDING = 2;
constexpr int SLOT_FIN_Y = 0, SLOT_FIN_Z = 9, SLOT_FIN_PREFIX = 18;  // word offsets inside the lane's slot

LAMD_HD u8 schnorr_stage1(const gej &R, const u32 rw[8], u32 *slot) {
  if (R.inf) return 0;
  const fe z2 = fe_sqr(R.z);
  const fe rf = fe_from_words(rw);  // r < p checked in prep
  if (!fe_equal(fe_mul(rf, z2), R.x, 1)) return 0;
  const fe z = fe_norm_weak(R.z);
#pragma unroll
  for (int i = 0; i < 9; i++) {
    slot[SLOT_FIN_Y + i] = R.y.n[i];
    slot[SLOT_FIN_Z + i] = z.n[i];
  }
  return SCHNORR_PENDING;
}
LAMD_HD void schnorr_final_thread(size_t first, size_t stride, size_t n, u32 *slots, u8 *out, size_t slot_words = SLOT_WORDS) {
  fe acc = fe_set_int(1);
  size_t last = first;
  bool any = false;
#pragma unroll 1
  for (size_t i = first; i < n; i += stride) {
    if (out[i] != SCHNORR_PENDING) continue;
    u32 *slot = slots + i * slot_words;
#pragma unroll
    for (int k = 0; k < 9; k++) slot[SLOT_FIN_PREFIX + k] = acc.n[k];  // product of the pending Z before i
    acc = fe_mul(acc, slot_load_raw(slot + SLOT_FIN_Z));
    last = i;
    any = true;
  }
  if (!any) return;
  fe inv = fe_inv(acc);
#pragma unroll 1
  for (size_t i = last;; i -= stride) {
    if (out[i] == SCHNORR_PENDING) {
      const u32 *slot = slots + i * slot_words;
      const fe zi = fe_mul(inv, slot_load_raw(slot + SLOT_FIN_PREFIX));
      inv = fe_mul(inv, slot_load_raw(slot + SLOT_FIN_Z));
      const fe y = fe_normalize(fe_mul(slot_load_raw(slot + SLOT_FIN_Y), fe_mul(fe_sqr(zi), zi)));
      out[i] = (y.n[0] & 1) == 0;
    }
    if (i == first) break;
  }
}

// BIP-340 acceptance in one piece (single-lane form used by the self-test): R != inf, y(R) even, x(R) == r
LAMD_HD bool schnorr_final(const gej &R, const u32 rw[8]) {
  if (R.inf) return false;
  const fe z2 = fe_sqr(R.z);
  const fe rf = fe_from_words(rw);  // r < p checked in prep
  if (!fe_equal(fe_mul(rf, z2), R.x, 1)) return false;
  const fe zi = fe_inv(fe_norm_weak(R.z));
  const fe y = fe_normalize(fe_mul(R.y, fe_mul(fe_sqr(zi), zi)));
  return (y.n[0] & 1) == 0;
}

}  // namespace lamd

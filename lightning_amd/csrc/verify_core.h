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
//            then 11 lookups in the 24-bit-window table of G; final x (and y-parity) check
//
// Reference semantics: secp256k1_ecdsa_verify / secp256k1_schnorrsig_verify /
// secp256k1_ec_pubkey_parse / secp256k1_xonly_pubkey_parse as called from
// bitcoin/signature.c:188,422,425, common/node_id.c:24, bitcoin/pubkey.c:19.
#pragma once
#include "group.h"
#include "scalar.h"
#include "sha256.h"

namespace lamd {

enum { MODE_ECDSA = 0, MODE_SCHNORR = 1, MODE_RECOVER = 2 };

// ---- field inversion by division steps (scalar.h s30_inverse with the modulus p): ~10^4 instructions against the 255 squarings +
// 15 multiplications of fe_inv.  Used where ONE inversion is shared by a group of rows (BIP-340 parity stage, key recovery) or sits
// on a small batch's latency path; variable time (public data).
#define LAMD_S30_P {0x3FFFFC2F, 0x3FFFFFFB, 0x3FFFFFFF, 0x3FFFFFFF, 0x3FFFFFFF, 0x3FFFFFFF, 0x3FFFFFFF, 0x3FFFFFFF, 0xFFFF}
LAMD_HD fe fe_inv_var(const fe &a) {
  const int32_t pl[9] = LAMD_S30_P;
  u32 aw[8], w[9];
  fe_to_words(aw, fe_normalize(a));
  s30_inverse(w, aw, pl, 0x2DDACACFu);  // p^-1 mod 2^30
  // + 32 p = 2^261 - 2^37 - 32 * 977, modulo 2^288: the sum is in [0, 64 p)
  const u32 p32[9] = {0xFFFF85E0u, 0xFFFFFFDFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0x1Fu};
  u64 cy = 0;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    cy += (u64)w[i] + p32[i];
    w[i] = (u32)cy;
    cy >>= 32;
  }
  // fold the bits above 2^256 (2^256 = 2^32 + 977 mod p), twice: q < 64 first, then the carry of that addition
  u32 q = w[8];
#pragma unroll
  for (int round = 0; round < 2; round++) {
    u64 k = (u64)w[0] + (u64)q * 977u;
    w[0] = (u32)k;
    k = (k >> 32) + (u64)w[1] + q;
    w[1] = (u32)k;
    k >>= 32;
#pragma unroll
    for (int i = 2; i < 8; i++) {
      k += w[i];
      w[i] = (u32)k;
      k >>= 32;
    }
    q = (u32)k;
  }
  return fe_from_words(w);  // any value below 2^256 is a magnitude-1 element
}

// ---- per-signature record produced by prep, consumed by ecmult (80 bytes, 16-byte aligned)
struct prep_rec {
  u32 u1[8];   // scalar for G (little-endian words)
  u32 k1[4];   // |k1| + 0x88..8  (see glv_half)
  u32 k2[4];
  u32 flags;   // PREP_*
  u32 pad[3];
};
enum { PREP_VALID = 1, PREP_K1NEG = 2, PREP_K2NEG = 4, PREP_K1TOP = 8, PREP_K2TOP = 16 };

// Table entries (the ladder's per-lane table, the per-key combs, the static G table) hold their coordinates as 8 x 32-bit words
// (canonical) and the loads re-pack them into the nine 29-bit limbs the field arithmetic computes in: ~16 VALU instructions per
// coordinate, 98 coordinates per signature = 1.5 % of the table-driven kernel's instructions.  LAMD_TABLE_LIMBS=1 stores the nine
// limbs instead (one dword more per coordinate, no re-packing).  Measured on MI355X (round 3, tools/runs/next_ab_variants.sh,
// profiles/r03_ab_variants.txt): the limb layout is NOT faster -- isolated 1 M-row launch 3.31 / 3.37 ms against 3.29 / 3.28 ms for
// words: 72-byte G entries straddle 128-byte lines, 112-byte comb entries move 17 % more bytes, and memory waits (10 % of the wave
// cycles) grow by as much as the VALU work shrinks.  Words ship; the limb layout stays buildable and tested (host build).
#ifndef LAMD_TABLE_LIMBS
#define LAMD_TABLE_LIMBS 0
#endif
constexpr int TW = LAMD_TABLE_LIMBS ? 9 : 8;                         // words per stored coordinate
constexpr int SLOT_ENTRY_WORDS = LAMD_TABLE_LIMBS ? 28 : 24;         // x | beta*x | y (+ 1 word of padding: 16-byte aligned entries)
constexpr int ENT_X = 0, ENT_BX = TW, ENT_Y = 2 * TW;                // word offsets inside an entry
constexpr int GT_ENTRY_WORDS = 2 * TW;                               // static G table: x | y
constexpr int SLOT_H_OFF = 8 * SLOT_ENTRY_WORDS;                     // 6 x TW words of H_2..H_7
constexpr int SLOT_WORDS = LAMD_TABLE_LIMBS ? 288 : 256;             // scratch slot owned by one ladder lane

// Static table of G: window w, digit d -> d * 2^(BITS*w) * G as 64-byte affine words (d = 0 unused).
// 24-bit windows (11 windows, 11 GiB of the card's 288 GB of HBM) make u1*G eleven mixed additions.  Rounds 1-3 shipped 22 bits (12 windows,
// 3 GiB); the step is VALU instruction issue end to end, so one addition less (1 of 67 group operations) is 1.5 % of the dominant kernel:
// three alternating pairs of bench.py --ab on one box, 24 against 22 bits: cold 239.5 / 236.8 / 236.2 against 233.8 / 231.3 / 235.4 M verifies/s,
// chained launch 4.07 against 4.10-4.14 ms (profiles/r04_ab_variants.txt), and one random 64-byte read less per signature.  26 bits
// (10 windows, 43 GB) measured another 1 % in round 2 -- not worth 4x the memory.  16-bit windows (64 MiB, Infinity-Cache resident, 16 additions)
// are 4 % slower than 22: one random 64-byte read per ~1 400 instructions hides.
// The CPU test harness builds the same code with 8-bit windows to keep its table small.
#ifndef LAMD_GTABLE_WINDOW_BITS
#define LAMD_GTABLE_WINDOW_BITS 24
#endif
#ifndef LAMD_G_RUN_XYZZ
#define LAMD_G_RUN_XYZZ 1
#endif
#ifndef LAMD_UNROLL_HALF
#define LAMD_UNROLL_HALF 0
#endif
constexpr int GTABLE_WINDOW_BITS = LAMD_GTABLE_WINDOW_BITS;
constexpr int GTABLE_WINDOWS = (256 + GTABLE_WINDOW_BITS - 1) / GTABLE_WINDOW_BITS;
constexpr size_t GTABLE_ENTRIES = (size_t)GTABLE_WINDOWS << GTABLE_WINDOW_BITS;
constexpr size_t GTABLE_BYTES = GTABLE_ENTRIES * GT_ENTRY_WORDS * 4;
// window w of a 256-bit scalar (8 little-endian words); windows may straddle a word and run past bit 255
LAMD_HD u32 gtable_digit(const u32 k[8], int w) {
  const int bit = w * GTABLE_WINDOW_BITS, i = bit >> 5, sh = bit & 31;
  u32 v = k[i] >> sh;
  if ((32 % GTABLE_WINDOW_BITS) != 0 && sh + GTABLE_WINDOW_BITS > 32 && i + 1 < 8) v |= k[i + 1] << (32 - sh);
  return v & ((1u << GTABLE_WINDOW_BITS) - 1u);
}

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

// ---- seeded mixing (hash-table probes of the key de-duplication / cache, synthetic test data)
LAMD_HD u64 splitmix64(u64 x) {
  x += 0x9E3779B97F4A7C15ULL;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
  return x ^ (x >> 31);
}
// 8 little-endian words -> 32 big-endian bytes
LAMD_HD void store_words_be(u8 *dst, const u32 w[8]) {
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const u32 v = w[7 - i];
    dst[4 * i] = (u8)(v >> 24); dst[4 * i + 1] = (u8)(v >> 16); dst[4 * i + 2] = (u8)(v >> 8); dst[4 * i + 3] = (u8)v;
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

// Cheap pre-test of a signature's two scalars, used by the row-list builders (early reject): true = the preparation kernel is
// certain to reject this row (ECDSA: r or s outside [1, n-1], or s > n/2 -- secp256k1_ecdsa_signature_parse_compact + low-S rule;
// BIP-340: r >= p or s >= n), so it needs no lane of an ecmult wave.  Says nothing about rows it lets through.
LAMD_HD bool sig_certain_reject(const u8 *sig64, int mode) {
  u32 rw[8], sw[8];
  load_words_be(rw, sig64);
  load_words_be(sw, sig64 + 32);
  if (mode == MODE_SCHNORR) return words_ge_p(rw) | words_ge_n(sw);
  u32 rz = 0, sz = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) { rz |= rw[i]; sz |= sw[i]; }
  sc s;
#pragma unroll
  for (int i = 0; i < 8; i++) s.w[i] = sw[i];
  return words_ge_n(rw) | words_ge_n(sw) | (rz == 0) | (sz == 0) | sc_is_high(s);
}

LAMD_HD void ecdsa_prep_thread(size_t first, size_t stride, size_t n, const u8 *hash32, const u8 *sig64,
                               prep_rec *recs) {
  // prefix products, the inversion and u1, u2 run in the 9x29 representation (scalar.h); the record of row i keeps the prefix
  // product between the passes in its first nine words (u1[0..7], k1[0])
  sc29 acc = sc29_one();
  size_t last = first;
  bool any = false;
#pragma unroll 1
  for (size_t i = first; i < n; i += stride) {
    sc r, s;
    bool ok;
    ecdsa_load_rs(sig64 + 64 * i, &r, &s, &ok);
#pragma unroll
    for (int k = 0; k < 8; k++) recs[i].u1[k] = acc.n[k];  // product of the valid s before i
    recs[i].k1[0] = acc.n[8];
    if (ok) acc = sc29_mul(acc, sc29_from_sc(s));
    last = i;
    any = true;
  }
  if (!any) return;
  sc29 inv = sc29_from_sc(sc_inv_var(sc29_to_sc(acc)));  // division steps: ~12x fewer instructions than a^(n-2)
#pragma unroll 1
  for (size_t i = last;; i -= stride) {
    sc r, s;
    sc29 prefix;
    bool ok;
    ecdsa_load_rs(sig64 + 64 * i, &r, &s, &ok);
#pragma unroll
    for (int k = 0; k < 8; k++) prefix.n[k] = recs[i].u1[k];
    prefix.n[8] = recs[i].k1[0];
    prep_rec out;
#pragma unroll
    for (int k = 0; k < 8; k++) out.u1[k] = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) { out.k1[k] = 0x88888888u; out.k2[k] = 0x88888888u; }
    out.flags = 0;
    out.pad[0] = out.pad[1] = out.pad[2] = 0;
    if (ok) {
      const sc29 w = sc29_mul(inv, prefix);  // s_i^-1
      inv = sc29_mul(inv, sc29_from_sc(s));
      u32 zw[8];
      load_words_be(zw, hash32 + 32 * i);    // any 256-bit value: reduced by the multiplication
      const sc u1 = sc29_to_sc(sc29_mul(sc29_from_words(zw), w));
      const sc u2 = sc29_to_sc(sc29_mul(sc29_from_sc(r), w));
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
  const fe n = fe_normalize(a);
  if (LAMD_TABLE_LIMBS) {
#pragma unroll
    for (int i = 0; i < 9; i++) dst[i] = n.n[i];
  } else {
    u32 w[8];
    fe_to_words(w, n);
#pragma unroll
    for (int i = 0; i < 8; i++) dst[i] = w[i];
  }
}
LAMD_HD fe slot_load_raw(const u32 *src) {
  fe r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.n[i] = src[i];
  FE_SETMAG(r, 1);
  return r;
}
LAMD_HD fe slot_load_fe(const u32 *src) {
  if (LAMD_TABLE_LIMBS) return slot_load_raw(src);
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
  slot_store_fe(slot + ENT_Y, p.y);
  p = gej_double(p);
  slot_store_fe(slot + SLOT_ENTRY_WORDS + 0, p.x);
  slot_store_fe(slot + SLOT_ENTRY_WORDS + ENT_Y, p.y);
#pragma unroll 1
  for (int i = 2; i < NE; i++) {  // entry i = (i+1)Q = entry(i-1) + Q
    bool degenerate;
    fe h, rr;
    p = gej_add_ge_core(p, q, &degenerate, &h, &rr);  // (i)Q = +-Q is impossible for i in 2..NE: no degenerate case
    slot_store_fe(slot + i * SLOT_ENTRY_WORDS + 0, p.x);
    slot_store_fe(slot + i * SLOT_ENTRY_WORDS + ENT_Y, p.y);
    slot_store_fe(hbuf + (i - 2) * TW, h);
  }
  const fe zg = fe_norm_weak(p.z);
  const u32 betaw[8] = LAMD_BETA;
  const fe beta = fe_from_words(betaw);
  // the last entry already has Z = Zg
  {
    const fe x = slot_load_fe(slot + (NE - 1) * SLOT_ENTRY_WORDS);
    slot_store_fe(slot + (NE - 1) * SLOT_ENTRY_WORDS + ENT_BX, fe_mul(x, beta));
  }
  fe rho = fe_set_int(1);
#pragma unroll 1
  for (int i = NE - 2; i >= 0; i--) {
    // rho = Zg / Z_entry(i): entry i+1 = entry i + Q had Z_{i+1} = Z_i * H (H stored at index i-1 for i >= 1),
    // and entry 1 = 2Q has Z = Z_2, entry 0 = Q has Z = 1 so its ratio is Zg itself.
    if (i >= 1) rho = fe_mul(rho, slot_load_fe(hbuf + (i - 1) * TW));
    else rho = zg;
    const fe r2 = fe_sqr(rho);
    const fe r3 = fe_mul(r2, rho);
    const fe x = fe_mul(slot_load_fe(slot + i * SLOT_ENTRY_WORDS + 0), r2);
    const fe y = fe_mul(slot_load_fe(slot + i * SLOT_ENTRY_WORDS + ENT_Y), r3);
    slot_store_fe(slot + i * SLOT_ENTRY_WORDS + 0, x);
    slot_store_fe(slot + i * SLOT_ENTRY_WORDS + ENT_BX, fe_mul(x, beta));
    slot_store_fe(slot + i * SLOT_ENTRY_WORDS + ENT_Y, y);
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
      pt.x = slot_load_fe(e + (half ? ENT_BX : ENT_X));
      pt.y = slot_load_fe(e + ENT_Y);
      pt = ge_neg_if(pt, d < 0);
      acc = gej_add_ge(acc, pt, skip);
    }
  }
  // back from the isomorphic curve: (X, Y, Z) -> (X, Y, Z*Zg)
  acc.z = fe_mul(acc.z, zg);
  // + u1*G from the window table of G
#pragma unroll 1
  for (int w = 0; w < GTABLE_WINDOWS; w++) {
    const u32 d = gtable_digit(rec.u1, w);
    const bool skip = d == 0;
    const u32 *e = gtable + (((size_t)w << GTABLE_WINDOW_BITS) + d) * GT_ENTRY_WORDS;
    ge pt;
    pt.x = slot_load_fe(e);
    pt.y = slot_load_fe(e + TW);
    acc = gej_add_ge(acc, pt, skip);
  }
  return acc;
}

// One entry of the static G table: out = d * B (B = 2^(BITS*w) * G given as affine words), d >= 1.
LAMD_HD void gtable_compute_entry(u32 out[GT_ENTRY_WORDS], const u32 base[16], u32 d) {
  const ge b = ge_from_words(base, base + 8);
  gej acc = gej_infinity();
#pragma unroll 1
  for (int bit = GTABLE_WINDOW_BITS - 1; bit >= 0; bit--) {
    acc = gej_double(acc);
    acc = gej_add_ge(acc, b, ((d >> bit) & 1u) == 0);
  }
  const fe zi = fe_inv_var(acc.z);
  const fe zi2 = fe_sqr(zi);
  slot_store_fe(out, fe_mul(acc.x, zi2));
  slot_store_fe(out + TW, fe_mul(acc.y, fe_mul(zi2, zi)));
}

// ================================================================================================
// Keyed path: when a batch re-uses public keys (gossip node ids, the 483 HTLC signatures of one
// commitment_signed share remote_htlckey -- channeld/channeld.c:2224-2225), each distinct key gets ONE
// table in HBM, shared by all of its signatures: a signed comb (Lim-Lee) with T teeth spaced D bits apart.
//
// A GLV half-scalar |k| < 2^129 is made odd (k' = |k| | 1; an even |k| is repaired afterwards by one predicated
// addition of -+Q) and written with N = T*D digits s_i in {-1, +1}: k' = sum s_i 2^i with s_i = 2*b_i - 1 and b the
// bits of w = (|k| >> 1) | 2^(N-1).  Column j (0 <= j < D) gathers digits j, j+D, .., j+(T-1)D:
//      sum_i s_(iD+j) * B_i,  B_i = 2^(iD) * Q,
// which is +-(B_(T-1) + sum_(i<T-1) +-B_i): one of NE = 2^(T-1) table entries, negated when the top tooth is -1.
//      acc = inf; for j = D-1..0 { acc = 2*acc; acc += column_j(k1) ; acc += lambda*column_j(k2) }
// costs 2*D mixed additions and D-1 doublings for BOTH halves -- T = 7: 38 + 18 (was 56 + 30 with 5-bit windows at
// the same 64 entries per key), T = 10: 26 + 12 with 512 entries for heavily re-used keys.
// Entries are affine points (x | beta*x | y) of one per-key isomorphic curve y^2 = x^3 + 7*Zc^6 (shared Z, no
// inversion to build them); entry m = B_(T-1) + sum_(i<T-1) (m_i ? +B_i : -B_i); entry NE is Q itself.
constexpr int kc_spacing(int T) { return T == 7 ? 19 : T == 8 ? 17 : T == 9 ? 15 : T == 10 ? 13 : 12; }  // T*D >= 129
constexpr int kc_ne(int T) { return 1 << (T - 1); }
constexpr int KC_SUB_LOG = 4, KC_SUB = 1 << KC_SUB_LOG;            // entries built by one thread (a Gray-code chain)
constexpr int kc_nsub(int T) { return kc_ne(T) / KC_SUB; }
constexpr int kc_words(int T) { return (kc_ne(T) + 1) * SLOT_ENTRY_WORDS; }
constexpr int kc_stride(int T) { return kc_words(T) + 16; }        // + Zc (TW words), keeps 16-byte alignment
// scratch per key (words): P0 = entry 0 as a Jacobian point (27) | Zb (9) | C_i = 2*B_i, i < T-1, affine (18 each) |
// per chain: Z_chain (9), unify ratio (9), the 15 H values of its additions (9 each).  Stage 1 parks its 2T-1
// Jacobian points (36 words each) in the per-chain area before any chain uses it.
constexpr int KC_P0 = 0, KC_ZB = 27, KC_C = 36;
constexpr int kc_sub_off(int T) { return KC_C + (T - 1) * 18; }
constexpr int KC_SUB_WORDS = 18 + (KC_SUB - 1) * 9;
constexpr int kc_scratch_words(int T) { return kc_sub_off(T) + kc_nsub(T) * KC_SUB_WORDS; }
static_assert(kc_nsub(7) * KC_SUB_WORDS >= 13 * 36 && kc_nsub(8) * KC_SUB_WORDS >= 15 * 36, "stage-1 parking space");

LAMD_HD void store_raw(u32 *dst, const fe &a) {
#pragma unroll
  for (int i = 0; i < 9; i++) dst[i] = a.n[i];
}

// Building one key's table, in four stages so that stages 2 and 4 can run one thread per (key, chain):
//   1 kc_bases    per key    the doubling chain Q -> 2^((T-1)D) Q, picking up B_i and C_i = 2*B_i on the way; all of
//                            them brought to one Z (Zb) by products of the others' Z -- no inversion; P0 = B_(T-1) - sum B_i
//   2 kc_chain_fwd per chain 16 entries in Gray-code order from P0 + (the chain's fixed high teeth): one mixed addition of
//                            +-C_i per entry (Jacobian X, Y parked in the table, the H values in scratch)
//   3 kc_prefix   per key    Zc = Zb * prod Z_chain, per chain the product of the OTHER chains' Z; the entry of Q
//   4 kc_chain_bwd per chain walks the chain backwards multiplying up Zc / (Zb * Z_entry) from the H values:
//                            entry *= ratio^2 / ratio^3 and beta*x -- every entry an affine point of y^2 = x^3 + 7*Zc^6
// No addition here can be degenerate: every operand is (odd integer < 2^136)*Q +- (even integer < 2^136)*Q with Q of
// prime order n > 2^255.
template <int T>
LAMD_HD void kc_bases(u32 *scratch, const ge &q) {
  constexpr int D = kc_spacing(T), NPT = 2 * T - 1;
  u32 *tmp = scratch + kc_sub_off(T);  // point 2i = B_i, point 2i+1 = C_i; x | y | z | prefix
  gej b = gej_from_ge(q);
#pragma unroll 1
  for (int i = 0; i < NPT; i++) {
    store_raw(tmp + i * 36 + 0, b.x);
    store_raw(tmp + i * 36 + 9, b.y);
    store_raw(tmp + i * 36 + 18, fe_norm_weak(b.z));
    if (i == NPT - 1) break;
    const int nd = (i & 1) ? D - 1 : 1;
#pragma unroll 1
    for (int j = 0; j < nd; j++) b = gej_double(b);
  }
  fe acc = fe_set_int(1);
#pragma unroll 1
  for (int i = 0; i < NPT; i++) {
    store_raw(tmp + i * 36 + 27, acc);
    acc = fe_mul(acc, slot_load_raw(tmp + i * 36 + 18));
  }
  store_raw(scratch + KC_ZB, acc);
  fe suffix = fe_set_int(1);
#pragma unroll 1
  for (int i = NPT - 1; i >= 0; i--) {
    const fe ratio = fe_mul(suffix, slot_load_raw(tmp + i * 36 + 27));
    suffix = fe_mul(suffix, slot_load_raw(tmp + i * 36 + 18));
    const fe r2 = fe_sqr(ratio);
    const fe x = fe_mul(slot_load_raw(tmp + i * 36 + 0), r2);
    const fe y = fe_mul(slot_load_raw(tmp + i * 36 + 9), fe_mul(r2, ratio));
    u32 *dst = (i & 1) ? scratch + KC_C + (i >> 1) * 18 : tmp + i * 36;
    store_raw(dst + 0, x);
    store_raw(dst + 9, y);
  }
  gej p;
  p.x = slot_load_raw(tmp + (NPT - 1) * 36 + 0);
  p.y = slot_load_raw(tmp + (NPT - 1) * 36 + 9);
  p.z = fe_set_int(1);
  p.inf = false;
#pragma unroll 1
  for (int i = 0; i < T - 1; i++) {
    ge m;
    m.x = slot_load_raw(tmp + 2 * i * 36 + 0);
    m.y = fe_norm_weak(fe_neg(slot_load_raw(tmp + 2 * i * 36 + 9), 1));
    bool degenerate;
    fe h, rr;
    p = gej_add_ge_core(p, m, &degenerate, &h, &rr);
  }
  store_raw(scratch + KC_P0 + 0, p.x);
  store_raw(scratch + KC_P0 + 9, p.y);
  store_raw(scratch + KC_P0 + 18, fe_norm_weak(p.z));
}
template <int T>
LAMD_HD void kc_chain_fwd(u32 *tab, u32 *scratch, int sub) {
  u32 *sp = scratch + kc_sub_off(T) + sub * KC_SUB_WORDS;
  u32 *ent = tab + sub * KC_SUB * SLOT_ENTRY_WORDS;
  gej p;
  p.x = slot_load_raw(scratch + KC_P0 + 0);
  p.y = slot_load_raw(scratch + KC_P0 + 9);
  p.z = slot_load_raw(scratch + KC_P0 + 18);
  p.inf = false;
  bool degenerate;
  fe h, rr;
#pragma unroll 1
  for (int b = 0; b < T - 1 - KC_SUB_LOG; b++) {  // the chain's fixed high teeth
    if ((sub >> b) & 1) {
      ge c;
      c.x = slot_load_raw(scratch + KC_C + (KC_SUB_LOG + b) * 18 + 0);
      c.y = slot_load_raw(scratch + KC_C + (KC_SUB_LOG + b) * 18 + 9);
      p = gej_add_ge_core(p, c, &degenerate, &h, &rr);
      p.z = fe_norm_weak(p.z);
    }
  }
  slot_store_fe(ent + 0, p.x);
  slot_store_fe(ent + ENT_Y, p.y);
#pragma unroll 1
  for (int g = 1; g < KC_SUB; g++) {
    const int c = __builtin_ctz((unsigned)g), gr = g ^ (g >> 1);
    ge pt;
    pt.x = slot_load_raw(scratch + KC_C + c * 18 + 0);
    pt.y = slot_load_raw(scratch + KC_C + c * 18 + 9);
    pt = ge_neg_if(pt, ((gr >> c) & 1) == 0);  // tooth c goes - to +: add 2*B_c; + to -: subtract it
    p = gej_add_ge_core(p, pt, &degenerate, &h, &rr);
    slot_store_fe(ent + gr * SLOT_ENTRY_WORDS + 0, p.x);
    slot_store_fe(ent + gr * SLOT_ENTRY_WORDS + ENT_Y, p.y);
    store_raw(sp + 18 + (g - 1) * 9, h);
  }
  store_raw(sp + 0, fe_norm_weak(p.z));
}
template <int T>
LAMD_HD void kc_prefix(u32 *tab, u32 *scratch, const ge &q) {
  constexpr int NS = kc_nsub(T), NE = kc_ne(T);
  u32 *sp = scratch + kc_sub_off(T);
  fe acc = fe_set_int(1);
#pragma unroll 1
  for (int s = 0; s < NS; s++) {
    store_raw(sp + s * KC_SUB_WORDS + 9, acc);
    acc = fe_mul(acc, slot_load_raw(sp + s * KC_SUB_WORDS + 0));
  }
  const fe zc = fe_mul(acc, slot_load_raw(scratch + KC_ZB));
  slot_store_fe(tab + kc_words(T), zc);
  fe suffix = fe_set_int(1);
#pragma unroll 1
  for (int s = NS - 1; s >= 0; s--) {
    const fe ratio = fe_mul(suffix, slot_load_raw(sp + s * KC_SUB_WORDS + 9));
    suffix = fe_mul(suffix, slot_load_raw(sp + s * KC_SUB_WORDS + 0));
    store_raw(sp + s * KC_SUB_WORDS + 9, ratio);
  }
  const u32 betaw[8] = LAMD_BETA;
  const fe z2 = fe_sqr(zc);
  const fe x = fe_mul(q.x, z2);
  u32 *e = tab + NE * SLOT_ENTRY_WORDS;
  slot_store_fe(e + 0, x);
  slot_store_fe(e + ENT_BX, fe_mul(x, fe_from_words(betaw)));
  slot_store_fe(e + ENT_Y, fe_mul(q.y, fe_mul(z2, zc)));
}
template <int T>
LAMD_HD void kc_chain_bwd(u32 *tab, const u32 *scratch, int sub) {
  const u32 betaw[8] = LAMD_BETA;
  const fe beta = fe_from_words(betaw);
  const u32 *sp = scratch + kc_sub_off(T) + sub * KC_SUB_WORDS;
  u32 *ent = tab + sub * KC_SUB * SLOT_ENTRY_WORDS;
  fe rho = slot_load_raw(sp + 9);  // Zc / (Zb * Z_chain): the last entry's ratio
#pragma unroll 1
  for (int g = KC_SUB - 1; g >= 0; g--) {  // rho = Zc / (Zb * Z_entry(g)): entry g+1 = entry g +- C had Z_(g+1) = Z_g * H_(g+1)
    if (g != KC_SUB - 1) rho = fe_mul(rho, slot_load_raw(sp + 18 + g * 9));
    const int gr = g ^ (g >> 1);
    const fe r2 = fe_sqr(rho);
    const fe x = fe_mul(slot_load_fe(ent + gr * SLOT_ENTRY_WORDS + 0), r2);
    const fe y = fe_mul(slot_load_fe(ent + gr * SLOT_ENTRY_WORDS + ENT_Y), fe_mul(r2, rho));
    slot_store_fe(ent + gr * SLOT_ENTRY_WORDS + 0, x);
    slot_store_fe(ent + gr * SLOT_ENTRY_WORDS + ENT_BX, fe_mul(x, beta));
    slot_store_fe(ent + gr * SLOT_ENTRY_WORDS + ENT_Y, y);
  }
}
// sequential composition (CPU test harness; the engine launches the stages as separate kernels)
template <int T>
LAMD_HD void keytable_build(u32 *tab, u32 *scratch, const ge &q) {
  kc_bases<T>(scratch, q);
  for (int s = 0; s < kc_nsub(T); s++) kc_chain_fwd<T>(tab, scratch, s);
  kc_prefix<T>(tab, scratch, q);
  for (int s = 0; s < kc_nsub(T); s++) kc_chain_bwd<T>(tab, scratch, s);
}

// ---- The tree builder (round 6): the 2^(T-1) entries of a key as a doubling TREE of AFFINE additions with the inversions shared.
// Stage 1 (kc_bases) leaves P0 (Jacobian) and C_c = 2*B_c (affine) on the key's isomorphic curve E_b: y^2 = x^3 + 7*Zb^6 -- a curve whose
// coefficient no addition formula reads.  Entry m = P0 + sum over the set bits c of m of C_c, so with S_0 = {P0} level c makes
// S_(c+1) = S_c + (S_c + C_c): 2^c independent additions of ONE affine point, T - 1 levels, 2^(T-1) - 1 additions -- as many as the
// Gray-code chains make, but
//   * affine: 1 inversion + 2M + 1S per addition, the inversions of a whole level -- of every key the lane owns -- by Montgomery's trick
//     (3M per addition + ONE division-step inversion per level), so 6M + 1S per entry with its beta*x against 8M + 3S for the mixed addition;
//   * the entries come out affine on E_b with ONE Z (Zc = Zb): no Z products, no backward rescale (kc_prefix / kc_chain_bwd: 5M + 1S per entry).
// A lane owns `nk` keys (7 teeth: four, so that a level's inversion of ~1.3e4 instructions is shared by 4 * 2^c additions; 10 teeth: one -- its
// levels hold up to 256).  The prefix products of a level are parked in the keys' per-chain scratch areas (free after stage 1).
// No addition can be degenerate (see above: distinct odd/even multiples of a point of prime order), so no denominator is zero.
constexpr int KC_TREE_MAXK = 4;
struct kc_tree_keys {
  u32 *tab[KC_TREE_MAXK];
  u32 *scr[KC_TREE_MAXK];
  int n;
};
template <int T>
LAMD_HD void kc_tree_affine(const kc_tree_keys &K, const ge *qs) {
  constexpr int NE = kc_ne(T);
  static_assert(kc_nsub(T) * KC_SUB_WORDS >= (NE / 2) * 9, "prefix products of the last level fit the per-chain scratch");
  const u32 betaw[8] = LAMD_BETA;
  const fe beta = fe_from_words(betaw);
  // P0 of every key -> affine on its curve: one inversion for the lane
  fe acc = fe_set_int(1);
#pragma unroll 1
  for (int k = 0; k < K.n; k++) {
    store_raw(K.scr[k] + kc_sub_off(T), acc);
    acc = fe_mul(acc, slot_load_raw(K.scr[k] + KC_P0 + 18));
  }
  fe inv = fe_inv_var(acc);
#pragma unroll 1
  for (int k = K.n - 1; k >= 0; k--) {
    const fe zi = fe_mul(inv, slot_load_raw(K.scr[k] + kc_sub_off(T)));
    inv = fe_mul(inv, slot_load_raw(K.scr[k] + KC_P0 + 18));
    const fe zi2 = fe_sqr(zi);
    const fe x = fe_mul(slot_load_raw(K.scr[k] + KC_P0 + 0), zi2);
    const fe y = fe_mul(slot_load_raw(K.scr[k] + KC_P0 + 9), fe_mul(zi2, zi));
    slot_store_fe(K.tab[k] + ENT_X, x);
    slot_store_fe(K.tab[k] + ENT_BX, fe_mul(x, beta));
    slot_store_fe(K.tab[k] + ENT_Y, y);
  }
#pragma unroll 1
  for (int c = 0; c < T - 1; c++) {
    const int cnt = 1 << c;
    acc = fe_set_int(1);
#pragma unroll 1
    for (int k = 0; k < K.n; k++) {  // pass 1: the denominators x_C - x_P multiplied up, every prefix product parked
      const fe cx = slot_load_raw(K.scr[k] + KC_C + c * 18 + 0);
      u32 *wk = K.scr[k] + kc_sub_off(T);
#pragma unroll 1
      for (int m = 0; m < cnt; m++) {
        const fe d = fe_add(cx, fe_neg(slot_load_fe(K.tab[k] + m * SLOT_ENTRY_WORDS + ENT_X), 1));
        store_raw(wk + m * 9, acc);
        acc = fe_mul(acc, d);
      }
    }
    inv = fe_inv_var(acc);
#pragma unroll 1
    for (int k = K.n - 1; k >= 0; k--) {  // pass 2, backwards: 1 / d peeled off the running inverse, the new entry, its beta*x
      const fe cx = slot_load_raw(K.scr[k] + KC_C + c * 18 + 0), cy = slot_load_raw(K.scr[k] + KC_C + c * 18 + 9);
      const u32 *wk = K.scr[k] + kc_sub_off(T);
#pragma unroll 1
      for (int m = cnt - 1; m >= 0; m--) {
        const u32 *e = K.tab[k] + m * SLOT_ENTRY_WORDS;
        const fe xp = slot_load_fe(e + ENT_X), yp = slot_load_fe(e + ENT_Y);
        const fe d = fe_add(cx, fe_neg(xp, 1));
        const fe dinv = fe_mul(inv, slot_load_raw(wk + m * 9));
        inv = fe_mul(inv, d);
        const fe lam = fe_mul(fe_add(cy, fe_neg(yp, 1)), dinv);                 // (y_C - y_P) / (x_C - x_P)
        const fe x3 = fe_sqr_add(lam, fe_neg(fe_add(xp, cx), 2));               // lambda^2 - x_P - x_C
        const fe y3 = fe_mul_add(lam, fe_add(xp, fe_neg(x3, 1)), fe_neg(yp, 1)); // lambda * (x_P - x_3) - y_P
        u32 *o = K.tab[k] + (m | cnt) * SLOT_ENTRY_WORDS;
        slot_store_fe(o + ENT_X, x3);
        slot_store_fe(o + ENT_BX, fe_mul(x3, beta));
        slot_store_fe(o + ENT_Y, y3);
      }
    }
  }
#pragma unroll 1
  for (int k = 0; k < K.n; k++) {  // Zc = Zb, and the entry of Q itself on E_b
    const fe zb = slot_load_raw(K.scr[k] + KC_ZB);
    slot_store_fe(K.tab[k] + kc_words(T), zb);
    const fe z2 = fe_sqr(zb);
    const fe x = fe_mul(qs[k].x, z2);
    u32 *e = K.tab[k] + NE * SLOT_ENTRY_WORDS;
    slot_store_fe(e + ENT_X, x);
    slot_store_fe(e + ENT_BX, fe_mul(x, beta));
    slot_store_fe(e + ENT_Y, fe_mul(qs[k].y, fe_mul(z2, zb)));
  }
}
// sequential composition of the tree builder for nk keys (CPU test harness)
template <int T>
LAMD_HD void keytable_build_tree(u32 *const tabs[], u32 *const scratches[], const ge qs[], int nk) {
  kc_tree_keys K;
  K.n = nk;
  for (int k = 0; k < nk; k++) {
    K.tab[k] = tabs[k];
    K.scr[k] = scratches[k];
    kc_bases<T>(scratches[k], qs[k]);
  }
  kc_tree_affine<T>(K, qs);
}

// Comb recoding of one GLV half from its prep_rec form (|k| = mag + top*2^128 - 0x88..8): tooth i holds bits
// iD..iD+D-1 of w = (|k| >> 1) | 2^(N-1)
template <int T>
struct comb_half {
  u32 tooth[T];
  bool even;
};
template <int T>
LAMD_HD comb_half<T> comb_from_rec(const u32 mag[4], u32 top) {
  constexpr int D = kc_spacing(T), N = T * D;
  u32 t[6];
  u64 c = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const u64 d = (u64)mag[i] - 0x88888888u - c;
    t[i] = (u32)d;
    c = (d >> 32) & 1;
  }
  t[4] = top - (u32)c;
  t[5] = 0;
  comb_half<T> r;
  r.even = (t[0] & 1u) == 0;
  u32 w[6];
#pragma unroll
  for (int i = 0; i < 5; i++) w[i] = (t[i] >> 1) | (t[i + 1] << 31);
  w[5] = 0;
  w[(N - 1) >> 5] |= 1u << ((N - 1) & 31);
#pragma unroll
  for (int i = 0; i < T; i++) {
    const int word = (i * D) >> 5, sh = (i * D) & 31;
    u32 v = w[word] >> sh;
    if (sh + D > 32) v |= w[word + 1] << (32 - sh);
    r.tooth[i] = v & ((1u << D) - 1u);
  }
  return r;
}

// Both GLV halves as ODD magnitudes, so that the comb needs no repair additions: an even half is fixed in the scalar
// domain by adding a vector of the GLV lattice {(x, y): x + y*lambda = 0 mod n} -- (a1, b1) has both components odd,
// (a2, b2) = (even, odd), and (a1 - a2, b1 - b2) = (b1, b1 - a1) = (odd, even) (constants of glv_split, scalar.h) -- which
// leaves k1 + k2*lambda unchanged and the halves below 2^128 + 1.09 * 2^128 < 2^130 <= 2^(T*D) for both combs.
template <int T>
struct comb_pair {
  u32 tooth1[T], tooth2[T];
  bool n1, n2;  // signs of the (adjusted) halves
};
LAMD_HD void s160_add(u32 r[5], const u32 a[5], const u32 b[5]) {
  u64 c = 0;
#pragma unroll
  for (int i = 0; i < 5; i++) { c += (u64)a[i] + b[i]; r[i] = (u32)c; c >>= 32; }
}
LAMD_HD void s160_negate_if(u32 r[5], const u32 a[5], bool neg) {
  const u32 m = neg ? 0xFFFFFFFFu : 0u;
  u64 c = neg ? 1 : 0;
#pragma unroll
  for (int i = 0; i < 5; i++) { c += (u64)(a[i] ^ m); r[i] = (u32)c; c >>= 32; }
}
// the two halves after that adjustment, as signed 160-bit integers (two's complement): both odd, |k[h]| < 2^130
LAMD_HD void glv_odd_halves(const prep_rec &rec, u32 k[2][5]) {
  // |k_i| from the biased form (mag + top * 2^128 - 0x88..8), as signed 160-bit integers
#pragma unroll
  for (int h = 0; h < 2; h++) {
    const u32 *mag = h ? rec.k2 : rec.k1;
    const u32 top = (rec.flags & (h ? PREP_K2TOP : PREP_K1TOP)) ? 1u : 0u;
    u32 t[5];
    u64 c = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const u64 d = (u64)mag[i] - 0x88888888u - c;
      t[i] = (u32)d;
      c = (d >> 32) & 1;
    }
    t[4] = top - (u32)c;
    s160_negate_if(k[h], t, (rec.flags & (h ? PREP_K2NEG : PREP_K1NEG)) != 0);
  }
  const bool e1 = (k[0][0] & 1u) == 0, e2 = (k[1][0] & 1u) == 0;
  // lattice vectors as two's-complement words: A1 = a1, B1 = b1 = a1 - a2, A2 = a2, KY = b1 - a1
  const u32 A1[5] = {0x9284EB15u, 0xE86C90E4u, 0xA7D46BCDu, 0x3086D221u, 0x00000000u};
  const u32 B1[5] = {0xF5401B3Du, 0x90AB8056u, 0xFEF177D7u, 0x1BBC8129u, 0xFFFFFFFFu};
  const u32 A2[5] = {0x9D44CFD8u, 0x57C1108Du, 0xA8E2F3F6u, 0x14CA50F7u, 0x00000001u};
  const u32 KY[5] = {0x62BB3028u, 0xA83EEF72u, 0x571D0C09u, 0xEB35AF08u, 0xFFFFFFFEu};
  u32 ax[5], ay[5];
#pragma unroll
  for (int i = 0; i < 5; i++) {
    // both even: (a1, b1); k1 even only: (b1, b1 - a1); k2 even only: (a2, a1); both odd: nothing
    ax[i] = e1 ? (e2 ? A1[i] : B1[i]) : (e2 ? A2[i] : 0u);
    ay[i] = e1 ? (e2 ? B1[i] : KY[i]) : (e2 ? A1[i] : 0u);
  }
  s160_add(k[0], k[0], ax);
  s160_add(k[1], k[1], ay);
}
template <int T>
LAMD_HD comb_pair<T> comb_from_rec_odd(const prep_rec &rec) {
  constexpr int D = kc_spacing(T), N = T * D;
  static_assert(N >= 130, "comb must cover the adjusted halves");
  u32 k[2][5];
  glv_odd_halves(rec, k);
  comb_pair<T> r;
#pragma unroll
  for (int h = 0; h < 2; h++) {
    const bool neg = (k[h][4] >> 31) != 0;
    u32 m[6];
    s160_negate_if(m, k[h], neg);
    m[5] = 0;
    LAMD_ASSERT((m[0] & 1u) == 1u && m[4] < 4u);
    u32 w[6];
#pragma unroll
    for (int i = 0; i < 5; i++) w[i] = (m[i] >> 1) | (m[i + 1] << 31);
    w[5] = 0;
    w[(N - 1) >> 5] |= 1u << ((N - 1) & 31);
#pragma unroll
    for (int i = 0; i < T; i++) {
      const int word = (i * D) >> 5, sh = (i * D) & 31;
      u32 v = w[word] >> sh;
      if (sh + D > 32) v |= w[word + 1] << (32 - sh);
      (h ? r.tooth2 : r.tooth1)[i] = v & ((1u << D) - 1u);
    }
    (h ? r.n2 : r.n1) = neg;
  }
  return r;
}

// The hot form of the table-driven ecmult: both halves odd (no repair additions), the first column initialises the
// accumulator (no infinity handling), every addition is the bare formula (gej_add_ge_fast) and the G windows skip a zero
// digit by branching.  Degenerate events (an addition meeting +-its operand: adversarial scalars only, or the result being
// infinity) leave Z = 0, which the caller tests ONCE: *suspect = true means "verdict unknown, run ecmult_lane_keyed".
// (glds: experiment only, -DLAMD_G_LDS -- the north star's "LDS-staged precomputed G table": 52 windows x 32 entries x 64 B = 104 KB
// staged into the block's LDS by the kernel; u1*G then takes 52 additions from LDS instead of 12 from the 3 GiB table in HBM.
// Measured A/B in profiles/r03_ab_variants.txt; the shipped build has no such code.)
constexpr int GLDS_BITS = 5, GLDS_WINDOWS = (256 + GLDS_BITS - 1) / GLDS_BITS, GLDS_WORDS = GLDS_WINDOWS * (1 << GLDS_BITS) * 16;
template <int T>
LAMD_HD gexz ecmult_lane_keyed_fast(const prep_rec &rec, const u32 *tab, const u32 *gtable, bool *suspect, const u32 *glds = nullptr) {
  constexpr int D = kc_spacing(T), NE = kc_ne(T);
  const comb_pair<T> cp = comb_from_rec_odd<T>(rec);
  gej acc = gej_infinity();
#pragma unroll 1
  for (int j = D - 1; j >= 0; j--) {
    if (j != D - 1) acc = gej_double(acc);
    // (-DLAMD_UNROLL_HALF=1, an experiment: both additions of a column as straight-line code -- the accumulator's 27 registers are copied at
    // the column loop's back edge only, not after every addition; 12 KB more code.  profiles/r06_ab_variants.txt)
#if LAMD_UNROLL_HALF
#pragma unroll
#else
#pragma unroll 1
#endif
    for (int half = 0; half < 2; half++) {
      u32 m = 0;
#pragma unroll
      for (int i = 0; i < T; i++) m |= (((half ? cp.tooth2[i] : cp.tooth1[i]) >> j) & 1u) << i;
      const bool top = (m >> (T - 1)) & 1u;
      const u32 idx = (top ? m : ~m) & (u32)(NE - 1);
      const u32 *e = tab + idx * SLOT_ENTRY_WORDS;
      ge pt;
      pt.x = slot_load_fe(e + (half ? ENT_BX : ENT_X));
      pt.y = slot_load_fe(e + ENT_Y);
      pt = ge_neg_if_lazy(pt, top == (half ? cp.n2 : cp.n1));
#if defined(LAMD_TOUCH_NEXT) && defined(__HIP_DEVICE_COMPILE__)
      // experiment (not in the shipped build): touch the table entry of the NEXT addition before this one starts, so that its cache
      // lines are on their way from HBM while the ~1000 multiply-adds of this addition issue.  One dword per coordinate, result unused.
      if (half == 0 || j > 0) {
        const int nj = half ? j - 1 : j, nh = half ? 0 : 1;
        u32 nm = 0;
#pragma unroll
        for (int i = 0; i < T; i++) nm |= (((nh ? cp.tooth2[i] : cp.tooth1[i]) >> nj) & 1u) << i;
        const u32 nidx = (((nm >> (T - 1)) & 1u) ? nm : ~nm) & (u32)(NE - 1);
        const volatile u32 *ne_ = tab + nidx * SLOT_ENTRY_WORDS;
        (void)ne_[nh ? ENT_BX : ENT_X];
        (void)ne_[ENT_Y];
      }
#endif
      if (j == D - 1 && half == 0) {  // uniform across the wave: the first point is the accumulator
        acc.x = pt.x;
        acc.y = fe_norm_weak(pt.y);
        acc.z = fe_set_int(1);
        acc.inf = false;
      } else {
        acc = gej_add_ge_fast(acc, pt);
      }
    }
  }
  acc.z = fe_mul(acc.z, slot_load_fe(tab + kc_words(T)));
  // the windows of u1 come off a 256-bit shift register: no dynamic indexing of the scalar (which would put it in scratch)
  u32 uw[8];
#pragma unroll
  for (int i = 0; i < 8; i++) uw[i] = rec.u1[i];
#if defined(LAMD_G_LDS)
  if (glds) {
#pragma unroll 1
    for (int w = 0; w < GLDS_WINDOWS; w++) {
      const u32 d = uw[0] & ((1u << GLDS_BITS) - 1u);
#pragma unroll
      for (int i = 0; i < 7; i++) uw[i] = (uw[i] >> GLDS_BITS) | (uw[i + 1] << (32 - GLDS_BITS));
      uw[7] >>= GLDS_BITS;
      if (d != 0) {
        const u32 *e = glds + ((w << GLDS_BITS) + d) * 16;
        u32 xw[8], yw[8];
#pragma unroll
        for (int i = 0; i < 8; i++) { xw[i] = e[i]; yw[i] = e[8 + i]; }
        ge pt;
        pt.x = fe_from_words(xw);
        pt.y = fe_from_words(yw);
        acc = gej_add_ge_fast(acc, pt);
      }
    }
    *suspect = fe_is_zero(acc.z);
    return gexz_from_gej(acc);
  }
#endif
  // the G windows are a run of additions without a doubling: XYZZ coordinates (group.h), a squaring less per addition
  // (-DLAMD_G_RUN_XYZZ=0: the Jacobian run of rounds 1-5, converted at the end -- the A/B in profiles/r06_ab_variants.txt)
#if LAMD_G_RUN_XYZZ
  gexz xz = gexz_from_gej(acc);
#endif
#pragma unroll 1
  for (int w = 0; w < GTABLE_WINDOWS; w++) {
    const u32 d = uw[0] & ((1u << GTABLE_WINDOW_BITS) - 1u);
#pragma unroll
    for (int i = 0; i < 7; i++) uw[i] = (uw[i] >> GTABLE_WINDOW_BITS) | (uw[i + 1] << (32 - GTABLE_WINDOW_BITS));
    uw[7] >>= GTABLE_WINDOW_BITS;
    if (d != 0) {  // a zero digit (2^-22 per window) is a divergent skip, not a select
#if defined(LAMD_CLOCK_PROBE_G_HOT)
      // clock experiment (profiles/r06_clock.txt; verdicts are WRONG in this build): every window reads from the first 64 MiB of the table --
      // the same eleven additions and loads, served from the Infinity Cache instead of HBM
      const u32 *e = gtable + (size_t)(d & 0xFFFFFu) * GT_ENTRY_WORDS;
#else
      const u32 *e = gtable + (((size_t)w << GTABLE_WINDOW_BITS) + d) * GT_ENTRY_WORDS;
#endif
      ge pt;
      pt.x = slot_load_fe(e);
      pt.y = slot_load_fe(e + TW);
#if LAMD_G_RUN_XYZZ
      xz = gexz_add_ge_fast(xz, pt);
#else
      acc = gej_add_ge_fast(acc, pt);
#endif
    }
  }
#if !LAMD_G_RUN_XYZZ
  const gexz xz = gexz_from_gej(acc);
#endif
  *suspect = fe_is_zero(xz.zz);
  return xz;
}

// ---- the per-signature ladder in its hot form (keys without a table: the cold rows of a batch, every row of an all-distinct batch).
// ecmult_lane above adds with the complete formula under a zero-digit predicate, an infinity select and a degenerate-case test -- ~270 of
// the ~1 700 instructions of each of its 66 + 11 additions.  Here both halves are made ODD in the scalar domain (glv_odd_halves: a lattice
// vector, no repair additions) and written in signed odd digits: with w = (|k| >> 1) | 2^131, digit i = 2 * nibble_i(w) - 15 is one of
// +-1, +-3, .., +-15 and sum digit_i 16^i = |k| (33 digits cover the 130 bits).  No digit is zero, the top digit is +1 or +3 and initialises
// the accumulator, the table holds the eight odd multiples 1Q, 3Q, .., 15Q -- so every addition is the bare formula, and the G windows
// skip a zero digit by branching, exactly as in ecmult_lane_keyed_fast.  Degenerate events leave Z = 0: *suspect, and the caller runs
// ecmult_lane.  (An accumulator a*Q + b*lambda*Q meets +-d*Q or +-d*lambda*Q only if (a -+ d, b) resp. (a, b -+ d) is a vector of the GLV
// lattice: never for honest scalars, constructible for chosen ones -- hence the test, not an argument.)
// The table: 2Q is computed once, Q is carried to the isomorphic curve on which 2Q is affine, and seven mixed additions of 2Q give
// 3Q .. 15Q; the entries are rescaled to the last one's Z from the H values, as build_multiples does.  Returns the Z that maps the
// entries' curve back to secp256k1 (Zg * Z_2Q).
LAMD_HD fe build_odd_multiples8(u32 *slot, u32 *hbuf, const ge &q) {
  const gej d = gej_double(gej_from_ge(q));
  const fe zd = fe_norm_weak(d.z);
  const fe zd2 = fe_sqr(zd);
  ge dd;
  dd.x = d.x;
  dd.y = d.y;
  gej p;
  p.x = fe_mul(q.x, zd2);
  p.y = fe_mul(q.y, fe_mul(zd2, zd));
  p.z = fe_set_int(1);
  p.inf = false;
  slot_store_fe(slot + 0, p.x);
  slot_store_fe(slot + ENT_Y, p.y);
#pragma unroll 1
  for (int i = 1; i < 8; i++) {  // entry i = (2i+1)Q = entry(i-1) + 2Q: an odd and an even multiple never meet
    bool degenerate;
    fe h, rr;
    p = gej_add_ge_core(p, dd, &degenerate, &h, &rr);
    slot_store_fe(slot + i * SLOT_ENTRY_WORDS + 0, p.x);
    slot_store_fe(slot + i * SLOT_ENTRY_WORDS + ENT_Y, p.y);
    slot_store_fe(hbuf + (i - 1) * TW, h);
  }
  const fe zg = fe_norm_weak(p.z);
  const u32 betaw[8] = LAMD_BETA;
  const fe beta = fe_from_words(betaw);
  {
    const fe x = slot_load_fe(slot + 7 * SLOT_ENTRY_WORDS);
    slot_store_fe(slot + 7 * SLOT_ENTRY_WORDS + ENT_BX, fe_mul(x, beta));
  }
  fe rho = fe_set_int(1);
#pragma unroll 1
  for (int i = 6; i >= 0; i--) {  // rho = Zg / Z_entry(i) = H_(i+1) * .. * H_7 (Z_0 = 1 on the isomorphic curve)
    rho = i == 6 ? slot_load_fe(hbuf + 6 * TW) : fe_mul(rho, slot_load_fe(hbuf + i * TW));
    const fe r2 = fe_sqr(rho);
    const fe r3 = fe_mul(r2, rho);
    const fe x = fe_mul(slot_load_fe(slot + i * SLOT_ENTRY_WORDS + 0), r2);
    const fe y = fe_mul(slot_load_fe(slot + i * SLOT_ENTRY_WORDS + ENT_Y), r3);
    slot_store_fe(slot + i * SLOT_ENTRY_WORDS + 0, x);
    slot_store_fe(slot + i * SLOT_ENTRY_WORDS + ENT_BX, fe_mul(x, beta));
    slot_store_fe(slot + i * SLOT_ENTRY_WORDS + ENT_Y, y);
  }
  return fe_mul(zg, zd);
}
static_assert(SLOT_H_OFF + 7 * TW <= SLOT_WORDS, "the ladder's slot holds eight entries and seven H values");
LAMD_HD gej ecmult_lane_fast(const prep_rec &rec, const ge &q, u32 *slot, const u32 *gtable, bool *suspect) {
  const fe zg = build_odd_multiples8(slot, slot + SLOT_H_OFF, q);
  u32 k[2][5];
  glv_odd_halves(rec, k);
  // w = (|k| >> 1) | 2^131 per half, kept as a shift register whose top nibble (bits 128..131) is the next digit
  u32 w[2][5];
  bool neg[2];
#pragma unroll
  for (int h = 0; h < 2; h++) {
    neg[h] = (k[h][4] >> 31) != 0;
    u32 m[6];
    s160_negate_if(m, k[h], neg[h]);
    m[5] = 0;
    LAMD_ASSERT((m[0] & 1u) == 1u && m[4] < 4u);
#pragma unroll
    for (int i = 0; i < 5; i++) w[h][i] = (m[i] >> 1) | (m[i + 1] << 31);
    w[h][4] |= 8u;
  }
  gej acc;
#pragma unroll 1
  for (int i = 32; i >= 0; i--) {
    if (i != 32) {
#pragma unroll 1
      for (int j = 0; j < 4; j++) acc = gej_double(acc);
    }
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const u32 nib = w[h][4] & 15u;
#pragma unroll
      for (int t = 4; t > 0; t--) w[h][t] = (w[h][t] << 4) | (w[h][t - 1] >> 28);
      w[h][0] <<= 4;
      const bool dneg = nib < 8u;                    // digit = 2 * nib - 15
      const u32 idx = dneg ? 7u - nib : nib - 8u;    // (|digit| - 1) / 2
      const u32 *e = slot + idx * SLOT_ENTRY_WORDS;
      ge pt;
      pt.x = slot_load_fe(e + (h ? ENT_BX : ENT_X));
      pt.y = slot_load_fe(e + ENT_Y);
      pt = ge_neg_if_lazy(pt, dneg != neg[h]);
      if (i == 32 && h == 0) {  // uniform across the wave: the first point is the accumulator
        acc.x = pt.x;
        acc.y = fe_norm_weak(pt.y);
        acc.z = fe_set_int(1);
        acc.inf = false;
      } else {
        acc = gej_add_ge_fast(acc, pt);
      }
    }
  }
  acc.z = fe_mul(acc.z, zg);  // back from the isomorphic curve
  u32 uw[8];
#pragma unroll
  for (int i = 0; i < 8; i++) uw[i] = rec.u1[i];
#pragma unroll 1
  for (int wd = 0; wd < GTABLE_WINDOWS; wd++) {
    const u32 d = uw[0] & ((1u << GTABLE_WINDOW_BITS) - 1u);
#pragma unroll
    for (int i = 0; i < 7; i++) uw[i] = (uw[i] >> GTABLE_WINDOW_BITS) | (uw[i + 1] << (32 - GTABLE_WINDOW_BITS));
    uw[7] >>= GTABLE_WINDOW_BITS;
    if (d != 0) {
      const u32 *e = gtable + (((size_t)wd << GTABLE_WINDOW_BITS) + d) * GT_ENTRY_WORDS;
      ge pt;
      pt.x = slot_load_fe(e);
      pt.y = slot_load_fe(e + TW);
      acc = gej_add_ge_fast(acc, pt);
    }
  }
  *suspect = fe_is_zero(acc.z);
  return acc;
}

// R = u1*G + (k1 + k2*lambda)*Q from the key's comb table
template <int T>
LAMD_HD gej ecmult_lane_keyed(const prep_rec &rec, const u32 *tab, const u32 *gtable) {
  constexpr int D = kc_spacing(T), NE = kc_ne(T);
  const bool n1 = rec.flags & PREP_K1NEG, n2 = rec.flags & PREP_K2NEG;
  const comb_half<T> h1 = comb_from_rec<T>(rec.k1, (rec.flags & PREP_K1TOP) ? 1u : 0u);
  const comb_half<T> h2 = comb_from_rec<T>(rec.k2, (rec.flags & PREP_K2TOP) ? 1u : 0u);
  gej acc = gej_infinity();
#pragma unroll 1
  for (int j = D - 1; j >= 0; j--) {
    if (j != D - 1) acc = gej_double(acc);
#pragma unroll 1
    for (int half = 0; half < 2; half++) {
      u32 m = 0;
#pragma unroll
      for (int i = 0; i < T; i++) m |= (((half ? h2.tooth[i] : h1.tooth[i]) >> j) & 1u) << i;
      const bool top = (m >> (T - 1)) & 1u;
      const u32 idx = (top ? m : ~m) & (u32)(NE - 1);
      const u32 *e = tab + idx * SLOT_ENTRY_WORDS;
      ge pt;
      pt.x = slot_load_fe(e + (half ? ENT_BX : ENT_X));
      pt.y = slot_load_fe(e + ENT_Y);
      pt = ge_neg_if(pt, top == (half ? n2 : n1));  // column value is -entry when the top tooth is -1; times the half's sign
      acc = gej_add_ge(acc, pt, false);
    }
  }
  // an even |k| was evaluated as |k| + 1: take sign*Q (sign*lambda*Q) off again
#pragma unroll 1
  for (int half = 0; half < 2; half++) {
    const u32 *e = tab + NE * SLOT_ENTRY_WORDS;
    ge pt;
    pt.x = slot_load_fe(e + (half ? ENT_BX : ENT_X));
    pt.y = slot_load_fe(e + ENT_Y);
    pt = ge_neg_if(pt, !(half ? n2 : n1));
    acc = gej_add_ge(acc, pt, !(half ? h2.even : h1.even));
  }
  // back from the table's isomorphic curve: (X, Y, Z) -> (X, Y, Z*Zc)
  acc.z = fe_mul(acc.z, slot_load_fe(tab + kc_words(T)));
#pragma unroll 1
  for (int w = 0; w < GTABLE_WINDOWS; w++) {
    const u32 d = gtable_digit(rec.u1, w);
    const bool skip = d == 0;
    const u32 *e = gtable + (((size_t)w << GTABLE_WINDOW_BITS) + d) * GT_ENTRY_WORDS;
    ge pt;
    pt.x = slot_load_fe(e);
    pt.y = slot_load_fe(e + TW);
    acc = gej_add_ge(acc, pt, skip);
  }
  return acc;
}

// ================================================================================================
// Pairs first (round 4): the table-driven ecmult with HALF its mixed additions replaced by affine + affine additions that share
// one field inversion.  The ~49 points a verification sums are all affine table entries, and they come in natural pairs that sit
// at the same doubling level: the two GLV halves' entries of one comb column, and two windows of u1*G.  A pair (x1, y1) + (x2, y2)
// with the inverse of H = x2 - x1 at hand costs 2M + 1S (lambda = (y2 - y1)/H, x3 = lambda^2 - x1 - x2, y3 = lambda*(x1 - x3) - y1)
// and the sum is affine again, so it enters the accumulator by ONE mixed addition (8M + 3S) where the two points took two.
// The inverses come from Montgomery's trick over every pair of every row a lane owns in one batch (<= PAIRS_BMAX rows):
//   pass 1  (pairs_prefix_row)  walks the pairs in ascending order multiplying the H values up; every prefix product is parked
//           in a per-lane workspace in HBM (48 bytes each; the entries are only read for their x here);
//   invert  ONE variable-time inversion (division steps, fe_inv_var) per batch;
//   pass 2  (pairs_sum_row)  walks the pairs in DESCENDING order -- which is the Horner order of the comb: highest column first,
//           then the G windows -- peeling 1/H off the running inverse with two multiplications per pair.
// Per pair 1M + store (pass 1) and 4M + 1S + load (pass 2) against the 8M + 3S it removes; the inversion (~10^4 instructions)
// is shared by the batch.  T = 7: 24 pairs of a row's 49 points; T = 10: 18 of 37.
// Degenerate pairs (H = 0: the two entries equal or opposite -- no honest and, for the comb columns, no crafted scalar reaches
// that: entry multipliers are below 2^115, the GLV lattice has no vector that short) would zero the whole product: pass 1 tests
// the product after every ROW and hands such a row to the complete formulas (VERDICT_SUSPECT), leaving the batch intact.  A G
// window whose digit is zero (2^-24) has no entry: its pair is not formed, the other window's entry is added on its own.
constexpr int PAIRS_BMAX = 6;                                  // rows per lane and inversion
constexpr int PAIRS_WS_WORDS = 12;                             // a parked prefix product: 9 limbs in 48 bytes
constexpr int GT_PAIRS = GTABLE_WINDOWS / 2, GT_SINGLE = GTABLE_WINDOWS & 1;
constexpr int pairs_per_row(int T) { return GT_PAIRS + kc_spacing(T); }
constexpr int PAIRS_SLOTS = 1 + PAIRS_BMAX * pairs_per_row(7);  // slot 0 holds 1; T = 7 has the most pairs per row
static_assert(pairs_per_row(7) >= pairs_per_row(8) && pairs_per_row(7) >= pairs_per_row(10), "workspace sized for the 7-tooth comb");

LAMD_HD void pairs_ws_store(u32 *p, const fe &a) {
#if defined(__HIP_DEVICE_COMPILE__)
  uint4 *q = reinterpret_cast<uint4 *>(p);
  q[0] = make_uint4(a.n[0], a.n[1], a.n[2], a.n[3]);
  q[1] = make_uint4(a.n[4], a.n[5], a.n[6], a.n[7]);
  q[2] = make_uint4(a.n[8], 0u, 0u, 0u);
#else
  for (int i = 0; i < 9; i++) p[i] = a.n[i];
#endif
}
LAMD_HD fe pairs_ws_load(const u32 *p) {
  fe r;
#if defined(__HIP_DEVICE_COMPILE__)
  const uint4 *q = reinterpret_cast<const uint4 *>(p);
  const uint4 a = q[0], b = q[1];
  r.n[0] = a.x; r.n[1] = a.y; r.n[2] = a.z; r.n[3] = a.w;
  r.n[4] = b.x; r.n[5] = b.y; r.n[6] = b.z; r.n[7] = b.w;
  r.n[8] = p[8];
#else
  for (int i = 0; i < 9; i++) r.n[i] = p[i];
#endif
  FE_SETMAG(r, 1);
  return r;
}

template <int T>
LAMD_HD u32 comb_column(const u32 *tooth, int j) {
  u32 m = 0;
#pragma unroll
  for (int i = 0; i < T; i++) m |= ((tooth[i] >> j) & 1u) << i;
  return m;
}

// (x1, y1) + (x2, y2) with 1/(x2 - x1) = inv * pprev, where inv is the inverse of the product of the H values of this pair and
// every pair before it; inv becomes the inverse of the product before this pair.  y1, y2 magnitude <= 2 (ge_neg_if_lazy).
LAMD_HD ge pair_add_affine(const fe &x1, const fe &y1, const fe &x2, const fe &y2, const fe &pprev, fe &inv) {
  const fe h = fe_sub(x2, x1, 1);                                  // (3)
  const fe ih = fe_mul(inv, pprev);
  inv = fe_mul(inv, h);
  const fe lam = fe_mul(fe_sub(y2, y1, 2), ih);                    // (5) * (1)
  ge r;
  r.x = fe_sqr_add(lam, fe_neg(fe_add(x1, x2), 2));                // lambda^2 - x1 - x2
  r.y = fe_mul_add(lam, fe_sub(x1, r.x, 1), fe_neg(y1, 2));        // lambda*(x1 - x3) - y1
  return r;
}

// pass 1 over one row: multiplies the H values of its pairs (G pairs first, then the comb's columns in ascending order) onto P
// and parks every prefix product: pair k of the row at ws + k * ws_stride (words)
template <int T>
LAMD_HD fe pairs_prefix_row(const prep_rec &rec, const u32 *tab, const u32 *gtable, fe P, u32 *ws, size_t ws_stride) {
  constexpr int D = kc_spacing(T), NE = kc_ne(T);
#pragma unroll 1
  for (int p = 0; p < GT_PAIRS; p++) {
    const u32 d0 = gtable_digit(rec.u1, 2 * p), d1 = gtable_digit(rec.u1, 2 * p + 1);
    if (d0 != 0 && d1 != 0) {
      const u32 *e0 = gtable + (((size_t)(2 * p) << GTABLE_WINDOW_BITS) + d0) * GT_ENTRY_WORDS;
      const u32 *e1 = gtable + (((size_t)(2 * p + 1) << GTABLE_WINDOW_BITS) + d1) * GT_ENTRY_WORDS;
      P = fe_mul(P, fe_sub(slot_load_fe(e1), slot_load_fe(e0), 1));
    }
    pairs_ws_store(ws + (size_t)p * ws_stride, P);
  }
  const comb_pair<T> cp = comb_from_rec_odd<T>(rec);
#pragma unroll 1
  for (int j = 0; j < D; j++) {
    const u32 m1 = comb_column<T>(cp.tooth1, j), m2 = comb_column<T>(cp.tooth2, j);
    const u32 i1 = (((m1 >> (T - 1)) & 1u) ? m1 : ~m1) & (u32)(NE - 1), i2 = (((m2 >> (T - 1)) & 1u) ? m2 : ~m2) & (u32)(NE - 1);
    const fe x1 = slot_load_fe(tab + i1 * SLOT_ENTRY_WORDS + ENT_X), x2 = slot_load_fe(tab + i2 * SLOT_ENTRY_WORDS + ENT_BX);
    P = fe_mul(P, fe_sub(x2, x1, 1));
    pairs_ws_store(ws + (size_t)(GT_PAIRS + j) * ws_stride, P);
  }
  return P;
}

// pass 2 over one row: R = u1*G + (k1 + k2*lambda)*Q.  ws as in pass 1 (the slot before the row's first holds the product
// before the row); inv: in, the inverse of the product through this row's last pair; out, of the product before its first.
// Bare formulas throughout, *suspect as ecmult_lane_keyed_fast.
template <int T>
LAMD_HD gej pairs_sum_row(const prep_rec &rec, const u32 *tab, const u32 *gtable, fe &inv, const u32 *ws, size_t ws_stride, bool *suspect) {
  constexpr int D = kc_spacing(T), NE = kc_ne(T);
  const comb_pair<T> cp = comb_from_rec_odd<T>(rec);
  gej acc = gej_infinity();
#pragma unroll 1
  for (int j = D - 1; j >= 0; j--) {
    const u32 m1 = comb_column<T>(cp.tooth1, j), m2 = comb_column<T>(cp.tooth2, j);
    const bool t1 = (m1 >> (T - 1)) & 1u, t2 = (m2 >> (T - 1)) & 1u;
    const u32 *e1 = tab + ((t1 ? m1 : ~m1) & (u32)(NE - 1)) * SLOT_ENTRY_WORDS, *e2 = tab + ((t2 ? m2 : ~m2) & (u32)(NE - 1)) * SLOT_ENTRY_WORDS;
    ge a, b;
    a.x = slot_load_fe(e1 + ENT_X);
    a.y = slot_load_fe(e1 + ENT_Y);
    b.x = slot_load_fe(e2 + ENT_BX);
    b.y = slot_load_fe(e2 + ENT_Y);
    a = ge_neg_if_lazy(a, t1 == cp.n1);
    b = ge_neg_if_lazy(b, t2 == cp.n2);
    const fe pprev = pairs_ws_load(ws + ((ptrdiff_t)(GT_PAIRS + j) - 1) * (ptrdiff_t)ws_stride);
    const ge s = pair_add_affine(a.x, a.y, b.x, b.y, pprev, inv);
    if (j == D - 1) {  // uniform across the wave: the first sum is the accumulator
      acc.x = s.x;
      acc.y = s.y;
      acc.z = fe_set_int(1);
      acc.inf = false;
    } else {
      acc = gej_add_ge_fast(gej_double(acc), s);
    }
  }
  acc.z = fe_mul(acc.z, slot_load_fe(tab + kc_words(T)));  // back from the table's isomorphic curve
#pragma unroll 1
  for (int p = GT_PAIRS - 1; p >= 0; p--) {
    const u32 d0 = gtable_digit(rec.u1, 2 * p), d1 = gtable_digit(rec.u1, 2 * p + 1);
    const u32 *e0 = gtable + (((size_t)(2 * p) << GTABLE_WINDOW_BITS) + d0) * GT_ENTRY_WORDS;
    const u32 *e1 = gtable + (((size_t)(2 * p + 1) << GTABLE_WINDOW_BITS) + d1) * GT_ENTRY_WORDS;
    if (d0 != 0 && d1 != 0) {
      const fe pprev = pairs_ws_load(ws + ((ptrdiff_t)p - 1) * (ptrdiff_t)ws_stride);
      const ge s = pair_add_affine(slot_load_fe(e0), slot_load_fe(e0 + TW), slot_load_fe(e1), slot_load_fe(e1 + TW), pprev, inv);
      acc = gej_add_ge_fast(acc, s);
    } else {  // a window without an entry: no pair was formed in pass 1
      if (d0 != 0) { ge pt; pt.x = slot_load_fe(e0); pt.y = slot_load_fe(e0 + TW); acc = gej_add_ge_fast(acc, pt); }
      if (d1 != 0) { ge pt; pt.x = slot_load_fe(e1); pt.y = slot_load_fe(e1 + TW); acc = gej_add_ge_fast(acc, pt); }
    }
  }
  if (GT_SINGLE) {
    const u32 d = gtable_digit(rec.u1, GTABLE_WINDOWS - 1);
    if (d != 0) {
      const u32 *e = gtable + (((size_t)(GTABLE_WINDOWS - 1) << GTABLE_WINDOW_BITS) + d) * GT_ENTRY_WORDS;
      ge pt;
      pt.x = slot_load_fe(e);
      pt.y = slot_load_fe(e + TW);
      acc = gej_add_ge_fast(acc, pt);
    }
  }
  *suspect = fe_is_zero(acc.z);
  return acc;
}

// One lane's batch of nb <= PAIRS_BMAX rows through the three stages.  row(b, first, &rec, &tab): the row's scalars and its key's
// table, false when row b has no work (past the end of the list, scalars that failed the preparation); `first` tells the pass-1
// call from the pass-2 call.  done(b, R, suspect): the row's result; suspect as ecmult_lane_keyed_fast (R is garbage then).
// ws: the lane's PAIRS_SLOTS parking slots, ws_stride words apart.
#if defined(LAMD_PAIRS_CLOCK) && defined(__HIPCC__)
// experiment build: where a lane's time goes (pass 1 | inversion | pass 2), summed over all lanes in 100 MHz ticks (tools/pairs_clock_probe.py)
__device__ unsigned long long g_pairs_clk[8];   // [0..2] phase sums | [3] max lane total | [4] min start | [5] max end | [6] max pass 1 | [7] max pass 2
#endif
#if defined(LAMD_PAIRS_CLOCK) && defined(__HIP_DEVICE_COMPILE__)
#define LAMD_PCLK(k) do { const unsigned long long now_ = wall_clock64(); atomicAdd(&g_pairs_clk[k], now_ - pclk_); if ((k) == 0) atomicMax(&g_pairs_clk[6], now_ - pclk_); \
    if ((k) == 2) { atomicMax(&g_pairs_clk[7], now_ - pclk_); atomicMax(&g_pairs_clk[3], now_ - pclk0_); atomicMax(&g_pairs_clk[5], now_); } pclk_ = now_; } while (0)
#define LAMD_PCLK_START unsigned long long pclk_ = wall_clock64(); const unsigned long long pclk0_ = pclk_; atomicMin(&g_pairs_clk[4], pclk_);
#else
#define LAMD_PCLK(k) ((void)0)
#define LAMD_PCLK_START
#endif
template <int T, class RowF, class DoneF>
LAMD_HD void pairs_batch(int nb, const u32 *gtable, u32 *ws, size_t ws_stride, RowF row, DoneF done) {
  constexpr int NP = pairs_per_row(T);
  LAMD_PCLK_START
  fe P = fe_set_int(1);
  pairs_ws_store(ws, P);
  u32 good = 0;
  int ng = 0;
#pragma unroll 1
  for (int b = 0; b < nb; b++) {
    const prep_rec *rec;
    const u32 *tab;
    if (!row(b, true, &rec, &tab)) continue;
    const fe before = P;
    P = pairs_prefix_row<T>(*rec, tab, gtable, P, ws + (size_t)(1 + ng * NP) * ws_stride, ws_stride);
    if (fe_is_zero(P)) {  // a pair of equal or opposite entries: this row is not for the bare formulas
      P = before;
      done(b, gej_infinity(), true);
      continue;
    }
    good |= 1u << b;
    ng++;
  }
  if (ng == 0) return;
  LAMD_PCLK(0);
  fe inv = fe_inv_var(P);
  LAMD_PCLK(1);
#pragma unroll 1
  for (int b = nb - 1; b >= 0; b--) {
    if (!((good >> b) & 1u)) continue;
    ng--;
    const prep_rec *rec;
    const u32 *tab;
    row(b, false, &rec, &tab);
    bool suspect;
    const gej R = pairs_sum_row<T>(*rec, tab, gtable, inv, ws + (size_t)(1 + ng * NP) * ws_stride, ws_stride, &suspect);
    done(b, R, suspect);
  }
  LAMD_PCLK(2);
}

// ================================================================================================
// Task split (latency path: k_small_verify, 64 rows per block).  One row per LANE and one TASK per WAVE: a lone signature on one
// lane is a chain of ~10^5 dependent instructions, and a wave issues one instruction every ~4.3 cycles however few lanes are
// live -- so the verification is cut into independent partial sums that different waves (different SIMDs) compute at the same
// time, and one wave merges them with complete Jacobian additions:
//     R = u1*G  +  [ (H1lo + H1hi) + (H2lo + H2hi) ] * (Z scale of the table's isomorphic curve)
//   comb of T teeth, D columns, split at column J:  Hxlo = sum_{j<J} 2^j C_j,  Hxhi = 2^J * sum_{j>=J} 2^(j-J) C_j   (x = GLV half)
//   ladder (key without a table):                   H1lo = k1*Q, H2lo = k2*lambda*Q, H1hi = H2hi = infinity
// Every addition here is the COMPLETE mixed addition (gej_add_ge): no suspect rows, no second pass.
enum { ST_G = 0, ST_H1LO = 1, ST_H1HI = 2, ST_H2LO = 3, ST_H2HI = 4, ST_TASKS = 5 };
constexpr int kc_split_col(int T) { return T == 7 ? 12 : T == 10 ? 8 : (kc_spacing(T) * 5) / 8; }  // balances doublings + additions of the two parts

LAMD_HD gej small_task_g(const prep_rec &rec, const u32 *gtable, int w_lo = 0, int w_hi = GTABLE_WINDOWS) {
  gej acc = gej_infinity();
#pragma unroll 1
  for (int w = w_lo; w < w_hi; w++) {
    const u32 d = gtable_digit(rec.u1, w);
    const u32 *e = gtable + (((size_t)w << GTABLE_WINDOW_BITS) + d) * GT_ENTRY_WORDS;
    ge pt;
    pt.x = slot_load_fe(e);
    pt.y = slot_load_fe(e + TW);
    acc = gej_add_ge(acc, pt, d == 0);
  }
  return acc;
}
// task in ST_H1LO..ST_H2HI; the result lives on the table's isomorphic curve
template <int T>
LAMD_HD gej small_task_comb(const prep_rec &rec, const u32 *tab, int task) {
  constexpr int D = kc_spacing(T), NE = kc_ne(T), J = kc_split_col(T);
  const comb_pair<T> cp = comb_from_rec_odd<T>(rec);
  const bool second = task >= ST_H2LO, hi = task == ST_H1HI || task == ST_H2HI;
  u32 tooth[T];
#pragma unroll
  for (int i = 0; i < T; i++) tooth[i] = second ? cp.tooth2[i] : cp.tooth1[i];
  const bool neg = second ? cp.n2 : cp.n1;
  const int jtop = hi ? D - 1 : J - 1, jbot = hi ? J : 0;
  gej acc = gej_infinity();
#pragma unroll 1
  for (int j = jtop; j >= jbot; j--) {
    if (j != jtop) acc = gej_double(acc);
    u32 m = 0;
#pragma unroll
    for (int i = 0; i < T; i++) m |= ((tooth[i] >> j) & 1u) << i;
    const bool top = (m >> (T - 1)) & 1u;
    const u32 idx = (top ? m : ~m) & (u32)(NE - 1);
    const u32 *e = tab + idx * SLOT_ENTRY_WORDS;
    ge pt;
    pt.x = slot_load_fe(e + (second ? ENT_BX : ENT_X));
    pt.y = slot_load_fe(e + ENT_Y);
    pt = ge_neg_if(pt, top == neg);  // the column is -entry when the top tooth is -1; times the sign of the (adjusted) half
    acc = gej_add_ge(acc, pt, false);
  }
  if (hi) {
#pragma unroll 1
    for (int k = 0; k < J; k++) acc = gej_double(acc);
  }
  return acc;
}
// The same comb part with the shape known only at run time (T = 7 or 10).  k_small_verify keeps one row per lane, and the lanes of a wave
// may hold keys with 7-tooth and keys with 10-tooth tables -- a commitment_signed is one signature under the funding key (7 teeth) and 483
// under the htlc key (10): with one instantiation per shape the wave runs both bodies one after the other (0.33 ms for that batch against
// 0.14 ms for its 483 htlc rows alone); with run-time bounds it runs the longer of the two loops once.
LAMD_HD gej small_task_comb_rt(const prep_rec &rec, const u32 *tab, int task, int T) {
  const comb_pair<7> c7 = comb_from_rec_odd<7>(rec);
  const comb_pair<10> c10 = comb_from_rec_odd<10>(rec);
  const bool ten = T == 10, second = task >= ST_H2LO, hi = task == ST_H1HI || task == ST_H2HI;
  u32 tooth[10];
#pragma unroll
  for (int i = 0; i < 10; i++) {
    const u32 t10 = second ? c10.tooth2[i] : c10.tooth1[i];
    const u32 t7 = i < 7 ? (second ? c7.tooth2[i < 7 ? i : 0] : c7.tooth1[i < 7 ? i : 0]) : 0u;
    tooth[i] = ten ? t10 : t7;
  }
  const bool neg = ten ? (second ? c10.n2 : c10.n1) : (second ? c7.n2 : c7.n1);
  const int D = ten ? kc_spacing(10) : kc_spacing(7), J = ten ? kc_split_col(10) : kc_split_col(7), topbit = ten ? 9 : 6;
  const u32 ne_mask = ten ? (u32)(kc_ne(10) - 1) : (u32)(kc_ne(7) - 1);
  const int jtop = hi ? D - 1 : J - 1, jbot = hi ? J : 0;
  gej acc = gej_infinity();
#pragma unroll 1
  for (int j = jtop; j >= jbot; j--) {
    if (j != jtop) acc = gej_double(acc);
    u32 m = 0;
#pragma unroll
    for (int i = 0; i < 10; i++) m |= ((tooth[i] >> j) & 1u) << i;
    const bool top = (m >> topbit) & 1u;
    const u32 idx = (top ? m : ~m) & ne_mask;
    const u32 *e = tab + idx * SLOT_ENTRY_WORDS;
    ge pt;
    pt.x = slot_load_fe(e + (second ? ENT_BX : ENT_X));
    pt.y = slot_load_fe(e + ENT_Y);
    pt = ge_neg_if(pt, top == neg);
    acc = gej_add_ge(acc, pt, false);
  }
  if (hi) {
#pragma unroll 1
    for (int k = 0; k < J; k++) acc = gej_double(acc);
  }
  return acc;
}
// one GLV half over the lane's own 8-entry table (build_q_table): k1*Q (second = false) or k2*lambda*Q
LAMD_HD gej small_task_ladder(const prep_rec &rec, const u32 *slot, bool second) {
  const bool neg = rec.flags & (second ? PREP_K2NEG : PREP_K1NEG);
  const u32 top = (rec.flags & (second ? PREP_K2TOP : PREP_K1TOP)) ? 1u : 0u;
  gej acc = gej_infinity();
#pragma unroll 1
  for (int i = 32; i >= 0; i--) {
    if (i != 32) {
#pragma unroll 1
      for (int j = 0; j < 4; j++) acc = gej_double(acc);
    }
    int d = second ? glv_digit(rec.k2, top, i) : glv_digit(rec.k1, top, i);
    if (neg) d = -d;
    const bool skip = d == 0;
    const int a = d < 0 ? -d : d;
    const u32 *e = slot + (skip ? 0 : a - 1) * SLOT_ENTRY_WORDS;
    ge pt;
    pt.x = slot_load_fe(e + (second ? ENT_BX : ENT_X));
    pt.y = slot_load_fe(e + ENT_Y);
    pt = ge_neg_if(pt, d < 0);
    acc = gej_add_ge(acc, pt, skip);
  }
  return acc;
}
// Which windows of u1*G a task wave adds up next to its comb / ladder part (k_small_verify runs FOUR task waves, one per SIMD of
// the CU -- a fifth wave would share a SIMD with another and stretch both): a row with a comb table spreads the 11 windows over the four waves (3 + 3 + 3 + 2),
// a ladder row gives them to the two waves that have no ladder half to compute.
LAMD_HD void small_g_windows(int task /*0..3*/, bool ladder, int *w_lo, int *w_hi) {
  constexpr int Q = (GTABLE_WINDOWS + 3) / 4, H = (GTABLE_WINDOWS + 1) / 2;
  if (!ladder) { *w_lo = task * Q; *w_hi = (task + 1) * Q; }
  else if (task == 1) { *w_lo = 0; *w_hi = H; }
  else if (task == 3) { *w_lo = H; *w_hi = GTABLE_WINDOWS; }
  else { *w_lo = *w_hi = 0; }
  if (*w_hi > GTABLE_WINDOWS) *w_hi = GTABLE_WINDOWS;
  if (*w_lo > *w_hi) *w_lo = *w_hi;
}
// the merge as the kernel stages it (three levels, the two sums of a level on different waves):
//   level 1  P01 = P0 + P1, P23 = P2 + P3, G01 = G0 + G1, G23 = G2 + G3      level 2  S = (P01 + P23) * zscale, G = G01 + G23      level 3  R = S + G
LAMD_HD gej small_merge4(const gej *p /*[4] comb / ladder parts (isomorphic curve)*/, const gej *g /*[4] parts of u1*G*/, const fe &zscale) {
  gej s = gej_add_var(gej_add_var(p[0], p[1]), gej_add_var(p[2], p[3]));
  if (!s.inf) s.z = fe_mul(fe_norm_weak(s.z), zscale);
  return gej_add_var(s, gej_add_var(gej_add_var(g[0], g[1]), gej_add_var(g[2], g[3])));
}
// parts[ST_TASKS]; zscale = Zc of the comb table / Zg of the ladder table
LAMD_HD gej small_merge(const gej *parts, const fe &zscale) {
  gej s = gej_add_var(gej_add_var(parts[ST_H1LO], parts[ST_H1HI]), gej_add_var(parts[ST_H2LO], parts[ST_H2HI]));
  if (!s.inf) s.z = fe_mul(fe_norm_weak(s.z), zscale);  // back from the isomorphic curve: (X, Y, Z) -> (X, Y, Z * zscale)
  return gej_add_var(s, parts[ST_G]);
}
// BIP-340 acceptance for one row (the batched form shares the inversion over 16 rows: schnorr_stage1 / schnorr_final_thread)
LAMD_HD bool schnorr_accept_one(const gej &R, const u32 rw[8]) {
  if (R.inf) return false;
  const fe z = fe_norm_weak(R.z);
  const fe z2 = fe_sqr(z);
  if (!fe_equal(fe_mul(fe_from_words(rw), z2), R.x, 1)) return false;
  const fe zi = fe_inv_var(z);
  const fe y = fe_normalize(fe_mul(R.y, fe_mul(fe_sqr(zi), zi)));
  return (y.n[0] & 1) == 0;
}

// p - n (129 bits): r + n < p  <=>  r < p - n
#define LAMD_P_MINUS_N {0x2FC9BAEEu, 0x402DA172u, 0x50B75FC4u, 0x45512319u, 1u, 0u, 0u, 0u}

// ECDSA acceptance: R != inf and x(R) mod n == r, tested without inversion
// r*Z^2 == X, or (r + n)*Z^2 == X when r + n < p (x(R) mod n == r, secp256k1_ecdsa_sig_verify), for a finite R given by X and Z^2
LAMD_HD bool ecdsa_final_xzz(const fe &x, const fe &z2, const u32 rw[8]);
LAMD_HD bool ecdsa_final(const gej &R, const u32 rw[8]) {
  if (R.inf) return false;
  return ecdsa_final_xzz(R.x, fe_sqr(R.z), rw);
}
// the hot form's result (XYZZ, never flagged infinite: a result at infinity has ZZ = 0 and went to the careful launch as SUSPECT)
LAMD_HD bool ecdsa_final(const gexz &R, const u32 rw[8]) { return ecdsa_final_xzz(R.x, R.zz, rw); }
LAMD_HD bool ecdsa_final_xzz(const fe &x, const fe &z2, const u32 rw[8]) {
  const fe rf = fe_from_words(rw);  // r < n < p
  bool ok = fe_equal(fe_mul(rf, z2), x, 1);
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
    ok |= fe_equal(fe_mul(fe_from_words(t), z2), x, 1);
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
// the hot form's result (XYZZ): stage 2 computes Y' * Z'^-3 from what is parked, so Y' = Y * ZZZ and Z' = ZZ make that Y * ZZZ / ZZ^3 =
// Y / ZZZ (ZZ^3 = ZZZ^2) -- one multiplication here, for the rows whose x matches only, and stage 2 stays what it is
LAMD_HD u8 schnorr_stage1(const gexz &R, const u32 rw[8], u32 *slot) {
  const fe rf = fe_from_words(rw);  // r < p checked in prep
  if (!fe_equal(fe_mul(rf, R.zz), R.x, 1)) return 0;
  const fe y = fe_mul(R.y, R.zzz);
#pragma unroll
  for (int i = 0; i < 9; i++) {
    slot[SLOT_FIN_Y + i] = y.n[i];
    slot[SLOT_FIN_Z + i] = R.zz.n[i];
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
  fe inv = fe_inv_var(acc);
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

// ---- double SHA-256 over byte strings (gossip tails, BIP143 preimages: bitcoin/shadouble.c:7-11)
// SHA256(SHA256(m)) where the first `done` bytes of m (a multiple of 64) are already absorbed into st; p = the rest
LAMD_HD void sha256d_finish(u32 st[8], const u8 *p, size_t len, size_t done, u8 out32[32]) {
  u32 w[16];
  size_t off = 0;
  for (; off + 64 <= len; off += 64) {
    for (int i = 0; i < 16; i++) w[i] = load_be32(p + off + 4 * i);
    sha256_compress(st, w);
  }
  const size_t rem = len - off;
  for (int i = 0; i < 16; i++) {
    u32 v = 0;
    for (int b = 0; b < 4; b++) {
      const size_t k = (size_t)i * 4 + b;
      const u32 byte = k < rem ? p[off + k] : (k == rem ? 0x80u : 0u);
      v = (v << 8) | byte;
    }
    w[i] = v;
  }
  if (rem >= 56) {
    sha256_compress(st, w);
    for (int i = 0; i < 16; i++) w[i] = 0;
  }
  w[14] = (u32)(((u64)(done + len) * 8) >> 32);
  w[15] = (u32)((u64)(done + len) * 8);
  sha256_compress(st, w);
  // second hash over the 32-byte digest
  for (int i = 0; i < 8; i++) w[i] = st[i];
  w[8] = 0x80000000u;
  for (int i = 9; i < 15; i++) w[i] = 0;
  w[15] = 256;
  u32 st2[8] = LAMD_SHA256_IV;
  sha256_compress(st2, w);
  for (int i = 0; i < 8; i++) {
    out32[4 * i] = (u8)(st2[i] >> 24); out32[4 * i + 1] = (u8)(st2[i] >> 16);
    out32[4 * i + 2] = (u8)(st2[i] >> 8); out32[4 * i + 3] = (u8)st2[i];
  }
}
LAMD_HD void sha256d_bytes(const u8 *p, size_t len, u8 out32[32]) {
  u32 st[8] = LAMD_SHA256_IV;
  sha256d_finish(st, p, len, 0, out32);
}


// ---- BIP143 signature hash on the device: what bitcoin_tx_hash_for_sig() (bitcoin/signature.c:120-151) obtains from
// libwally's wally_tx_get_btc_signature_hash(..., WALLY_TX_FLAG_USE_WITNESS) -- always the segwit form, whichever script
// check_tx_sig() picked (:198-199).  Streaming SHA-256 over the pieces, nothing is materialised:
//   nVersion | hashPrevouts | hashSequence | outpoint | varint(len) script | amount | nSequence | hashOutputs | nLockTime | type
//   hashPrevouts = 0 with ANYONECANPAY; hashSequence = 0 with ANYONECANPAY, SINGLE or NONE; hashOutputs = all outputs (ALL), output
//   [input index] (SINGLE, 0 when there is none) or 0 (NONE)
struct sha_stream {
  u32 st[8];
  u32 w[16];
  u32 fill;   // bytes in w
  u64 total;  // bytes absorbed
};
LAMD_HD void shs_init(sha_stream *s) {
  const u32 iv[8] = LAMD_SHA256_IV;
  for (int i = 0; i < 8; i++) s->st[i] = iv[i];
  for (int i = 0; i < 16; i++) s->w[i] = 0;
  s->fill = 0;
  s->total = 0;
}
LAMD_HD void shs_update(sha_stream *s, const u8 *p, size_t n) {
  size_t k = 0;
#if !defined(__HIP_DEVICE_COMPILE__)
  // host: once the block buffer is empty, whole blocks go from the caller's bytes straight into the compression (a commitment transaction's 20 KB of outputs)
  if (n >= 128) {
    for (; s->fill != 0; k++) {
      s->w[s->fill >> 2] |= (u32)p[k] << (24 - 8 * (s->fill & 3));
      if (++s->fill == 64) {
        sha256_compress(s->st, s->w);
        for (int i = 0; i < 16; i++) s->w[i] = 0;
        s->fill = 0;
      }
    }
    for (; k + 64 <= n; k += 64) {
      for (int i = 0; i < 16; i++) s->w[i] = load_be32(p + k + 4 * i);
      sha256_compress(s->st, s->w);
    }
    for (int i = 0; i < 16; i++) s->w[i] = 0;
  }
#endif
  for (; k < n; k++) {
    s->w[s->fill >> 2] |= (u32)p[k] << (24 - 8 * (s->fill & 3));
    if (++s->fill == 64) {
      sha256_compress(s->st, s->w);
      for (int i = 0; i < 16; i++) s->w[i] = 0;
      s->fill = 0;
    }
  }
  s->total += n;
}
LAMD_HD void shs_update_le(sha_stream *s, u64 v, int bytes) {
  u8 b[8];
  for (int i = 0; i < bytes; i++) b[i] = (u8)(v >> (8 * i));
  shs_update(s, b, (size_t)bytes);
}
LAMD_HD void shs_final_double(sha_stream *s, u8 out32[32]) {  // SHA256(SHA256(everything absorbed))
  const u64 bits = s->total * 8;
  const u8 pad = 0x80;
  shs_update(s, &pad, 1);
  const u8 zero = 0;
  while (s->fill != 56) shs_update(s, &zero, 1);
  s->w[14] = (u32)(bits >> 32);
  s->w[15] = (u32)bits;
  sha256_compress(s->st, s->w);
  u32 w[16];
  for (int i = 0; i < 8; i++) w[i] = s->st[i];
  w[8] = 0x80000000u;
  for (int i = 9; i < 15; i++) w[i] = 0;
  w[15] = 256;
  u32 st2[8] = LAMD_SHA256_IV;
  sha256_compress(st2, w);
  for (int i = 0; i < 8; i++) {
    out32[4 * i] = (u8)(st2[i] >> 24); out32[4 * i + 1] = (u8)(st2[i] >> 16);
    out32[4 * i + 2] = (u8)(st2[i] >> 8); out32[4 * i + 3] = (u8)st2[i];
  }
}
// Bitcoin's CompactSize at p (at most `max` bytes available): bytes consumed (0 = truncated)
LAMD_HD size_t tx_compact_size(const u8 *p, size_t max, u64 *val) {
  if (max < 1) return 0;
  if (p[0] < 0xfd) { *val = p[0]; return 1; }
  const size_t width = p[0] == 0xfd ? 2 : (p[0] == 0xfe ? 4 : 8);
  if (max < 1 + width) return 0;
  u64 v = 0;
  for (size_t i = 0; i < width; i++) v |= (u64)p[1 + i] << (8 * i);
  *val = v;
  return 1 + width;
}
LAMD_HD void shs_update_compact_size(sha_stream *s, u64 v) {
  if (v < 0xfd) shs_update_le(s, v, 1);
  else if (v <= 0xffff) { shs_update_le(s, 0xfd, 1); shs_update_le(s, v, 2); }
  else if (v <= 0xffffffffull) { shs_update_le(s, 0xfe, 1); shs_update_le(s, v, 4); }
  else { shs_update_le(s, 0xff, 1); shs_update_le(s, v, 8); }
}
// One transaction as flat bytes: inputs = n_in x (txid 32 | vout u32 LE | nSequence u32 LE); outputs = the n_out outputs in wire
// form (amount u64 LE | CompactSize | scriptPubKey) back to back.
struct tx_view {
  u32 version, locktime;
  const u8 *inputs;
  u32 n_in;
  const u8 *outputs;
  size_t outputs_len;
  u32 n_out;
};
// false: the template is inconsistent (input index out of range -- the reference asserts, :213 -- or outputs that do not parse)
LAMD_HD bool bip143_sighash(const tx_view &t, u32 in_idx, const u8 *script, size_t script_len, u64 amount, u32 sighash_type, u8 out32[32]) {
  if (in_idx >= t.n_in) return false;
  const bool acp = (sighash_type & 0x80u) != 0;
  const u32 base = sighash_type & 0x1fu;
  const bool single = base == 3, none = base == 2;
  u8 hp[32], hs[32], ho[32];
  for (int i = 0; i < 32; i++) hp[i] = hs[i] = ho[i] = 0;
  sha_stream s;
  if (!acp) {
    shs_init(&s);
    for (u32 i = 0; i < t.n_in; i++) shs_update(&s, t.inputs + 40 * (size_t)i, 36);
    shs_final_double(&s, hp);
  }
  if (!acp && !single && !none) {
    shs_init(&s);
    for (u32 i = 0; i < t.n_in; i++) shs_update(&s, t.inputs + 40 * (size_t)i + 36, 4);
    shs_final_double(&s, hs);
  }
  // walk the outputs once: they must parse, and SINGLE needs the boundaries of output [in_idx]
  size_t pos = 0, one_off = 0, one_len = 0;
  for (u32 k = 0; k < t.n_out; k++) {
    if (t.outputs_len - pos < 8) return false;
    u64 sl;
    const size_t l = tx_compact_size(t.outputs + pos + 8, t.outputs_len - pos - 8, &sl);
    if (!l || sl > t.outputs_len - pos - 8 - l) return false;
    const size_t len = 8 + l + (size_t)sl;
    if (k == in_idx) { one_off = pos; one_len = len; }
    pos += len;
  }
  if (pos != t.outputs_len) return false;
  if (single) {
    if (in_idx < t.n_out) {
      shs_init(&s);
      shs_update(&s, t.outputs + one_off, one_len);
      shs_final_double(&s, ho);
    }
  } else if (!none) {
    shs_init(&s);
    shs_update(&s, t.outputs, t.outputs_len);
    shs_final_double(&s, ho);
  }
  shs_init(&s);
  shs_update_le(&s, t.version, 4);
  shs_update(&s, hp, 32);
  shs_update(&s, hs, 32);
  shs_update(&s, t.inputs + 40 * (size_t)in_idx, 36);
  shs_update_compact_size(&s, script_len);
  shs_update(&s, script, script_len);
  shs_update_le(&s, amount, 8);
  shs_update(&s, t.inputs + 40 * (size_t)in_idx + 36, 4);
  shs_update(&s, ho, 32);
  shs_update_le(&s, t.locktime, 4);
  shs_update_le(&s, sighash_type, 4);
  shs_final_double(&s, out32);
  return true;
}

// ---- gossip framing: what fromwire_channel_announcement / _node_announcement / _channel_update (generated from
// wire/peer_wire.csv:344-381) reject before gossipd/sigcheck.c runs, so that the batch entry points can take raw wire bytes.
enum { GOSSIP_CANN = 256, GOSSIP_NANN = 257, GOSSIP_CUPD = 258 };
// BigSize (BOLT #1, common/bigsize.c:53-104): bytes consumed, 0 = truncated or not minimally encoded
LAMD_HD size_t wire_bigsize(const u8 *p, size_t max, u64 *val) {
  if (max < 1) return 0;
  const u32 t = p[0];
  if (t < 0xfd) { *val = t; return 1; }
  const size_t width = t == 0xfd ? 2 : (t == 0xfe ? 4 : 8);
  if (max < 1 + width) return 0;
  u64 v = 0;
  for (size_t i = 0; i < width; i++) v = (v << 8) | p[1 + i];
  *val = v;
  const u64 floor = t == 0xfd ? 0xfdull : (t == 0xfe ? 0x10000ull : 0x100000000ull);
  return v < floor ? 0 : 1 + width;
}
// node_ann_tlvs (peer_wire.csv:367-369) under fromwire_tlv's rules (wire/tlvstream.c:144-300): strictly increasing types,
// lengths inside the message, unknown even types fail, unknown odd ones are skipped; record 1 (option_will_fund) is a
// lease_rates: u16 u16 u16 u32 tu32 = 10..14 bytes, the tu32 minimal (wire/fromwire.c:115-150)
LAMD_HD bool wire_node_ann_tlvs_ok(const u8 *p, size_t max) {
  bool first = true;
  u64 prev = 0;
  while (max > 0) {
    u64 type, length;
    size_t l = wire_bigsize(p, max, &type);
    if (!l) return false;
    p += l; max -= l;
    if (!first && type <= prev) return false;
    first = false; prev = type;
    if (type != 1 && (type & 1) == 0) return false;
    l = wire_bigsize(p, max, &length);
    if (!l) return false;
    p += l; max -= l;
    if (length > max) return false;
    if (type == 1 && (length < 10 || length > 14 || (length > 10 && p[10] == 0))) return false;
    p += length; max -= (size_t)length;
  }
  return true;
}
struct gossip_frame {
  u32 type;
  bool bad;           // fromwire_* would fail on the framing alone (signature ranges / key validity are checked elsewhere)
  size_t signed_off;  // the signed region is [signed_off, len)
  size_t keyoff;      // channel_announcement: the four keys; node_announcement: node_id
};
LAMD_HD gossip_frame gossip_parse_frame(const u8 *m, size_t len) {
  gossip_frame f;
  f.bad = len < 2;
  f.type = f.bad ? 0 : (((u32)m[0] << 8) | m[1]);
  f.signed_off = 66;
  f.keyoff = 0;
  if (f.type == GOSSIP_CANN) {
    f.signed_off = 258;
    f.bad |= len < 260;
    if (!f.bad) {
      const size_t flen = ((size_t)m[258] << 8) | m[259];
      f.keyoff = 260 + flen + 32 + 8;
      f.bad |= len < f.keyoff + 4 * 33;
    }
  } else if (f.type == GOSSIP_NANN) {
    f.bad |= len < 68;
    if (!f.bad) {
      const size_t flen = ((size_t)m[66] << 8) | m[67];
      f.keyoff = 68 + flen + 4;
      // node_id 33 | rgb_color 3 | alias 32 | addrlen u16 | addresses | tlvs
      f.bad |= len < f.keyoff + 70;
      if (!f.bad) {
        const size_t end = f.keyoff + 70 + (((size_t)m[f.keyoff + 68] << 8) | m[f.keyoff + 69]);
        f.bad |= len < end;
        if (!f.bad) f.bad |= !wire_node_ann_tlvs_ok(m + end, len - end);
      }
    }
  } else if (f.type == GOSSIP_CUPD) {
    f.bad |= len < 138;  // every fixed field of peer_wire.csv:370-381; trailing bytes are tolerated (and signed)
  } else {
    f.bad = true;
  }
  return f;
}

}  // namespace lamd
#include "bolt12.h"
namespace lamd {

// ---- fee grind (onchaind/onchaind.c:388-438 grind_htlc_tx_fee): ONE signature and key, many candidate fees.  Every
// candidate changes output 0's amount, hence hashOutputs, hence the sighash z -- but r, s and Q stay: with w = 1/s,
// R = (z*w)*G + (r*w)*Q, so (r*w)*Q is computed once (grind_prepare, the ordinary GLV ladder) and a candidate costs two
// small double-SHA256, one scalar multiplication and the G-table additions.
constexpr int GRIND_MAX_OUTPUTS = 192;  // serialised outputs that go into hashOutputs (an HTLC tx has one 43-byte output)
constexpr int GRIND_MAX_TAIL = 128;     // preimage bytes from the last 64-byte boundary before hashOutputs to the end
struct grind_setup {
  u32 valid;
  u32 sinv[8], rw[8], px[8], py[8];  // 1/s, r, affine (r/s)*Q
  u32 mid[8];                        // SHA-256 state after the preimage's leading whole blocks
};
LAMD_HD void grind_prepare(grind_setup *out, const u8 *sig64, const u8 *pub33, const u8 *pre, u32 lead_blocks, u32 *slot, const u32 *gtable) {
  grind_setup g;
  sc r, s;
  bool ok;
  ecdsa_load_rs(sig64, &r, &s, &ok);
  u32 qx[8], qy[8];
  ok &= parse_pubkey(pub33, 33, qx, qy);
  g.valid = ok;
  u32 st[8] = LAMD_SHA256_IV;
  for (u32 b = 0; b < lead_blocks; b++) {
    u32 w[16];
    for (int i = 0; i < 16; i++) w[i] = load_be32(pre + 64 * (size_t)b + 4 * i);
    sha256_compress(st, w);
  }
  for (int i = 0; i < 8; i++) { g.mid[i] = st[i]; g.sinv[i] = g.rw[i] = g.px[i] = g.py[i] = 0; }
  if (ok) {
    const sc sinv = sc_inv_var(s);
    const sc u2 = sc_mul(r, sinv);
    glv_half h1, h2;
    glv_split(&h1, &h2, u2);
    prep_rec rec;
    for (int i = 0; i < 8; i++) rec.u1[i] = 0;
    for (int i = 0; i < 4; i++) { rec.k1[i] = h1.mag[i]; rec.k2[i] = h2.mag[i]; }
    rec.flags = PREP_VALID | (h1.neg ? PREP_K1NEG : 0) | (h2.neg ? PREP_K2NEG : 0) | (h1.top ? PREP_K1TOP : 0) | (h2.top ? PREP_K2TOP : 0);
    const gej P = ecmult_lane(rec, ge_from_words(qx, qy), slot, gtable);
    if (P.inf) {
      g.valid = 0;  // unreachable: r/s != 0 and Q has prime order
    } else {
      const fe zi = fe_inv_var(P.z), zi2 = fe_sqr(zi);
      fe_to_words(g.px, fe_normalize(fe_mul(P.x, zi2)));
      fe_to_words(g.py, fe_normalize(fe_mul(P.y, fe_mul(zi2, zi))));
      for (int i = 0; i < 8; i++) { g.sinv[i] = sinv.w[i]; g.rw[i] = r.w[i]; }
    }
  }
  *out = g;
}
// does candidate c (feerate min_rate + c) verify?  false also for the candidates the reference's loop never checks
LAMD_HD bool grind_candidate(u32 c, u32 min_rate, u64 weight, u64 input_sat, const u8 *tail, u32 tail_len, u32 lead_bytes,
                             const u8 *outputs, u32 outputs_len, const grind_setup &setup, const u32 *gtable) {
  const u64 rate = (u64)min_rate + c;
  const u64 fee = rate * weight / 1000;                                    // amount_tx_fee(), common/amount.c:698-707
  if (c > 0 && (rate - 1) * weight / 1000 == fee) return false;            // "don't check same fee twice"
  if (fee > input_sat) return false;                                       // amount_sat_sub() fails: the reference stops here
  const u64 amount = input_sat - fee;
  u8 buf[GRIND_MAX_OUTPUTS > GRIND_MAX_TAIL ? GRIND_MAX_OUTPUTS : GRIND_MAX_TAIL];
  u8 h[32];
  for (u32 i = 0; i < outputs_len; i++) buf[i] = i < 8 ? (u8)(amount >> (8 * i)) : outputs[i];
  sha256d_bytes(buf, outputs_len, h);                                      // hashOutputs
  const u32 ho = tail_len - 40;                                            // ... sits 40 bytes before the end of the preimage
  for (u32 i = 0; i < tail_len; i++) buf[i] = (i >= ho && i < ho + 32) ? h[i - ho] : tail[i];
  u32 st[8];
  for (int i = 0; i < 8; i++) st[i] = setup.mid[i];
  sha256d_finish(st, buf, tail_len, lead_bytes, h);                        // the sighash
  u32 zw[8];
  load_words_be(zw, h);
  sc sinv;
  for (int i = 0; i < 8; i++) sinv.w[i] = setup.sinv[i];
  const sc u1 = sc_mul(sc_from_words(zw, nullptr), sinv);
  u32 rw[8], pw[16];
  for (int i = 0; i < 8; i++) { rw[i] = setup.rw[i]; pw[i] = setup.px[i]; pw[8 + i] = setup.py[i]; }
  gej acc = gej_from_ge(ge_from_words(pw, pw + 8));
#pragma unroll 1
  for (int w = 0; w < GTABLE_WINDOWS; w++) {
    const u32 d = gtable_digit(u1.w, w);
    const u32 *e = gtable + (((size_t)w << GTABLE_WINDOW_BITS) + d) * GT_ENTRY_WORDS;
    ge pt;
    pt.x = slot_load_fe(e);
    pt.y = slot_load_fe(e + TW);
    acc = gej_add_ge(acc, pt, d == 0);
  }
  return ecdsa_final(acc, rw);
}

// ---- ECDSA public-key recovery (secp256k1_ecdsa_recover as reached from common/bolt11.c:1041-1046 and
// lightningd/signmessage.c:193): Q = (s/r)*R - (z/r)*G with R = the point whose x is r (+ n when recid & 2) and whose y
// has parity recid & 1.  It is the verification ecmult with (u1, u2, key) = (-z/r, s/r, R): the prep below writes the same
// prep_rec and R as a 33-byte compressed key for k_keys; the ecmult kernel parks the Jacobian result and k_recover_final
// brings it to affine with one shared inversion per thread.
// Fails (ok = 0) exactly where libsecp256k1 does: r or s >= n (recoverable_signature_parse_compact), recid > 3, r or s = 0,
// recid & 2 with r >= p - n, x not on the curve, Q = infinity.  No low-S rule here.
LAMD_HD void recover_load(const u8 *sig64, u8 recid, sc *r, sc *s, bool *ok) {
  u32 rw[8], sw[8];
  load_words_be(rw, sig64);
  load_words_be(sw, sig64 + 32);
  bool v = !words_ge_n(rw) & !words_ge_n(sw) & (recid < 4);
#pragma unroll
  for (int i = 0; i < 8; i++) { r->w[i] = rw[i]; s->w[i] = sw[i]; }
  v &= !sc_is_zero(*r) & !sc_is_zero(*s);
  if (recid & 2) {
    const u32 pmn[8] = LAMD_P_MINUS_N;
    v &= !words_ge(rw, pmn);
  }
  *ok = v;
}
LAMD_HD void recover_prep_thread(size_t first, size_t stride, size_t n, const u8 *hash32, const u8 *sig64, const u8 *recid,
                                 prep_rec *recs, u8 *rkey33) {
  sc acc;
#pragma unroll
  for (int i = 0; i < 8; i++) acc.w[i] = (i == 0);
  size_t last = first;
  bool any = false;
#pragma unroll 1
  for (size_t i = first; i < n; i += stride) {
    sc r, s;
    bool ok;
    recover_load(sig64 + 64 * i, recid[i], &r, &s, &ok);
#pragma unroll
    for (int k = 0; k < 8; k++) recs[i].u1[k] = acc.w[k];  // product of the valid r before i
    if (ok) acc = sc_mul(acc, r);
    last = i;
    any = true;
  }
  if (!any) return;
  sc inv = sc_inv_var(acc);
#pragma unroll 1
  for (size_t i = last;; i -= stride) {
    sc r, s, prefix;
    bool ok;
    recover_load(sig64 + 64 * i, recid[i], &r, &s, &ok);
#pragma unroll
    for (int k = 0; k < 8; k++) prefix.w[k] = recs[i].u1[k];
    prep_rec out;
#pragma unroll
    for (int k = 0; k < 8; k++) out.u1[k] = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) { out.k1[k] = 0x88888888u; out.k2[k] = 0x88888888u; }
    out.flags = 0;
    out.pad[0] = out.pad[1] = out.pad[2] = 0;
    u8 *key = rkey33 + 33 * i;
    for (int b = 0; b < 33; b++) key[b] = 0;  // prefix 0: k_keys rejects the row
    if (ok) {
      const sc rinv = sc_mul(inv, prefix);
      inv = sc_mul(inv, r);
      u32 zw[8];
      load_words_be(zw, hash32 + 32 * i);
      const sc z = sc_from_words(zw, nullptr);
      const sc u1 = sc_neg(sc_mul(z, rinv));
      const sc u2 = sc_mul(s, rinv);
      glv_half h1, h2;
      glv_split(&h1, &h2, u2);
#pragma unroll
      for (int k = 0; k < 8; k++) out.u1[k] = u1.w[k];
#pragma unroll
      for (int k = 0; k < 4; k++) { out.k1[k] = h1.mag[k]; out.k2[k] = h2.mag[k]; }
      out.flags = PREP_VALID | (h1.neg ? PREP_K1NEG : 0) | (h2.neg ? PREP_K2NEG : 0) | (h1.top ? PREP_K1TOP : 0) |
                  (h2.top ? PREP_K2TOP : 0);
      // x(R) = r, or r + n (< p: checked in recover_load)
      u32 xw[8];
      u64 c = 0;
      const u32 nw[8] = LAMD_SC_N;
#pragma unroll
      for (int k = 0; k < 8; k++) {
        c += (u64)r.w[k] + ((recid[i] & 2) ? nw[k] : 0u);
        xw[k] = (u32)c;
        c >>= 32;
      }
      key[0] = 2 + (recid[i] & 1);
      for (int k = 0; k < 8; k++) {
        const u32 v = xw[7 - k];
        key[1 + 4 * k] = (u8)(v >> 24); key[2 + 4 * k] = (u8)(v >> 16); key[3 + 4 * k] = (u8)(v >> 8); key[4 + 4 * k] = (u8)v;
      }
    }
    recs[i] = out;
    if (i == first) break;
  }
}
constexpr int SLOT_REC_X = 0, SLOT_REC_Y = 9, SLOT_REC_Z = 18, SLOT_REC_PREFIX = 27;
LAMD_HD u8 recover_stage1(const gej &Q, u32 *slot) {
  if (Q.inf) return 0;
  const fe z = fe_norm_weak(Q.z);
#pragma unroll
  for (int k = 0; k < 9; k++) { slot[SLOT_REC_X + k] = Q.x.n[k]; slot[SLOT_REC_Y + k] = Q.y.n[k]; slot[SLOT_REC_Z + k] = z.n[k]; }
  return SCHNORR_PENDING;
}
// out[i]: SCHNORR_PENDING -> 1 with the compressed key in pub33[i]; everything else -> 0 and a zeroed key
LAMD_HD void recover_final_thread(size_t first, size_t stride, size_t n, u32 *slots, u8 *out, u8 *pub33) {
  fe acc = fe_set_int(1);
  size_t last = first;
  bool any = false;
#pragma unroll 1
  for (size_t i = first; i < n; i += stride) {
    last = i;
    any = true;
    if (out[i] != SCHNORR_PENDING) continue;
    u32 *slot = slots + i * SLOT_WORDS;
#pragma unroll
    for (int k = 0; k < 9; k++) slot[SLOT_REC_PREFIX + k] = acc.n[k];
    acc = fe_mul(acc, slot_load_raw(slot + SLOT_REC_Z));
  }
  if (!any) return;
  fe inv = fe_inv_var(acc);
#pragma unroll 1
  for (size_t i = last;; i -= stride) {
    u8 *key = pub33 + 33 * i;
    if (out[i] == SCHNORR_PENDING) {
      const u32 *slot = slots + i * SLOT_WORDS;
      const fe zi = fe_mul(inv, slot_load_raw(slot + SLOT_REC_PREFIX));
      inv = fe_mul(inv, slot_load_raw(slot + SLOT_REC_Z));
      const fe zi2 = fe_sqr(zi);
      u32 xw[8];
      fe_to_words(xw, fe_normalize(fe_mul(slot_load_raw(slot + SLOT_REC_X), zi2)));
      const fe y = fe_normalize(fe_mul(slot_load_raw(slot + SLOT_REC_Y), fe_mul(zi2, zi)));
      key[0] = 2 + (y.n[0] & 1);
      for (int k = 0; k < 8; k++) {
        const u32 v = xw[7 - k];
        key[1 + 4 * k] = (u8)(v >> 24); key[2 + 4 * k] = (u8)(v >> 16); key[3 + 4 * k] = (u8)(v >> 8); key[4 + 4 * k] = (u8)v;
      }
      out[i] = 1;
    } else {
      for (int b = 0; b < 33; b++) key[b] = 0;
      out[i] = 0;
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
  const fe zi = fe_inv_var(R.z);
  const fe y = fe_normalize(fe_mul(R.y, fe_mul(fe_sqr(zi), zi)));
  return (y.n[0] & 1) == 0;
}


// ================================================================================================
// Per-row front ends of the callers that hash first -- one source for the kernels (k_txsig_hash, k_txsig_tx_hash, k_gossip_expand,
// k_gossip_reduce) and for the host-side latency paths of lamd_check_tx_sig_*batch / lamd_sigcheck_gossip_batch (a few rows per call:
// hashing on the host, the rows through k_small_verify)
// one row of check_tx_sig: the sighash-type gate (only SIGHASH_ALL, or SINGLE|ANYONECANPAY with a witness script) and the double SHA-256
// of the caller's preimage; shared by the kernel and the host-side latency path (a few rows: lamd_check_tx_sig_batch)
LAMD_HD bool txsig_hash_one(const u8 *pre, size_t len, u8 sighash_type, bool has_witness, u8 hash32[32]) {
  const bool pass = sighash_type == 1 || (sighash_type == 0x83 && has_witness);
  u8 h[32];
  if (pass) sha256d_bytes(pre, len, h);
  for (int b = 0; b < 32; b++) hash32[b] = pass ? h[b] : 0;
  return pass;
}
// ---- check_tx_sig from transaction templates: the BIP143 hash of bitcoin_tx_hash_for_sig() (bitcoin/signature.c:120-151) computed
// here from flat template arrays (verify_core.h "BIP143 signature hash on the device"), plus the sighash-type gate of :206-211
LAMD_HD bool txsig_tx_hash_one(size_t i, const u32 *version, const u32 *locktime, const u8 *inputs40, const u64 *in_off, const u32 *input_num,
                               const u64 *amount, const u8 *outputs, const u64 *out_off, const u32 *n_outputs, const u8 *scripts,
                               const u64 *script_off, const u8 *sighash_type, const u8 *has_witness, u8 hash32[32]) {
  const u8 t = sighash_type[i];
  bool pass = t == 1 || (t == 0x83 && has_witness[i]);
  u8 h[32];
  for (int b = 0; b < 32; b++) h[b] = 0;
  if (pass) {
    tx_view tv;
    tv.version = version[i];
    tv.locktime = locktime[i];
    tv.inputs = inputs40 + 40 * in_off[i];
    tv.n_in = (u32)(in_off[i + 1] - in_off[i]);
    tv.outputs = outputs + out_off[i];
    tv.outputs_len = (size_t)(out_off[i + 1] - out_off[i]);
    tv.n_out = n_outputs[i];
    pass = bip143_sighash(tv, input_num[i], scripts + script_off[i], (size_t)(script_off[i + 1] - script_off[i]), amount[i], t, h);
  }
  for (int b = 0; b < 32; b++) hash32[b] = h[b];
  return pass;
}
// one message -> its signature rows (hash of the signed tail, signature k, signer k; strides 32 / 64 / 33) and whether it is malformed
// (framing, or a signature whose r or s is out of range: fromwire_secp256k1_ecdsa_signature fails).  Shared by the kernel below and by the
// host-side latency path of lamd_sigcheck_gossip_batch (a handful of messages: the rows go through k_small_verify).
LAMD_HD bool gossip_expand_one(const u8 *m, size_t len, const u8 *node_id33, size_t nrows, u8 *hash32, u8 *sig64, u8 *pub33) {
  const gossip_frame fr = gossip_parse_frame(m, len);
  const u32 type = fr.type;
  // a channel_update without a signer cannot be decided: the message is reported malformed (-1), nothing is read through the null pointer
  bool bad = fr.bad | (type == GOSSIP_CUPD && node_id33 == nullptr);
  const size_t signed_off = fr.signed_off, keyoff = fr.keyoff;
  u8 h[32];
  if (!bad) sha256d_bytes(m + signed_off, len - signed_off, h);
  for (size_t k = 0; k < nrows; k++) {
    u8 *hd = hash32 + 32 * k, *sd = sig64 + 64 * k, *pd = pub33 + 33 * k;
    if (bad) {
      for (int b = 0; b < 32; b++) hd[b] = 0;
      for (int b = 0; b < 64; b++) sd[b] = 0;
      for (int b = 0; b < 33; b++) pd[b] = 0;
      continue;
    }
    const u8 *sp = m + 2 + 64 * k;
    const u8 *kp = (type == GOSSIP_CUPD) ? node_id33 : m + keyoff + 33 * k;
    u32 rw[8], sw[8];
    load_words_be(rw, sp);
    load_words_be(sw, sp + 32);
    if (words_ge_n(rw) | words_ge_n(sw)) bad = true;  // fromwire_secp256k1_ecdsa_signature fails: malformed
    for (int b = 0; b < 32; b++) hd[b] = h[b];
    for (int b = 0; b < 64; b++) sd[b] = sp[b];
    for (int b = 0; b < 33; b++) pd[b] = kp[b];
  }
  return bad;
}
// verdicts of a message's rows -> the message's verdict: -1 malformed (incl. a bitcoin key that does not parse: fromwire_pubkey),
// 0 all signatures good, k = the FIRST failing signature (sigcheck.c:78-113 order)
LAMD_HD int gossip_reduce_one(size_t nrows, const u8 *ok, const u8 *keyok, bool malformed) {
  bool bad = malformed;
  if (!bad && nrows == 4) bad = !keyok[2] | !keyok[3];  // fromwire_pubkey on bitcoin_key_1/2
  if (bad) return -1;
  int v = 0;
  for (size_t k = nrows; k-- > 0;)
    if (!ok[k]) v = (int)k + 1;
  return v;
}

}  // namespace lamd

// CPU build of the DEVICE arithmetic (lightning_amd/csrc/*.h compiled for the host with
// magnitude checking on).  Test infrastructure only: lets the CPU test-suite drive exactly the
// code the HIP kernels run (field, scalar, GLV, table build, ladder, final checks) against
// Python big-ints and the oracle before anything touches a GPU.  Not a product path: the
// shipped library (liblightning_amd.so) contains no CPU verification code.
#define LAMD_CHECK_MAG 1
#ifndef LAMD_GTABLE_WINDOW_BITS
#define LAMD_GTABLE_WINDOW_BITS 8
#endif
#include <string>
#include "../lightning_amd/csrc/verify_core.h"
#include <stdlib.h>
#include <string.h>
#include <vector>
using namespace lamd;

static void be_to_words(u32 w[8], const u8 *b) { for (int i = 0; i < 8; i++) w[7 - i] = load_be32(b + 4 * i); }
static void words_to_be(u8 *b, const u32 w[8]) { for (int i = 0; i < 8; i++) { u32 v = w[7 - i]; b[4*i] = v >> 24; b[4*i+1] = v >> 16; b[4*i+2] = v >> 8; b[4*i+3] = v; } }

static fe fe_from_raw(const u32 *limbs, int mag) { fe a; for (int i = 0; i < 9; i++) a.n[i] = limbs[i]; a.mag = mag; fe_verify(a); return a; }
static void fe_to_raw(u32 *limbs, int *mag, const fe &a) { for (int i = 0; i < 9; i++) limbs[i] = a.n[i]; *mag = a.mag; }

extern "C" {
// op codes: 0 add, 1 neg(a, mag), 2 mul, 3 sqr, 4 norm_weak, 5 carry, 6 normalize, 7 mul_int(k=b_mag), 8 is_zero (out[0])
int dm_fe_op(int op, const u32 *a, int amag, const u32 *b, int bmag, u32 *out, int *omag) {
  fe x = fe_from_raw(a, amag), y, r;
  if (op == 0 || op == 2) y = fe_from_raw(b, bmag);
  switch (op) {
    case 0: r = fe_add(x, y); break;
    case 1: r = fe_neg(x, amag); break;
    case 2: r = fe_mul(x, y); break;
    case 3: r = fe_sqr(x); break;
    case 4: r = fe_norm_weak(x); break;
    case 5: r = fe_carry(x); break;
    case 6: r = fe_normalize(x); break;
    case 7: r = fe_mul_int(x, (u32)bmag); break;
    case 8: r = fe_zero(); r.n[0] = fe_is_zero(x); break;
    default: return -1;
  }
  fe_to_raw(out, omag, r);
  return 0;
}
void dm_fe_from_be(const u8 *b, u32 *limbs) { u32 w[8]; be_to_words(w, b); fe a = fe_from_words(w); for (int i = 0; i < 9; i++) limbs[i] = a.n[i]; }
void dm_fe_to_be(const u32 *limbs, int mag, u8 *b) { fe a = fe_from_raw(limbs, mag); u32 w[8]; fe_to_words(w, fe_normalize(a)); words_to_be(b, w); }
void dm_fe_inv(const u8 *a, u8 *out) { u32 w[8]; be_to_words(w, a); fe r = fe_inv(fe_from_words(w)); fe_to_words(w, fe_normalize(r)); words_to_be(out, w); }
int dm_fe_sqrt(const u8 *a, u8 *out) { u32 w[8]; be_to_words(w, a); fe x = fe_from_words(w); fe r = fe_sqrt_candidate(x); int ok = fe_equal(x, fe_sqr(r), 1); fe_to_words(w, fe_normalize(r)); words_to_be(out, w); return ok; }
int dm_words_ge_p(const u8 *a) { u32 w[8]; be_to_words(w, a); return words_ge_p(w); }

void dm_sc_mul(const u8 *a, const u8 *b, u8 *out) { sc x, y; be_to_words(x.w, a); be_to_words(y.w, b); sc r = sc_mul(x, y); words_to_be(out, r.w); }
void dm_sc_inv(const u8 *a, u8 *out) { sc x; be_to_words(x.w, a); sc r = sc_inv(x); words_to_be(out, r.w); }
void dm_sc_neg(const u8 *a, u8 *out) { sc x; be_to_words(x.w, a); sc r = sc_neg(x); words_to_be(out, r.w); }
int dm_sc_from(const u8 *a, u8 *out) { u32 w[8]; be_to_words(w, a); bool of; sc r = sc_from_words(w, &of); words_to_be(out, r.w); return of; }
int dm_sc_is_high(const u8 *a) { sc x; be_to_words(x.w, a); return sc_is_high(x); }
// out: k1mag[16 bytes LE words], top1, neg1, k2mag, top2, neg2 as u32[12]
void dm_glv_split(const u8 *k, u32 *out) { sc x; be_to_words(x.w, k); glv_half h1, h2; glv_split(&h1, &h2, x);
  for (int i = 0; i < 4; i++) { out[i] = h1.mag[i]; out[6 + i] = h2.mag[i]; } out[4] = h1.top; out[5] = h1.neg; out[10] = h2.top; out[11] = h2.neg; }
void dm_bip340_challenge(const u8 *r, const u8 *pk, const u8 *m, u8 *out) {
  u32 rb[8], pb[8], mb[8], o[8];
  for (int i = 0; i < 8; i++) { rb[i] = load_be32(r + 4*i); pb[i] = load_be32(pk + 4*i); mb[i] = load_be32(m + 4*i); }
  bip340_challenge(o, rb, pb, mb);
  for (int i = 0; i < 8; i++) { out[4*i] = o[i] >> 24; out[4*i+1] = o[i] >> 16; out[4*i+2] = o[i] >> 8; out[4*i+3] = o[i]; }
}
int dm_parse_pubkey(const u8 *p, int len, u8 *out64) { u32 qx[8], qy[8]; bool ok = parse_pubkey(p, len, qx, qy); words_to_be(out64, qx); words_to_be(out64 + 32, qy); return ok; }

// ---- static G table (8-bit windows in this build), built with the same entry function the device uses
static std::vector<u32> g_table;
void dm_init(void) {
  if (!g_table.empty()) return;
  g_table.assign(GTABLE_ENTRIES * GT_ENTRY_WORDS, 0);
  const u32 gx[8] = LAMD_GX, gy[8] = LAMD_GY;
  u32 base[16];
  memcpy(base, gx, 32); memcpy(base + 8, gy, 32);
  for (int w = 0; w < GTABLE_WINDOWS; w++) {
    for (u32 d = 1; d < (1u << GTABLE_WINDOW_BITS); d++)
      gtable_compute_entry(&g_table[(((size_t)w << GTABLE_WINDOW_BITS) + d) * GT_ENTRY_WORDS], base, d);
    // next base = 2^BITS * base
    gej b = gej_from_ge(ge_from_words(base, base + 8));
    for (int i = 0; i < GTABLE_WINDOW_BITS; i++) b = gej_double(b);
    const fe zi = fe_inv(fe_norm_weak(b.z)); const fe zi2 = fe_sqr(zi);
    fe_to_words(base, fe_normalize(fe_mul(b.x, zi2)));
    fe_to_words(base + 8, fe_normalize(fe_mul(b.y, fe_mul(zi2, zi))));
  }
}
// table entry as 64 big-endian bytes
void dm_gtable_entry(int w, u32 d, u8 *out64) {
  dm_init();
  const u32 *e = &g_table[(((size_t)w << GTABLE_WINDOW_BITS) + d) * GT_ENTRY_WORDS];
  u32 xw[8], yw[8];
  fe_to_words(xw, fe_normalize(slot_load_fe(e)));
  fe_to_words(yw, fe_normalize(slot_load_fe(e + TW)));
  words_to_be(out64, xw);
  words_to_be(out64 + 32, yw);
}

// Whole pipeline exactly as the kernels stage it.  threads = how many "prep threads" share the batch
// (exercises the Montgomery batch inversion with different group sizes).
void dm_ecdsa_verify_batch(size_t n, const u8 *hash32, const u8 *sig64, const u8 *pub, int publen, int pubstride, u8 *out, size_t threads) {
  dm_init();
  std::vector<prep_rec> recs(n);
  for (size_t t = 0; t < threads; t++) ecdsa_prep_thread(t, threads, n, hash32, sig64, recs.data());
  std::vector<u32> slot(SLOT_WORDS);
  for (size_t i = 0; i < n; i++) {
    u32 qx[8], qy[8], rw[8];
    bool ok = parse_pubkey(pub + (size_t)pubstride * i, publen, qx, qy);
    ok &= (recs[i].flags & PREP_VALID) != 0;
    if (ok) {
      const gej R = ecmult_lane(recs[i], ge_from_words(qx, qy), slot.data(), g_table.data());
      be_to_words(rw, sig64 + 64 * i);
      ok = ecdsa_final(R, rw);
    }
    out[i] = ok;
  }
}
#ifndef DM_NO_KEYED
static size_t g_suspects;  // how often the bare-formula ecmult reported Z = 0 and the complete form decided
size_t dm_suspects(int reset) { const size_t v = g_suspects; if (reset) g_suspects = 0; return v; }
// keyed path: one window table per row's key (no sharing here -- this is an arithmetic test), then the table-driven ecmult
}  // extern "C"
// which builder makes the key tables of this harness: 0 = the Gray-code chains (keytable_build), nk = 1..4: the affine tree (keytable_build_tree) with nk
// keys per lane -- the key under test LAST, the lane's other keys 2Q, 4Q, .. (so that the shared inversions really mix keys)
static int g_tree_builder = 0;
extern "C" void dm_set_table_builder(int nk) { g_tree_builder = nk < 0 ? 0 : nk > KC_TREE_MAXK ? KC_TREE_MAXK : nk; }
template <int T>
static void build_table(u32 *tab, u32 *scratch, const ge &q) {
  if (!g_tree_builder) { keytable_build<T>(tab, scratch, q); return; }
  const int nk = g_tree_builder;
  std::vector<std::vector<u32>> tabs(nk, std::vector<u32>(kc_stride(T))), scrs(nk, std::vector<u32>(kc_scratch_words(T)));
  std::vector<ge> qs(nk);
  u32 *tp[KC_TREE_MAXK], *sp[KC_TREE_MAXK];
  gej p = gej_from_ge(q);
  for (int k = nk - 1; k >= 0; k--) {
    if (k == nk - 1) qs[k] = q;
    else {
      p = gej_double(p);
      const fe zi = fe_inv(fe_norm_weak(p.z)), zi2 = fe_sqr(zi);
      qs[k].x = fe_normalize(fe_mul(p.x, zi2));
      qs[k].y = fe_normalize(fe_mul(p.y, fe_mul(zi2, zi)));
    }
    tp[k] = tabs[k].data();
    sp[k] = scrs[k].data();
  }
  keytable_build_tree<T>(tp, sp, qs.data(), nk);
  memcpy(tab, tabs[nk - 1].data(), sizeof(u32) * kc_stride(T));
}
template <int T>
static void verify_keyed_t(int mode, size_t n, const u8 *a32, const u8 *sig64, const u8 *key, int keylen, u8 *out) {
  dm_init();
  std::vector<u32> tab(kc_stride(T)), scratch(kc_scratch_words(T)), fin(n * 32);
  std::vector<prep_rec> recs(n);
  if (mode == MODE_ECDSA) ecdsa_prep_thread(0, 1, n, a32, sig64, recs.data());
  for (size_t i = 0; i < n; i++) {
    if (mode == MODE_SCHNORR) schnorr_prep_one(a32 + 32 * i, key + (size_t)keylen * i, sig64 + 64 * i, &recs[i]);
    u32 qx[8], qy[8], rw[8];
    bool ok = parse_pubkey(key + (size_t)keylen * i, keylen, qx, qy);
    ok &= (recs[i].flags & PREP_VALID) != 0;
    out[i] = 0;
    if (ok) {
      build_table<T>(tab.data(), scratch.data(), ge_from_words(qx, qy));
      // as the kernels stage it: the bare-formula form first, the complete form only when it reports Z = 0
      bool suspect;
      const gexz R = ecmult_lane_keyed_fast<T>(recs[i], tab.data(), g_table.data(), &suspect);
      be_to_words(rw, sig64 + 64 * i);
      if (suspect) {
        g_suspects++;
        const gej Rc = ecmult_lane_keyed<T>(recs[i], tab.data(), g_table.data());
        out[i] = mode == MODE_ECDSA ? (u8)ecdsa_final(Rc, rw) : schnorr_stage1(Rc, rw, &fin[i * 32]);
      } else {
        out[i] = mode == MODE_ECDSA ? (u8)ecdsa_final(R, rw) : schnorr_stage1(R, rw, &fin[i * 32]);
      }
    }
  }
  if (mode == MODE_SCHNORR) schnorr_final_thread(0, 1, n, fin.data(), out, 32);
}
// the pairs-first form (verify_core.h "Pairs first"): rows in batches of `batch` <= PAIRS_BMAX per inversion, every row's key with a table of its own
template <int T>
static void verify_pairs_t(int mode, size_t n, int batch, const u8 *a32, const u8 *sig64, const u8 *key, int keylen, u8 *out) {
  dm_init();
  std::vector<u32> tabs((size_t)PAIRS_BMAX * kc_stride(T)), scratch(kc_scratch_words(T)), fin(n * 32), ws((size_t)PAIRS_SLOTS * PAIRS_WS_WORDS);
  std::vector<prep_rec> recs(n);
  if (mode == MODE_ECDSA) ecdsa_prep_thread(0, 1, n, a32, sig64, recs.data());
  else for (size_t i = 0; i < n; i++) schnorr_prep_one(a32 + 32 * i, key + (size_t)keylen * i, sig64 + 64 * i, &recs[i]);
  if (batch < 1) batch = 1;
  if (batch > PAIRS_BMAX) batch = PAIRS_BMAX;
  for (size_t base = 0; base < n; base += batch) {
    const int nb = (int)std::min<size_t>(batch, n - base);
    bool have[PAIRS_BMAX];
    for (int b = 0; b < nb; b++) {
      const size_t i = base + b;
      u32 qx[8], qy[8];
      have[b] = parse_pubkey(key + (size_t)keylen * i, keylen, qx, qy) && (recs[i].flags & PREP_VALID) != 0;
      out[i] = 0;
      if (have[b]) build_table<T>(&tabs[(size_t)b * kc_stride(T)], scratch.data(), ge_from_words(qx, qy));
    }
    pairs_batch<T>(nb, g_table.data(), ws.data(), PAIRS_WS_WORDS,
                   [&](int b, bool, const prep_rec **rec, const u32 **tab) {
                     *rec = &recs[base + b];
                     *tab = &tabs[(size_t)b * kc_stride(T)];
                     return have[b];
                   },
                   [&](int b, gej R, bool suspect) {
                     const size_t i = base + b;
                     if (suspect) { g_suspects++; R = ecmult_lane_keyed<T>(recs[i], &tabs[(size_t)b * kc_stride(T)], g_table.data()); }
                     u32 rw[8];
                     be_to_words(rw, sig64 + 64 * i);
                     out[i] = mode == MODE_ECDSA ? (u8)ecdsa_final(R, rw) : schnorr_stage1(R, rw, &fin[i * 32]);
                   });
  }
  if (mode == MODE_SCHNORR) schnorr_final_thread(0, 1, n, fin.data(), out, 32);
}
template <int T>
static void keytable_entry_t(const u32 *qx, const u32 *qy, int idx, u8 *out96) {
  std::vector<u32> tab(kc_stride(T)), scratch(kc_scratch_words(T));
  build_table<T>(tab.data(), scratch.data(), ge_from_words(qx, qy));
  // entries are affine on the key's isomorphic curve: x = x_true * Zc^2, y = y_true * Zc^3
  const fe zc = slot_load_fe(&tab[kc_words(T)]);
  const fe zi = fe_inv(zc), zi2 = fe_sqr(zi), zi3 = fe_mul(zi2, zi);
  const u32 *e = &tab[idx * SLOT_ENTRY_WORDS];
  u32 w[8];
  fe_to_words(w, fe_normalize(fe_mul(slot_load_fe(e), zi2))); words_to_be(out96, w);
  fe_to_words(w, fe_normalize(fe_mul(slot_load_fe(e + ENT_Y), zi3))); words_to_be(out96 + 32, w);
  fe_to_words(w, fe_normalize(fe_mul(slot_load_fe(e + ENT_BX), zi2))); words_to_be(out96 + 64, w);
}
// The latency path's task split (verify_core.h "Task split"): every task evaluated on its own, merged with complete additions --
// T = 7 / 10: comb tasks over the key's table; T = 0: the two ladder halves over the lane's own 8-entry table
template <int T>
static void verify_split_t(int mode, size_t n, const u8 *a32, const u8 *sig64, const u8 *key, int keylen, u8 *out) {
  dm_init();
  std::vector<u32> tab(kc_stride(T ? T : 7)), scratch(kc_scratch_words(T ? T : 7)), slot(SLOT_WORDS);
  std::vector<prep_rec> recs(n);
  for (size_t i = 0; i < n; i++) {
    if (mode == MODE_ECDSA) ecdsa_prep_thread(i, n, n, a32, sig64, recs.data());   // one row per "lane", as k_small_verify runs it
    else schnorr_prep_one(a32 + 32 * i, key + (size_t)keylen * i, sig64 + 64 * i, &recs[i]);
    u32 qx[8], qy[8], rw[8];
    bool ok = parse_pubkey(key + (size_t)keylen * i, keylen, qx, qy);
    ok &= (recs[i].flags & PREP_VALID) != 0;
    out[i] = 0;
    if (!ok) continue;
    gej parts[ST_TASKS];
    fe zscale;
    parts[ST_G] = small_task_g(recs[i], g_table.data());
    if (T) {
      build_table<(T ? T : 7)>(tab.data(), scratch.data(), ge_from_words(qx, qy));
      bool rt_same = true;
      for (int t = ST_H1LO; t <= ST_H2HI; t++) {
        parts[t] = small_task_comb<(T ? T : 7)>(recs[i], tab.data(), t);
        // the form k_small_verify runs (shape known at run time) must give the same part, coordinate for coordinate
        const gej q = small_task_comb_rt(recs[i], tab.data(), t, T ? T : 7);
        rt_same &= q.inf == parts[t].inf;
        if (!q.inf && !parts[t].inf)
          rt_same &= fe_equal(fe_norm_weak(q.x), fe_norm_weak(parts[t].x), 1) && fe_equal(fe_norm_weak(q.y), fe_norm_weak(parts[t].y), 1) &&
                     fe_equal(fe_norm_weak(q.z), fe_norm_weak(parts[t].z), 1);
        parts[t] = q;
      }
      if (!rt_same) { out[i] = 0xEF; continue; }   // a verdict no test expects
      zscale = slot_load_fe(&tab[kc_words(T ? T : 7)]);
    } else {
      zscale = build_q_table(slot.data(), ge_from_words(qx, qy));
      parts[ST_H1LO] = small_task_ladder(recs[i], slot.data(), false);
      parts[ST_H2LO] = small_task_ladder(recs[i], slot.data(), true);
      parts[ST_H1HI] = parts[ST_H2HI] = gej_infinity();
    }
    gej R = small_merge(parts, zscale);
    {  // ... and as k_small_verify stages it: four task waves, u1*G spread over them, a three-level merge -- must describe the same point
      gej p4[4] = {parts[ST_H1LO], parts[ST_H1HI], parts[ST_H2LO], parts[ST_H2HI]}, g4[4];
      for (int t = 0; t < 4; t++) {
        int lo, hi;
        small_g_windows(t, T == 0, &lo, &hi);
        g4[t] = small_task_g(recs[i], g_table.data(), lo, hi);
      }
      const gej R4 = small_merge4(p4, g4, zscale);
      bool same = R.inf == R4.inf;
      if (same && !R.inf) {
        const fe z1 = fe_norm_weak(R.z), z2 = fe_norm_weak(R4.z), z1s = fe_sqr(z1), z2s = fe_sqr(z2);
        same = fe_equal(fe_mul(R.x, z2s), fe_mul(R4.x, z1s), 1) && fe_equal(fe_mul(R.y, fe_mul(z2s, z2)), fe_mul(R4.y, fe_mul(z1s, z1)), 1);
      }
      if (!same) { out[i] = 0xEE; continue; }   // a verdict no test expects
      R = R4;
    }
    be_to_words(rw, sig64 + 64 * i);
    out[i] = mode == MODE_ECDSA ? (u8)ecdsa_final(R, rw) : (u8)schnorr_accept_one(R, rw);
  }
}
extern "C" void dm_verify_split(int mode, int T, size_t n, const u8 *a32, const u8 *sig64, const u8 *key, int keylen, u8 *out) {
  if (T == 7) verify_split_t<7>(mode, n, a32, sig64, key, keylen, out);
  else if (T == 10) verify_split_t<10>(mode, n, a32, sig64, key, keylen, out);
  else verify_split_t<0>(mode, n, a32, sig64, key, keylen, out);
}
// the per-row front ends the kernels share with the host-side latency paths (verify_core.h, last section)
extern "C" int dm_gossip_expand(const u8 *m, size_t len, const u8 *node_id33, size_t nrows, u8 *hash32, u8 *sig64, u8 *pub33) {
  return gossip_expand_one(m, len, node_id33, nrows, hash32, sig64, pub33) ? 1 : 0;
}
extern "C" int dm_gossip_reduce(size_t nrows, const u8 *ok, const u8 *keyok, int malformed) { return gossip_reduce_one(nrows, ok, keyok, malformed != 0); }
extern "C" int dm_txsig_hash(const u8 *pre, size_t len, int sighash_type, int has_witness, u8 *hash32) {
  return txsig_hash_one(pre, len, (u8)sighash_type, has_witness != 0, hash32) ? 1 : 0;
}
extern "C" {
int dm_comb_spacing(int T) { return kc_spacing(T); }
void dm_verify_pairs(int mode, int T, size_t n, int batch, const u8 *a32, const u8 *sig64, const u8 *key, int keylen, u8 *out) {
  if (T == 7) verify_pairs_t<7>(mode, n, batch, a32, sig64, key, keylen, out);
  else if (T == 8) verify_pairs_t<8>(mode, n, batch, a32, sig64, key, keylen, out);
  else verify_pairs_t<10>(mode, n, batch, a32, sig64, key, keylen, out);
}
void dm_verify_keyed(int mode, int T, size_t n, const u8 *a32, const u8 *sig64, const u8 *key, int keylen, u8 *out) {
  if (T == 7) verify_keyed_t<7>(mode, n, a32, sig64, key, keylen, out);
  else if (T == 8) verify_keyed_t<8>(mode, n, a32, sig64, key, keylen, out);
  else if (T == 9) verify_keyed_t<9>(mode, n, a32, sig64, key, keylen, out);
  else verify_keyed_t<10>(mode, n, a32, sig64, key, keylen, out);
}
// comb table entry idx (0..2^(T-1): the last one is Q itself) of a key as 64 affine bytes + 32 bytes beta*x
void dm_keytable_entry(const u8 *key33, int T, int idx, u8 *out96) {
  u32 qx[8], qy[8];
  parse_pubkey(key33, 33, qx, qy);
  if (T == 7) keytable_entry_t<7>(qx, qy, idx, out96);
  else if (T == 8) keytable_entry_t<8>(qx, qy, idx, out96);
  else if (T == 9) keytable_entry_t<9>(qx, qy, idx, out96);
  else keytable_entry_t<10>(qx, qy, idx, out96);
}
// R = u1*G + u2*Q through the GLV split, the key's comb table and the table-driven ecmult; returns 0 for infinity
}  // extern "C"
template <int T>
static int ecmult_keyed_t(const u32 *qx, const u32 *qy, const u8 *u1, const u8 *u2, u8 *out64) {
  dm_init();
  std::vector<u32> tab(kc_stride(T)), scratch(kc_scratch_words(T));
  build_table<T>(tab.data(), scratch.data(), ge_from_words(qx, qy));
  prep_rec rec;
  sc k; be_to_words(k.w, u2);
  be_to_words(rec.u1, u1);
  glv_half h1, h2;
  glv_split(&h1, &h2, k);
  for (int i = 0; i < 4; i++) { rec.k1[i] = h1.mag[i]; rec.k2[i] = h2.mag[i]; }
  rec.flags = PREP_VALID | (h1.neg ? PREP_K1NEG : 0) | (h2.neg ? PREP_K2NEG : 0) | (h1.top ? PREP_K1TOP : 0) | (h2.top ? PREP_K2TOP : 0);
  bool suspect;
  const gexz Rx = ecmult_lane_keyed_fast<T>(rec, tab.data(), g_table.data(), &suspect);
  const gej Rc = ecmult_lane_keyed<T>(rec, tab.data(), g_table.data());
  const gej R = Rc;
  if (suspect) {
    g_suspects++;
  } else {
    // the two forms must describe the same point: X1*Z2^2 == X2*ZZ1, Y1*Z2^3 == Y2*ZZZ1 -- and the hot form's ZZ, ZZZ must be a square and a cube
    // of ONE Z (ZZ^3 == ZZZ^2), or the BIP-340 parity stage would be fed a y of another curve
    if (Rc.inf) return -1;
    const fe z2 = fe_norm_weak(Rc.z), z2s = fe_sqr(z2);
    if (!fe_equal(fe_mul(Rx.x, z2s), fe_mul(Rc.x, Rx.zz), 1)) return -1;
    if (!fe_equal(fe_mul(Rx.y, fe_mul(z2s, z2)), fe_mul(Rc.y, Rx.zzz), 1)) return -1;
    if (!fe_equal(fe_mul(fe_sqr(Rx.zz), Rx.zz), fe_sqr(Rx.zzz), 1)) return -1;
  }
  if (R.inf) return 0;
  const fe zi = fe_inv(fe_norm_weak(R.z)), zi2 = fe_sqr(zi);
  u32 w[8];
  fe_to_words(w, fe_normalize(fe_mul(R.x, zi2))); words_to_be(out64, w);
  fe_to_words(w, fe_normalize(fe_mul(R.y, fe_mul(zi2, zi)))); words_to_be(out64 + 32, w);
  return 1;
}
extern "C" {
int dm_ecmult_keyed(int T, const u8 *key33, const u8 *u1, const u8 *u2, u8 *out64) {
  u32 qx[8], qy[8];
  parse_pubkey(key33, 33, qx, qy);
  if (T == 7) return ecmult_keyed_t<7>(qx, qy, u1, u2, out64);
  if (T == 8) return ecmult_keyed_t<8>(qx, qy, u1, u2, out64);
  if (T == 9) return ecmult_keyed_t<9>(qx, qy, u1, u2, out64);
  return ecmult_keyed_t<10>(qx, qy, u1, u2, out64);
}
#endif  // DM_NO_KEYED
// R = u1*G + u2*Q through the per-signature ladder: the hot form (signed odd digits, bare additions, one Z == 0 test: ecmult_lane_fast) AND the complete
// form (ecmult_lane); -1 if they describe different points, 0 for infinity, 1 with the affine point otherwise.  dm_ladder_suspects: how often the
// hot form reported Z = 0 and the complete form decided.
static size_t g_ladder_suspects;
size_t dm_ladder_suspects(int reset) { const size_t v = g_ladder_suspects; if (reset) g_ladder_suspects = 0; return v; }
int dm_ecmult_ladder(const u8 *key33, const u8 *u1, const u8 *u2, u8 *out64) {
  dm_init();
  u32 qx[8], qy[8];
  if (!parse_pubkey(key33, 33, qx, qy)) return -2;
  const ge q = ge_from_words(qx, qy);
  prep_rec rec;
  sc k; be_to_words(k.w, u2);
  be_to_words(rec.u1, u1);
  glv_half h1, h2;
  glv_split(&h1, &h2, k);
  for (int i = 0; i < 4; i++) { rec.k1[i] = h1.mag[i]; rec.k2[i] = h2.mag[i]; }
  rec.flags = PREP_VALID | (h1.neg ? PREP_K1NEG : 0) | (h2.neg ? PREP_K2NEG : 0) | (h1.top ? PREP_K1TOP : 0) | (h2.top ? PREP_K2TOP : 0);
  std::vector<u32> slot(SLOT_WORDS), slot2(SLOT_WORDS);
  bool suspect;
  gej R = ecmult_lane_fast(rec, q, slot.data(), g_table.data(), &suspect);
  const gej Rc = ecmult_lane(rec, q, slot2.data(), g_table.data());
  if (suspect) {
    g_ladder_suspects++;
    R = Rc;
  } else {
    if (Rc.inf) return -1;
    const fe z1 = fe_norm_weak(R.z), z2 = fe_norm_weak(Rc.z), z1s = fe_sqr(z1), z2s = fe_sqr(z2);
    if (!fe_equal(fe_mul(R.x, z2s), fe_mul(Rc.x, z1s), 1)) return -1;
    if (!fe_equal(fe_mul(R.y, fe_mul(z2s, z2)), fe_mul(Rc.y, fe_mul(z1s, z1)), 1)) return -1;
  }
  if (R.inf) return 0;
  const fe zi = fe_inv(fe_norm_weak(R.z)), zi2 = fe_sqr(zi);
  u32 w[8];
  fe_to_words(w, fe_normalize(fe_mul(R.x, zi2))); words_to_be(out64, w);
  fe_to_words(w, fe_normalize(fe_mul(R.y, fe_mul(zi2, zi)))); words_to_be(out64 + 32, w);
  return 1;
}
// the ladder's table of odd multiples: entry e (0..7) as x | beta*x | y (canonical, 32 bytes each) and the Z that maps its curve back
void dm_odd_table(const u8 *key33, u8 *out_entries /*8 * 96*/, u8 *zg32) {
  dm_init();
  u32 qx[8], qy[8];
  parse_pubkey(key33, 33, qx, qy);
  std::vector<u32> slot(SLOT_WORDS);
  const fe zg = build_odd_multiples8(slot.data(), slot.data() + SLOT_H_OFF, ge_from_words(qx, qy));
  u32 w[8];
  fe_to_words(w, fe_normalize(zg)); words_to_be(zg32, w);
  for (int e = 0; e < 8; e++)
    for (int c = 0; c < 3; c++) {
      fe_to_words(w, fe_normalize(slot_load_fe(slot.data() + e * SLOT_ENTRY_WORDS + c * TW)));
      words_to_be(out_entries + 96 * e + 32 * c, w);
    }
}
// public-key recovery exactly as the kernels stage it: prep (batch inversion over `threads` owners) -> key parse of R ->
// ladder -> shared-inversion final stage
void dm_recover_batch(size_t n, const u8 *hash32, const u8 *sig64, const u8 *recid, u8 *pub33, u8 *ok, size_t threads) {
  dm_init();
  std::vector<prep_rec> recs(n);
  std::vector<u8> rkey(n * 33);
  for (size_t t = 0; t < threads; t++) recover_prep_thread(t, threads, n, hash32, sig64, recid, recs.data(), rkey.data());
  std::vector<u32> slots(n * SLOT_WORDS);
  for (size_t i = 0; i < n; i++) {
    u32 qx[8], qy[8];
    bool v = parse_pubkey(&rkey[33 * i], 33, qx, qy);
    v &= (recs[i].flags & PREP_VALID) != 0;
    ok[i] = 0;
    if (v) {
      const gej Q = ecmult_lane(recs[i], ge_from_words(qx, qy), &slots[i * SLOT_WORDS], g_table.data());
      ok[i] = recover_stage1(Q, &slots[i * SLOT_WORDS]);
    }
  }
  for (size_t t = 0; t < threads; t++) recover_final_thread(t, threads, n, slots.data(), ok, pub33);
}
// fee grind as the engine stages it (lamd_grind_htlc_tx_fee): prepare once, then every candidate feerate; returns 1 and the
// lowest matching feerate / its fee, or 0
int dm_grind(const u8 *pre, size_t pre_len, const u8 *outputs, size_t outputs_len, uint64_t input_sat, uint64_t weight, u32 min_rate,
             u32 max_rate, const u8 *sig64, int sighash_type, int has_witness, const u8 *pub33, u32 *rate_out, uint64_t *fee_out) {
  dm_init();
  if (!(sighash_type == 1 || (sighash_type == 0x83 && has_witness))) return 0;
  if (max_rate < min_rate) return 0;
  const size_t lead = ((pre_len - 40) / 64) * 64, tail_len = pre_len - lead;
  std::vector<u32> slot(SLOT_WORDS);
  grind_setup g;
  grind_prepare(&g, sig64, pub33, pre, (u32)(lead / 64), slot.data(), g_table.data());
  if (!g.valid) return 0;
  for (u32 c = 0; c <= max_rate - min_rate; c++)
    if (grind_candidate(c, min_rate, weight, input_sat, pre + lead, (u32)tail_len, (u32)lead, outputs, (u32)outputs_len, g, g_table.data())) {
      *rate_out = min_rate + c;
      *fee_out = (uint64_t)(min_rate + c) * weight / 1000;
      return 1;
    }
  return 0;
}
// two-stage form exactly as the kernels run it (shared inversion over `threads` owners)
void dm_schnorr_verify_batch2(size_t n, const u8 *msg32, const u8 *pk32, const u8 *sig64, u8 *out, size_t threads) {
  dm_init();
  std::vector<u32> slots(n * SLOT_WORDS);
  for (size_t i = 0; i < n; i++) {
    prep_rec rec;
    schnorr_prep_one(msg32 + 32 * i, pk32 + 32 * i, sig64 + 64 * i, &rec);
    u32 qx[8], qy[8], rw[8];
    bool ok = parse_pubkey(pk32 + 32 * i, 32, qx, qy);
    ok &= (rec.flags & PREP_VALID) != 0;
    out[i] = 0;
    if (ok) {
      const gej R = ecmult_lane(rec, ge_from_words(qx, qy), &slots[i * SLOT_WORDS], g_table.data());
      be_to_words(rw, sig64 + 64 * i);
      out[i] = schnorr_stage1(R, rw, &slots[i * SLOT_WORDS]);
    }
  }
  for (size_t t = 0; t < threads; t++) schnorr_final_thread(t, threads, n, slots.data(), out);
}
void dm_schnorr_verify_batch(size_t n, const u8 *msg32, const u8 *pk32, const u8 *sig64, u8 *out) {
  dm_init();
  std::vector<u32> slot(SLOT_WORDS);
  for (size_t i = 0; i < n; i++) {
    prep_rec rec;
    schnorr_prep_one(msg32 + 32 * i, pk32 + 32 * i, sig64 + 64 * i, &rec);
    u32 qx[8], qy[8], rw[8];
    bool ok = parse_pubkey(pk32 + 32 * i, 32, qx, qy);
    ok &= (rec.flags & PREP_VALID) != 0;
    if (ok) {
      const gej R = ecmult_lane(rec, ge_from_words(qx, qy), slot.data(), g_table.data());
      be_to_words(rw, sig64 + 64 * i);
      ok = schnorr_final(R, rw);
    }
    out[i] = ok;
  }
}
}

// ---- gossip framing (verify_core.h "gossip framing"): bad flag, type, signed region offset, key offset
extern "C" int dm_gossip_frame(const u8 *m, size_t len, u32 *type, size_t *signed_off, size_t *keyoff) {
  const gossip_frame f = gossip_parse_frame(m, len);
  *type = f.type; *signed_off = f.signed_off; *keyoff = f.keyoff;
  return f.bad;
}

// ---- the device fuzzer's lane function with magnitude assertions on (lamd_fuzz_field runs the same function on the GPU)
#include "../lightning_amd/csrc/fuzz.h"
extern "C" uint64_t dm_fuzz_lane(uint64_t seed, uint64_t lane, int iters) { return fuzz_lane(seed, lane, iters); }

// ---- BIP143 sighash of a flat transaction template (verify_core.h "BIP143 signature hash on the device")
extern "C" int dm_bip143(u32 version, u32 locktime, const u8 *inputs, u32 n_in, const u8 *outputs, size_t outputs_len, u32 n_out, u32 in_idx,
                         const u8 *script, size_t script_len, uint64_t amount, u32 sighash_type, u8 *out32) {
  tx_view t;
  t.version = version; t.locktime = locktime; t.inputs = inputs; t.n_in = n_in; t.outputs = outputs; t.outputs_len = outputs_len; t.n_out = n_out;
  return bip143_sighash(t, in_idx, script, script_len, amount, sighash_type, out32);
}

// ---- BOLT #12 merkle root + signature hash of a TLV stream (bolt12.h)
extern "C" int dm_bolt12(const u8 *tlv, size_t len, const char *messagename, const char *fieldname, u8 *root32, u8 *sighash32) {
  bolt12_mids mids;
  const u8 leaf[6] = {'L', 'n', 'L', 'e', 'a', 'f'}, branch[8] = {'L', 'n', 'B', 'r', 'a', 'n', 'c', 'h'};
  bolt12_tag_midstate(leaf, 6, leaf, 0, mids.leaf);
  bolt12_tag_midstate(branch, 8, branch, 0, mids.branch);
  const std::string t2 = std::string(messagename) + fieldname;
  bolt12_tag_midstate((const u8 *)"lightning", 9, (const u8 *)t2.data(), t2.size(), mids.sig);
  if (!bolt12_merkle_root(tlv, len, mids, root32)) return 0;
  bolt12_sighash(mids, root32, sighash32);
  return 1;
}

// ---- 9x29 scalar arithmetic of the ECDSA preparation (scalar.h sc29_*): inputs are arbitrary 256-bit values, outputs canonical
extern "C" void dm_sc29_mul(const u8 *a, const u8 *b, u8 *out) {
  u32 x[8], y[8];
  be_to_words(x, a); be_to_words(y, b);
  const sc r = sc29_to_sc(sc29_mul(sc29_from_words(x), sc29_from_words(y)));
  words_to_be(out, r.w);
}
extern "C" void dm_sc29_inv(const u8 *a, u8 *out) {
  u32 x[8];
  be_to_words(x, a);
  const sc r = sc29_to_sc(sc29_inv(sc29_from_words(x)));
  words_to_be(out, r.w);
}
extern "C" void dm_sc29_chain(const u8 *a, const u8 *b, int steps, u8 *out) {  // lazy values fed back without canonicalising
  u32 x[8], y[8];
  be_to_words(x, a); be_to_words(y, b);
  sc29 p = sc29_from_words(x), q = sc29_from_words(y);
  for (int i = 0; i < steps; i++) { const sc29 t = sc29_mul(p, q); p = q; q = t; }
  const sc r = sc29_to_sc(q);
  words_to_be(out, r.w);
}
extern "C" void dm_sc_inv_var(const u8 *a, u8 *out) { sc x; be_to_words(x.w, a); sc r = sc_inv_var(x); words_to_be(out, r.w); }
extern "C" void dm_fe_inv_var(const u8 *a, u8 *out) {
  u32 w[8];
  be_to_words(w, a);
  const fe r = fe_inv_var(fe_from_words(w));
  u32 o[8];
  fe_to_words(o, fe_normalize(r));
  words_to_be(out, o);
}

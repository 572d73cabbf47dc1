// liblightning_amd_testgen.so: synthetic workloads for tests/ and bench.py -- NOT part of the product library.
// The signer kernels below hold private keys (seeded), sign on the device with the engine's own G table and write rows /
// wire messages in the formats the verification entry points take: the role devtools/mkgossip.c:131-147,235-322 plays for the
// reference's gossip benchmarks.  Declared in include/lightning_amd_testgen.h; reaches the engine only through its public
// and diagnostic ABI (lamd_stream, lamd_get_info, lamd_debug_gtable).
#include <hip/hip_runtime.h>

#include "../../include/lightning_amd.h"
#include "../../include/lightning_amd_debug.h"
#include "../../include/lightning_amd_testgen.h"
#include "verify_core.h"

using namespace lamd;

static inline unsigned blocks_for(size_t n) { return (unsigned)((n + 255) / 256); }
// the engine's stream (work is ordered with the verification calls that follow on the same context) and its G table
static bool bind(lamd_ctx *ctx, hipStream_t *st, const u32 **gtable) {
  lamd_info info;
  if (lamd_get_info(ctx, &info) != LAMD_OK || hipSetDevice(info.device) != hipSuccess) return false;
  *st = (hipStream_t)lamd_stream(ctx);
  *gtable = (const u32 *)lamd_debug_gtable(ctx);
  return *gtable != nullptr;
}

LAMD_HD void rand_words(u32 w[8], u64 seed, u64 idx, u64 stream) {
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const u64 v = splitmix64(seed ^ splitmix64(idx * 4 + j + (stream << 56)));
    w[2 * j] = (u32)v;
    w[2 * j + 1] = (u32)(v >> 32);
  }
}
LAMD_HD sc rand_scalar(u64 seed, u64 idx, u64 stream) {
  u32 w[8];
  rand_words(w, seed, idx, stream);
  sc s = sc_from_words(w, nullptr);
  if (sc_is_zero(s)) s.w[0] = 1;
  return s;
}
// k*G as canonical affine words
LAMD_HD void gmul_affine(u32 xw[8], u32 yw[8], const sc &k, const u32 *gtable) {
  gej acc = gej_infinity();
#pragma unroll 1
  for (int w = 0; w < GTABLE_WINDOWS; w++) {
    const u32 d = gtable_digit(k.w, w);
    const u32 *e = gtable + (((size_t)w << GTABLE_WINDOW_BITS) + d) * GT_ENTRY_WORDS;
    ge pt;
    pt.x = slot_load_fe(e);
    pt.y = slot_load_fe(e + TW);
    acc = gej_add_ge(acc, pt, d == 0);
  }
  const fe zi = fe_inv(fe_norm_weak(acc.z));
  const fe zi2 = fe_sqr(zi);
  fe_to_words(xw, fe_normalize(fe_mul(acc.x, zi2)));
  fe_to_words(yw, fe_normalize(fe_mul(acc.y, fe_mul(zi2, zi))));
}
LAMD_HD sc sc_add_mod(const sc &a, const sc &b) {
  u32 t[8];
  u64 c = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) { c += (u64)a.w[i] + b.w[i]; t[i] = (u32)c; c >>= 32; }
  u32 d[8];
  words_sub_n(d, t);
  const bool ge = (c != 0) | words_ge_n(t);
  sc r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.w[i] = ge ? d[i] : t[i];
  return r;
}

// key of row i: rows are cut into groups of `group` consecutive rows that share one key (group = 0: every row draws its
// key independently); the group's key index is a seeded hash modulo nkeys
LAMD_HD u64 gen_key_index(u64 seed, u64 i, u64 nkeys, u64 group) {
  const u64 g = group ? i / group : i;
  return splitmix64(seed ^ splitmix64(g + (7ULL << 56))) % nkeys;
}
__global__ void __launch_bounds__(256) k_gen_ecdsa(size_t n, u64 seed, u64 nkeys, u64 group, int publen, const u32 *__restrict__ gtable,
                                                   u8 *__restrict__ hash32, u8 *__restrict__ sig64, u8 *__restrict__ pub) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const u64 ki = gen_key_index(seed, i, nkeys, group);
  const sc d = rand_scalar(seed, ki, 1);
  const sc k = rand_scalar(seed, i, 2);
  u32 zw[8];
  rand_words(zw, seed, i, 3);
  u32 qx[8], qy[8], rx[8], ry[8];
  gmul_affine(qx, qy, d, gtable);
  gmul_affine(rx, ry, k, gtable);
  const sc r = sc_from_words(rx, nullptr);
  const sc z = sc_from_words(zw, nullptr);
  sc s = sc_mul(sc_inv_var(k), sc_add_mod(z, sc_mul(r, d)));
  if (sc_is_high(s)) s = sc_neg(s);
  store_words_be(hash32 + 32 * i, zw);
  store_words_be(sig64 + 64 * i, r.w);
  store_words_be(sig64 + 64 * i + 32, s.w);
  u8 *p = pub + (size_t)publen * i;
  if (publen == 65) {
    p[0] = 4;
    store_words_be(p + 1, qx);
    store_words_be(p + 33, qy);
  } else {
    p[0] = 2 + (qy[0] & 1);
    store_words_be(p + 1, qx);
  }
}

__global__ void __launch_bounds__(256) k_gen_schnorr(size_t n, u64 seed, u64 nkeys, u64 group, const u32 *__restrict__ gtable,
                                                     u8 *__restrict__ msg32, u8 *__restrict__ pk32, u8 *__restrict__ sig64) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const u64 ki = gen_key_index(seed, i, nkeys, group);
  sc d = rand_scalar(seed, ki, 1);
  sc k = rand_scalar(seed, i, 2);
  u32 mw[8];
  rand_words(mw, seed, i, 3);
  u32 px[8], py[8], rx[8], ry[8];
  gmul_affine(px, py, d, gtable);
  gmul_affine(rx, ry, k, gtable);
  if (py[0] & 1) d = sc_neg(d);
  if (ry[0] & 1) k = sc_neg(k);
  u32 rb[8], pb[8], mb[8], eh[8], ew[8];
#pragma unroll
  for (int j = 0; j < 8; j++) { rb[j] = rx[7 - j]; pb[j] = px[7 - j]; mb[j] = mw[7 - j]; }
  bip340_challenge(eh, rb, pb, mb);
#pragma unroll
  for (int j = 0; j < 8; j++) ew[j] = eh[7 - j];
  const sc e = sc_from_words(ew, nullptr);
  const sc s = sc_add_mod(k, sc_mul(e, d));
  store_words_be(msg32 + 32 * i, mw);
  store_words_be(pk32 + 32 * i, px);
  store_words_be(sig64 + 64 * i, rx);
  store_words_be(sig64 + 64 * i + 32, s.w);
}

// ---- synthetic gossip (shape of devtools/mkgossip.c:131-147,235-322): n_cann channel_announcements (432 B, no
// features) followed by n_cupd channel_updates (138 B) signed by one of the referenced channel's nodes
LAMD_HD void sign_ecdsa_words(u32 rw[8], u32 sw[8], const u32 zw[8], const sc &d, const sc &k, const u32 *gtable) {
  u32 rx[8], ry[8];
  gmul_affine(rx, ry, k, gtable);
  const sc r = sc_from_words(rx, nullptr);
  const sc z = sc_from_words(zw, nullptr);
  sc s = sc_mul(sc_inv_var(k), sc_add_mod(z, sc_mul(r, d)));
  if (sc_is_high(s)) s = sc_neg(s);
#pragma unroll
  for (int i = 0; i < 8; i++) { rw[i] = r.w[i]; sw[i] = s.w[i]; }
}
LAMD_HD void pubkey33(u8 out[33], const sc &d, const u32 *gtable) {
  u32 qx[8], qy[8];
  gmul_affine(qx, qy, d, gtable);
  out[0] = 2 + (qy[0] & 1);
  store_words_be(out + 1, qx);
}
LAMD_HD void gossip_chan_nodes(u64 seed, u64 c, u64 n_nodes, u64 *a, u64 *b) {
  *a = splitmix64(seed ^ splitmix64(c + (10ULL << 56))) % n_nodes;
  *b = splitmix64(seed ^ splitmix64(c + (11ULL << 56))) % n_nodes;
  if (*b == *a) *b = (*a + 1) % n_nodes;
}
constexpr size_t CANN_LEN = 432, CUPD_LEN = 138;
__global__ void __launch_bounds__(256) k_gen_gossip(size_t n_cann, size_t n_cupd, u64 seed, u64 n_nodes, const u32 *__restrict__ gtable,
                                                    u8 *__restrict__ msgs, u8 *__restrict__ ids) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_cann + n_cupd) return;
  const u8 chain[32] = {0x6f, 0xe2, 0x8c, 0x0a, 0xb6, 0xf1, 0xb3, 0x72, 0xc1, 0xa6, 0xa2, 0x46, 0xae, 0x63, 0xf7, 0x4f,
                        0x93, 0x1e, 0x83, 0x65, 0xe1, 0x5a, 0x08, 0x9c, 0x68, 0xd6, 0x19, 0x00, 0x00, 0x00, 0x00, 0x00};
  u8 h[32];
  u32 zw[8], rw[8], sw[8];
  if (i < n_cann) {
    u8 *m = msgs + i * CANN_LEN;
    u64 a, b;
    gossip_chan_nodes(seed, i, n_nodes, &a, &b);
    sc d[4] = {rand_scalar(seed, a, 1), rand_scalar(seed, b, 1), rand_scalar(seed, 2 * i, 4), rand_scalar(seed, 2 * i + 1, 4)};
    u8 k0[33], k1[33];
    pubkey33(k0, d[0], gtable);
    pubkey33(k1, d[1], gtable);
    bool swap = false;  // BOLT #7: node_id_1 is the lexicographically lesser
    for (int j = 0; j < 33; j++)
      if (k0[j] != k1[j]) { swap = k0[j] > k1[j]; break; }
    if (swap) { const sc t = d[0]; d[0] = d[1]; d[1] = t; }
    m[0] = 0x01; m[1] = 0x00;
    u8 *tail = m + 258;
    tail[0] = 0; tail[1] = 0;
    for (int j = 0; j < 32; j++) tail[2 + j] = chain[j];
    for (int j = 0; j < 8; j++) tail[34 + j] = (u8)((u64)i >> (8 * (7 - j)));
    for (int j = 0; j < 33; j++) { tail[42 + j] = swap ? k1[j] : k0[j]; tail[75 + j] = swap ? k0[j] : k1[j]; }
    pubkey33(tail + 108, d[2], gtable);
    pubkey33(tail + 141, d[3], gtable);
    sha256d_bytes(tail, CANN_LEN - 258, h);
    load_words_be(zw, h);
    for (int j = 0; j < 4; j++) {
      sign_ecdsa_words(rw, sw, zw, d[j], rand_scalar(seed, 4 * i + j, 5), gtable);
      store_words_be(m + 2 + 64 * j, rw);
      store_words_be(m + 2 + 64 * j + 32, sw);
    }
    for (int j = 0; j < 33; j++) ids[i * 33 + j] = 0;
  } else {
    const size_t u = i - n_cann;
    u8 *m = msgs + n_cann * CANN_LEN + u * CUPD_LEN;
    const u64 c = splitmix64(seed ^ splitmix64(u + (12ULL << 56))) % (n_cann ? n_cann : 1);
    u64 a, b;
    gossip_chan_nodes(seed, c, n_nodes, &a, &b);
    const u64 side = splitmix64(seed ^ splitmix64(u + (13ULL << 56))) & 1;
    const sc d = rand_scalar(seed, side ? b : a, 1);
    pubkey33(ids + i * 33, d, gtable);
    // the direction bit of channel_flags says whether the signer is node_id_1 or node_id_2 of the announcement, i.e. the lesser or
    // the greater of the two keys (BOLT #7; gossmap_manage.c:920-922 picks the verification key by it)
    u8 other[33];
    pubkey33(other, rand_scalar(seed, side ? a : b, 1), gtable);
    bool signer_greater = false;
    for (int j = 0; j < 33; j++)
      if (ids[i * 33 + j] != other[j]) { signer_greater = ids[i * 33 + j] > other[j]; break; }
    m[0] = 0x01; m[1] = 0x02;
    u8 *body = m + 66;
    for (int j = 0; j < 32; j++) body[j] = chain[j];
    for (int j = 0; j < 8; j++) body[32 + j] = (u8)(c >> (8 * (7 - j)));
    u32 rndw[8];
    rand_words(rndw, seed, u, 6);
    for (int j = 0; j < 32; j++) body[40 + j] = (u8)(rndw[j >> 2] >> (8 * (j & 3)));
    body[44] = 1;               // message_flags: option_channel_htlc_max
    body[45] = signer_greater ? 1 : 0;  // channel_flags: direction
    sha256d_bytes(body, CUPD_LEN - 66, h);
    load_words_be(zw, h);
    sign_ecdsa_words(rw, sw, zw, d, rand_scalar(seed, u, 7), gtable);
    store_words_be(m + 2, rw);
    store_words_be(m + 34, sw);
  }
}

// ---- synthetic workloads
extern "C" int lamd_gen_ecdsa_device(lamd_ctx *ctx, size_t n, uint64_t seed, size_t nkeys, size_t group, size_t publen, void *d_hash32,
                                     void *d_sig64, void *d_pub) {
  if (!ctx) return LAMD_ERR_ARG;
  hipStream_t st;
  const u32 *gtable;
  if (!bind(ctx, &st, &gtable)) return LAMD_ERR_HIP;
  if (!d_hash32 || !d_sig64 || !d_pub || (publen != 33 && publen != 65) || nkeys == 0) return LAMD_ERR_ARG;
  if (n == 0) return LAMD_OK;
  hipLaunchKernelGGL(k_gen_ecdsa, dim3(blocks_for(n)), dim3(256), 0, st, n, (u64)seed, (u64)nkeys, (u64)group,
                     (int)publen, gtable, (u8 *)d_hash32, (u8 *)d_sig64, (u8 *)d_pub);
  if (hipGetLastError() != hipSuccess) return LAMD_ERR_HIP;
  return LAMD_OK;
}
extern "C" int lamd_gen_schnorr_device(lamd_ctx *ctx, size_t n, uint64_t seed, size_t nkeys, size_t group, void *d_msg32,
                                       void *d_xonly32, void *d_sig64) {
  if (!ctx) return LAMD_ERR_ARG;
  hipStream_t st;
  const u32 *gtable;
  if (!bind(ctx, &st, &gtable)) return LAMD_ERR_HIP;
  if (!d_msg32 || !d_xonly32 || !d_sig64 || nkeys == 0) return LAMD_ERR_ARG;
  if (n == 0) return LAMD_OK;
  hipLaunchKernelGGL(k_gen_schnorr, dim3(blocks_for(n)), dim3(256), 0, st, n, (u64)seed, (u64)nkeys, (u64)group,
                     gtable, (u8 *)d_msg32, (u8 *)d_xonly32, (u8 *)d_sig64);
  if (hipGetLastError() != hipSuccess) return LAMD_ERR_HIP;
  return LAMD_OK;
}
extern "C" int lamd_gen_gossip_device(lamd_ctx *ctx, size_t n_cann, size_t n_cupd, uint64_t seed, size_t n_nodes, void *d_msgs,
                                      void *d_node_ids33) {
  if (!ctx) return LAMD_ERR_ARG;
  hipStream_t st;
  const u32 *gtable;
  if (!bind(ctx, &st, &gtable)) return LAMD_ERR_HIP;
  if (!d_msgs || !d_node_ids33 || n_nodes < 2 || (n_cupd && !n_cann)) return LAMD_ERR_ARG;
  if (n_cann + n_cupd == 0) return LAMD_OK;
  hipLaunchKernelGGL(k_gen_gossip, dim3(blocks_for(n_cann + n_cupd)), dim3(256), 0, st, n_cann, n_cupd, (u64)seed, (u64)n_nodes,
                     gtable, (u8 *)d_msgs, (u8 *)d_node_ids33);
  if (hipGetLastError() != hipSuccess) return LAMD_ERR_HIP;
  return LAMD_OK;
}

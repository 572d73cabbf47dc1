// Host mirror of the reference's prototypes over the C ABI (see cln_shim.h).  Host code here is
// framing only: SHA-256 of message tails / preimages, DER and compact parsing, error strings.
// Every elliptic-curve decision is made by the HIP kernels behind lamd_*.
#include "../../include/cln_shim.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/lightning_amd.h"

static lamd_ctx *g_ctx;
static bool g_ctx_owned;
static std::string g_err;

extern "C" bool lamd_shim_setup(void) {
  if (g_ctx) return true;
  const char *dev = getenv("LAMD_DEVICE");
  lamd_ctx *c = nullptr;
  const int rc = lamd_init(&c, dev ? atoi(dev) : 0);
  if (rc != LAMD_OK) {
    g_err = std::string("lamd_init failed (") + std::to_string(rc) + "): " + (c ? lamd_last_error(c) : "no device");
    if (c) lamd_shutdown(c);
    return false;
  }
  g_ctx = c;
  g_ctx_owned = true;
  return true;
}
extern "C" void lamd_shim_use_context(void *p) {
  if (g_ctx && g_ctx_owned) lamd_shutdown(g_ctx);
  g_ctx = (lamd_ctx *)p;
  g_ctx_owned = false;
}
extern "C" void lamd_shim_shutdown(void) {
  if (g_ctx && g_ctx_owned) lamd_shutdown(g_ctx);
  g_ctx = nullptr;
}

// ---- tal-style arrays (ccan/tal keeps the length with the allocation; so does this)
struct tal_hdr { size_t len; size_t magic; };
static const size_t TAL_MAGIC = 0x7A11ED0C0FFEE000ull;
extern "C" u8 *shim_tal_dup(const tal_t *, const u8 *src, size_t len) {
  tal_hdr *h = (tal_hdr *)malloc(sizeof(tal_hdr) + (len ? len : 1));
  if (!h) return nullptr;
  h->len = len;
  h->magic = TAL_MAGIC;
  if (len) memcpy(h + 1, src, len);
  return (u8 *)(h + 1);
}
// Named shim_tal_bytelen, NOT tal_bytelen: linked next to the real ccan/tal inside lightningd a second global `tal_bytelen` would
// either be a duplicate definition or interpose ccan's (which would then read this header layout on genuine tal arrays, or the
// other way round).  In-tree build: compile with -DLAMD_SHIM_WITH_CCAN_TAL and the lengths come from ccan's own tal_bytelen().
// A pointer that is not one of shim_tal_dup()'s fails closed: SHIM_TAL_FOREIGN, and the caller reports "does not verify".
#if defined(LAMD_SHIM_WITH_CCAN_TAL)
extern "C" size_t tal_bytelen(const void *ptr);  // ccan/tal/tal.h
extern "C" size_t shim_tal_bytelen(const void *ptr) { return ptr ? tal_bytelen(ptr) : 0; }
#else
extern "C" size_t shim_tal_bytelen(const void *ptr) {
  if (!ptr) return 0;
  const tal_hdr *h = (const tal_hdr *)ptr - 1;
  if (h->magic != TAL_MAGIC) return SHIM_TAL_FOREIGN;
  return h->len;
}
#endif
extern "C" void shim_tal_free(const void *ptr) {
  if (ptr) free((tal_hdr *)ptr - 1);
}
extern "C" const char *lamd_shim_last_error(void) { return g_err.c_str(); }

// ---- SHA-256 (FIPS 180-4), host framing only
static const uint32_t K256[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be,
    0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa,
    0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85,
    0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3,
    0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f,
    0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
static inline uint32_t ror(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
static void sha256_block(uint32_t st[8], const u8 *b) {
  uint32_t w[64];
  for (int i = 0; i < 16; i++) w[i] = ((uint32_t)b[4 * i] << 24) | ((uint32_t)b[4 * i + 1] << 16) | ((uint32_t)b[4 * i + 2] << 8) | b[4 * i + 3];
  for (int i = 16; i < 64; i++)
    w[i] = w[i - 16] + (ror(w[i - 15], 7) ^ ror(w[i - 15], 18) ^ (w[i - 15] >> 3)) + w[i - 7] + (ror(w[i - 2], 17) ^ ror(w[i - 2], 19) ^ (w[i - 2] >> 10));
  uint32_t a = st[0], bb = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
  for (int i = 0; i < 64; i++) {
    const uint32_t t1 = h + (ror(e, 6) ^ ror(e, 11) ^ ror(e, 25)) + ((e & f) ^ (~e & g)) + K256[i] + w[i];
    const uint32_t t2 = (ror(a, 2) ^ ror(a, 13) ^ ror(a, 22)) + ((a & bb) ^ (a & c) ^ (bb & c));
    h = g; g = f; f = e; e = d + t1; d = c; c = bb; bb = a; a = t1 + t2;
  }
  st[0] += a; st[1] += bb; st[2] += c; st[3] += d; st[4] += e; st[5] += f; st[6] += g; st[7] += h;
}
static void sha256_host(const u8 *p, size_t len, u8 out[32]) {
  uint32_t st[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
  size_t off = 0;
  for (; off + 64 <= len; off += 64) sha256_block(st, p + off);
  u8 tail[128] = {0};
  const size_t rem = len - off;
  memcpy(tail, p + off, rem);
  tail[rem] = 0x80;
  const size_t tl = rem + 9 <= 64 ? 64 : 128;
  const uint64_t bits = (uint64_t)len * 8;
  for (int i = 0; i < 8; i++) tail[tl - 1 - i] = (u8)(bits >> (8 * i));
  sha256_block(st, tail);
  if (tl == 128) sha256_block(st, tail + 64);
  for (int i = 0; i < 8; i++) { out[4 * i] = st[i] >> 24; out[4 * i + 1] = st[i] >> 16; out[4 * i + 2] = st[i] >> 8; out[4 * i + 3] = st[i]; }
}
extern "C" void sha256_double(struct sha256_double *shadouble, const void *p, size_t len) {
  u8 h[32];
  sha256_host((const u8 *)p, len, h);
  sha256_host(h, 32, shadouble->sha.u.u8);
}

// ---- keys
static bool parse_key(const u8 *ser, size_t len, secp256k1_pubkey *out) {
  if (!g_ctx && !lamd_shim_setup()) return false;
  u8 ok = 0;
  const int rc = lamd_pubkey_parse_batch(g_ctx, 1, ser, len, len, out->data, &ok);
  if (rc != LAMD_OK) { g_err = lamd_last_error(g_ctx); return false; }
  return ok != 0;
}
extern "C" bool pubkey_from_der(const u8 *der, size_t len, struct pubkey *key) {
  if (len != PUBKEY_CMPR_LEN) return false;  // bitcoin/pubkey.c:16
  return parse_key(der, len, &key->pubkey);
}
extern "C" void pubkey_to_der(u8 der[PUBKEY_CMPR_LEN], const struct pubkey *key) {
  der[0] = 2 + (key->pubkey.data[63] & 1);
  memcpy(der + 1, key->pubkey.data, 32);
}
extern "C" bool pubkey_from_node_id(struct pubkey *key, const struct node_id *id) { return parse_key(id->k, sizeof(id->k), &key->pubkey); }

// ---- signatures
static const u8 ORDER_N[32] = {0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFE,
                               0xBA, 0xAE, 0xDC, 0xE6, 0xAF, 0x48, 0xA0, 0x3B, 0xBF, 0xD2, 0x5E, 0x8C, 0xD0, 0x36, 0x41, 0x41};
static bool below_n(const u8 v[32]) { return memcmp(v, ORDER_N, 32) < 0; }

extern "C" bool fromwire_secp256k1_ecdsa_signature(const u8 compact[64], secp256k1_ecdsa_signature *sig) {
  if (!below_n(compact) || !below_n(compact + 32)) return false;  // secp256k1_ecdsa_signature_parse_compact, wire/fromwire.c:196
  memcpy(sig->data, compact, 64);
  return true;
}

// strict DER (what secp256k1_ecdsa_signature_parse_der accepts); out-of-range integers parse as 0
static bool der_len(size_t *out, const u8 **p, const u8 *end) {
  if (*p >= end) return false;
  const unsigned b1 = *((*p)++);
  if (b1 == 0xFF) return false;
  if (!(b1 & 0x80)) { *out = b1; return true; }
  if (b1 == 0x80) return false;
  size_t left = b1 & 0x7F;
  if (left > (size_t)(end - *p)) return false;
  if (**p == 0) return false;
  if (left > sizeof(size_t)) return false;
  size_t v = 0;
  while (left-- > 0) v = (v << 8) | *((*p)++);
  if (v > (size_t)(end - *p) || v < 128) return false;
  *out = v;
  return true;
}
static bool der_int(u8 out32[32], const u8 **p, const u8 *end) {
  size_t rlen;
  if (*p == end || **p != 0x02) return false;
  (*p)++;
  if (!der_len(&rlen, p, end)) return false;
  if (rlen == 0 || rlen > (size_t)(end - *p)) return false;
  if ((*p)[0] == 0x00 && rlen > 1 && !((*p)[1] & 0x80)) return false;
  if ((*p)[0] == 0xFF && rlen > 1 && ((*p)[1] & 0x80)) return false;
  bool overflow = ((*p)[0] & 0x80) != 0;
  const u8 *s = *p;
  size_t l = rlen;
  if (l > 0 && s[0] == 0) { l--; s++; }
  if (l > 32) overflow = true;
  memset(out32, 0, 32);
  if (!overflow) {
    memcpy(out32 + 32 - l, s, l);
    if (!below_n(out32)) overflow = true;
  }
  if (overflow) memset(out32, 0, 32);
  *p += rlen;
  return true;
}
static bool sighash_type_valid(int t) { return t == SIGHASH_ALL || t == (SIGHASH_SINGLE | SIGHASH_ANYONECANPAY); }
// the engine's sighash gate sees one byte: an int the reference's gate rejects (bitcoin/signature.c:206-211; 0x101 would truncate to SIGHASH_ALL)
// travels as 0, which no gate accepts -- the row is bad, in the reference's order
static u8 sighash_byte(int t) { return sighash_type_valid(t) ? (u8)t : 0; }

extern "C" bool signature_from_der(const u8 *der, size_t len, struct bitcoin_signature *sig) {
  if (len < 1) return false;
  const u8 *p = der, *end = der + len - 1;
  size_t rlen;
  if (p == end || *(p++) != 0x30) return false;
  if (!der_len(&rlen, &p, end) || rlen != (size_t)(end - p)) return false;
  if (!der_int(sig->s.data, &p, end) || !der_int(sig->s.data + 32, &p, end) || p != end) return false;
  sig->sighash_type = (enum sighash_type)der[len - 1];
  return sighash_type_valid(der[len - 1]);
}

// ---- checks
extern "C" bool check_signed_hash(const struct sha256_double *hash, const secp256k1_ecdsa_signature *signature, const struct pubkey *key) {
  if (!g_ctx && !lamd_shim_setup()) return false;
  u8 pub65[65];
  pub65[0] = 4;
  memcpy(pub65 + 1, key->pubkey.data, 64);
  const int rc = lamd_check_signed_hash(g_ctx, hash->sha.u.u8, signature->data, pub65, 65);
  if (rc < 0) g_err = lamd_last_error(g_ctx);
  return rc == 1;
}
extern "C" bool check_signed_hash_nodeid(const struct sha256_double *hash, const secp256k1_ecdsa_signature *signature, const struct node_id *id) {
  if (!g_ctx && !lamd_shim_setup()) return false;
  // common/node_id.c:72-80: parse the 33-byte id, then check_signed_hash -- both happen on the device in one call
  const int rc = lamd_check_signed_hash_nodeid(g_ctx, hash->sha.u.u8, signature->data, id->k);
  if (rc < 0) g_err = lamd_last_error(g_ctx);
  return rc == 1;
}
extern "C" bool check_schnorr_sig(const struct sha256 *hash, const secp256k1_pubkey *pubkey, const struct bip340sig *sig) {
  if (!g_ctx && !lamd_shim_setup()) return false;
  u8 raw[PUBKEY_CMPR_LEN];
  raw[0] = 2 + (pubkey->data[63] & 1);  // bitcoin/signature.c:417: serialise compressed ...
  memcpy(raw + 1, pubkey->data, 32);
  const int rc = lamd_check_schnorr_sig(g_ctx, hash->u.u8, raw, sig->u8);  // ... :422 drops the parity byte
  if (rc < 0) g_err = lamd_last_error(g_ctx);
  return rc == 1;
}
extern "C" bool check_tx_sig_preimage(const u8 *bip143_preimage, size_t preimage_len, const u8 *witness_script, const struct pubkey *key,
                                      const struct bitcoin_signature *sig) {
  // bitcoin/signature.c:206-211: only SIGHASH_ALL, or SINGLE|ANYONECANPAY with a witness script
  if (sig->sighash_type != SIGHASH_ALL) {
    if (!witness_script) return false;
    if (sig->sighash_type != (SIGHASH_SINGLE | SIGHASH_ANYONECANPAY)) return false;
  }
  struct sha256_double hash;
  sha256_double(&hash, bip143_preimage, preimage_len);
  return check_signed_hash(&hash, &sig->s, key);
}
static void put_compact_size(std::string &o, uint64_t v) {
  if (v < 0xfd) o.push_back((char)v);
  else if (v <= 0xffff) { o.push_back((char)0xfd); o.push_back((char)v); o.push_back((char)(v >> 8)); }
  else { o.push_back((char)0xfe); for (int i = 0; i < 4; i++) o.push_back((char)(v >> (8 * i))); }
}
// the two byte strings the device hashes of a transaction: inputs (txid | vout | nSequence, 40 bytes each) and outputs in wire form
static bool flatten_tx(const struct bitcoin_tx *tx, std::string &in, std::string &out) {
  for (size_t i = 0; i < tx->num_inputs; i++) {
    in.append((const char *)tx->inputs[i].txid, 32);
    for (int b = 0; b < 4; b++) in.push_back((char)(tx->inputs[i].index >> (8 * b)));
    for (int b = 0; b < 4; b++) in.push_back((char)(tx->inputs[i].sequence >> (8 * b)));
  }
  for (size_t i = 0; i < tx->num_outputs; i++) {
    for (int b = 0; b < 8; b++) out.push_back((char)(tx->outputs[i].amount_sat >> (8 * b)));
    const size_t sl = shim_tal_bytelen(tx->outputs[i].script);
    if (sl == SHIM_TAL_FOREIGN) { g_err = "check_tx_sig: output script is not a tal array"; return false; }
    put_compact_size(out, sl);
    out.append((const char *)tx->outputs[i].script, sl);
  }
  return true;
}
extern "C" bool check_tx_sig(const struct bitcoin_tx *tx, size_t input_num, const u8 *redeemscript, const u8 *witness_script,
                             const struct pubkey *key, const struct bitcoin_signature *sig) {
  const bool use_segwit = witness_script != nullptr;                   // bitcoin/signature.c:198-199
  const u8 *script = use_segwit ? witness_script : redeemscript;
  // :206-211 -- rejected before anything is hashed (the device applies the same gate again)
  if (sig->sighash_type != SIGHASH_ALL) {
    if (!witness_script) return false;
    if (sig->sighash_type != (SIGHASH_SINGLE | SIGHASH_ANYONECANPAY)) return false;
  }
  if (input_num >= tx->num_inputs) abort();                            // assert(input_num < tx->wtx->num_inputs), :213
  if (!g_ctx && !lamd_shim_setup()) return false;
  // flatten the template: that is all the host does -- hashPrevouts/Sequence/Outputs, the preimage and SHA256d run on the device
  std::string in, out;
  if (!flatten_tx(tx, in, out)) return false;
  const uint32_t version = tx->version, locktime = tx->locktime, inum = (uint32_t)input_num, nout = (uint32_t)tx->num_outputs;
  const size_t script_len = shim_tal_bytelen(script);
  if (script_len == SHIM_TAL_FOREIGN) { g_err = "check_tx_sig: script is not a tal array"; return false; }
  const uint64_t in_off[2] = {0, tx->num_inputs}, out_off[2] = {0, out.size()}, sc_off[2] = {0, script_len};
  const uint64_t amount = tx->inputs[input_num].amount_sat;
  const u8 type = (u8)sig->sighash_type, wit = use_segwit ? 1 : 0;
  u8 pub65[65], ok = 0;
  pub65[0] = 4;
  memcpy(pub65 + 1, key->pubkey.data, 64);
  const u8 dummy = 0;
  const int rc = lamd_check_tx_sig_tx_batch(g_ctx, 1, &version, &locktime, (const u8 *)in.data(), in_off, &inum, &amount,
                                            out.empty() ? &dummy : (const u8 *)out.data(), out_off, &nout, script ? script : &dummy, sc_off, &type, &wit,
                                            sig->s.data, pub65, 65, 65, &ok);
  if (rc != LAMD_OK) { g_err = lamd_last_error(g_ctx); return false; }
  return ok != 0;
}

// ---- channeld/channeld.c:2171-2232 as ONE call (see cln_shim.h check_commit_sigs)
static std::string hex(const u8 *p, size_t n);
static std::string der_hex(const secp256k1_ecdsa_signature *sig);
// fmt_bitcoin_signature (bitcoin/signature.c:336-343): the DER signature with the sighash byte appended, in hex
static std::string fmt_bitcoin_signature_(const struct bitcoin_signature *sig) {
  const u8 t = (u8)sig->sighash_type;
  return der_hex(&sig->s) + hex(&t, 1);
}
// fmt_bitcoin_tx (bitcoin/tx.c:712-718): hex of linearize_tx() -- a transaction channel_txs() built has empty scriptSigs and no witness stack
// (the signature just received sits in the PSBT, :2147-2151), so this is the plain serialisation
static std::string fmt_bitcoin_tx_(const struct bitcoin_tx *tx) {
  std::string o;
  for (int b = 0; b < 4; b++) o.push_back((char)(tx->version >> (8 * b)));
  put_compact_size(o, tx->num_inputs);
  for (size_t i = 0; i < tx->num_inputs; i++) {
    o.append((const char *)tx->inputs[i].txid, 32);
    for (int b = 0; b < 4; b++) o.push_back((char)(tx->inputs[i].index >> (8 * b)));
    o.push_back(0);
    for (int b = 0; b < 4; b++) o.push_back((char)(tx->inputs[i].sequence >> (8 * b)));
  }
  put_compact_size(o, tx->num_outputs);
  for (size_t i = 0; i < tx->num_outputs; i++) {
    for (int b = 0; b < 8; b++) o.push_back((char)(tx->outputs[i].amount_sat >> (8 * b)));
    const size_t sl = shim_tal_bytelen(tx->outputs[i].script);
    put_compact_size(o, sl == SHIM_TAL_FOREIGN ? 0 : sl);
    if (sl != SHIM_TAL_FOREIGN) o.append((const char *)tx->outputs[i].script, sl);
  }
  for (int b = 0; b < 4; b++) o.push_back((char)(tx->locktime >> (8 * b)));
  return hex((const u8 *)o.data(), o.size());
}
static std::string fmt_pubkey_(const struct pubkey *key) {
  u8 der[PUBKEY_CMPR_LEN];
  pubkey_to_der(der, key);
  return hex(der, sizeof der);
}
static const char *dup(const tal_t *ctx, const std::string &s);
extern "C" const char *check_commit_sigs(const tal_t *ctx, uint64_t local_index, const struct bitcoin_tx *const *txs, const u8 *funding_wscript,
                                         const struct pubkey *remote_funding, const struct bitcoin_signature *commit_sig, const u8 *const *htlc_wscripts,
                                         const struct pubkey *remote_htlckey, const struct bitcoin_signature *htlc_sigs, uint32_t feerate,
                                         const char *commit_warning_tail) {
  const size_t txs_bytes = shim_tal_bytelen(txs), sigs_bytes = shim_tal_bytelen(htlc_sigs), fw_len = shim_tal_bytelen(funding_wscript);
  if (txs_bytes == SHIM_TAL_FOREIGN || sigs_bytes == SHIM_TAL_FOREIGN || fw_len == SHIM_TAL_FOREIGN || txs_bytes < sizeof(void *))
    return dup(ctx, "engine error: check_commit_sigs: txs / htlc_sigs / funding_wscript is not a tal array");
  const size_t n_txs = txs_bytes / sizeof(void *), n_sigs = sigs_bytes / sizeof(struct bitcoin_signature);
  if (!g_ctx && !lamd_shim_setup()) return dup(ctx, "engine error: " + g_err);
  // every signature the reference's loop could reach goes into ONE device call: the commitment signature and, when the count is the expected
  // one, the HTLC signatures (with another count the reference warns before it looks at any of them, :2203-2206)
  const size_t n_htlc = n_sigs == n_txs - 1 ? n_sigs : 0;
  std::vector<std::string> ins(1 + n_htlc), outs(1 + n_htlc);
  std::vector<lamd_tx_template> tm(1 + n_htlc);
  std::vector<u8> sigs64(64 * (n_htlc ? n_htlc : 1)), types(n_htlc ? n_htlc : 1);
  for (size_t i = 0; i < 1 + n_htlc; i++) {
    const struct bitcoin_tx *tx = txs[i];
    if (tx->num_inputs < 1) abort();  // assert(input_num < tx->wtx->num_inputs), bitcoin/signature.c:213
    if (!flatten_tx(tx, ins[i], outs[i])) return dup(ctx, "engine error: " + g_err);
    const u8 *ws = i ? htlc_wscripts[i - 1] : funding_wscript;
    const size_t wl = shim_tal_bytelen(ws);
    if (wl == SHIM_TAL_FOREIGN) return dup(ctx, "engine error: check_commit_sigs: a witness script is not a tal array");
    lamd_tx_template &t = tm[i];
    t.version = tx->version; t.locktime = tx->locktime;
    t.inputs40 = (const u8 *)ins[i].data(); t.n_inputs = (uint32_t)tx->num_inputs; t.input_num = 0; t.amount_sat = tx->inputs[0].amount_sat;
    t.outputs = (const u8 *)outs[i].data(); t.outputs_len = outs[i].size(); t.n_outputs = (uint32_t)tx->num_outputs;
    t.script = ws; t.script_len = wl;
    if (i) {
      memcpy(&sigs64[64 * (i - 1)], htlc_sigs[i - 1].s.data, 64);
      types[i - 1] = sighash_byte(htlc_sigs[i - 1].sighash_type);
    }
  }
  u8 fund33[PUBKEY_CMPR_LEN], htlc33[PUBKEY_CMPR_LEN];
  pubkey_to_der(fund33, remote_funding);
  pubkey_to_der(htlc33, remote_htlckey);
  int64_t first_bad = 0;
  const int rc = lamd_check_commitment_signed(g_ctx, &tm[0], fund33, commit_sig->s.data, sighash_byte(commit_sig->sighash_type), n_htlc, n_htlc ? &tm[1] : nullptr, htlc33,
                                              sigs64.data(), types.data(), &first_bad, nullptr);
  if (rc != LAMD_OK) { g_err = lamd_last_error(g_ctx); return dup(ctx, "engine error: " + g_err); }
  if (first_bad == 0)  // :2171-2193
    return dup(ctx, "Bad commit_sig signature " + std::to_string(local_index) + " " + fmt_bitcoin_signature_(commit_sig) + " for tx " + fmt_bitcoin_tx_(txs[0]) +
                        " wscript " + hex(funding_wscript, fw_len) + " key " + fmt_pubkey_(remote_funding) + " feerate " + std::to_string(feerate) +
                        (commit_warning_tail ? commit_warning_tail : ""));
  if (n_sigs != n_txs - 1)  // :2203-2206
    return dup(ctx, "Expected " + std::to_string(n_txs - 1) + " htlc sigs, not " + std::to_string(n_sigs));
  if (first_bad > 0) {  // :2224-2231
    const size_t i = (size_t)first_bad - 1;
    return dup(ctx, "Bad commit_sig signature " + fmt_bitcoin_signature_(&htlc_sigs[i]) + " for htlc " + fmt_bitcoin_tx_(txs[1 + i]) + " wscript " +
                        hex(htlc_wscripts[i], shim_tal_bytelen(htlc_wscripts[i])) + " key " + fmt_pubkey_(remote_htlckey));
  }
  return nullptr;
}

// ---- BOLT #12: the fields are re-serialised (towire of type, length, value: bolt12_merkle.c:33-40) and hashed on the device
static void put_bigsize(std::string &o, uint64_t v) {
  if (v < 0xfd) o.push_back((char)v);
  else if (v <= 0xffff) { o.push_back((char)0xfd); for (int i = 1; i >= 0; i--) o.push_back((char)(v >> (8 * i))); }
  else if (v <= 0xffffffffull) { o.push_back((char)0xfe); for (int i = 3; i >= 0; i--) o.push_back((char)(v >> (8 * i))); }
  else { o.push_back((char)0xff); for (int i = 7; i >= 0; i--) o.push_back((char)(v >> (8 * i))); }
}
static bool serialise_fields(const struct tlv_field *fields, std::string &o) {
  const size_t bytes = shim_tal_bytelen(fields);
  if (bytes == SHIM_TAL_FOREIGN) { g_err = "bolt12: fields is not a tal array"; return false; }
  const size_t n = bytes / sizeof(struct tlv_field);
  for (size_t i = 0; i < n; i++) {
    put_bigsize(o, fields[i].numtype);
    put_bigsize(o, fields[i].length);
    o.append((const char *)fields[i].value, fields[i].length);
  }
  return true;
}
static bool bolt12_hashes(const struct tlv_field *fields, const char *messagename, const char *fieldname, u8 *merkle32, u8 *sighash32) {
  if (!g_ctx && !lamd_shim_setup()) return false;
  std::string st;
  if (!serialise_fields(fields, st)) return false;
  const uint64_t off[2] = {0, st.size()};
  u8 ok = 0, dummy = 0;
  const int rc = lamd_bolt12_merkle_batch(g_ctx, 1, st.empty() ? &dummy : (const u8 *)st.data(), off, messagename, fieldname, merkle32, sighash32, &ok);
  if (rc != LAMD_OK) { g_err = lamd_last_error(g_ctx); return false; }
  return ok != 0;
}
extern "C" void merkle_tlv(const struct tlv_field *fields, struct sha256 *merkle) {
  if (!bolt12_hashes(fields, "", "", merkle->u.u8, nullptr)) memset(merkle->u.u8, 0, 32);  // "a distinctive all-zeroes" (bolt12_merkle.c:297-299)
}
extern "C" void sighash_from_merkle(const char *messagename, const char *fieldname, const struct sha256 *merkle, struct sha256 *sighash) {
  const std::string tag = std::string("lightning") + messagename + fieldname;  // bip340_sighash_init, bitcoin/signature.c:389-405
  u8 buf[96];
  sha256_host((const u8 *)tag.data(), tag.size(), buf);
  memcpy(buf + 32, buf, 32);
  memcpy(buf + 64, merkle->u.u8, 32);
  sha256_host(buf, sizeof buf, sighash->u.u8);
}
extern "C" bool bolt12_check_signature(const struct tlv_field *fields, const char *messagename, const char *fieldname, const struct pubkey *key,
                                       const struct bip340sig *sig) {
  if (!g_ctx && !lamd_shim_setup()) return false;
  std::string st;
  if (!serialise_fields(fields, st)) return false;
  const uint64_t off[2] = {0, st.size()};
  u8 der[PUBKEY_CMPR_LEN], ok = 0, dummy = 0;
  pubkey_to_der(der, key);
  const int rc = lamd_bolt12_check_signature_batch(g_ctx, 1, st.empty() ? &dummy : (const u8 *)st.data(), off, messagename, fieldname, der, 33, sig->u8, &ok);
  if (rc != LAMD_OK) { g_err = lamd_last_error(g_ctx); return false; }
  return ok != 0;
}

extern "C" int lamd_secp256k1_ecdsa_recoverable_signature_parse_compact(const void *, secp256k1_ecdsa_recoverable_signature *sig,
                                                                    const unsigned char *input64, int recid) {
  if (recid < 0 || recid > 3 || !below_n(input64) || !below_n(input64 + 32)) {
    memset(sig->data, 0, sizeof(sig->data));
    return 0;
  }
  memcpy(sig->data, input64, 64);
  sig->data[64] = (unsigned char)recid;
  return 1;
}
extern "C" int lamd_secp256k1_ecdsa_recover(const void *, secp256k1_pubkey *pubkey, const secp256k1_ecdsa_recoverable_signature *sig,
                                       const unsigned char *msghash32) {
  memset(pubkey->data, 0, sizeof(pubkey->data));
  if (!g_ctx && !lamd_shim_setup()) return 0;
  u8 key33[33], ok = 0;
  const int rc = lamd_ecdsa_recover_batch(g_ctx, 1, msghash32, sig->data, sig->data + 64, key33, &ok);
  if (rc < 0) g_err = lamd_last_error(g_ctx);
  if (rc != LAMD_OK || !ok) return 0;
  return parse_key(key33, 33, pubkey) ? 1 : 0;
}
// common/bolt11.c:1026-1027: drops the recovery id (upstream returns 1 unconditionally)
extern "C" int lamd_secp256k1_ecdsa_recoverable_signature_convert(const void *, secp256k1_ecdsa_signature *sig,
                                                              const secp256k1_ecdsa_recoverable_signature *sigin) {
  memcpy(sig->data, sigin->data, 64);
  return 1;
}
// common/bolt11.c:1055, lightningd/dual_open_control.c:2254, bitcoin/signature.c:188: the library call itself
extern "C" int lamd_secp256k1_ecdsa_verify(const void *, const secp256k1_ecdsa_signature *sig, const unsigned char *msghash32,
                                      const secp256k1_pubkey *pubkey) {
  struct sha256_double h;
  memcpy(h.sha.u.u8, msghash32, 32);
  struct pubkey k;
  k.pubkey = *pubkey;
  return check_signed_hash(&h, sig, &k) ? 1 : 0;
}
extern "C" void node_id_from_pubkey(struct node_id *id, const struct pubkey *key) { pubkey_to_der(id->k, key); }

extern "C" bool grind_htlc_tx_fee(uint64_t *fee_sat, const u8 *bip143_preimage, size_t preimage_len, const u8 *outputs, size_t outputs_len,
                                  uint64_t input_sat, const struct bitcoin_signature *remotesig, const u8 *wscript, uint64_t weight,
                                  uint32_t min_possible_feerate, uint32_t max_possible_feerate, const struct pubkey *other_htlc_key) {
  if (!g_ctx && !lamd_shim_setup()) return false;
  u8 der[PUBKEY_CMPR_LEN];
  pubkey_to_der(der, other_htlc_key);
  uint32_t rate = 0;
  uint64_t fee = 0;
  const int rc = lamd_grind_htlc_tx_fee(g_ctx, bip143_preimage, preimage_len, outputs, outputs_len, input_sat, weight, min_possible_feerate,
                                        max_possible_feerate, remotesig->s.data, (uint8_t)remotesig->sighash_type, wscript != nullptr, der, &rate, &fee);
  if (rc < 0) g_err = lamd_last_error(g_ctx);
  if (rc != 1) return false;
  *fee_sat = fee;
  return true;
}

// ---- gossip veneer
static std::string hex(const u8 *p, size_t n) {
  static const char *d = "0123456789abcdef";
  std::string s;
  s.reserve(2 * n);
  for (size_t i = 0; i < n; i++) { s.push_back(d[p[i] >> 4]); s.push_back(d[p[i] & 15]); }
  return s;
}
// fmt_secp256k1_ecdsa_signature prints the DER serialisation (bitcoin/signature.c:325-335)
static std::string der_hex(const secp256k1_ecdsa_signature *sig) {
  u8 out[72];
  size_t n = 0;
  auto put_int = [&](const u8 *v) {
    size_t skip = 0;
    while (skip < 31 && v[skip] == 0) skip++;
    const bool pad = v[skip] & 0x80;
    out[n++] = 0x02;
    out[n++] = (u8)(32 - skip + (pad ? 1 : 0));
    if (pad) out[n++] = 0;
    memcpy(out + n, v + skip, 32 - skip);
    n += 32 - skip;
  };
  n = 2;
  put_int(sig->data);
  put_int(sig->data + 32);
  out[0] = 0x30;
  out[1] = (u8)(n - 2);
  return hex(out, n);
}
// The error string is a tal string (the reference: tal_fmt(ctx, ...), gossipd/sigcheck.c:36-41).  In-tree (-DLAMD_SHIM_WITH_CCAN_TAL) it is
// tal_strdup()ed onto the caller's ctx; stand-alone it is a shim_tal_dup() array the caller owns (shim_tal_free()).
#if defined(LAMD_SHIM_WITH_CCAN_TAL)
extern "C" char *tal_strdup_(const tal_t *ctx, const char *p, const char *label);  // ccan/tal/str/str.h:20
static const char *dup(const tal_t *ctx, const std::string &s) { return tal_strdup_(ctx, s.c_str(), "char[]"); }
#else
static const char *dup(const tal_t *ctx, const std::string &s) { return (const char *)shim_tal_dup(ctx, (const u8 *)s.c_str(), s.size() + 1); }
#endif
// gossipd/sigcheck.c:9-164 decides on its ARGUMENTS: the hash of the message tail, then check_signed_hash_nodeid(hash, sig_i, id_i) /
// check_signed_hash(hash, sig_i, key_i) in order with an early return.  It never parses the message and never calls it malformed.  So does
// this: SHA256d of the tail on the host, the rows (hash, PASSED signature, PASSED key) through ONE launch of the latency path, the first
// row that fails names the string.  Keys travel in their 33-byte form (a struct pubkey is serialised as pubkey_to_der() does; the device
// decompresses it -- for every struct pubkey a parser produced that is the same point), so a channel_announcement is one 4-row call.
// -2: engine error (fails closed: a non-NULL string), -1: every row verifies, else the index of the first row that does not.
static int first_bad_row(const struct sha256_double *hash, int n, const secp256k1_ecdsa_signature *const *sigs, const u8 (*keys33)[PUBKEY_CMPR_LEN]) {
  if (!g_ctx && !lamd_shim_setup()) return -2;
  u8 hs[4 * 32], sg[4 * 64], ok[4] = {0, 0, 0, 0};
  for (int i = 0; i < n; i++) {
    memcpy(hs + 32 * i, hash->sha.u.u8, 32);
    memcpy(sg + 64 * i, sigs[i]->data, 64);
  }
  const int rc = lamd_verify_ecdsa_batch(g_ctx, (size_t)n, hs, sg, &keys33[0][0], PUBKEY_CMPR_LEN, PUBKEY_CMPR_LEN, ok);
  if (rc != LAMD_OK) { g_err = lamd_last_error(g_ctx); return -2; }
  for (int i = 0; i < n; i++)
    if (!ok[i]) return i;
  return -1;
}
static std::string bad(const char *what, const secp256k1_ecdsa_signature *sig, const struct sha256_double *h, const u8 *msg, size_t len, const char *kind) {
  return std::string(what) + " " + der_hex(sig) + " hash " + hex(h->sha.u.u8, 32) + " on " + kind + " " + hex(msg, len);
}
// the reference hashes tal_count(msg) - offset bytes and would run off the end of a message shorter than its offset (its callers have parsed the
// message before, so that cannot happen there); here such a tail is empty
static void tail_hash(struct sha256_double *h, const u8 *msg, size_t len, size_t offset) {
  sha256_double(h, msg + (offset < len ? offset : len), offset < len ? len - offset : 0);
}
extern "C" const char *sigcheck_channel_update_len(const tal_t *ctx, const struct node_id *node_id, const secp256k1_ecdsa_signature *node_sig,
                                                   const u8 *update, size_t len) {
  struct sha256_double hash;
  tail_hash(&hash, update, len, 66);  // 2 byte msg type + 64 byte signature (sigcheck.c:29-33)
  const secp256k1_ecdsa_signature *sigs[1] = {node_sig};
  u8 keys[1][PUBKEY_CMPR_LEN];
  memcpy(keys[0], node_id->k, PUBKEY_CMPR_LEN);
  const int v = first_bad_row(&hash, 1, sigs, keys);
  if (v == -1) return nullptr;
  if (v == -2) return dup(ctx, "engine error: " + g_err);
  return dup(ctx, bad("Bad signature for", node_sig, &hash, update, len, "channel_update"));
}
extern "C" const char *sigcheck_channel_announcement_len(const tal_t *ctx, const struct node_id *node1_id, const struct node_id *node2_id,
                                                         const struct pubkey *bitcoin1_key, const struct pubkey *bitcoin2_key,
                                                         const secp256k1_ecdsa_signature *node1_sig, const secp256k1_ecdsa_signature *node2_sig,
                                                         const secp256k1_ecdsa_signature *bitcoin1_sig, const secp256k1_ecdsa_signature *bitcoin2_sig,
                                                         const u8 *announcement, size_t len) {
  struct sha256_double hash;
  tail_hash(&hash, announcement, len, 258);  // 2 byte msg type + 256 byte signatures (sigcheck.c:71-76)
  const secp256k1_ecdsa_signature *sigs[4] = {node1_sig, node2_sig, bitcoin1_sig, bitcoin2_sig};
  u8 keys[4][PUBKEY_CMPR_LEN];
  memcpy(keys[0], node1_id->k, PUBKEY_CMPR_LEN);     // :78  check_signed_hash_nodeid(&hash, node1_sig, node1_id)
  memcpy(keys[1], node2_id->k, PUBKEY_CMPR_LEN);     // :87
  pubkey_to_der(keys[2], bitcoin1_key);              // :96  check_signed_hash(&hash, bitcoin1_sig, bitcoin1_key)
  pubkey_to_der(keys[3], bitcoin2_key);              // :105
  const int v = first_bad_row(&hash, 4, sigs, keys);
  if (v == -1) return nullptr;
  if (v == -2) return dup(ctx, "engine error: " + g_err);
  static const char *names[4] = {"Bad node_signature_1", "Bad node_signature_2", "Bad bitcoin_signature_1", "Bad bitcoin_signature_2"};
  return dup(ctx, bad(names[v], sigs[v], &hash, announcement, len, "channel_announcement"));
}
extern "C" const char *sigcheck_node_announcement_len(const tal_t *ctx, const struct node_id *node_id, const secp256k1_ecdsa_signature *node_sig,
                                                      const u8 *node_announcement, size_t len) {
  struct sha256_double hash;
  tail_hash(&hash, node_announcement, len, 66);  // sigcheck.c:136-141
  const secp256k1_ecdsa_signature *sigs[1] = {node_sig};
  u8 keys[1][PUBKEY_CMPR_LEN];
  memcpy(keys[0], node_id->k, PUBKEY_CMPR_LEN);  // "If node_id is invalid, it fails here" (:142): the device's key parse decides
  const int v = first_bad_row(&hash, 1, sigs, keys);
  if (v == -1) return nullptr;
  if (v == -2) return dup(ctx, "engine error: " + g_err);
  return dup(ctx, bad("Bad signature for", node_sig, &hash, node_announcement, len, "node_announcement"));
}
// ---- the reference's own prototypes (gossipd/sigcheck.h:7-28): the message is a tal array, its length travels with the pointer
// (tal_count(msg), gossipd/sigcheck.c:30-33,73-76,138-141).  A pointer whose length cannot be read fails closed.
static const char *not_tal(const tal_t *ctx, const char *kind) { return dup(ctx, std::string("engine error: ") + kind + " is not a tal array"); }
extern "C" const char *sigcheck_channel_update(const tal_t *ctx, const struct node_id *node_id, const secp256k1_ecdsa_signature *node_sig,
                                               const u8 *update) {
  const size_t len = shim_tal_bytelen(update);
  if (len == SHIM_TAL_FOREIGN) return not_tal(ctx, "channel_update");
  return sigcheck_channel_update_len(ctx, node_id, node_sig, update, len);
}
extern "C" const char *sigcheck_channel_announcement(const tal_t *ctx, const struct node_id *node1_id, const struct node_id *node2_id,
                                                     const struct pubkey *bitcoin1_key, const struct pubkey *bitcoin2_key,
                                                     const secp256k1_ecdsa_signature *node1_sig, const secp256k1_ecdsa_signature *node2_sig,
                                                     const secp256k1_ecdsa_signature *bitcoin1_sig, const secp256k1_ecdsa_signature *bitcoin2_sig,
                                                     const u8 *announcement) {
  const size_t len = shim_tal_bytelen(announcement);
  if (len == SHIM_TAL_FOREIGN) return not_tal(ctx, "channel_announcement");
  return sigcheck_channel_announcement_len(ctx, node1_id, node2_id, bitcoin1_key, bitcoin2_key, node1_sig, node2_sig, bitcoin1_sig, bitcoin2_sig,
                                           announcement, len);
}
extern "C" const char *sigcheck_node_announcement(const tal_t *ctx, const struct node_id *node_id, const secp256k1_ecdsa_signature *node_sig,
                                                  const u8 *node_announcement) {
  const size_t len = shim_tal_bytelen(node_announcement);
  if (len == SHIM_TAL_FOREIGN) return not_tal(ctx, "node_announcement");
  return sigcheck_node_announcement_len(ctx, node_id, node_sig, node_announcement, len);
}

// BOLT #12 signature front end for one TLV stream per lane: merkle_tlv() + sighash_from_merkle() of the reference
// (common/bolt12_merkle.c:48-318, tag construction bitcoin/signature.c:389-405), producing the 32-byte message that
// check_schnorr_sig() verifies (common/bolt12.c:80-92).  Included by verify_core.h (uses its streaming SHA-256).
//
//   H(tag, msg)  = SHA256(SHA256(tag) || SHA256(tag) || msg)                       -- the tag block is a midstate here
//   leaf_i       = H("LnBranch", ordered(H("LnLeaf", tlv_i), H("LnNonce" || first-tlv, type_i)))
//   tree         = the reference's oversized power-of-two tree whose absent right subtrees pass the left one through (:186-225,
//                  :293-296).  Because every pair is ordered before hashing, that equals merging complete subtrees as a binary
//                  counter does (a stack of (level, hash); equal levels merge) and folding what is left from the top
//   message      = H("lightning" || messagename || fieldname, root)
// Signature fields (types 240..1000) are no leaves (:20-23).  The stream must obey fromwire_tlv's generic rules (BigSize type
// and length minimally encoded, lengths inside the stream, types strictly increasing: wire/tlvstream.c:144-300) -- the
// reference hashes the re-serialised fields (:33-40), which for such a stream are the bytes themselves.
#pragma once

namespace lamd {

LAMD_HD void shs_init_mid(sha_stream *s, const u32 st[8]) {  // continue after one absorbed 64-byte block
  for (int i = 0; i < 8; i++) s->st[i] = st[i];
  for (int i = 0; i < 16; i++) s->w[i] = 0;
  s->fill = 0;
  s->total = 64;
}
LAMD_HD void shs_final(sha_stream *s, u8 out32[32]) {  // single SHA-256 of everything absorbed
  const u64 bits = s->total * 8;
  const u8 pad = 0x80, zero = 0;
  shs_update(s, &pad, 1);
  while (s->fill != 56) shs_update(s, &zero, 1);
  s->w[14] = (u32)(bits >> 32);
  s->w[15] = (u32)bits;
  sha256_compress(s->st, s->w);
  for (int i = 0; i < 8; i++) {
    out32[4 * i] = (u8)(s->st[i] >> 24); out32[4 * i + 1] = (u8)(s->st[i] >> 16);
    out32[4 * i + 2] = (u8)(s->st[i] >> 8); out32[4 * i + 3] = (u8)s->st[i];
  }
}
// state after SHA256(tag) || SHA256(tag); the tag is given in up to two pieces
LAMD_HD void bolt12_tag_midstate(const u8 *tag1, size_t len1, const u8 *tag2, size_t len2, u32 st[8]) {
  sha_stream s;
  u8 h[32];
  shs_init(&s);
  shs_update(&s, tag1, len1);
  shs_update(&s, tag2, len2);
  shs_final(&s, h);
  shs_init(&s);
  shs_update(&s, h, 32);
  shs_update(&s, h, 32);  // 64 bytes: compressed, the state is the midstate
  for (int i = 0; i < 8; i++) st[i] = s.st[i];
}
struct bolt12_mids {
  u32 leaf[8], branch[8], sig[8];  // "LnLeaf", "LnBranch", "lightning" || messagename || fieldname
};
// H("LnBranch", lesser || greater) (merkle_pair, :98-112)
LAMD_HD void bolt12_pair(const u32 branch_mid[8], const u8 a[32], const u8 b[32], u8 out[32]) {
  int cmp = 0;
  for (int i = 0; i < 32 && cmp == 0; i++) cmp = (int)a[i] - (int)b[i];
  sha_stream s;
  shs_init_mid(&s, branch_mid);
  shs_update(&s, cmp > 0 ? b : a, 32);
  shs_update(&s, cmp > 0 ? a : b, 32);
  shs_final(&s, out);
}
constexpr int BOLT12_STACK = 24;  // complete subtrees pending: one per bit of the leaf count
// false: the stream breaks the TLV rules, or holds no leaf
LAMD_HD bool bolt12_merkle_root(const u8 *tlv, size_t len, const bolt12_mids &m, u8 root[32]) {
  u8 stack[BOLT12_STACK][32];
  int level[BOLT12_STACK];
  int sp = 0;
  u32 nonce_mid[8];
  bool have_leaf = false;
  bool first = true;
  u64 prev = 0;
  size_t pos = 0;
  while (pos < len) {
    const size_t start = pos;
    u64 type, length;
    size_t l = wire_bigsize(tlv + pos, len - pos, &type);
    if (!l) return false;
    const size_t tlen = l;
    pos += l;
    if (!first && type <= prev) return false;
    first = false;
    prev = type;
    l = wire_bigsize(tlv + pos, len - pos, &length);
    if (!l) return false;
    pos += l;
    if (length > len - pos) return false;
    pos += (size_t)length;
    // the nonce tag comes from the last record seen before the first leaf exists (merkle_tlv_full_, :267-269)
    if (!have_leaf) {
      const u8 ln[7] = {'L', 'n', 'N', 'o', 'n', 'c', 'e'};
      bolt12_tag_midstate(ln, 7, tlv + start, pos - start, nonce_mid);
    }
    if (type >= 240 && type <= 1000) continue;
    u8 leaf[32], nonce[32], node[32];
    sha_stream s;
    shs_init_mid(&s, m.leaf);
    shs_update(&s, tlv + start, pos - start);
    shs_final(&s, leaf);
    shs_init_mid(&s, nonce_mid);
    shs_update(&s, tlv + start, tlen);  // the type as it is encoded (1-9 bytes)
    shs_final(&s, nonce);
    bolt12_pair(m.branch, leaf, nonce, node);
    have_leaf = true;
    int lv = 0;
    while (sp > 0 && level[sp - 1] == lv) {  // binary counter: two complete subtrees of one size become one of the next
      u8 t[32];
      bolt12_pair(m.branch, stack[sp - 1], node, t);
      for (int i = 0; i < 32; i++) node[i] = t[i];
      sp--;
      lv++;
    }
    if (sp >= BOLT12_STACK) return false;
    for (int i = 0; i < 32; i++) stack[sp][i] = node[i];
    level[sp++] = lv;
  }
  if (!have_leaf) return false;
  // what is left are complete subtrees of decreasing size: fold from the smallest
  for (int i = 0; i < 32; i++) root[i] = stack[sp - 1][i];
  for (int k = sp - 2; k >= 0; k--) {
    u8 t[32];
    bolt12_pair(m.branch, stack[k], root, t);
    for (int i = 0; i < 32; i++) root[i] = t[i];
  }
  return true;
}
// sighash_from_merkle (:308-318)
LAMD_HD void bolt12_sighash(const bolt12_mids &m, const u8 root[32], u8 out32[32]) {
  sha_stream s;
  shs_init_mid(&s, m.sig);
  shs_update(&s, root, 32);
  shs_final(&s, out32);
}

}  // namespace lamd

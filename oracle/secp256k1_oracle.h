/* CPU oracle for the signature-verification hot path -- TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may link or
 * call this.  The product (lightning_amd/csrc) never does.
 *
 * Plain-C restatement of what Core Lightning (reference: /root/reference, v26.06.6)
 * computes on this path.  The arithmetic below the veneer lives in libsecp256k1-zkp
 * (nested submodule of external/libwally-core, release line 1.4.0), which is an EMPTY,
 * un-vendored submodule in /root/reference -- so `oracle/_ref` cannot be built and the
 * library's algorithm is restated here from SEC1/SEC2, BIP-62 (low-S), BIP-340, BIP-143.
 *
 * Parity pinning: tests/test_oracle_golden.py checks every function here against
 *  (1) all literal vectors the reference's own tests hold for this path (KAT-G, KAT-O,
 *      KAT-B11: tests/golden/kat.json, provenance file:line recorded per vector),
 *  (2) the BIP-340 official vectors 0-14 (recalled offline; 0-4 self-authenticate by
 *      re-signing, 5-14 by their documented property),
 *  (3) the spec-level Python big-int model oracle/pyref.py on seeded random + edge rows,
 *  (4) OpenSSL (oracle/openssl_xcheck.c): its own ECDSA verification, and for BIP-340 and public-key recovery a third
 *      statement of the protocol over OpenSSL's generic curve arithmetic and SHA-256, on every golden vector and on
 *      seeded random / damaged rows.
 * libsecp256k1's own edge semantics (not exercised by any in-tree reference test) are
 * therefore pinned by (2)-(4), not by the reference: see DESIGN.md "parity status".
 */
#ifndef LIGHTNING_AMD_ORACLE_H
#define LIGHTNING_AMD_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* one-time table build (idempotent, called lazily by everything else) */
void orc_init(void);

/* ccan/ccan/crypto/sha256/sha256.c:87-250 (FIPS 180-4) and bitcoin/shadouble.c:7-11 */
void orc_sha256(const uint8_t *data, size_t len, uint8_t out32[32]);
void orc_sha256d(const uint8_t *data, size_t len, uint8_t out32[32]);

/* secp256k1_ec_pubkey_parse as reached from common/node_id.c:21-27 and bitcoin/pubkey.c:14-24.
 * len 33 (02/03) or 65 (04, hybrid 06/07).  1 = valid (out64 = affine X||Y big-endian). */
int orc_pubkey_parse(const uint8_t *pub, size_t len, uint8_t out64[64]);

/* secp256k1_ecdsa_signature_parse_compact (wire/fromwire.c:188-199): 0 iff r >= n or s >= n */
int orc_sig_parse_compact(const uint8_t sig64[64]);

/* secp256k1_ecdsa_signature_parse_der (strict DER) -> 64-byte compact (r||s); 1 = parsed.
 * As upstream, an out-of-range integer parses as 0 (and is then rejected by verify). */
int orc_sig_parse_der(const uint8_t *der, size_t len, uint8_t out64[64]);

/* signature_from_der, bitcoin/signature.c:310-323: DER + trailing sighash byte */
int orc_signature_from_der(const uint8_t *der, size_t len, uint8_t out64[64], int *sighash_type);

/* check_signed_hash, bitcoin/signature.c:174-192, on serialized inputs:
 * ok = parse_compact(sig64) && pubkey_parse(pub) && secp256k1_ecdsa_verify (low-S enforced).
 * With publen == 33 this is check_signed_hash_nodeid (common/node_id.c:72-80). */
int orc_ecdsa_verify(const uint8_t hash32[32], const uint8_t sig64[64],
		     const uint8_t *pub, size_t publen);

/* check_schnorr_sig, bitcoin/signature.c:408-430 == BIP-340 Verify with a 32-byte message */
int orc_schnorr_verify(const uint8_t msg32[32], const uint8_t xonly32[32],
		       const uint8_t sig64[64]);

/* secp256k1_ecdsa_recoverable_signature_parse_compact + secp256k1_ecdsa_recover (common/bolt11.c:1021-1046,
 * lightningd/signmessage.c:193): 1 and the compressed key, or 0 (and a zeroed key) where the library calls fail */
int orc_ecdsa_recover(const uint8_t hash32[32], const uint8_t sig64[64], int recid, uint8_t out33[33]);
void orc_ecdsa_recover_batch(size_t n, const uint8_t *hash32, const uint8_t *sig64, const uint8_t *recid, uint8_t *pub33,
			     uint8_t *ok, int nthreads);

/* gossipd/sigcheck.c on raw wire messages.  Return 0 = OK (reference returns NULL),
 * k>0 = the k-th signature is the first bad one (channel_announcement: 1 node_signature_1,
 * 2 node_signature_2, 3 bitcoin_signature_1, 4 bitcoin_signature_2; others: 1),
 * -1 = malformed (rejected by fromwire_* before sigcheck would run). */
int orc_sigcheck_channel_announcement(const uint8_t *msg, size_t len);
int orc_sigcheck_channel_update(const uint8_t *msg, size_t len, const uint8_t node_id33[33]);
int orc_sigcheck_node_announcement(const uint8_t *msg, size_t len);

/* batch drivers (OpenMP when nthreads > 1); pub stride = publen. out[i] in {0,1} */
void orc_ecdsa_verify_batch(size_t n, const uint8_t *hash32, const uint8_t *sig64,
			    const uint8_t *pub, size_t publen, uint8_t *out, int nthreads);
void orc_schnorr_verify_batch(size_t n, const uint8_t *msg32, const uint8_t *xonly32,
			      const uint8_t *sig64, uint8_t *out, int nthreads);

void orc_sigcheck_gossip_batch(size_t n, const uint8_t *msgs, const uint64_t *off, const uint8_t *node_ids33,
			       int8_t *out, int nthreads);

/* ---- test-vector generation only (the reference signs via hsmd; never on this path) ---- */
int orc_pubkey_create(const uint8_t seckey32[32], uint8_t out65[65]);
int orc_ecdsa_sign(const uint8_t hash32[32], const uint8_t seckey32[32],
		   const uint8_t nonce32[32], uint8_t sig64[64]);
int orc_schnorr_sign(const uint8_t msg32[32], const uint8_t seckey32[32],
		     const uint8_t aux32[32], uint8_t sig64[64]);

#ifdef __cplusplus
}
#endif
#endif

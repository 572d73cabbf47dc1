/* Host-side mirror of Core Lightning's own prototypes for the signature-check path, implemented
 * over the C ABI of liblightning_amd.so.  Same names, argument meaning and error behaviour as
 * the reference (v26.06.6):
 *
 *   bitcoin/signature.h:85-87    check_signed_hash()
 *   bitcoin/signature.h:120-124  check_tx_sig()            (see note below)
 *   bitcoin/signature.h:129-131  check_schnorr_sig()
 *   common/bolt11.c:1021-1057    lamd_secp256k1_ecdsa_recoverable_signature_parse_compact(), _convert(), lamd_secp256k1_ecdsa_recover(),
 *                                lamd_secp256k1_ecdsa_verify() (also lightningd/dual_open_control.c:2254) -- prefixed, see below
 *   onchaind/onchaind.c:388-438  grind_htlc_tx_fee()       (static there; its statics become arguments)
 *   bitcoin/signature.h:158-159  signature_from_der()      (bitcoin/signature.c:310-323)
 *   common/node_id.h:72-82       pubkey_from_node_id(), check_signed_hash_nodeid()
 *   bitcoin/pubkey.h             pubkey_from_der(), pubkey_to_der()
 *   bitcoin/shadouble.h:15       sha256_double()
 *   wire/fromwire.c:188-199      fromwire_secp256k1_ecdsa_signature()
 *   gossipd/sigcheck.h:7-28      sigcheck_channel_update/_channel_announcement/_node_announcement()
 *
 * The opaque types keep the reference's names and sizes; their CONTENT is this library's own
 * (the reference never looks inside them): secp256k1_ecdsa_signature.data = r||s big-endian,
 * secp256k1_pubkey.data = affine X||Y big-endian.
 *
 * Differences a maintainer has to know about (also in INTEGRATION.md):
 *  - tal: stand-alone, "tal arrays" are shim_tal_dup()'s (length in front of the data) and the sigcheck_* error strings are such
 *    arrays owned by the caller; built with -DLAMD_SHIM_WITH_CCAN_TAL (in-tree) lengths come from ccan's tal_bytelen() and the
 *    strings are tal_strdup()ed onto `ctx` -- the reference's ownership exactly.
 *  - check_tx_sig() has the reference's prototype (bitcoin/signature.h:120-124).  `struct bitcoin_tx` here is a plain
 *    mirror of what that path reads from the reference's libwally-backed one (version, locktime, inputs with the amounts
 *    the PSBT carries, outputs); the two script arguments are tal-style arrays whose length travels with the pointer
 *    (shim_tal_bytelen(); tal_bytelen() in the reference) -- make them with shim_tal_dup().  The BIP143 hash of
 *    bitcoin_tx_hash_for_sig() (:120-151) is computed on the device.
 *  - all elliptic-curve work (key decompression, verification) runs on the GPU through
 *    lamd_*; hashing and DER/compact parsing are host code.  Without a device every check
 *    fails closed (false / "engine error" string), there is no CPU verification path.
 */
#ifndef LIGHTNING_AMD_CLN_SHIM_H
#define LIGHTNING_AMD_CLN_SHIM_H
#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef uint8_t u8;
typedef void tal_t;

struct sha256 { union { uint32_t u32[8]; unsigned char u8[32]; } u; };      /* ccan/crypto/sha256 */
struct sha256_double { struct sha256 sha; };                                    /* bitcoin/shadouble.h:9-11 */
typedef struct { unsigned char data[64]; } secp256k1_ecdsa_signature;          /* opaque; here r||s */
typedef struct { unsigned char data[64]; } secp256k1_pubkey;                   /* opaque; here X||Y */
struct pubkey { secp256k1_pubkey pubkey; };                                     /* bitcoin/pubkey.h:15-18 */
struct node_id { u8 k[33]; };                                                   /* common/node_id.h:11-13 */
struct bip340sig { u8 u8[64]; };                                                /* bitcoin/signature.h:145-147 */

enum sighash_type { SIGHASH_ALL = 1, SIGHASH_NONE = 2, SIGHASH_SINGLE = 3, SIGHASH_ANYONECANPAY = 0x80 };
struct bitcoin_signature { secp256k1_ecdsa_signature s; enum sighash_type sighash_type; };
#define PUBKEY_CMPR_LEN 33

/* common/setup.c:38-64 analogue: creates the process-global engine context (device 0 unless
 * LAMD_DEVICE is set).  Returns false (and every later check fails closed) without a GPU. */
bool lamd_shim_setup(void);
/* adopt an engine context the process already owns instead of creating one (the caller keeps ownership) */
void lamd_shim_use_context(void *lamd_ctx_ptr);
void lamd_shim_shutdown(void);
const char *lamd_shim_last_error(void);

void sha256_double(struct sha256_double *shadouble, const void *p, size_t len);

bool pubkey_from_der(const u8 *der, size_t len, struct pubkey *key);
void pubkey_to_der(u8 der[PUBKEY_CMPR_LEN], const struct pubkey *key);
bool pubkey_from_node_id(struct pubkey *key, const struct node_id *id);

/* returns false (the reference calls fromwire_fail) iff r >= n or s >= n */
bool fromwire_secp256k1_ecdsa_signature(const u8 compact[64], secp256k1_ecdsa_signature *sig);
bool signature_from_der(const u8 *der, size_t len, struct bitcoin_signature *sig);

bool check_signed_hash(const struct sha256_double *hash, const secp256k1_ecdsa_signature *signature,
		       const struct pubkey *key);
bool check_signed_hash_nodeid(const struct sha256_double *hash, const secp256k1_ecdsa_signature *signature,
			      const struct node_id *id);
bool check_schnorr_sig(const struct sha256 *hash, const secp256k1_pubkey *pubkey, const struct bip340sig *sig);
/* tal-style byte arrays: the length is stored in front of the data; shim_tal_bytelen() reads it (the role of ccan/tal's
 * tal_bytelen() in the reference -- deliberately NOT that name, so that the mirror can be linked next to the real ccan/tal;
 * built with -DLAMD_SHIM_WITH_CCAN_TAL it forwards to ccan's).  A pointer that did not come from shim_tal_dup() gives
 * SHIM_TAL_FOREIGN and the check that was handed it fails closed (false), it does not abort(). */
#define SHIM_TAL_FOREIGN ((size_t)-1)
u8 *shim_tal_dup(const tal_t *ctx, const u8 *src, size_t len);
size_t shim_tal_bytelen(const void *ptr);
void shim_tal_free(const void *ptr);

/* what check_tx_sig()/bitcoin_tx_hash_for_sig() read from the reference's struct bitcoin_tx (bitcoin/tx.h: wtx + psbt) */
struct bitcoin_tx_input {
	u8 txid[32];         /* as serialised in the transaction (and hashed) */
	uint32_t index;      /* vout */
	uint32_t sequence;
	uint64_t amount_sat; /* psbt_input_get_amount(tx->psbt, in) */
};
struct bitcoin_tx_output {
	uint64_t amount_sat;
	const u8 *script;    /* scriptPubKey, tal-style array */
};
struct bitcoin_tx {
	uint32_t version, locktime;
	size_t num_inputs, num_outputs;
	const struct bitcoin_tx_input *inputs;
	const struct bitcoin_tx_output *outputs;
};
/* bitcoin/signature.h:120-124, same prototype, argument order and meaning: witness NULL = hash `subscript`, and then only
 * SIGHASH_ALL is accepted (bitcoin/signature.c:198-211). */
bool check_tx_sig(const struct bitcoin_tx *tx, size_t input_num,
		  const u8 *subscript,
		  const u8 *witness,
		  const struct pubkey *key,
		  const struct bitcoin_signature *sig);
/* channeld/channeld.c:2171-2232 (handle_peer_commit_sig) as ONE call: the commitment signature under remote_funding over txs[0] (:2171), the
 * count of HTLC signatures (:2203), then htlc_sigs[i] under remote_htlckey over txs[1+i] with htlc_wscripts[i] (:2224) -- all 1 + N signatures go
 * to the device as one batch (lamd_check_commitment_signed), the answer is the reference's: NULL when it would have gone on, else the text it
 * hands peer_failed_warn() for the FIRST check that fails, character for character:
 *   "Bad commit_sig signature <local_index> <sig> for tx <tx> wscript <hex> key <key> feerate <feerate><commit_warning_tail>"
 *   "Expected <n> htlc sigs, not <m>"
 *   "Bad commit_sig signature <sig> for htlc <tx> wscript <hex> key <key>"
 * (fmt_bitcoin_signature: DER + sighash byte; fmt_bitcoin_tx: hex of the serialised transaction; fmt_pubkey: 33 bytes).
 * txs: tal-style array of pointers, as channel_txs() returns it (:2144); htlc_sigs: tal-style array; htlc_wscripts[i]: what
 * bitcoin_tx_output_get_witscript(tmpctx, txs[0], txs[i+1]->wtx->inputs[0].index) gives the reference (:2215); commit_warning_tail: the rest
 * of the first warning after the feerate (". Outpoint %s, funding_sats: %s, funding_txid: %s, inflight splice count: %zu" -- values the
 * check never reads), may be NULL.  The string is allocated like the sigcheck_* ones. */
const char *check_commit_sigs(const tal_t *ctx, uint64_t local_index, const struct bitcoin_tx *const *txs, const u8 *funding_wscript,
			      const struct pubkey *remote_funding, const struct bitcoin_signature *commit_sig,
			      const u8 *const *htlc_wscripts, const struct pubkey *remote_htlckey,
			      const struct bitcoin_signature *htlc_sigs, uint32_t feerate, const char *commit_warning_tail);
/* the round-1 form, kept for callers that already hold the serialised BIP143 preimage */
bool check_tx_sig_preimage(const u8 *bip143_preimage, size_t preimage_len, const u8 *witness_script,
			   const struct pubkey *key, const struct bitcoin_signature *sig);

/* BOLT #12 (common/bolt12.c:80-92, common/bolt12_merkle.h:8-12,70-79).  `fields` is a tal-style array of struct tlv_field
 * (wire/tlvstream.h:16-26; make it with shim_tal_dup(): tal_count = shim_tal_bytelen / sizeof) in stream order; the merkle tree and the tagged
 * hash are computed on the device from the re-serialised fields, the verification is check_schnorr_sig()'s. */
struct tlv_field {
	const void *meta;   /* const struct tlv_record_type *: unused here */
	uint64_t numtype;
	size_t length;
	u8 *value;
};
void merkle_tlv(const struct tlv_field *fields, struct sha256 *merkle);
void sighash_from_merkle(const char *messagename, const char *fieldname, const struct sha256 *merkle, struct sha256 *sighash);
bool bolt12_check_signature(const struct tlv_field *fields, const char *messagename, const char *fieldname, const struct pubkey *key,
			    const struct bip340sig *sig);

/* The four libsecp256k1 calls the reference makes directly on this path -- public-key recovery (common/bolt11.c:1021-1046,
 * lightningd/signmessage.c:193) and the BOLT #11 `n`-field / dual-open verification (common/bolt11.c:1026-1027,1055,
 * lightningd/dual_open_control.c:2254) -- with the library's prototypes and return conventions (1 = ok, 0 = failure; the context
 * argument is accepted and ignored), under a lamd_ PREFIX.  They are NOT exported under libsecp256k1's own names: lightningd, libwally
 * and bitcoin/signature.c all link the real library, whose opaque secp256k1_pubkey / secp256k1_ecdsa_signature have another memory
 * layout than this mirror's (X||Y and r||s big-endian) -- a second global `secp256k1_ecdsa_verify` would be a duplicate definition
 * or, interposed, would be handed the real library's objects and fail every check.  A translation unit that wants the reference's
 * call sites to compile unchanged against THIS header's types defines LAMD_SHIM_LIBSECP_NAMES before including it (macros, no symbols).
 * The opaque recoverable signature holds r||s||recid; convert drops the recovery id (returns 1 always, as upstream);
 * lamd_secp256k1_ecdsa_verify() is check_signed_hash() without the struct wrappers -- 1 iff r, s in [1, n-1], s <= n/2 (upstream
 * rejects high-S: bitcoin/signature.c:185-187) and the signature verifies; msghash32 is used as given (32 bytes, reduced mod n). */
typedef struct { unsigned char data[65]; } secp256k1_ecdsa_recoverable_signature;
int lamd_secp256k1_ecdsa_recoverable_signature_parse_compact(const void *ctx, secp256k1_ecdsa_recoverable_signature *sig,
							      const unsigned char *input64, int recid);
int lamd_secp256k1_ecdsa_recover(const void *ctx, secp256k1_pubkey *pubkey, const secp256k1_ecdsa_recoverable_signature *sig,
				 const unsigned char *msghash32);
int lamd_secp256k1_ecdsa_recoverable_signature_convert(const void *ctx, secp256k1_ecdsa_signature *sig,
						       const secp256k1_ecdsa_recoverable_signature *sigin);
int lamd_secp256k1_ecdsa_verify(const void *ctx, const secp256k1_ecdsa_signature *sig, const unsigned char *msghash32,
				const secp256k1_pubkey *pubkey);
#ifdef LAMD_SHIM_LIBSECP_NAMES
#define secp256k1_ecdsa_recoverable_signature_parse_compact lamd_secp256k1_ecdsa_recoverable_signature_parse_compact
#define secp256k1_ecdsa_recover lamd_secp256k1_ecdsa_recover
#define secp256k1_ecdsa_recoverable_signature_convert lamd_secp256k1_ecdsa_recoverable_signature_convert
#define secp256k1_ecdsa_verify lamd_secp256k1_ecdsa_verify
#endif
void node_id_from_pubkey(struct node_id *id, const struct pubkey *key);   /* common/node_id.c:12-19 */

/* onchaind/onchaind.c:388-438.  The reference's file-scope state becomes arguments (min/max_possible_feerate,
 * keyset->other_htlc_key) and the transaction is given as its BIP143 preimage, the serialised outputs that hashOutputs
 * covers and the input amount (see lamd_grind_htlc_tx_fee).  true: *fee = the fee whose signature check passed, found at
 * the lowest feerate of the range that produces it -- where the reference's ascending loop stops.  (On failure the
 * reference leaves the last fee it tried in *fee; here *fee is untouched.) */
bool grind_htlc_tx_fee(uint64_t *fee_sat, const u8 *bip143_preimage, size_t preimage_len, const u8 *outputs, size_t outputs_len,
		       uint64_t input_sat, const struct bitcoin_signature *remotesig, const u8 *wscript, uint64_t weight,
		       uint32_t min_possible_feerate, uint32_t max_possible_feerate, const struct pubkey *other_htlc_key);

/* gossipd/sigcheck.h:7-28 -- the reference's prototypes, token for token (tests/test_abi.py diffs them against the reference header).
 * The message is a tal array, as everywhere in the reference (sigcheck.c:30-33 hashes tal_count(msg) - 66 bytes): its length is read
 * with shim_tal_bytelen() -- ccan's tal_bytelen() in a -DLAMD_SHIM_WITH_CCAN_TAL build -- and a pointer whose length cannot be read fails
 * closed ("... is not a tal array").  NULL = OK, else the reference's exact wording ("Bad node_signature_1 <der-hex> hash <hex> on
 * channel_announcement <hex>", ...): a tal string -- tal_strdup()ed onto `ctx` in a -DLAMD_SHIM_WITH_CCAN_TAL build (sigcheck.c:36-41
 * tal_fmt(ctx, ...)); stand-alone a shim_tal_dup()ed one whose owner is the caller (shim_tal_free(); `ctx` has no allocator behind it). */
const char *sigcheck_channel_update(const tal_t *ctx,
				    const struct node_id *node_id,
				    const secp256k1_ecdsa_signature *node_sig,
				    const u8 *update);
const char *sigcheck_channel_announcement(const tal_t *ctx,
					  const struct node_id *node1_id,
					  const struct node_id *node2_id,
					  const struct pubkey *bitcoin1_key,
					  const struct pubkey *bitcoin2_key,
					  const secp256k1_ecdsa_signature *node1_sig,
					  const secp256k1_ecdsa_signature *node2_sig,
					  const secp256k1_ecdsa_signature *bitcoin1_sig,
					  const secp256k1_ecdsa_signature *bitcoin2_sig,
					  const u8 *announcement);
const char *sigcheck_node_announcement(const tal_t *ctx,
				       const struct node_id *node_id,
				       const secp256k1_ecdsa_signature *node_sig,
				       const u8 *node_announcement);
/* The same for callers that hold a plain (pointer, length) pair -- a message still sitting in a receive buffer: msg_len replaces
 * tal_count(msg); the string is allocated as above. */
const char *sigcheck_channel_update_len(const tal_t *ctx, const struct node_id *node_id,
					const secp256k1_ecdsa_signature *node_sig, const u8 *update, size_t msg_len);
const char *sigcheck_channel_announcement_len(const tal_t *ctx, const struct node_id *node1_id, const struct node_id *node2_id,
					      const struct pubkey *bitcoin1_key, const struct pubkey *bitcoin2_key,
					      const secp256k1_ecdsa_signature *node1_sig, const secp256k1_ecdsa_signature *node2_sig,
					      const secp256k1_ecdsa_signature *bitcoin1_sig, const secp256k1_ecdsa_signature *bitcoin2_sig,
					      const u8 *announcement, size_t msg_len);
const char *sigcheck_node_announcement_len(const tal_t *ctx, const struct node_id *node_id,
					   const secp256k1_ecdsa_signature *node_sig, const u8 *node_announcement, size_t msg_len);

#ifdef __cplusplus
}
#endif
#endif

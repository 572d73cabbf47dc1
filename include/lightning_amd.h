/* lightning_amd -- MI355X (gfx950) batched secp256k1 signature verification.
 *
 * C ABI of liblightning_amd.so: the drop-in boundary for Core Lightning's signature-check
 * hot path.  Every entry point names the reference interface it replaces (paths relative to
 * the Core Lightning tree, v26.06.6).  Plain pointers and sizes only; all byte formats are the
 * SERIALISED forms the reference handles at its own boundaries (the opaque in-memory
 * secp256k1_pubkey / secp256k1_ecdsa_signature layouts are implementation-defined and never
 * cross this ABI):
 *     hash / sighash / BIP-340 message : 32 raw bytes   (struct sha256_double, bitcoin/shadouble.h:9-11)
 *     ECDSA signature                  : 64-byte compact big-endian r||s (wire/fromwire.c:188-199)
 *     public key                       : 33-byte SEC1 compressed (struct node_id, common/node_id.h:11-13;
 *                                        pubkey_to_der, bitcoin/pubkey.c:26-34) or 65-byte uncompressed
 *     BIP-340 signature / x-only key   : 64 / 32 raw bytes (struct bip340sig, bitcoin/signature.h:145-147)
 *
 * Verdict convention: ok[i] = 1 iff the reference would have accepted, i.e.
 *     parse_ok(signature) && parse_ok(key) && verify_ok
 * so a host-side parse failure and a device-side reject are indistinguishable from the
 * reference's combined outcome.  There is NO CPU fallback: without a working HIP device every
 * call returns LAMD_ERR_NO_DEVICE / LAMD_ERR_HIP and the caller must fail closed.
 *
 * Threading: a context owns one device, one stream and its staging buffers; calls on one
 * context must be serialised by the caller (the reference's daemons are single-threaded:
 * gossipd/gossipd.c:620-625, channeld/channeld.c:7063-7121).  Use one context per thread/GPU.
 */
#ifndef LIGHTNING_AMD_H
#define LIGHTNING_AMD_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lamd_ctx lamd_ctx;

enum {
	LAMD_OK = 0,
	LAMD_ERR_NO_DEVICE = -1, /* no usable HIP device (or wrong architecture) */
	LAMD_ERR_HIP = -2,       /* a HIP call failed; lamd_last_error() has the text */
	LAMD_ERR_ARG = -3,       /* bad argument (NULL, unsupported key length, ...) */
	LAMD_ERR_NOMEM = -4,
	LAMD_ERR_STATE = -5      /* streaming API misuse (poll before flush, queue full, ...) */
};

/* A context is NOT thread-safe -- including the single-item veneers and every host-buffer call of <= 4096 rows, which share one pinned block
 * and one completion word per context: one context per calling thread, or serialise the calls (lamd_multi_* below runs one context per device
 * behind its own lock).  A latency-path call busy-waits for its completion word for at most LAMD_SPIN_US microseconds (default 2000), then
 * blocks in the runtime.  Every entry point returns a
 * LAMD_ERR_* code (< 0) or a documented non-negative value; lamd_last_error() describes the last failure. */
/* ---- lifecycle.  Replaces the process-global secp256k1_ctx set up in common/setup.c:58
 * (secp256k1_ctx = wally_get_secp_context(), common/utils.c:16): the one-time work here is the
 * upload/build of the static table of G multiples in HBM.
 * Hardware queues: the context runs its calls on several HIP streams ("lanes", LAMD_LANES) that only overlap when each has a
 * hardware queue of its own.  ROCm reads GPU_MAX_HW_QUEUES (default 4) at the runtime's FIRST call.  PRECONDITION for full
 * throughput: the host process exports GPU_MAX_HW_QUEUES=16 (or setenv()s it) BEFORE its first HIP call -- the library does not
 * touch the process environment.  lamd_init() records what it found (lamd_info.hw_queues_env; 0 = unset, i.e. the runtime's 4:
 * the lanes then share queues and the pipelined loop runs ~7 % slower, nothing else changes).  Create the context BEFORE an RCCL communicator (ncclCommInitRank / torch.distributed
 * init_process_group): RCCL's own streams otherwise take hardware queues first and the lanes end up sharing one (measured:
 * -8 % on the 2 M-row step).  Correctness does not depend on either (tests run at 4, 16 and 32 queues). */
int lamd_init(lamd_ctx **ctx, int device);
void lamd_shutdown(lamd_ctx *ctx);
const char *lamd_last_error(const lamd_ctx *ctx);
const char *lamd_version(void);

/* ---- batch verification, host buffers in, host verdicts out (H2D + kernels + D2H, synchronous).
 * A call of <= 4096 rows is ONE kernel launch (k_small_verify, a grid of 64-row blocks): the rows travel through pinned device-mapped
 * memory, each verification is split over eight waves, the verdict bytes come back the same way (LAMD_SMALL_KERNEL=0 sends such calls
 * down the general path).  A key such calls bring for the second time gets its comb table built and cached, so recurring keys (a peer's
 * node id, a channel's htlc key) are cache hits from their third sight on.
 *
 * lamd_verify_ecdsa_batch: n independent check_signed_hash() calls (bitcoin/signature.c:174-192,
 * decl bitcoin/signature.h:85-87).  publen is 33 or 65 for every key of the batch; key i is at
 * pub + i*pubstride.  With publen 33 this is check_signed_hash_nodeid() (common/node_id.c:72-80).
 * 65-byte keys may be 0x04 or hybrid 0x06/0x07, as secp256k1_ec_pubkey_parse accepts them. */
int lamd_verify_ecdsa_batch(lamd_ctx *ctx, size_t n, const uint8_t *hash32, const uint8_t *sig64,
			    const uint8_t *pub, size_t publen, size_t pubstride, uint8_t *ok);

/* n independent check_schnorr_sig() calls (bitcoin/signature.c:408-430, decl signature.h:129-131):
 * BIP-340 verification of a 32-byte message under an x-only key. */
int lamd_verify_schnorr_batch(lamd_ctx *ctx, size_t n, const uint8_t *msg32, const uint8_t *xonly32,
			      const uint8_t *sig64, uint8_t *ok);

/* ---- the same with every buffer already resident in HBM (device pointers), asynchronous: the work is ordered after
 * whatever is already queued on the context's stream (lamd_stream()) and the verdicts are complete after
 * lamd_synchronize(), or, without blocking the host, for a stream passed to lamd_stream_wait_results().  Successive
 * calls rotate over LAMD_LANES internal lanes (default 6; own streams and workspaces) so that one call's key
 * de-duplication and table building run under the ecmult kernels of the calls before it; LAMD_LANES=1 turns that off
 * (strictly one stream).  d_ok is never read and is written once per call, with final verdicts (one device-to-device copy at the
 * end of the call): calls in flight at the same time may be handed the same verdict buffer.
 * This is what bench.py times. */
int lamd_verify_ecdsa_batch_device(lamd_ctx *ctx, size_t n, const void *d_hash32, const void *d_sig64,
				   const void *d_pub, size_t publen, size_t pubstride, void *d_ok);
int lamd_verify_schnorr_batch_device(lamd_ctx *ctx, size_t n, const void *d_msg32, const void *d_xonly32,
				     const void *d_sig64, void *d_ok);
void *lamd_stream(lamd_ctx *ctx); /* hipStream_t */
int lamd_synchronize(lamd_ctx *ctx);
/* `stream` (hipStream_t) waits, on the device, for every verification submitted so far */
int lamd_stream_wait_results(lamd_ctx *ctx, void *stream);
/* The same in two phases: lamd_results_mark() remembers "everything submitted so far" in slot 0..15 (events on the lanes' streams,
 * nothing waits); lamd_stream_wait_mark() later makes `stream` wait for exactly that work.  A consumer that joins late -- after
 * the next batch has been submitted -- keeps its stream's wait short: a wait that sits in a hardware queue for a whole batch
 * holds up every other stream that shares the queue (bench.py's collective path: the all-gather of step k is issued after the
 * calls of step k+1). */
int lamd_results_mark(lamd_ctx *ctx, int slot);
/* ... "the call submitted last" only: ONE event, on the lane that ran it -- what the consumer of that call's verdict buffer needs (bench.py's
 * per-call all-gathers); lamd_stream_wait_mark() then waits for that one event. */
int lamd_results_mark_last(lamd_ctx *ctx, int slot);
int lamd_stream_wait_mark(lamd_ctx *ctx, int slot, void *stream);
/* verification submitted from now on waits, on the device, for what `stream` holds at this moment */
int lamd_wait_stream(lamd_ctx *ctx, void *stream);
/* the same for one event (hipEvent_t) the caller recorded: verification submitted from now on waits, on the device, for it */
int lamd_wait_event(lamd_ctx *ctx, void *event);

/* ---- single-item veneers with the reference's exact boolean semantics (1 = true, 0 = false,
 * < 0 = engine error: treat as failure).  They run a batch of one through the one-launch path above (0.17-0.21 ms under a key
 * the cache knows, ~0.9 ms at a key's first sight on one MI355X); callers on the hot path should batch. */
int lamd_check_signed_hash(lamd_ctx *ctx, const uint8_t hash32[32], const uint8_t sig64[64],
			   const uint8_t *pubkey, size_t publen);          /* bitcoin/signature.h:85-87 */
int lamd_check_signed_hash_nodeid(lamd_ctx *ctx, const uint8_t hash32[32], const uint8_t sig64[64],
				  const uint8_t node_id33[33]);             /* common/node_id.h:80-82 */
int lamd_check_schnorr_sig(lamd_ctx *ctx, const uint8_t hash32[32], const uint8_t pubkey33[33],
			   const uint8_t bip340sig64[64]);                   /* bitcoin/signature.h:129-131 */

/* ---- n independent check_tx_sig() calls (bitcoin/signature.c:194-221, decl signature.h:120-124) on caller-built
 * BIP143 preimages: preimage i = preimages[off[i]..off[i+1]) is what wally_tx_get_btc_signature_hash() hashes for
 * (tx, input, script, amount, sighash_type) -- signature.c:145-148.  The device applies the sighash-type gate
 * (:206-211: SIGHASH_ALL, or SINGLE|ANYONECANPAY only with a witness script), double-SHA256s the preimage and
 * verifies.  This is also the shape of onchaind's fee grind (one signature/key against many candidate
 * transactions, onchaind/onchaind.c:389-438): pass the same sig/key n times. */
int lamd_check_tx_sig_batch(lamd_ctx *ctx, size_t n, const uint8_t *preimages, const uint64_t *off,
			    const uint8_t *sighash_type, const uint8_t *has_witness_script,
			    const uint8_t *sig64, const uint8_t *pub, size_t publen, size_t pubstride, uint8_t *ok);

/* ---- the same from TRANSACTION TEMPLATES: n independent check_tx_sig(tx, input_num, script, witness_script, key, sig) calls
 * (bitcoin/signature.c:194-221) with the BIP143 signature hash of bitcoin_tx_hash_for_sig() (:120-151, libwally's
 * wally_tx_get_btc_signature_hash with WALLY_TX_FLAG_USE_WITNESS) computed ON THE DEVICE: hashPrevouts / hashSequence /
 * hashOutputs, the preimage and its double SHA-256 are streamed piecewise, nothing is serialised on the host.
 * Row i describes one (transaction, input, signature):
 *   version[i], locktime[i]
 *   inputs40 + 40*in_off[i] .. in_off[i+1]   : the transaction's inputs, 40 bytes each: txid (32, as hashed) | vout u32 LE | nSequence u32 LE
 *   input_num[i]                              : which input the signature is for (>= the number of inputs: verdict 0; the reference asserts)
 *   amount_sat[i]                             : that input's amount (psbt_input_get_amount)
 *   outputs + out_off[i] .. out_off[i+1]      : the n_outputs[i] outputs in wire form (amount u64 LE | CompactSize | scriptPubKey), back to back
 *   scripts + script_off[i] .. script_off[i+1]: the script the reference would hash -- the witness script, or the redeemscript when there is none
 *   sighash_type[i], has_witness_script[i]    : the gate of :206-211 (SIGHASH_ALL, or SINGLE|ANYONECANPAY only with a witness script)
 * All offsets are uint64, arrays of n+1 entries.  The Elements branch (:136-143) is out of scope. */
int lamd_check_tx_sig_tx_batch(lamd_ctx *ctx, size_t n, const uint32_t *version, const uint32_t *locktime,
			       const uint8_t *inputs40, const uint64_t *in_off, const uint32_t *input_num, const uint64_t *amount_sat,
			       const uint8_t *outputs, const uint64_t *out_off, const uint32_t *n_outputs,
			       const uint8_t *scripts, const uint64_t *script_off,
			       const uint8_t *sighash_type, const uint8_t *has_witness_script,
			       const uint8_t *sig64, const uint8_t *pub, size_t publen, size_t pubstride, uint8_t *ok);

/* ---- ONE commitment_signed as ONE call: the loop of channeld/channeld.c:2171-2232 (handle_peer_commit_sig).  The reference checks
 *   check_tx_sig(txs[0], 0, NULL, funding_wscript, &remote_funding, &commit_sig)            (:2171)
 * and then, for i = 0 .. tal_count(htlc_sigs) - 1,
 *   check_tx_sig(txs[1+i], 0, NULL, wscript_i, &remote_htlckey, &htlc_sigs[i])              (:2224)
 * one signature per call, failing the peer at the first bad one.  Here the 1 + N rows are one batch -- the largest natural batch of the
 * product, <= 1 + 483 signatures, N of them under ONE key: the BIP143 hashes of all 1 + N inputs are computed on the device from the
 * templates (a row with a long output / input list -- the commitment transaction's one output per HTLC -- is hashed on the host while the
 * rows are packed: SHA-256 is sequential and one lane would need milliseconds for it), a commitment of <= 4096 rows is one small copy and two
 * launches (k_txsig_tx_hash, then k_small_verify taking hashes and sighash-type gate from device memory), larger ones take the batch machinery.
 *   *first_bad = -1: every signature verifies;  0: the commitment signature does not;  1 + i: htlc_sigs[i] is the first that does not
 * -- the reference's order, so the caller prints exactly the warning the reference would ("Bad commit_sig signature ..." with or without
 * "for htlc": include/cln_shim.h check_commit_sigs()).  ok_rows (optional, 1 + n_htlc bytes): every row's verdict.
 * A transaction template is what check_tx_sig() / bitcoin_tx_hash_for_sig() (bitcoin/signature.c:120-151,194-221) read of one (transaction, input):
 * the fields of lamd_check_tx_sig_tx_batch(), one struct per row; `script` is the witness script (funding_wscript; the HTLC output's wscript). */
typedef struct lamd_tx_template {
	uint32_t version, locktime;
	const uint8_t *inputs40;   /* n_inputs x (txid 32, as hashed | vout u32 LE | nSequence u32 LE) */
	uint32_t n_inputs;
	uint32_t input_num;        /* the input the signature is for (0 for commitment and HTLC transactions) */
	uint64_t amount_sat;       /* that input's amount */
	const uint8_t *outputs;    /* the outputs in wire form, back to back (amount u64 LE | CompactSize | scriptPubKey) */
	uint64_t outputs_len;
	uint32_t n_outputs;
	const uint8_t *script;     /* the witness script the signature commits to */
	uint64_t script_len;
} lamd_tx_template;
int lamd_check_commitment_signed(lamd_ctx *ctx, const lamd_tx_template *commit_tx, const uint8_t remote_funding33[33],
				 const uint8_t commit_sig64[64], uint8_t commit_sighash_type,
				 size_t n_htlc, const lamd_tx_template *htlc_txs, const uint8_t remote_htlckey33[33],
				 const uint8_t *htlc_sigs64, const uint8_t *htlc_sighash_types,
				 int64_t *first_bad, uint8_t *ok_rows);

/* ---- BOLT #12 signatures: n independent bolt12_check_signature(fields, messagename, fieldname, key, sig) calls
 * (common/bolt12.c:80-92): merkle_tlv() over the TLV stream minus its signature fields (types 240..1000) and
 * sighash_from_merkle() (common/bolt12_merkle.c:227-318: H("LnLeaf"), H("LnNonce"|first-tlv), H("LnBranch"), tag
 * "lightning"|messagename|fieldname, bitcoin/signature.c:389-405) run on the device in front of the BIP-340 verification of
 * check_schnorr_sig() (:408-430).  Row i: the serialised TLV stream tlvs + off[i] .. off[i+1] (off: n+1 entries) as it came off
 * the wire -- the reference hashes the re-serialised fields, which for a stream fromwire_tlv() accepts (minimal BigSize, ascending
 * types) are these bytes; a stream breaking those rules gets verdict 0 -- key33 + keystride*i the signer (33-byte compressed key:
 * the parity byte is dropped as :417-422 do), sig64 + 64*i the signature. */
int lamd_bolt12_check_signature_batch(lamd_ctx *ctx, size_t n, const uint8_t *tlvs, const uint64_t *off, const char *messagename,
				      const char *fieldname, const uint8_t *key33, size_t keystride, const uint8_t *sig64, uint8_t *ok);
/* the two hashes alone: merkle32 + 32*i = merkle_tlv(), sighash32 + 32*i = sighash_from_merkle() (either may be NULL), ok[i] = the
 * stream obeys the TLV rules and holds at least one non-signature field */
int lamd_bolt12_merkle_batch(lamd_ctx *ctx, size_t n, const uint8_t *tlvs, const uint64_t *off, const char *messagename,
			     const char *fieldname, uint8_t *merkle32, uint8_t *sighash32, uint8_t *ok);

/* n independent secp256k1_ecdsa_recoverable_signature_parse_compact() + secp256k1_ecdsa_recover() calls as made by
 * common/bolt11.c:1021-1046 (invoices without an `n` field) and lightningd/signmessage.c:193: sig64 = r||s, recid 0..3.
 * pub33[i] = the recovered key, compressed (what node_id_from_pubkey() stores), ok[i] = 1; where the library calls would
 * fail (r or s >= n or zero, recid > 3, recid & 2 with r >= p - n, no curve point with that x, infinity) ok[i] = 0 and the
 * key bytes are zero.  There is no low-S rule on this path, as in the reference. */
int lamd_ecdsa_recover_batch(lamd_ctx *ctx, size_t n, const uint8_t *hash32, const uint8_t *sig64, const uint8_t *recid,
			     uint8_t *pub33, uint8_t *ok);
int lamd_ecdsa_recover_batch_device(lamd_ctx *ctx, size_t n, const void *d_hash32, const void *d_sig64, const void *d_recid,
				    void *d_pub33, void *d_ok);

/* grind_htlc_tx_fee() (onchaind/onchaind.c:388-438): find the feerate whose fee makes `sig64` (one remote HTLC signature,
 * one key) verify.  For feerate = min_feerate..max_feerate: fee = feerate * weight / 1000 (amount_tx_fee), equal consecutive
 * fees are tried once, fees above input_sat end the search; candidate = the transaction with output 0 paying
 * input_sat - fee.  The caller hands the BIP143 preimage of the transaction as it stands (any output-0 amount; the 32
 * bytes of hashOutputs sit 40 bytes before its end, bitcoin/signature.c:120-151) and the serialised outputs that
 * hashOutputs covers (amount of output 0 first).  All hashing and curve work runs on the device; (r/s)*Q is computed once
 * and each candidate costs two small double-SHA256 and the G-table additions.  Returns 1 = found (*feerate, *fee = the
 * lowest matching feerate and its fee, exactly the pair the reference's ascending loop stops at), 0 = none, < 0 error. */
int lamd_grind_htlc_tx_fee(lamd_ctx *ctx, const uint8_t *preimage, size_t preimage_len, const uint8_t *outputs,
			   size_t outputs_len, uint64_t input_sat, uint64_t weight, uint32_t min_feerate,
			   uint32_t max_feerate, const uint8_t sig64[64], uint8_t sighash_type, int has_witness_script,
			   const uint8_t pubkey33[33], uint32_t *feerate, uint64_t *fee);

/* ---- public-key parsing: n independent pubkey_from_der() / pubkey_from_node_id() calls
 * (bitcoin/pubkey.c:14-24, common/node_id.c:21-27 -> secp256k1_ec_pubkey_parse; publen 33 or 65),
 * or secp256k1_xonly_pubkey_parse with publen 32 (bitcoin/signature.c:422).  ok[i] = validity,
 * out64 + 64*i = affine X||Y big-endian (unspecified when invalid).  out64 may be NULL. */
int lamd_pubkey_parse_batch(lamd_ctx *ctx, size_t n, const uint8_t *pub, size_t publen, size_t pubstride,
			    uint8_t *out64, uint8_t *ok);

/* ---- gossip: gossipd/sigcheck.c on raw wire messages, batched.
 * msgs: the messages back to back; off[i]..off[i+1] delimits message i (off has n+1 entries).
 * kind is inferred from the 2-byte type (256 channel_announcement, 257 node_announcement,
 * 258 channel_update).  For channel_update the signer is not in the message (the reference looks
 * it up in the gossmap, gossipd/gossmap_manage.c:920-922): node_ids holds one 33-byte id per
 * message (ignored for the other kinds; may be NULL if the batch has no channel_update).
 * verdict[i]: 0 = OK (reference returns NULL); k > 0 = first bad signature, numbered as the
 * reference's messages name them -- channel_announcement: 1 "Bad node_signature_1",
 * 2 "Bad node_signature_2", 3 "Bad bitcoin_signature_1", 4 "Bad bitcoin_signature_2"
 * (gossipd/sigcheck.c:78-113); others: 1 "Bad signature for" (:35-41, :144-161);
 * -1 = malformed: what fromwire_* would have rejected before sigcheck runs (truncated message,
 * compact signature with r or s >= n, wire/fromwire.c:196-198; invalid bitcoin_key,
 * bitcoin/pubkey.c:102-113). */
int lamd_sigcheck_gossip_batch(lamd_ctx *ctx, size_t n, const uint8_t *msgs, const uint64_t *off,
			       const uint8_t *node_ids33, int8_t *verdict);

/* Device-resident variant (asynchronous on the context's stream).  d_off: uint64[n+1] byte offsets;
 * d_rowbase: uint64[n+1], d_rowbase[i] = number of signatures in messages 0..i-1 (4 per
 * channel_announcement, 1 otherwise), rows = d_rowbase[n]. */
int lamd_sigcheck_gossip_batch_device(lamd_ctx *ctx, size_t n, const void *d_msgs, const void *d_off,
				      const void *d_node_ids33, const void *d_rowbase, size_t rows,
				      void *d_verdict);
/* The spans form of the device-resident variant: message i = d_msgs[d_start[i], d_start[i] + d_len[i]) (uint64[n] each) -- any selection of a
 * resident message blob, in any order, as ONE call: what a rank of a replay cut per message kind verifies (its range of the channel_announcements
 * and its range of the channel_updates; lightning_amd/sharding.py segment_bounds).  d_node_ids33, d_rowbase, rows, d_verdict index the SELECTION. */
int lamd_sigcheck_gossip_spans_device(lamd_ctx *ctx, size_t n, const void *d_msgs, const void *d_start, const void *d_len,
				      const void *d_node_ids33, const void *d_rowbase, size_t rows, void *d_verdict);

/* ---- streaming front end for callers that produce triples one at a time (channeld's
 * commitment_signed loop, channeld/channeld.c:2171,2215-2232; gossip ingest).  Triples are
 * appended to a pinned staging set; flush launches everything queued so far as one batch (asynchronous) and opens the
 * next set, so queueing continues while flushes are in flight: up to 9 flushes may be outstanding (LAMD_ERR_STATE beyond
 * that, until one is collected), successive flushes run on alternating lanes.  A flush's three host-to-device copies go down
 * a copy stream of their own in flush order, so with more flushes outstanding than lanes (4) the rows of the next flush are
 * already in HBM when its lane comes free.  A flush of <= 4096 rows (one commitment_signed) is ONE launch of the latency kernel over the
 * pinned staging rows themselves -- no copies at all; its verdicts arrive through lamd_poll / lamd_wait like any other flush's.  poll/wait return the verdicts of the
 * OLDEST outstanding flush, in submission (ticket) order. */
/* A ticket is the position of the triple's verdict in the vector its flush returns: tickets count 0, 1, 2, ... across
 * the kinds inside the open staging set and restart at 0 after every lamd_flush() (fewer than 2^30 triples per flush). */
int lamd_queue_ecdsa(lamd_ctx *ctx, const uint8_t hash32[32], const uint8_t sig64[64],
		     const uint8_t *pubkey, size_t publen); /* returns the ticket (>= 0) or an error */
int lamd_queue_schnorr(lamd_ctx *ctx, const uint8_t msg32[32], const uint8_t xonly32[32],
		       const uint8_t sig64[64]);
/* n triples at once (row strides 32, 64, pubstride): returns the first ticket, the others follow consecutively */
int lamd_queue_ecdsa_batch(lamd_ctx *ctx, size_t n, const uint8_t *hash32, const uint8_t *sig64, const uint8_t *pubkey,
			   size_t publen, size_t pubstride);
int lamd_queue_schnorr_batch(lamd_ctx *ctx, size_t n, const uint8_t *msg32, const uint8_t *xonly32, const uint8_t *sig64);
/* Zero-copy producer form: queues n triples (keylen 33 / 65: ECDSA with compressed / uncompressed keys; 32: BIP-340 with x-only keys) and
 * returns WHERE their rows live in the pinned staging set -- n rows of 32, 64 and keylen bytes, contiguous -- instead of copying them from
 * a caller-owned buffer.  The caller writes the rows before lamd_flush(); the pointers are valid until the next lamd_queue_* / lamd_flush call
 * on this context (a later push may move the set).  Returns the first ticket like the batch forms. */
int lamd_queue_reserve(lamd_ctx *ctx, size_t n, size_t keylen, uint8_t **hash32, uint8_t **sig64, uint8_t **key);
/* In-place form for a host that ALREADY holds the rows in pinned memory (lamd_served: its clients' shared blocks): the n triples (keys packed, publen
 * / 32 bytes each) get tickets like the batch forms, but their bytes are not copied -- they cross the bus from the caller's buffers when the set is
 * flushed.  The buffers must stay unchanged until the flush that carries the rows has been collected (lamd_poll / lamd_wait).  Batches of <= 4 096
 * rows (the latency kernel reads the staging rows themselves) and rows in memory that is NOT pinned end to end are copied as by lamd_queue_*_batch.
 * lamd_host_register() pins a range for every device (hipHostRegister, portable): register whole, page-aligned blocks, once, and keep them
 * registered while rows of theirs are queued.
 * Both may be called from any thread while another drives the context.  (Replaces nothing in the reference: the producer side of SURVEY.md 8(f)'s
 * sidecar, channeld/channeld.c:7063-7121 being one process per channel.) */
int lamd_queue_ecdsa_batch_inplace(lamd_ctx *ctx, size_t n, const uint8_t *hash32, const uint8_t *sig64, const uint8_t *pubkey, size_t publen);
int lamd_queue_schnorr_batch_inplace(lamd_ctx *ctx, size_t n, const uint8_t *msg32, const uint8_t *xonly32, const uint8_t *sig64);
int lamd_host_register(lamd_ctx *ctx, void *p, size_t bytes);
int lamd_host_unregister(lamd_ctx *ctx, void *p);
/* The NUMA node device `device` hangs on (sysfs), -1 when unknown: run the producer threads and allocate the buffers that feed a device there. */
int lamd_device_numa_node(int device);
int lamd_flush(lamd_ctx *ctx);
/* 1 = finished (ok[0..*n) filled, tickets in submission order), 0 = still running, < 0 error */
int lamd_poll(lamd_ctx *ctx, uint8_t *ok, size_t cap, size_t *n);
int lamd_wait(lamd_ctx *ctx, uint8_t *ok, size_t cap, size_t *n);

/* ---- several MI355X behind one host process (the sidecar that serves a node's channeld / gossipd processes owns every GPU).
 * lamd_multi_init(): one engine context and one host thread per device (devices == NULL: 0 .. n_devices-1), then an RCCL communicator over them
 * (librccl.so is dlopen()ed here, never by a single-GPU process).  A call cuts its rows into n_devices contiguous ranges on GROUP boundaries
 * (lamd_shard_bounds: a channel_announcement's four signatures / a commitment's 1 + 483 rows stay on one device, so a per-key table is built
 * once), copies and verifies every range on its device (the ranges' host-to-device copies run side by side, piece k+1 of a range under the
 * kernels of piece k), all-gathers the verdict bytes ON THE DEVICES (ncclAllGather over xGMI, shards padded to the largest: every device
 * ends with the whole vector) and copies the vector to the host once, from the first device.  A gossip batch made of a few runs of one kind
 * (a replay: its channel_announcements, then its channel_updates) is cut RUN BY RUN -- device i takes range i of every run, in one engine call --
 * so that every device holds the same mix; the verdicts still come back in the caller's order.  Verdicts are exactly those of the
 * single-device calls (lamd_verify_ecdsa_batch, lamd_verify_schnorr_batch, lamd_sigcheck_gossip_batch); with n_devices == 1 the same path
 * runs on one GPU, collective included.  One call at a time per lamd_multi (internally locked); < 0 = LAMD_ERR_*, lamd_multi_last_error(). */
typedef struct lamd_multi lamd_multi;
int lamd_multi_init(lamd_multi **m, const int *devices, int n_devices);
void lamd_multi_shutdown(lamd_multi *m);
const char *lamd_multi_last_error(const lamd_multi *m);
int lamd_multi_devices(const lamd_multi *m);
lamd_ctx *lamd_multi_ctx(lamd_multi *m, int i);   /* device i's engine context (statistics, lamd_get_info); not for concurrent verification calls */
/* group_rows: rows that must stay together (1 = any row boundary; 484 = one commitment_signed; 4 = the rows of a channel_announcement) */
int lamd_multi_verify_ecdsa_batch(lamd_multi *m, size_t n, const uint8_t *hash32, const uint8_t *sig64, const uint8_t *pub, size_t publen,
				  size_t pubstride, size_t group_rows, uint8_t *ok);
int lamd_multi_verify_schnorr_batch(lamd_multi *m, size_t n, const uint8_t *msg32, const uint8_t *xonly32, const uint8_t *sig64, size_t group_rows,
				    uint8_t *ok);
/* raw wire messages as for lamd_sigcheck_gossip_batch: sharded by MESSAGE, balanced by signatures (4 per channel_announcement) */
int lamd_multi_sigcheck_gossip_batch(lamd_multi *m, size_t n, const uint8_t *msgs, const uint64_t *off, const uint8_t *node_ids33, int8_t *verdict);
/* The partition itself: n_groups groups of group_rows[g] rows each (NULL: one row each) into n_shards contiguous ranges of whole groups,
 * balanced by rows; bounds_groups[0..n_shards] (and, if not NULL, bounds_rows[0..n_shards]) receive the cut points.  Shards may be empty. */
int lamd_shard_bounds(size_t n_groups, const uint32_t *group_rows, int n_shards, size_t *bounds_groups, size_t *bounds_rows);

/* Synthetic signed test traffic (the signer kernels tests/ and bench.py use) is NOT in this library: see
 * include/lightning_amd_testgen.h -> liblightning_amd_testgen.so. */

/* Diagnostics (device self-test, arithmetic fuzzers, work-buffer peeks) live in lightning_amd_debug.h: they are exported by the
 * same library but are not part of the drop-in boundary. */

/* ---- introspection for benchmarks / tests */
typedef struct {
	int device;
	int compute_units;
	char arch[64];
	size_t gtable_bytes;
	double last_kernel_ms[4]; /* prep, keys (+ key tables), ecmult, BIP-340 parity stage of the last launch sequence when timing is on */
	/* counts of the last chunk that took the keyed path (read back asynchronously: exact after lamd_synchronize()) */
	size_t last_unique_keys;  /* distinct public keys among the rows the cache did not know */
	size_t last_hot_rows;     /* rows verified against per-key comb tables (the rest took the per-signature ladder) */
	int last_keyed;           /* 0: ladder only; else the largest comb used (7 or 10 teeth) */
	int last_mode;            /* 0: the last chunk was ECDSA, 1: BIP-340 */
	int lanes;                /* number of lanes (LAMD_LANES, default 6; 1 = strictly one stream) */
	size_t last_cache_hits;   /* rows whose key already had a table in the cache */
	size_t last_cold_rows;    /* rows that took the ladder */
	size_t last_new_tables;   /* comb tables built by that chunk */
	size_t last_suspect_rows; /* rows re-decided by the complete addition formulas (crafted scalars / result at infinity) */
	int cache_enabled;        /* LAMD_CACHE (default 1): comb tables persist across calls */
	size_t cache_entries, cache_capacity; /* keys with a cached table (as last read back) / LAMD_CACHE_KEYS + LAMD_CACHE_KEYS10 */
	size_t cache_resets;      /* how often the bounded cache was emptied because it filled up */
	/* the dominant kernel by itself: HIP events right around every large table-driven ecmult launch ([0] ECDSA, [1] BIP-340)
	 * of this lane since lamd_set_timing(ctx, 1); summed by lamd_synchronize() */
	double keyed_ecmult_ms_sum[2];
	size_t keyed_ecmult_launches[2];
	int hw_queues_env;        /* GPU_MAX_HW_QUEUES as lamd_init() found it (0 = unset: the runtime's default of 4; < 16 costs overlap between the lanes) */
	int queue_sets;           /* staging sets of the streaming queue = the most flushes that may be outstanding */
} lamd_info;
int lamd_get_info(lamd_ctx *ctx, lamd_info *info);
int lamd_get_lane_info(lamd_ctx *ctx, int lane, lamd_info *info); /* the last call that ran on lane 0 .. lanes-1 */
int lamd_set_timing(lamd_ctx *ctx, int enable); /* record HIP events around each kernel group; (re)starts the keyed_ecmult_* sums */
/* Empties the key-table cache (drains the device first).  The cache is bounded (LAMD_CACHE_KEYS 7-tooth tables, 2^20 by
 * default = 6.3 GB of HBM; LAMD_CACHE_KEYS10 10-tooth tables, 2^16 = 3.2 GB) and empties itself when it fills up;
 * benchmarks call this to measure the cold path. */
int lamd_cache_clear(lamd_ctx *ctx);
/* Scheduling of the large (>= 65 536 rows) table-driven ecmult launches of successive calls.  0 (default; LAMD_ECMULT_CHAIN): they overlap
 * whenever the lanes let them -- highest throughput (230 against 224 M verifies/s in the pipelined loop), each launch stretched by its
 * neighbour (6.9 ms against 3.1 ms by itself).  1: a launch waits for the one submitted before it -- the kernel saturates the VALU issue
 * port by itself, so what runs under it is only the other lanes' front end (3.9 ms in the loop). */
int lamd_set_ecmult_chain(lamd_ctx *ctx, int enable);
/* Rows per launch sequence ("chunk") of the device-pointer and streaming calls that follow: a call of more rows is cut into chunks that alternate
 * between the call's lane and its peer lane, so that chunk k+1's front end (key de-duplication, tables, row lists) runs under chunk k's ecmult
 * launch.  Default 2^22 (rows == 0 restores it; clamped to [4096, 2^22]).  A caller whose call is the only thing the device has to do -- one rank's
 * shard of a strong-scaling job -- gains by cutting it in two; a caller that keeps several calls in flight leaves it alone (the calls overlap). */
int lamd_set_chunk_rows(lamd_ctx *ctx, size_t rows);

#ifdef __cplusplus
}
#endif
#endif /* LIGHTNING_AMD_H */

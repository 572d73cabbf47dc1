/* lightning_amd -- the shared-service front: ONE process (lamd_served) owns the engine context of a GPU, many
 * client processes use it through liblightning_amd_client.so.
 *
 * Why: Core Lightning runs one channeld per channel (channeld/channeld.c:7019-7129) next to gossipd, lightningd and the
 * plugins, every one of them a single-threaded process that calls check_signed_hash() / check_tx_sig() inline
 * (SURVEY.md 8(b) "callers").  An engine context costs 11 GiB of HBM (the static G table) plus its table pools and 0.36 s to
 * create -- one per daemon process does not scale.  The server holds the one context; a client process holds a socket and a
 * shared-memory block.  Requests that are waiting at the same moment are MERGED: the ECDSA rows of every client go to the device
 * as one lamd_verify_ecdsa_batch call, the BIP-340 rows as one lamd_verify_schnorr_batch call, the commitment_signed validations and
 * check_tx_sig batches (33-byte keys) as one lamd_check_tx_sig_tx_batch call (eight channelds validating a commitment_signed each =
 * one hashing launch + one 3872-row verification launch); the verdicts are scattered back per client.  That is the batching the north star asks the C host for, across process boundaries.
 *
 * liblightning_amd_client.so exports the entry points of include/lightning_amd.h that the mirror (include/cln_shim.h) and the
 * gossip ingest use, with the SAME prototypes and meaning:
 *     lamd_init / lamd_shutdown / lamd_last_error / lamd_version
 *     lamd_verify_ecdsa_batch, lamd_verify_schnorr_batch, lamd_check_signed_hash, lamd_check_signed_hash_nodeid, lamd_check_schnorr_sig
 *     lamd_pubkey_parse_batch, lamd_sigcheck_gossip_batch, lamd_check_tx_sig_tx_batch, lamd_check_commitment_signed
 *     lamd_bolt12_check_signature_batch, lamd_bolt12_merkle_batch, lamd_ecdsa_recover_batch, lamd_grind_htlc_tx_fee
 * so liblightning_amd_cln_client.so -- the mirror linked against the client instead of the engine -- gives a daemon the reference's
 * own prototypes over the service with no source change.  lamd_init(&ctx, device) connects (LAMD_SERVED_SOCKET, else
 * $XDG_RUNTIME_DIR/lamd_served.sock; `device` is ignored: the server chose it) and fails with LAMD_ERR_NO_DEVICE when no server answers:
 * the mirror then fails closed, as it does without a GPU.  No verification happens in the client: it frames bytes.
 *
 * STREAMING (round 6; channeld keeps commitment_signed validations in flight, channeld/channeld.c:7063-7121): the client library also exports
 *     lamd_queue_ecdsa, lamd_queue_schnorr, lamd_queue_ecdsa_batch, lamd_queue_schnorr_batch, lamd_queue_reserve, lamd_flush, lamd_poll, lamd_wait
 * with the engine's meaning (include/lightning_amd.h "streaming"): triples are queued locally, lamd_flush() hands the set to the server as ONE
 * asynchronous LAMD_SRV_OP_FLUSH request in a shared block of its own (up to LAMD_SRV_FLUSH_SLOTS flushes outstanding per connection) and returns
 * at once, lamd_poll / lamd_wait return the verdicts of the OLDEST outstanding flush.  The server queues the rows of every client's waiting
 * flushes into the engine's own pinned staging set, flushes it as one engine batch (up to eight in flight per device: copies and kernels of
 * successive flushes overlap as they do for an in-process producer) and scatters the verdict vector back per client when the engine reports it.
 *
 * SEVERAL GPUs (round 6): `lamd_served --devices 0,1,..` holds one engine context and one engine thread per device.  A request goes to the
 * device its FIRST KEY hashes to (funding key of a commitment, first row's key of a batch or flush), so the rows of one channel, one peer,
 * one gossip signer keep meeting the same device's key-table cache; requests without a key (statistics) go to device 0.
 *
 * Wire format:
 *   connect -> client sends a memfd (SCM_RIGHTS) + its size (scalar[0]) and slot (scalar[1]) in an LAMD_SRV_OP_SHM request; both sides mmap it.
 *              Slot 0 is the block of the synchronous calls (one in flight per connection), slots 1..LAMD_SRV_FLUSH_SLOTS carry one flush each.
 *   request  = struct lamd_srv_req over the socket (`slot` = the block its sections lie in); the argument arrays ("sections") lie in that
 *              block back to back, each aligned to 16 bytes, in the order the operation defines (lamd_served.cpp, op table)
 *   reply    = struct lamd_srv_rep over the socket; output sections follow the input sections in the block.  seq = 0 for a synchronous
 *              request, the flush's sequence number (scalar[0] of its request, >= 1) for LAMD_SRV_OP_FLUSH: replies to flushes arrive whenever
 *              the engine is done, also between a synchronous request and its reply.
 * Both structs are plain little-endian host structs: client and server are processes of one machine.
 *
 * Trust: the socket path is LAMD_SERVED_SOCKET, else $XDG_RUNTIME_DIR/lamd_served.sock -- there is NO default under /tmp: whoever binds a
 * world-writable path first would answer every check with "good".  The server binds under umask 0177 (mode 0600 from the start), refuses to replace a
 * path it does not own, and drops connections whose SO_PEERCRED uid is not its own (LAMD_SERVED_UID names another); the client checks the same of
 * the server and of the socket file's owner before it sends a row (lamd_init fails with LAMD_ERR_NO_DEVICE otherwise).  The server validates every
 * length of a request and works on ITS OWN COPY of every offset array (the block stays writable by its client while the request runs): a hostile
 * client of the same user can get wrong answers for itself, not crash the server or read another client's rows (one block per connection); an
 * engine error inside a merged call is retried request by request, so that only the offending client sees it.
 */
#ifndef LIGHTNING_AMD_SERVED_H
#define LIGHTNING_AMD_SERVED_H
#include <stdint.h>

#define LAMD_SRV_MAGIC 0x4C414D44u /* "LAMD" */
#define LAMD_SRV_SOCKET_NAME "lamd_served.sock" /* under $XDG_RUNTIME_DIR when LAMD_SERVED_SOCKET is unset */
#define LAMD_SRV_MAX_SECTIONS 20
#define LAMD_SRV_FLUSH_SLOTS 8   /* flushes a connection may have outstanding (the engine itself allows nine) */
#define LAMD_SRV_MAX_DEVICES 8

enum lamd_srv_op {
	LAMD_SRV_OP_SHM = 1,          /* (re)attach the shared block: fd by SCM_RIGHTS, scalar[0] = its size */
	LAMD_SRV_OP_ECDSA = 2,        /* lamd_verify_ecdsa_batch: n, scalar[0] = publen; in: hash32[n] sig64[n] pub[n*publen]; out: ok[n].  MERGED across clients */
	LAMD_SRV_OP_SCHNORR = 3,      /* lamd_verify_schnorr_batch: in: msg32 xonly32 sig64; out: ok[n].  MERGED across clients */
	LAMD_SRV_OP_PUBKEY_PARSE = 4, /* lamd_pubkey_parse_batch: scalar[0] = publen; in: pub; out: xy64[n] ok[n] */
	LAMD_SRV_OP_GOSSIP = 5,       /* lamd_sigcheck_gossip_batch: scalar[0] = node ids present; in: msgs off[n+1] ids33[n]; out: verdict[n] */
	LAMD_SRV_OP_TXSIG_TX = 6,     /* lamd_check_tx_sig_tx_batch: scalar[0] = publen; in: the fifteen arrays in prototype order; out: ok[n].  MERGED (publen 33) */
	LAMD_SRV_OP_COMMITMENT = 7,   /* lamd_check_commitment_signed: n = 1 + n_htlc rows as TXSIG_TX arrays + keys + sigs; out: first_bad (8) ok[n].  MERGED with 6 */
	LAMD_SRV_OP_BOLT12_CHECK = 8, /* lamd_bolt12_check_signature_batch: in: tlvs off[n+1] messagename fieldname key33[n] sig64[n]; out: ok[n] */
	LAMD_SRV_OP_BOLT12_MERKLE = 9,/* lamd_bolt12_merkle_batch: scalar[0] = sighash wanted; in: tlvs off messagename fieldname; out: merkle32[n] sighash32[n] ok[n] */
	LAMD_SRV_OP_RECOVER = 10,     /* lamd_ecdsa_recover_batch: in: hash32 sig64 recid[n]; out: pub33[n] ok[n] */
	LAMD_SRV_OP_GRIND = 11,       /* lamd_grind_htlc_tx_fee: scalars = input_sat weight min max sighash_type has_witness; in: preimage outputs sig64 pub33; out: rate(4) fee(8) found(4) */
	LAMD_SRV_OP_STATS = 12,       /* out: struct lamd_srv_stats */
	LAMD_SRV_OP_FLUSH = 13        /* ASYNCHRONOUS: one flushed staging set.  n rows, scalar[0] = sequence number (>= 1); in: runs[] (u64: keylen << 32 | rows; keylen 33 / 65
	                               * = ECDSA, 32 = BIP-340, in ticket order) hash32[n] sig64[n] keys (packed run by run); out: ok[n] in ticket order */
};

struct lamd_srv_req {
	uint32_t magic, op;
	uint64_t n;
	uint64_t scalar[6];
	uint32_t n_sections, slot; /* slot: which shared block of the connection holds the sections (0 = synchronous calls) */
	uint64_t section_len[LAMD_SRV_MAX_SECTIONS];
};
struct lamd_srv_rep {
	uint32_t magic;
	int32_t rc;          /* the engine call's return value (LAMD_OK, 0/1 for the single-item veneers, < 0 errors) */
	uint64_t out_offset; /* where the output sections start in the shared block */
	uint64_t seq;        /* 0: reply to a synchronous request; else the sequence number of the flush this answers */
	char err[208];       /* lamd_last_error() of the server's context when rc < 0 */
};
struct lamd_srv_stats {
	uint64_t requests, engine_calls, merged_requests, merged_rows, largest_merge_requests, clients_now, clients_total;
	uint64_t flushes, flush_rows, engine_flushes, largest_engine_flush_requests; /* client flushes, their rows, engine flushes they were merged into */
	uint64_t devices, rows_by_device[LAMD_SRV_MAX_DEVICES];                      /* rows verified per device (key affinity) */
	uint64_t flush_rows_in_place, pinned_blocks_now; /* flush rows that crossed the bus from the clients' own (pinned) blocks; blocks pinned right now */
};
#endif

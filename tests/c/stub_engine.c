/* A stand-in "engine" for lamd_served on a machine without a GPU (tests/test_served.py): the entry points the server binds, with verdicts that
 * are a fixed function of the input bytes, so that the test can tell whether every client got the verdicts of ITS rows back after the server
 * merged the requests of several clients into one call.  Test infrastructure: nothing here verifies a signature. */
#include <stdlib.h>
#include <string.h>

#include "lightning_amd.h"

#include <pthread.h>

#define STUB_SETS 16
#define STUB_DEFERRED 64
/* rows queued in place: their verdicts are computed when the flush is COLLECTED, from the caller's buffers -- a host that lets go of (or rewrites) a
 * buffer before it has collected the flush gets the wrong verdicts here */
struct stub_deferred { size_t pos, n, publen; const uint8_t *a, *b, *c; };
struct stub_set { uint8_t *ok; size_t n, cap; int polled; struct stub_deferred late[STUB_DEFERRED]; int n_late; };
struct lamd_ctx { int device; unsigned long calls, rows, largest, inplace_rows; char err[64]; struct stub_set open, closed[STUB_SETS]; int head, count; };

int lamd_init(lamd_ctx **ctx, int device) {
	*ctx = calloc(1, sizeof **ctx);
	(*ctx)->device = device;
	if (device == 99) { strcpy((*ctx)->err, "stub: no such device"); return LAMD_ERR_NO_DEVICE; }
	return LAMD_OK;
}
void lamd_shutdown(lamd_ctx *ctx) { free(ctx); }
const char *lamd_last_error(const lamd_ctx *ctx) { return ctx->err; }
static void note(lamd_ctx *c, size_t n) { c->calls++; c->rows += n; if (n > c->largest) c->largest = n; }

int lamd_verify_ecdsa_batch(lamd_ctx *ctx, size_t n, const uint8_t *h, const uint8_t *s, const uint8_t *p, size_t publen, size_t stride, uint8_t *ok) {
	note(ctx, n);
	for (size_t i = 0; i < n; i++) ok[i] = (h[32 * i] ^ s[64 * i + 63] ^ p[stride * i + publen - 1]) & 1;
	if (n && h[0] == 0xEE && h[1] == 0xEE) { strcpy(ctx->err, "stub: poisoned batch"); return LAMD_ERR_HIP; }
	return LAMD_OK;
}
int lamd_verify_schnorr_batch(lamd_ctx *ctx, size_t n, const uint8_t *m, const uint8_t *x, const uint8_t *s, uint8_t *ok) {
	note(ctx, n);
	for (size_t i = 0; i < n; i++) ok[i] = (m[32 * i + 1] ^ x[32 * i + 2] ^ s[64 * i + 3]) & 1;
	return LAMD_OK;
}
int lamd_pubkey_parse_batch(lamd_ctx *ctx, size_t n, const uint8_t *pub, size_t publen, size_t stride, uint8_t *out64, uint8_t *ok) {
	note(ctx, n);
	for (size_t i = 0; i < n; i++) {
		ok[i] = pub[stride * i] == 2 || pub[stride * i] == 3 || pub[stride * i] == 4;
		if (out64) { memset(out64 + 64 * i, 0, 64); memcpy(out64 + 64 * i, pub + stride * i + 1, 32); out64[64 * i + 63] = pub[stride * i] & 1; }
	}
	(void)publen;
	return LAMD_OK;
}
int lamd_sigcheck_gossip_batch(lamd_ctx *ctx, size_t n, const uint8_t *msgs, const uint64_t *off, const uint8_t *ids, int8_t *verdict) {
	note(ctx, n);
	for (size_t i = 0; i < n; i++) {
		const size_t len = off[i + 1] - off[i];
		verdict[i] = len < 3 ? -1 : (int8_t)((msgs[off[i] + 2] & 3) + (ids ? ids[33 * i] & 1 : 0));
	}
	return LAMD_OK;
}
static uint8_t tx_row(size_t i, const uint32_t *version, const uint8_t *inputs40, const uint64_t *in_off, const uint64_t *amount, const uint8_t *outputs, const uint64_t *out_off,
		      const uint8_t *scripts, const uint64_t *sc_off, uint8_t type, const uint8_t *sig, const uint8_t *pub) {
	uint8_t v = (uint8_t)(version[i] ^ amount[i] ^ type ^ sig[0] ^ pub[1]);
	if (in_off[i + 1] > in_off[i]) v ^= inputs40[40 * in_off[i]];
	if (out_off[i + 1] > out_off[i]) v ^= outputs[out_off[i + 1] - 1];
	if (sc_off[i + 1] > sc_off[i]) v ^= scripts[sc_off[i]];
	return v & 1;
}
int lamd_check_tx_sig_tx_batch(lamd_ctx *ctx, size_t n, const uint32_t *version, const uint32_t *locktime, const uint8_t *inputs40, const uint64_t *in_off,
			       const uint32_t *input_num, const uint64_t *amount_sat, const uint8_t *outputs, const uint64_t *out_off, const uint32_t *n_outputs,
			       const uint8_t *scripts, const uint64_t *script_off, const uint8_t *sighash_type, const uint8_t *has_witness_script,
			       const uint8_t *sig64, const uint8_t *pub, size_t publen, size_t pubstride, uint8_t *ok) {
	note(ctx, n);
	(void)locktime; (void)input_num; (void)n_outputs; (void)has_witness_script; (void)publen;
	for (size_t i = 0; i < n; i++)
		ok[i] = tx_row(i, version, inputs40, in_off, amount_sat, outputs, out_off, scripts, script_off, sighash_type[i], sig64 + 64 * i, pub + pubstride * i);
	return LAMD_OK;
}
int lamd_check_commitment_signed(lamd_ctx *ctx, const lamd_tx_template *commit_tx, const uint8_t remote_funding33[33], const uint8_t commit_sig64[64],
				 uint8_t commit_sighash_type, size_t n_htlc, const lamd_tx_template *htlc_txs, const uint8_t remote_htlckey33[33],
				 const uint8_t *htlc_sigs64, const uint8_t *htlc_sighash_types, int64_t *first_bad, uint8_t *ok_rows) {
	note(ctx, 1 + n_htlc);
	*first_bad = -1;
	for (size_t i = 0; i <= n_htlc; i++) {
		const lamd_tx_template *t = i ? &htlc_txs[i - 1] : commit_tx;
		const uint8_t *sig = i ? htlc_sigs64 + 64 * (i - 1) : commit_sig64, *key = i ? remote_htlckey33 : remote_funding33;
		const uint8_t type = i ? htlc_sighash_types[i - 1] : commit_sighash_type;
		uint8_t v = (uint8_t)(t->version ^ t->amount_sat ^ type ^ sig[0] ^ key[1]);
		if (t->n_inputs) v ^= t->inputs40[0];
		if (t->outputs_len) v ^= t->outputs[t->outputs_len - 1];
		if (t->script_len) v ^= t->script[0];
		v &= 1;
		if (ok_rows) ok_rows[i] = v;
		if (!v && *first_bad < 0) *first_bad = (int64_t)i;
	}
	return LAMD_OK;
}
int lamd_bolt12_check_signature_batch(lamd_ctx *ctx, size_t n, const uint8_t *tlvs, const uint64_t *off, const char *messagename, const char *fieldname,
				      const uint8_t *key33, size_t keystride, const uint8_t *sig64, uint8_t *ok) {
	note(ctx, n);
	for (size_t i = 0; i < n; i++) ok[i] = (uint8_t)((tlvs[off[i]] ^ messagename[0] ^ fieldname[0] ^ key33[keystride * i + 5] ^ sig64[64 * i]) & 1);
	return LAMD_OK;
}
int lamd_bolt12_merkle_batch(lamd_ctx *ctx, size_t n, const uint8_t *tlvs, const uint64_t *off, const char *messagename, const char *fieldname, uint8_t *merkle32,
			     uint8_t *sighash32, uint8_t *ok) {
	note(ctx, n);
	for (size_t i = 0; i < n; i++) {
		memset(merkle32 + 32 * i, tlvs[off[i]], 32);
		if (sighash32) memset(sighash32 + 32 * i, (uint8_t)(messagename[0] + fieldname[0]), 32);
		ok[i] = 1;
	}
	return LAMD_OK;
}
int lamd_ecdsa_recover_batch(lamd_ctx *ctx, size_t n, const uint8_t *hash32, const uint8_t *sig64, const uint8_t *recid, uint8_t *pub33, uint8_t *ok) {
	note(ctx, n);
	for (size_t i = 0; i < n; i++) {
		pub33[33 * i] = 2 + (recid[i] & 1);
		for (int b = 0; b < 32; b++) pub33[33 * i + 1 + b] = hash32[32 * i + b] ^ sig64[64 * i + b];
		ok[i] = recid[i] < 4;
	}
	return LAMD_OK;
}
int lamd_grind_htlc_tx_fee(lamd_ctx *ctx, const uint8_t *preimage, size_t preimage_len, const uint8_t *outputs, size_t outputs_len, uint64_t input_sat, uint64_t weight,
			   uint32_t min_feerate, uint32_t max_feerate, const uint8_t sig64[64], uint8_t sighash_type, int has_witness_script, const uint8_t pubkey33[33],
			   uint32_t *feerate, uint64_t *fee) {
	note(ctx, 1);
	(void)preimage; (void)outputs; (void)sig64; (void)pubkey33; (void)sighash_type;
	if (!has_witness_script || min_feerate > max_feerate) return 0;
	*feerate = min_feerate + (uint32_t)((preimage_len + outputs_len) % (max_feerate - min_feerate + 1));
	*fee = (uint64_t)*feerate * weight / 1000 + input_sat % 7;
	return 1;
}

/* ---- the streaming queue (include/lightning_amd.h "streaming"): the open set collects verdict bytes in ticket order, lamd_flush closes it,
 * lamd_poll / lamd_wait hand the OLDEST closed set back.  Verdicts are the functions of the synchronous calls above; lamd_poll reports the
 * oldest flush "still running" once (the server must come back for it), as a busy engine would. */
#define g_open (ctx->open)
#define g_closed (ctx->closed)
#define g_head (ctx->head)
#define g_count (ctx->count)
static void push_ok(lamd_ctx *ctx, uint8_t v) {
	if (g_open.n == g_open.cap) { g_open.cap = g_open.cap ? 2 * g_open.cap : 1024; g_open.ok = realloc(g_open.ok, g_open.cap); }
	g_open.ok[g_open.n++] = v;
}
int lamd_queue_ecdsa_batch(lamd_ctx *ctx, size_t n, const uint8_t *h, const uint8_t *s, const uint8_t *p, size_t publen, size_t stride) {
	const int first = (int)g_open.n;
	if (n && h[0] == 0xEE && h[1] == 0xEE) { strcpy(ctx->err, "stub: poisoned batch"); return LAMD_ERR_HIP; }
	for (size_t i = 0; i < n; i++) push_ok(ctx, (h[32 * i] ^ s[64 * i + 63] ^ p[stride * i + publen - 1]) & 1);
	ctx->rows += n;
	return first;
}
int lamd_queue_schnorr_batch(lamd_ctx *ctx, size_t n, const uint8_t *m, const uint8_t *x, const uint8_t *s) {
	const int first = (int)g_open.n;
	for (size_t i = 0; i < n; i++) push_ok(ctx, (m[32 * i + 1] ^ x[32 * i + 2] ^ s[64 * i + 3]) & 1);
	ctx->rows += n;
	return first;
}
/* the registered ranges are the process's, as hipHostRegister(portable)'s are; rows queued in place must lie inside one (the stub is stricter than the
 * engine, which accepts unpinned memory and is slow on it: the server under test must register what it queues in place).  STUB_REFUSE_REGISTER=1: the
 * runtime refuses every range, as it may for a mapping it cannot pin -- the server must fall back to the copying form. */
static pthread_mutex_t reg_mu = PTHREAD_MUTEX_INITIALIZER;
static struct { const uint8_t *p; size_t bytes; } reg[256];
static int n_reg;
int lamd_host_register(lamd_ctx *ctx, void *p, size_t bytes) {
	const char *refuse = getenv("STUB_REFUSE_REGISTER");
	if (!ctx || !p || !bytes) return LAMD_ERR_ARG;
	if (refuse && refuse[0] == '1') return LAMD_ERR_HIP;
	pthread_mutex_lock(&reg_mu);
	int rc = LAMD_ERR_HIP;
	if (n_reg < 256) { reg[n_reg].p = p; reg[n_reg++].bytes = bytes; rc = LAMD_OK; }
	pthread_mutex_unlock(&reg_mu);
	return rc;
}
int lamd_host_unregister(lamd_ctx *ctx, void *p) {
	int rc = LAMD_ERR_HIP;
	if (!ctx || !p) return LAMD_ERR_ARG;
	pthread_mutex_lock(&reg_mu);
	for (int i = 0; i < n_reg; i++)
		if (reg[i].p == p) { reg[i] = reg[--n_reg]; rc = LAMD_OK; break; }
	pthread_mutex_unlock(&reg_mu);
	return rc;
}
static int registered(const uint8_t *p, size_t bytes) {
	int ok = 0;
	pthread_mutex_lock(&reg_mu);
	for (int i = 0; i < n_reg && !ok; i++) ok = p >= reg[i].p && p + bytes <= reg[i].p + reg[i].bytes;
	pthread_mutex_unlock(&reg_mu);
	return ok;
}
static int push_late(lamd_ctx *ctx, size_t n, const uint8_t *a, const uint8_t *b, const uint8_t *c, size_t alen, size_t blen, size_t clen, size_t publen) {
	const int first = (int)g_open.n;
	if (!registered(a, alen * n) || !registered(b, blen * n) || !registered(c, clen * n)) { strcpy(ctx->err, "stub: in-place rows outside registered memory"); return LAMD_ERR_ARG; }
	if (g_open.n_late == STUB_DEFERRED) { strcpy(ctx->err, "stub: too many in-place batches in one set"); return LAMD_ERR_STATE; }
	g_open.late[g_open.n_late++] = (struct stub_deferred){g_open.n, n, publen, a, b, c};
	for (size_t i = 0; i < n; i++) push_ok(ctx, 0xAA);
	ctx->rows += n;
	ctx->inplace_rows += n;
	return first;
}
int lamd_queue_ecdsa_batch_inplace(lamd_ctx *ctx, size_t n, const uint8_t *h, const uint8_t *s, const uint8_t *p, size_t publen) {
	if (n && h[0] == 0xEE && h[1] == 0xEE) { strcpy(ctx->err, "stub: poisoned batch"); return LAMD_ERR_HIP; }
	return push_late(ctx, n, h, s, p, 32, 64, publen, publen);
}
int lamd_queue_schnorr_batch_inplace(lamd_ctx *ctx, size_t n, const uint8_t *m, const uint8_t *x, const uint8_t *s) {
	return push_late(ctx, n, m, x, s, 32, 32, 64, 0);
}
int lamd_device_numa_node(int device) { return device == 2 ? -1 : 0; }   /* (device 2: a platform that does not say) */
int lamd_flush(lamd_ctx *ctx) {
	if (!g_open.n) return LAMD_OK;
	if (g_count == STUB_SETS) { strcpy(ctx->err, "stub: too many flushes outstanding"); return LAMD_ERR_STATE; }
	g_closed[(g_head + g_count++) % STUB_SETS] = g_open;
	memset(&g_open, 0, sizeof g_open);
	note(ctx, 0);
	return LAMD_OK;
}
static int take(lamd_ctx *ctx, uint8_t *ok, size_t cap, size_t *n) {
	if (g_closed[g_head].n > cap) return LAMD_ERR_ARG;
	for (int k = 0; k < g_closed[g_head].n_late; k++) {
		const struct stub_deferred *d = &g_closed[g_head].late[k];
		for (size_t i = 0; i < d->n; i++)
			g_closed[g_head].ok[d->pos + i] = d->publen ? (d->a[32 * i] ^ d->b[64 * i + 63] ^ d->c[d->publen * i + d->publen - 1]) & 1
							       : (d->a[32 * i + 1] ^ d->b[32 * i + 2] ^ d->c[64 * i + 3]) & 1;
	}
	memcpy(ok, g_closed[g_head].ok, g_closed[g_head].n);
	*n = g_closed[g_head].n;
	free(g_closed[g_head].ok);
	memset(&g_closed[g_head], 0, sizeof g_closed[0]);
	g_head = (g_head + 1) % STUB_SETS;
	g_count--;
	return 1;
}
int lamd_poll(lamd_ctx *ctx, uint8_t *ok, size_t cap, size_t *n) {
	if (!g_count) return LAMD_ERR_STATE;
	if (!g_closed[g_head].polled++) return 0;
	return take(ctx, ok, cap, n);
}
int lamd_wait(lamd_ctx *ctx, uint8_t *ok, size_t cap, size_t *n) {
	if (!g_count) return LAMD_ERR_STATE;
	return take(ctx, ok, cap, n);
}

/* Link-compatibility check of the mirror: the two sigcheck_channel_announcement() calls of the reference's own unit test
 * (gossipd/test/run-check_channel_announcement.c:77-85 and :100-108) compiled against include/cln_shim.h UNCHANGED -- same
 * argument list, the message a tal array whose length travels with the pointer, the error string the reference's.
 * Everything around the two calls (hex decoding, the fromwire/towire of a channel_announcement, wire/peer_wire.csv:344-356)
 * is this test's own scaffolding with the reference's names, so that the call statements read as they do there.
 *
 *     run_check_channel_announcement <hex of the 435-byte message at run-check_channel_announcement.c:62>
 * prints the two error strings; exit 0 iff they name node_signature_1 and node_signature_2 as the reference asserts (:84-85, :107-108). */
#include <assert.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "cln_shim.h"

struct bitcoin_blkid { u8 id[32]; };
struct short_channel_id { uint64_t u64; };
static const tal_t *tmpctx = NULL;

static u8 *tal_hexdata(const tal_t *ctx, const char *str, size_t slen) {
	u8 *raw = malloc(slen / 2 + 1), *ret;
	for (size_t i = 0; i + 1 < slen; i += 2) {
		unsigned v;
		if (sscanf(str + i, "%2x", &v) != 1) abort();
		raw[i / 2] = (u8)v;
	}
	ret = shim_tal_dup(ctx, raw, slen / 2);
	free(raw);
	return ret;
}

/* channel_announcement: type(2) | 4 x signature(64) | flen(2) | features | chain_hash(32) | scid(8) | 2 x node_id(33) | 2 x bitcoin_key(33) */
static bool fromwire_channel_announcement(const tal_t *ctx, const void *p, secp256k1_ecdsa_signature *node_signature_1,
					  secp256k1_ecdsa_signature *node_signature_2, secp256k1_ecdsa_signature *bitcoin_signature_1,
					  secp256k1_ecdsa_signature *bitcoin_signature_2, u8 **features, struct bitcoin_blkid *chain_hash,
					  struct short_channel_id *short_channel_id, struct node_id *node_id_1, struct node_id *node_id_2,
					  struct pubkey *bitcoin_key_1, struct pubkey *bitcoin_key_2) {
	const u8 *m = p;
	const size_t len = shim_tal_bytelen(p);
	size_t flen, o;
	if (len == SHIM_TAL_FOREIGN || len < 260 || m[0] != 1 || m[1] != 0) return false;
	flen = ((size_t)m[258] << 8) | m[259];
	if (len != 260 + flen + 32 + 8 + 4 * 33) return false;
	if (!fromwire_secp256k1_ecdsa_signature(m + 2, node_signature_1) || !fromwire_secp256k1_ecdsa_signature(m + 66, node_signature_2) ||
	    !fromwire_secp256k1_ecdsa_signature(m + 130, bitcoin_signature_1) || !fromwire_secp256k1_ecdsa_signature(m + 194, bitcoin_signature_2))
		return false;
	*features = shim_tal_dup(ctx, m + 260, flen);
	o = 260 + flen;
	memcpy(chain_hash->id, m + o, 32);
	short_channel_id->u64 = 0;
	for (int i = 0; i < 8; i++) short_channel_id->u64 = (short_channel_id->u64 << 8) | m[o + 32 + i];
	o += 40;
	memcpy(node_id_1->k, m + o, 33);
	memcpy(node_id_2->k, m + o + 33, 33);
	return pubkey_from_der(m + o + 66, 33, bitcoin_key_1) && pubkey_from_der(m + o + 99, 33, bitcoin_key_2);
}

static void towire_sig(u8 *out, const secp256k1_ecdsa_signature *sig) { memcpy(out, sig->data, 64); /* the mirror's opaque form is r||s */ }
static u8 *towire_channel_announcement(const tal_t *ctx, const secp256k1_ecdsa_signature *node_signature_1,
				       const secp256k1_ecdsa_signature *node_signature_2, const secp256k1_ecdsa_signature *bitcoin_signature_1,
				       const secp256k1_ecdsa_signature *bitcoin_signature_2, const u8 *features, const struct bitcoin_blkid *chain_hash,
				       struct short_channel_id short_channel_id, const struct node_id *node_id_1, const struct node_id *node_id_2,
				       const struct pubkey *bitcoin_key_1, const struct pubkey *bitcoin_key_2) {
	const size_t flen = features ? shim_tal_bytelen(features) : 0;
	const size_t len = 260 + flen + 40 + 132;
	u8 *m = calloc(1, len), *ret;
	size_t o;
	m[0] = 1;
	towire_sig(m + 2, node_signature_1);
	towire_sig(m + 66, node_signature_2);
	towire_sig(m + 130, bitcoin_signature_1);
	towire_sig(m + 194, bitcoin_signature_2);
	m[258] = (u8)(flen >> 8);
	m[259] = (u8)flen;
	if (flen) memcpy(m + 260, features, flen);
	o = 260 + flen;
	memcpy(m + o, chain_hash->id, 32);
	for (int i = 0; i < 8; i++) m[o + 32 + i] = (u8)(short_channel_id.u64 >> (56 - 8 * i));
	o += 40;
	memcpy(m + o, node_id_1->k, 33);
	memcpy(m + o + 33, node_id_2->k, 33);
	pubkey_to_der(m + o + 66, bitcoin_key_1);
	pubkey_to_der(m + o + 99, bitcoin_key_2);
	ret = shim_tal_dup(ctx, m, len);
	free(m);
	return ret;
}

int main(int argc, char *argv[])
{
	struct bitcoin_blkid chain_hash;
	u8 *features;
	const char *err;
	secp256k1_ecdsa_signature node_signature_1, node_signature_2;
	secp256k1_ecdsa_signature bitcoin_signature_1, bitcoin_signature_2;
	struct short_channel_id short_channel_id;
	struct node_id node_id_1, node_id_2;
	struct pubkey bitcoin_key_1, bitcoin_key_2;
	const u8 *cannounce;
	int rc = 0;

	if (argc != 2) return 2;
	if (!lamd_shim_setup()) { /* common_setup(argv[0]) */
		printf("no engine: %s\n", lamd_shim_last_error());
		return 3;
	}
	cannounce = tal_hexdata(tmpctx, argv[1], strlen(argv[1]));
	if (!fromwire_channel_announcement(cannounce, cannounce,
					   &node_signature_1,
					   &node_signature_2,
					   &bitcoin_signature_1,
					   &bitcoin_signature_2,
					   &features,
					   &chain_hash,
					   &short_channel_id,
					   &node_id_1,
					   &node_id_2,
					   &bitcoin_key_1,
					   &bitcoin_key_2))
		abort();

	/* ---- gossipd/test/run-check_channel_announcement.c:77-85 */
	err = sigcheck_channel_announcement(cannounce,
					    &node_id_1, &node_id_2,
					    &bitcoin_key_1, &bitcoin_key_2,
					    &node_signature_1, &node_signature_2,
					    &bitcoin_signature_1,
					    &bitcoin_signature_2,
					    cannounce);
	printf("%s\n", err ? err : "(null)");
	if (!err || !strstr(err, "Bad node_signature_1")) rc |= 1;

	/* Turns out they didn't include the feature bit at all. */
	cannounce = towire_channel_announcement(tmpctx,
						&node_signature_1,
						&node_signature_2,
						&bitcoin_signature_1,
						&bitcoin_signature_2,
						NULL,
						&chain_hash,
						short_channel_id,
						&node_id_1,
						&node_id_2,
						&bitcoin_key_1,
						&bitcoin_key_2);
	/* ---- :100-108 */
	err = sigcheck_channel_announcement(cannounce,
					    &node_id_1, &node_id_2,
					    &bitcoin_key_1, &bitcoin_key_2,
					    &node_signature_1, &node_signature_2,
					    &bitcoin_signature_1,
					    &bitcoin_signature_2,
					    cannounce);
	printf("%s\n", err ? err : "(null)");
	if (!err || !strstr(err, "Bad node_signature_2")) rc |= 4;

	lamd_shim_shutdown(); /* common_shutdown() */
	return rc;
}

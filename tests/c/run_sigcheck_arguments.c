/* The mirror decides on its ARGUMENTS, as gossipd/sigcheck.c:45-115 does: the message is only hashed (msg + 258 .. end).
 *
 *     run_sigcheck_arguments <hex of a valid channel_announcement A> <hex of another valid channel_announcement B>
 *
 * (a) A with every argument parsed from A                          -> NULL
 * (b) A with node2_sig := B's node_signature_2                     -> "Bad node_signature_2 <DER of B's signature> hash <SHA256d(A + 258)> on channel_announcement <A>"
 * (c) A with its four embedded signatures overwritten by 0xff.. (not even in range), arguments still the ones parsed from A -> NULL
 * (d) A with bitcoin2_key := B's bitcoin_key_2                     -> "Bad bitcoin_signature_2 ..."
 * exit 0 iff all four hold; the strings are printed. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "cln_shim.h"

static u8 *tal_hexdata(const char *str) {
	const size_t slen = strlen(str);
	u8 *raw = malloc(slen / 2 + 1), *ret;
	for (size_t i = 0; i + 1 < slen; i += 2) {
		unsigned v;
		if (sscanf(str + i, "%2x", &v) != 1) abort();
		raw[i / 2] = (u8)v;
	}
	ret = shim_tal_dup(NULL, raw, slen / 2);
	free(raw);
	return ret;
}

struct cann {
	secp256k1_ecdsa_signature sig[4];
	struct node_id id[2];
	struct pubkey key[2];
};
/* fromwire_channel_announcement (wire/peer_wire.csv:344-356), the fields sigcheck needs */
static void parse(const u8 *m, struct cann *c) {
	const size_t len = shim_tal_bytelen(m);
	const size_t flen = ((size_t)m[258] << 8) | m[259];
	const size_t o = 260 + flen + 40;
	if (len != o + 132) abort();
	for (int i = 0; i < 4; i++)
		if (!fromwire_secp256k1_ecdsa_signature(m + 2 + 64 * i, &c->sig[i])) abort();
	memcpy(c->id[0].k, m + o, 33);
	memcpy(c->id[1].k, m + o + 33, 33);
	if (!pubkey_from_der(m + o + 66, 33, &c->key[0]) || !pubkey_from_der(m + o + 99, 33, &c->key[1])) abort();
}

int main(int argc, char *argv[])
{
	struct cann a, b;
	const char *err;
	u8 *ma, *mb, *garbled;
	int rc = 0;

	if (argc != 3) return 2;
	if (!lamd_shim_setup()) {
		printf("no engine: %s\n", lamd_shim_last_error());
		return 3;
	}
	ma = tal_hexdata(argv[1]);
	mb = tal_hexdata(argv[2]);
	parse(ma, &a);
	parse(mb, &b);

	err = sigcheck_channel_announcement(NULL, &a.id[0], &a.id[1], &a.key[0], &a.key[1], &a.sig[0], &a.sig[1], &a.sig[2], &a.sig[3], ma);
	printf("(a) %s\n", err ? err : "(null)");
	if (err) rc |= 1;

	err = sigcheck_channel_announcement(NULL, &a.id[0], &a.id[1], &a.key[0], &a.key[1], &a.sig[0], &b.sig[1], &a.sig[2], &a.sig[3], ma);
	printf("(b) %s\n", err ? err : "(null)");
	if (!err || strncmp(err, "Bad node_signature_2 30", 23) != 0) rc |= 2;

	garbled = shim_tal_dup(NULL, ma, shim_tal_bytelen(ma));
	memset(garbled + 2, 0xff, 256);
	err = sigcheck_channel_announcement(NULL, &a.id[0], &a.id[1], &a.key[0], &a.key[1], &a.sig[0], &a.sig[1], &a.sig[2], &a.sig[3], garbled);
	printf("(c) %s\n", err ? err : "(null)");
	if (err) rc |= 4;

	err = sigcheck_channel_announcement(NULL, &a.id[0], &a.id[1], &a.key[0], &b.key[1], &a.sig[0], &a.sig[1], &a.sig[2], &a.sig[3], ma);
	printf("(d) %s\n", err ? err : "(null)");
	if (!err || strncmp(err, "Bad bitcoin_signature_2 30", 26) != 0) rc |= 8;
	return rc;
}

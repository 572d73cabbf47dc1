/* BASELINE.md 3, leg C0: the reference's real CPU path -- libsecp256k1, called exactly as bitcoin/signature.c:188
 * (secp256k1_ecdsa_verify after fromwire's secp256k1_ecdsa_signature_parse_compact, wire/fromwire.c:196, and
 * secp256k1_ec_pubkey_parse, bitcoin/pubkey.c:19) and :422-429 (secp256k1_xonly_pubkey_parse + secp256k1_schnorrsig_verify)
 * call it -- IF a libsecp256k1.so can be dlopen()ed on this machine.  The reference tree's own copy is an empty submodule
 * (external/libwally-core) and this image ships none, so here every function reports "unavailable" (-1); on a node that has
 * the library bench.py's cpu_baseline gains its "reference" leg and tests/test_libsecp_xcheck.py its third opinion.
 * TEST INFRASTRUCTURE ONLY (oracle/): never linked into the product. */
#include <dlfcn.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { unsigned char data[64]; } blob64;
static void *h_lib, *h_ctx;
static void *(*p_ctx_create)(unsigned int);
static int (*p_sig_parse)(const void *, blob64 *, const unsigned char *);
static int (*p_pub_parse)(const void *, blob64 *, const unsigned char *, size_t);
static int (*p_verify)(const void *, const blob64 *, const unsigned char *, const blob64 *);
static int (*p_xonly_parse)(const void *, blob64 *, const unsigned char *);
static int (*p_schnorr_verify)(const void *, const unsigned char *, const unsigned char *, size_t, const blob64 *);
static char g_path[512];

/* 1 = loaded (ECDSA entry points present), 0 = no library.  path NULL: $LAMD_LIBSECP256K1, then the usual sonames. */
int secpdl_open(const char *path)
{
	static const char *names[] = {"libsecp256k1.so", "libsecp256k1.so.6", "libsecp256k1.so.5", "libsecp256k1.so.2", "libsecp256k1.so.1",
				      "libsecp256k1.so.0", "libsecp256k1_zkp.so", NULL};
	if (h_lib) return 1;
	if (!path) path = getenv("LAMD_LIBSECP256K1");
	if (path && *path) h_lib = dlopen(path, RTLD_NOW | RTLD_LOCAL);
	for (int i = 0; !h_lib && !(path && *path) && names[i]; i++) {
		h_lib = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
		if (h_lib) path = names[i];
	}
	if (!h_lib) return 0;
	*(void **)&p_ctx_create = dlsym(h_lib, "secp256k1_context_create");
	*(void **)&p_sig_parse = dlsym(h_lib, "secp256k1_ecdsa_signature_parse_compact");
	*(void **)&p_pub_parse = dlsym(h_lib, "secp256k1_ec_pubkey_parse");
	*(void **)&p_verify = dlsym(h_lib, "secp256k1_ecdsa_verify");
	*(void **)&p_xonly_parse = dlsym(h_lib, "secp256k1_xonly_pubkey_parse");       /* optional modules */
	*(void **)&p_schnorr_verify = dlsym(h_lib, "secp256k1_schnorrsig_verify");
	if (!p_ctx_create || !p_sig_parse || !p_pub_parse || !p_verify) { dlclose(h_lib); h_lib = NULL; return 0; }
	h_ctx = p_ctx_create(0x0101 /* SECP256K1_CONTEXT_VERIFY */);
	strncpy(g_path, path, sizeof(g_path) - 1);
	return h_ctx != NULL;
}
const char *secpdl_path(void) { return h_lib ? g_path : ""; }
int secpdl_has_schnorr(void) { return h_lib && p_xonly_parse && p_schnorr_verify; }

/* n x check_signed_hash(): returns 0, or -1 when no library is loaded (ok[] untouched) */
int secpdl_ecdsa_verify_batch(size_t n, const uint8_t *hash32, const uint8_t *sig64, const uint8_t *pub, size_t publen, size_t pubstride,
			      uint8_t *ok)
{
	if (!h_lib && !secpdl_open(NULL)) return -1;
	for (size_t i = 0; i < n; i++) {
		blob64 sig, key;
		ok[i] = p_sig_parse(h_ctx, &sig, sig64 + 64 * i) && p_pub_parse(h_ctx, &key, pub + pubstride * i, publen) &&
			p_verify(h_ctx, &sig, hash32 + 32 * i, &key) == 1;
	}
	return 0;
}
/* n x check_schnorr_sig() on x-only keys */
int secpdl_schnorr_verify_batch(size_t n, const uint8_t *msg32, const uint8_t *xonly32, const uint8_t *sig64, uint8_t *ok)
{
	if ((!h_lib && !secpdl_open(NULL)) || !secpdl_has_schnorr()) return -1;
	for (size_t i = 0; i < n; i++) {
		blob64 key;
		ok[i] = p_xonly_parse(h_ctx, &key, xonly32 + 32 * i) && p_schnorr_verify(h_ctx, sig64 + 64 * i, msg32 + 32 * i, 32, &key) == 1;
	}
	return 0;
}

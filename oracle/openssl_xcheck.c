/* Independent ECDSA cross-check through OpenSSL's generic secp256k1 (NID_secp256k1).
 * TEST INFRASTRUCTURE ONLY.  OpenSSL accepts high-S and does not range-check the way
 * libsecp256k1 does, so callers layer those rules on top before comparing
 * (tests/test_oracle_golden.py).  Returns 1 valid, 0 invalid, -1 on setup/parse error. */
#include <openssl/bn.h>
#include <openssl/ec.h>
#include <openssl/ecdsa.h>
#include <openssl/obj_mac.h>
#include <stddef.h>
#include <stdint.h>

int ossl_ecdsa_verify(const uint8_t hash32[32], const uint8_t sig64[64], const uint8_t *pub, size_t publen)
{
	int ret = -1;
	EC_KEY *key = EC_KEY_new_by_curve_name(NID_secp256k1);
	if (!key) return -1;
	const EC_GROUP *grp = EC_KEY_get0_group(key);
	EC_POINT *pt = EC_POINT_new(grp);
	ECDSA_SIG *sig = ECDSA_SIG_new();
	BIGNUM *r = BN_bin2bn(sig64, 32, NULL), *s = BN_bin2bn(sig64 + 32, 32, NULL);
	if (!pt || !sig || !r || !s) goto out;
	if (!EC_POINT_oct2point(grp, pt, pub, publen, NULL)) goto out;
	if (!EC_KEY_set_public_key(key, pt)) goto out;
	if (!ECDSA_SIG_set0(sig, r, s)) goto out;
	r = s = NULL;
	ret = ECDSA_do_verify(hash32, 32, sig, key);
	if (ret < 0) ret = 0; /* OpenSSL reports r/s out of range as an error: a reject */
out:
	BN_free(r); BN_free(s);
	ECDSA_SIG_free(sig); EC_POINT_free(pt); EC_KEY_free(key);
	return ret;
}

/* ---- the same idea for BIP-340 and for public-key recovery: OpenSSL only supplies generic curve arithmetic
 * (EC_POINT_mul, point decompression) and SHA-256; the protocol rules are spelled out here, a third time, independently of
 * oracle/pyref.py and oracle/secp256k1_oracle.c. */
#include <openssl/sha.h>
#include <string.h>

static EC_GROUP *grp_new(void) { return EC_GROUP_new_by_curve_name(NID_secp256k1); }

/* point with this x and the given y parity; 0 if x is not on the curve / >= p */
static int lift(const EC_GROUP *g, EC_POINT *out, const BIGNUM *x, int odd, BN_CTX *bc)
{
	return EC_POINT_set_compressed_coordinates(g, out, x, odd, bc) == 1;
}

/* BIP-340 Verify(pk, m, sig) for 32-byte m.  1 valid, 0 invalid, -1 setup error */
int ossl_schnorr_verify(const uint8_t msg32[32], const uint8_t pk32[32], const uint8_t sig64[64])
{
	int ret = -1;
	EC_GROUP *g = grp_new();
	BN_CTX *bc = BN_CTX_new();
	if (!g || !bc) goto out0;
	BN_CTX_start(bc);
	BIGNUM *p = BN_CTX_get(bc), *n = BN_CTX_get(bc), *px = BN_CTX_get(bc), *r = BN_CTX_get(bc), *s = BN_CTX_get(bc), *e = BN_CTX_get(bc),
	       *x = BN_CTX_get(bc), *y = BN_CTX_get(bc);
	EC_POINT *P = EC_POINT_new(g), *R = EC_POINT_new(g);
	if (!y || !P || !R) goto out;
	EC_GROUP_get_curve(g, p, NULL, NULL, bc);
	EC_GROUP_get_order(g, n, bc);
	BN_bin2bn(pk32, 32, px); BN_bin2bn(sig64, 32, r); BN_bin2bn(sig64 + 32, 32, s);
	ret = 0;
	if (BN_cmp(px, p) >= 0 || !lift(g, P, px, 0, bc)) goto out;          /* lift_x(pk): even y */
	if (BN_cmp(r, p) >= 0 || BN_cmp(s, n) >= 0) goto out;
	{
		/* e = int(hash_BIP0340/challenge(r || pk || m)) mod n */
		uint8_t tag[32], buf[64 + 96], h[32];
		SHA256((const unsigned char *)"BIP0340/challenge", 17, tag);
		memcpy(buf, tag, 32); memcpy(buf + 32, tag, 32);
		memcpy(buf + 64, sig64, 32); memcpy(buf + 96, pk32, 32); memcpy(buf + 128, msg32, 32);
		SHA256(buf, sizeof buf, h);
		BN_bin2bn(h, 32, e);
		BN_mod(e, e, n, bc);
	}
	BN_sub(e, n, e);                                                      /* -e mod n (e = 0 gives n: times P is infinity, fine) */
	if (!EC_POINT_mul(g, R, s, P, e, bc)) { ret = -1; goto out; }         /* R = s*G + (-e)*P */
	if (EC_POINT_is_at_infinity(g, R)) goto out;
	if (!EC_POINT_get_affine_coordinates(g, R, x, y, bc)) { ret = -1; goto out; }
	if (BN_is_odd(y) || BN_cmp(x, r) != 0) goto out;
	ret = 1;
out:
	EC_POINT_free(P); EC_POINT_free(R);
	BN_CTX_end(bc);
out0:
	BN_CTX_free(bc); EC_GROUP_free(g);
	return ret;
}

/* secp256k1_ecdsa_recoverable_signature_parse_compact + secp256k1_ecdsa_recover.  1 = recovered (out33 compressed key),
 * 0 = the library calls would fail, -1 setup error */
int ossl_ecdsa_recover(const uint8_t hash32[32], const uint8_t sig64[64], int recid, uint8_t out33[33])
{
	int ret = -1;
	EC_GROUP *g = grp_new();
	BN_CTX *bc = BN_CTX_new();
	if (!g || !bc) goto out0;
	BN_CTX_start(bc);
	BIGNUM *p = BN_CTX_get(bc), *n = BN_CTX_get(bc), *r = BN_CTX_get(bc), *s = BN_CTX_get(bc), *z = BN_CTX_get(bc), *x = BN_CTX_get(bc),
	       *ri = BN_CTX_get(bc), *u1 = BN_CTX_get(bc), *u2 = BN_CTX_get(bc);
	EC_POINT *R = EC_POINT_new(g), *Q = EC_POINT_new(g);
	if (!u2 || !R || !Q) goto out;
	EC_GROUP_get_curve(g, p, NULL, NULL, bc);
	EC_GROUP_get_order(g, n, bc);
	BN_bin2bn(sig64, 32, r); BN_bin2bn(sig64 + 32, 32, s); BN_bin2bn(hash32, 32, z);
	ret = 0;
	if (recid < 0 || recid > 3 || BN_cmp(r, n) >= 0 || BN_cmp(s, n) >= 0 || BN_is_zero(r) || BN_is_zero(s)) goto out;
	BN_copy(x, r);
	if (recid & 2) {
		BN_add(x, x, n);
		if (BN_cmp(x, p) >= 0) goto out;
	}
	if (!lift(g, R, x, recid & 1, bc)) goto out;
	BN_mod(z, z, n, bc);
	if (!BN_mod_inverse(ri, r, n, bc)) { ret = -1; goto out; }
	BN_mod_mul(u1, z, ri, n, bc);
	if (!BN_is_zero(u1)) BN_sub(u1, n, u1);                                /* -z/r */
	BN_mod_mul(u2, s, ri, n, bc);                                          /*  s/r */
	if (!EC_POINT_mul(g, Q, u1, R, u2, bc)) { ret = -1; goto out; }        /* Q = u1*G + u2*R */
	if (EC_POINT_is_at_infinity(g, Q)) goto out;
	if (EC_POINT_point2oct(g, Q, POINT_CONVERSION_COMPRESSED, out33, 33, bc) != 33) { ret = -1; goto out; }
	ret = 1;
out:
	EC_POINT_free(R); EC_POINT_free(Q);
	BN_CTX_end(bc);
out0:
	BN_CTX_free(bc); EC_GROUP_free(g);
	return ret;
}

/* ---- BASELINE.md 3, leg C2: n rows of "OpenSSL ECDSA_do_verify on NID_secp256k1 + libsecp256k1's acceptance rules" (r, s in
 * [1, n-1], s <= n/2 -- bitcoin/signature.c:185-187 -- checked here, in front of the library call), one thread, rows exactly as
 * lamd_verify_ecdsa_batch takes them.  bench.py's cpu_baseline times it (a second, independent CPU point: OpenSSL's generic
 * curve code, no secp256k1-specific tricks); tests compare its verdicts with the oracle's. */
static const uint8_t ORDER_N[32] = {0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFE,
				    0xBA, 0xAE, 0xDC, 0xE6, 0xAF, 0x48, 0xA0, 0x3B, 0xBF, 0xD2, 0x5E, 0x8C, 0xD0, 0x36, 0x41, 0x41};
static const uint8_t HALF_N[32] = {0x7F, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF,
				   0x5D, 0x57, 0x6E, 0x73, 0x57, 0xA4, 0x50, 0x1D, 0xDF, 0xE9, 0x2F, 0x46, 0x68, 0x1B, 0x20, 0xA0};
static int is_zero32(const uint8_t *a) { uint8_t o = 0; for (int i = 0; i < 32; i++) o |= a[i]; return o == 0; }
void ossl_ecdsa_verify_rules_batch_mt(size_t n, const uint8_t *hash32, const uint8_t *sig64, const uint8_t *pub, size_t publen, size_t pubstride,
				      uint8_t *ok, int nthreads)
{
	long i;
#pragma omp parallel for schedule(dynamic, 64) num_threads(nthreads > 0 ? nthreads : 1)
	for (i = 0; i < (long)n; i++) {
		const uint8_t *sg = sig64 + 64 * i;
		ok[i] = 0;
		if (is_zero32(sg) || is_zero32(sg + 32) || memcmp(sg, ORDER_N, 32) >= 0 || memcmp(sg + 32, HALF_N, 32) > 0) continue;
		ok[i] = ossl_ecdsa_verify(hash32 + 32 * i, sg, pub + pubstride * i, publen) == 1;
	}
}
void ossl_ecdsa_verify_rules_batch(size_t n, const uint8_t *hash32, const uint8_t *sig64, const uint8_t *pub, size_t publen, size_t pubstride,
				   uint8_t *ok)
{
	ossl_ecdsa_verify_rules_batch_mt(n, hash32, sig64, pub, publen, pubstride, ok, 1);
}
/* n x BIP-340 through ossl_schnorr_verify() (the protocol spelled out over OpenSSL's generic point arithmetic) */
void ossl_schnorr_verify_batch_mt(size_t n, const uint8_t *msg32, const uint8_t *pk32, const uint8_t *sig64, uint8_t *ok, int nthreads)
{
	long i;
#pragma omp parallel for schedule(dynamic, 64) num_threads(nthreads > 0 ? nthreads : 1)
	for (i = 0; i < (long)n; i++) ok[i] = ossl_schnorr_verify(msg32 + 32 * i, pk32 + 32 * i, sig64 + 64 * i) == 1;
}

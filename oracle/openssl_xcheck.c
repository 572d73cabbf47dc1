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

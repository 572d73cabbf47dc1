/* Seeded generator of signed rows covering every synthesised edge class of SURVEY.md 8(c) ("Edge-case ECDSA classes to synthesise",
 * the BIP-340 classes of cfg3) -- TEST INFRASTRUCTURE ONLY.  Rows are signed with the oracle's own signer (orc_ecdsa_sign /
 * orc_schnorr_sign), then damaged according to the row's class; cls[i] names the class and expect[i] the verdict the class fixes BY
 * CONSTRUCTION (a bit flip turns a valid signature into a valid one with probability 2^-128).  tests/ require
 *     C oracle == OpenSSL (+ libsecp256k1's rules) == construction      on >= 10^6 such rows (CPU),
 *     HIP engine == C oracle == construction                            on the same rows (GPU).
 * splitmix64 per row: row i depends on (seed, i) only, so any slice can be regenerated and OpenMP order does not matter. */
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include "secp256k1_oracle.h"

static uint64_t sm64(uint64_t *s) { uint64_t z = (*s += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
static void rnd32(uint64_t *s, uint8_t out[32]) { for (int i = 0; i < 4; i++) { uint64_t v = sm64(s); memcpy(out + 8 * i, &v, 8); } }
static const uint8_t N_BE[32] = {0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFE,
				 0xBA, 0xAE, 0xDC, 0xE6, 0xAF, 0x48, 0xA0, 0x3B, 0xBF, 0xD2, 0x5E, 0x8C, 0xD0, 0x36, 0x41, 0x41};
static const uint8_t P_BE[32] = {0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF,
				 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFE, 0xFF, 0xFF, 0xFC, 0x2F};
/* r = a - b (big-endian 32 bytes), a >= b */
static void sub_be(uint8_t r[32], const uint8_t a[32], const uint8_t b[32]) { int bw = 0; for (int i = 31; i >= 0; i--) { int v = a[i] - b[i] - bw; bw = v < 0; r[i] = (uint8_t)(v + (bw << 8)); } }
static void add_small_be(uint8_t r[32], const uint8_t a[32], uint64_t k) { unsigned c = 0; for (int i = 31; i >= 0; i--) { unsigned v = a[i] + (unsigned)(k & 0xFF) + c; k >>= 8; r[i] = (uint8_t)v; c = v >> 8; } }
static void seckey(uint64_t *s, uint8_t d[32]) { do { rnd32(s, d); d[0] &= 0x7F; } while (!(d[0] | d[1] | d[2] | d[3] | d[31])); }

enum { EC_VALID = 0, EC_VALID_HASH_GE_N, EC_FLIP_HASH, EC_FLIP_R, EC_FLIP_S, EC_HIGH_S, EC_WRONG_KEY, EC_ZERO_R, EC_ZERO_S, EC_R_GE_N, EC_S_GE_N,
       EC_KEY_OFF_CURVE, EC_KEY_BAD_PREFIX, EC_KEY_X_GE_P, EC_KEY_WRONG_PARITY, EC_KEY_HYBRID_OK, EC_KEY_HYBRID_BAD, EC_NCLASSES };
/* share of each class per 32 rows: 16 valid, one or two of each damage */
static const uint8_t EC_PLAN[32] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 2, 3, 4, 5, 5, 6, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
int orc_edge_nclasses_ecdsa(void) { return EC_NCLASSES; }

/* publen 33 or 65.  cls / expect: one byte per row. */
void orc_gen_ecdsa_edge_batch(uint64_t seed, size_t n, size_t publen, uint8_t *hash32, uint8_t *sig64, uint8_t *pub, uint8_t *cls, uint8_t *expect, int nthreads)
{
	orc_init();
	long i;
#pragma omp parallel for schedule(dynamic, 64) num_threads(nthreads > 0 ? nthreads : 1)
	for (i = 0; i < (long)n; i++) {
		uint64_t s = seed ^ (0xD1B54A32D192ED03ull * (uint64_t)(i + 1));
		uint8_t d[32], k[32], h[32], sg[64], p65[65], *P = pub + publen * i;
		int c = EC_PLAN[(i + (sm64(&s) & 31)) & 31], ok = 1;
		seckey(&s, d); seckey(&s, k); rnd32(&s, h);
		if (c == EC_VALID_HASH_GE_N) { memset(h, 0xFF, 16); h[15] = 0xFE; h[16] = 0xBA; h[17] = 0xAE; h[18] = 0xDC; h[19] = 0xE7; /* > n: reduced mod n by signer and verifier alike */ }
		orc_ecdsa_sign(h, d, k, sg);
		orc_pubkey_create(d, p65);
		if (publen == 65) memcpy(P, p65, 65);
		else { P[0] = 2 + (p65[64] & 1); memcpy(P + 1, p65 + 1, 32); }
		const unsigned bit = (unsigned)(sm64(&s) & 0xFF);
		switch (c) {
		case EC_FLIP_HASH: h[bit >> 3] ^= 1 << (bit & 7); ok = 0; break;
		case EC_FLIP_R: sg[bit >> 3] ^= 1 << (bit & 7); ok = 0; break;
		case EC_FLIP_S: sg[32 + (bit >> 3)] ^= 1 << (bit & 7); ok = 0; break;
		case EC_HIGH_S: { uint8_t t[32]; sub_be(t, N_BE, sg + 32); memcpy(sg + 32, t, 32); ok = 0; break; }
		case EC_WRONG_KEY: { uint8_t d2[32], q[65]; seckey(&s, d2); orc_pubkey_create(d2, q); if (publen == 65) memcpy(P, q, 65); else { P[0] = 2 + (q[64] & 1); memcpy(P + 1, q + 1, 32); } ok = 0; break; }
		case EC_ZERO_R: memset(sg, 0, 32); ok = 0; break;
		case EC_ZERO_S: memset(sg + 32, 0, 32); ok = 0; break;
		case EC_R_GE_N: add_small_be(sg, N_BE, sm64(&s) >> (8 + (bit & 31))); ok = 0; break;       /* n, n+1, ... : compact-parse failure */
		case EC_S_GE_N: add_small_be(sg + 32, N_BE, sm64(&s) >> (8 + (bit & 31))); ok = 0; break;
		case EC_KEY_OFF_CURVE:
			if (publen == 65) P[33 + (bit >> 3)] ^= 1 << (bit & 7);                                 /* Y damaged */
			else { /* an x with no square root: walk x until the oracle refuses it */
				uint8_t t[64]; do { add_small_be(P + 1, P + 1, 1); } while (orc_pubkey_parse(P, 33, t)); }
			ok = 0; break;
		case EC_KEY_BAD_PREFIX: P[0] = publen == 65 ? (uint8_t)(bit & 1 ? 0x05 : 0x00) : (uint8_t)(bit & 1 ? 0x04 : 0x05); ok = 0; break;
		case EC_KEY_X_GE_P: add_small_be(P + 1, P_BE, sm64(&s) & 0x3FF); if (publen == 65) { /* keep Y: parse must fail on x alone */ } ok = 0; break;
		case EC_KEY_WRONG_PARITY:
			if (publen == 33) P[0] ^= 1;                                                            /* the other lifting: a different, valid key */
			else { uint8_t t[32]; sub_be(t, P_BE, P + 33); memcpy(P + 33, t, 32); }                 /* -Q: on the curve, wrong key */
			ok = 0; break;
		case EC_KEY_HYBRID_OK: if (publen == 65) P[0] = 6 + (P[64] & 1); break;                         /* 0x06/0x07 with the right parity bit: accepted */
		case EC_KEY_HYBRID_BAD: if (publen == 65) { P[0] = 7 - (P[64] & 1); ok = 0; } else { P[0] = 6 + (bit & 1); ok = 0; } break;
		default: break;
		}
		memcpy(hash32 + 32 * i, h, 32);
		memcpy(sig64 + 64 * i, sg, 64);
		cls[i] = (uint8_t)c;
		expect[i] = (uint8_t)ok;
	}
}

enum { SC_VALID = 0, SC_FLIP_MSG, SC_FLIP_R, SC_FLIP_S, SC_R_GE_P, SC_S_GE_N, SC_KEY_NOT_LIFTABLE, SC_NEG_S, SC_WRONG_KEY, SC_KEY_X_GE_P, SC_ZERO_SIG, SC_NCLASSES };
static const uint8_t SC_PLAN[32] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 5, 6, 6, 7, 7, 8, 8, 9, 10};
int orc_edge_nclasses_schnorr(void) { return SC_NCLASSES; }
void orc_gen_schnorr_edge_batch(uint64_t seed, size_t n, uint8_t *msg32, uint8_t *xonly32, uint8_t *sig64, uint8_t *cls, uint8_t *expect, int nthreads)
{
	orc_init();
	long i;
#pragma omp parallel for schedule(dynamic, 64) num_threads(nthreads > 0 ? nthreads : 1)
	for (i = 0; i < (long)n; i++) {
		uint64_t s = seed ^ (0xD1B54A32D192ED03ull * (uint64_t)(i + 1));
		uint8_t d[32], aux[32], m[32], sg[64], p65[65], *X = xonly32 + 32 * i;
		int c = SC_PLAN[(i + (sm64(&s) & 31)) & 31], ok = 1;
		seckey(&s, d); rnd32(&s, aux); rnd32(&s, m);
		orc_schnorr_sign(m, d, aux, sg);
		orc_pubkey_create(d, p65);
		memcpy(X, p65 + 1, 32);
		const unsigned bit = (unsigned)(sm64(&s) & 0xFF);
		switch (c) {
		case SC_FLIP_MSG: m[bit >> 3] ^= 1 << (bit & 7); ok = 0; break;
		case SC_FLIP_R: sg[bit >> 3] ^= 1 << (bit & 7); ok = 0; break;
		case SC_FLIP_S: sg[32 + (bit >> 3)] ^= 1 << (bit & 7); ok = 0; break;
		case SC_R_GE_P: add_small_be(sg, P_BE, sm64(&s) & 0x3FF); ok = 0; break;
		case SC_S_GE_N: add_small_be(sg + 32, N_BE, sm64(&s) >> (8 + (bit & 31))); ok = 0; break;
		case SC_KEY_NOT_LIFTABLE: { uint8_t t[64], c33[33]; c33[0] = 2; do { add_small_be(X, X, 1); memcpy(c33 + 1, X, 32); } while (orc_pubkey_parse(c33, 33, t)); ok = 0; break; }
		case SC_NEG_S: { uint8_t t[32]; sub_be(t, N_BE, sg + 32); memcpy(sg + 32, t, 32); ok = 0; break; }
		case SC_WRONG_KEY: { uint8_t d2[32], q[65]; seckey(&s, d2); orc_pubkey_create(d2, q); memcpy(X, q + 1, 32); ok = 0; break; }
		case SC_KEY_X_GE_P: add_small_be(X, P_BE, sm64(&s) & 0x3FF); ok = 0; break;
		case SC_ZERO_SIG: memset(sg, 0, bit & 1 ? 64 : 32); ok = 0; break;   /* r = 0 (x = 0 is not on the curve) / the all-zero signature */
		default: break;
		}
		memcpy(msg32 + 32 * i, m, 32);
		memcpy(sig64 + 64 * i, sg, 64);
		cls[i] = (uint8_t)c;
		expect[i] = (uint8_t)ok;
	}
}

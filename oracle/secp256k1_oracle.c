/* CPU oracle -- TEST INFRASTRUCTURE ONLY (see secp256k1_oracle.h for scope, provenance
 * and how parity is pinned).  Representation is deliberately different from the HIP
 * path (4x64-bit limbs + unsigned __int128, fully reduced after every operation, wNAF
 * Strauss without the GLV endomorphism) so that a shared arithmetic bug is unlikely.
 */
#include "secp256k1_oracle.h"
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef uint64_t u64;

/* ------------------------------------------------------------------ 256-bit helpers */
typedef struct { u64 d[4]; } u256; /* little-endian limbs */

static int u256_cmp(const u256 *a, const u256 *b)
{
	for (int i = 3; i >= 0; i--) {
		if (a->d[i] < b->d[i]) return -1;
		if (a->d[i] > b->d[i]) return 1;
	}
	return 0;
}
static int u256_is_zero(const u256 *a) { return (a->d[0] | a->d[1] | a->d[2] | a->d[3]) == 0; }
static u64 u256_add(u256 *r, const u256 *a, const u256 *b)
{
	u128 c = 0;
	for (int i = 0; i < 4; i++) { c += (u128)a->d[i] + b->d[i]; r->d[i] = (u64)c; c >>= 64; }
	return (u64)c;
}
static u64 u256_sub(u256 *r, const u256 *a, const u256 *b)
{
	u64 borrow = 0;
	for (int i = 0; i < 4; i++) {
		u128 t = (u128)a->d[i] - b->d[i] - borrow;
		r->d[i] = (u64)t;
		borrow = (u64)(t >> 64) & 1;
	}
	return borrow;
}
static void u256_from_be(u256 *r, const uint8_t b[32])
{
	for (int i = 0; i < 4; i++) {
		u64 v = 0;
		for (int j = 0; j < 8; j++) v = (v << 8) | b[(3 - i) * 8 + j];
		r->d[i] = v;
	}
}
static void u256_to_be(uint8_t b[32], const u256 *a)
{
	for (int i = 0; i < 4; i++)
		for (int j = 0; j < 8; j++) b[(3 - i) * 8 + j] = (uint8_t)(a->d[i] >> (56 - 8 * j));
}
static void mul_4x4(u64 t[8], const u256 *a, const u256 *b)
{
	memset(t, 0, 8 * sizeof(u64));
	for (int i = 0; i < 4; i++) {
		u64 carry = 0;
		for (int j = 0; j < 4; j++) {
			u128 acc = (u128)a->d[i] * b->d[j] + t[i + j] + carry;
			t[i + j] = (u64)acc;
			carry = (u64)(acc >> 64);
		}
		t[i + 4] = carry;
	}
}

/* ------------------------------------------------------------------ field mod p (SEC2 secp256k1) */
typedef u256 fe; /* always in [0, p) */
static const fe FE_P = {{0xFFFFFFFEFFFFFC2FULL, 0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFFFFFFFFFFULL}};
#define FE_PC 0x1000003D1ULL /* 2^256 - p */
static const fe FE_ONE = {{1, 0, 0, 0}};
static const fe FE_SEVEN = {{7, 0, 0, 0}};

static void fe_fix(fe *r, u64 carry)
{
	/* r + carry*2^256, with carry in {0,1}, brought into [0,p) */
	if (carry || u256_cmp(r, &FE_P) >= 0) {
		u256 t;
		u256_sub(&t, r, &FE_P); /* mod 2^256: exactly r + 2^256*carry - p */
		*r = t;
	}
}
static void fe_add(fe *r, const fe *a, const fe *b) { u64 c = u256_add(r, a, b); fe_fix(r, c); }
static void fe_sub(fe *r, const fe *a, const fe *b)
{
	if (u256_sub(r, a, b)) { u256 t; u256_add(&t, r, &FE_P); *r = t; }
}
static void fe_neg(fe *r, const fe *a)
{
	if (u256_is_zero(a)) { *r = *a; return; }
	u256_sub(r, &FE_P, a);
}
static void fe_mul(fe *r, const fe *a, const fe *b)
{
	u64 t[8];
	mul_4x4(t, a, b);
	/* lo + hi * (2^256 mod p) */
	u128 acc = 0;
	u64 lo[4];
	for (int i = 0; i < 4; i++) {
		acc += (u128)t[4 + i] * FE_PC + t[i];
		lo[i] = (u64)acc;
		acc >>= 64;
	}
	u64 c = (u64)acc; /* < 2^34 */
	acc = (u128)c * FE_PC + lo[0];
	r->d[0] = (u64)acc; acc >>= 64;
	for (int i = 1; i < 4; i++) { acc += lo[i]; r->d[i] = (u64)acc; acc >>= 64; }
	fe_fix(r, (u64)acc);
}
static void fe_reduce512(fe *r, const u64 t[8])
{
	u128 acc = 0;
	u64 lo[4];
	for (int i = 0; i < 4; i++) {
		acc += (u128)t[4 + i] * FE_PC + t[i];
		lo[i] = (u64)acc;
		acc >>= 64;
	}
	u64 c = (u64)acc;
	acc = (u128)c * FE_PC + lo[0];
	r->d[0] = (u64)acc; acc >>= 64;
	for (int i = 1; i < 4; i++) { acc += lo[i]; r->d[i] = (u64)acc; acc >>= 64; }
	fe_fix(r, (u64)acc);
}
static void fe_sqr(fe *r, const fe *a)
{
	/* cross products once, doubled, plus the four squares */
	u64 t[8] = {0};
	for (int i = 0; i < 4; i++) {
		u64 carry = 0;
		for (int j = i + 1; j < 4; j++) {
			u128 acc = (u128)a->d[i] * a->d[j] + t[i + j] + carry;
			t[i + j] = (u64)acc;
			carry = (u64)(acc >> 64);
		}
		t[i + 4] = carry;
	}
	u64 top = 0;
	for (int i = 0; i < 8; i++) { u64 nt = t[i] >> 63; t[i] = (t[i] << 1) | top; top = nt; }
	u64 carry = 0;
	for (int i = 0; i < 4; i++) {
		u128 sq = (u128)a->d[i] * a->d[i];
		u128 acc = (u128)t[2 * i] + (u64)sq + carry;
		t[2 * i] = (u64)acc;
		acc = (u128)t[2 * i + 1] + (u64)(sq >> 64) + (u64)(acc >> 64);
		t[2 * i + 1] = (u64)acc;
		carry = (u64)(acc >> 64);
	}
	fe_reduce512(r, t);
}
static void fe_mul_int(fe *r, const fe *a, unsigned k)
{
	fe acc = {{0, 0, 0, 0}}, t = *a;
	while (k) { if (k & 1) fe_add(&acc, &acc, &t); fe_add(&t, &t, &t); k >>= 1; }
	*r = acc;
}
static void fe_pow(fe *r, const fe *a, const u256 *e)
{
	/* left-to-right 4-bit fixed window */
	fe tbl[16];
	tbl[0] = FE_ONE; tbl[1] = *a;
	for (int i = 2; i < 16; i++) fe_mul(&tbl[i], &tbl[i - 1], a);
	fe acc = FE_ONE;
	for (int i = 63; i >= 0; i--) {
		for (int k = 0; k < 4; k++) fe_sqr(&acc, &acc);
		unsigned nib = (unsigned)(e->d[i / 16] >> ((i % 16) * 4)) & 15;
		if (nib) fe_mul(&acc, &acc, &tbl[nib]);
	}
	*r = acc;
}
static void fe_inv(fe *r, const fe *a)
{
	static const u256 e = {{0xFFFFFFFEFFFFFC2DULL, 0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFFFFFFFFFFULL}}; /* p-2 */
	fe_pow(r, a, &e);
}
static int fe_sqrt(fe *r, const fe *a)
{
	/* p = 3 mod 4: candidate a^((p+1)/4); 1 iff it squares back to a */
	static const u256 e = {{0xFFFFFFFFBFFFFF0CULL, 0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFFFFFFFFFFULL, 0x3FFFFFFFFFFFFFFFULL}};
	fe c, c2;
	fe_pow(&c, a, &e);
	fe_sqr(&c2, &c);
	*r = c;
	return u256_cmp(&c2, a) == 0;
}
static int fe_from_be(fe *r, const uint8_t b[32]) /* 0 if >= p */
{
	u256_from_be(r, b);
	return u256_cmp(r, &FE_P) < 0;
}

/* ------------------------------------------------------------------ scalars mod n */
typedef u256 sc; /* always in [0, n) */
static const sc SC_N = {{0xBFD25E8CD0364141ULL, 0xBAAEDCE6AF48A03BULL, 0xFFFFFFFFFFFFFFFEULL, 0xFFFFFFFFFFFFFFFFULL}};
static const sc SC_HALF_N = {{0xDFE92F46681B20A0ULL, 0x5D576E7357A4501DULL, 0xFFFFFFFFFFFFFFFFULL, 0x7FFFFFFFFFFFFFFFULL}};
static const u64 SC_NC[3] = {0x402DA1732FC9BEBFULL, 0x4551231950B75FC4ULL, 1ULL}; /* 2^256 - n */
static const sc SC_ONE = {{1, 0, 0, 0}};

static void sc_reduce_wide(sc *r, const u64 *t, int nl)
{
	u64 cur[12] = {0};
	memcpy(cur, t, (size_t)nl * sizeof(u64));
	int n = nl;
	while (n > 4 && cur[n - 1] == 0) n--;
	while (n > 4) {
		/* cur = lo + hi * (2^256 mod n) */
		u64 nxt[12] = {0};
		int hl = n - 4;
		for (int i = 0; i < hl; i++) {
			u64 carry = 0;
			for (int j = 0; j < 3; j++) {
				u128 acc = (u128)cur[4 + i] * SC_NC[j] + nxt[i + j] + carry;
				nxt[i + j] = (u64)acc;
				carry = (u64)(acc >> 64);
			}
			nxt[i + 3] += carry; /* cannot overflow: slot was zero */
		}
		u128 c = 0;
		int m = hl + 3 + 1;
		if (m < 5) m = 5;
		for (int i = 0; i < m; i++) {
			c += (u128)nxt[i] + (i < 4 ? cur[i] : 0);
			nxt[i] = (u64)c;
			c >>= 64;
		}
		memcpy(cur, nxt, sizeof(cur));
		n = m;
		while (n > 4 && cur[n - 1] == 0) n--;
	}
	memcpy(r->d, cur, 4 * sizeof(u64));
	while (u256_cmp(r, &SC_N) >= 0) { u256 tt; u256_sub(&tt, r, &SC_N); *r = tt; }
}
static int sc_from_be(sc *r, const uint8_t b[32]) /* returns overflow flag; r = value mod n */
{
	u256_from_be(r, b);
	int over = u256_cmp(r, &SC_N) >= 0;
	if (over) { u256 t; u256_sub(&t, r, &SC_N); *r = t; }
	return over;
}
static void sc_mul(sc *r, const sc *a, const sc *b)
{
	u64 t[8];
	mul_4x4(t, a, b);
	sc_reduce_wide(r, t, 8);
}
static void sc_add(sc *r, const sc *a, const sc *b)
{
	u64 t[5];
	u256 s;
	t[4] = u256_add(&s, a, b);
	memcpy(t, s.d, 32);
	sc_reduce_wide(r, t, 5);
}
static void sc_neg(sc *r, const sc *a)
{
	if (u256_is_zero(a)) { *r = *a; return; }
	u256_sub(r, &SC_N, a);
}
static void sc_inv(sc *r, const sc *a)
{
	static const u256 e = {{0xBFD25E8CD036413FULL, 0xBAAEDCE6AF48A03BULL, 0xFFFFFFFFFFFFFFFEULL, 0xFFFFFFFFFFFFFFFFULL}}; /* n-2 */
	sc tbl[16];
	tbl[0] = SC_ONE; tbl[1] = *a;
	for (int i = 2; i < 16; i++) sc_mul(&tbl[i], &tbl[i - 1], a);
	sc acc = SC_ONE;
	for (int i = 63; i >= 0; i--) {
		for (int k = 0; k < 4; k++) sc_mul(&acc, &acc, &acc);
		unsigned nib = (unsigned)(e.d[i / 16] >> ((i % 16) * 4)) & 15;
		if (nib) sc_mul(&acc, &acc, &tbl[nib]);
	}
	*r = acc;
}

/* ------------------------------------------------------------------ group (y^2 = x^3 + 7) */
typedef struct { fe x, y; int inf; } ge;
typedef struct { fe x, y, z; int inf; } gej;

static const ge GE_G = {
	{{0x59F2815B16F81798ULL, 0x029BFCDB2DCE28D9ULL, 0x55A06295CE870B07ULL, 0x79BE667EF9DCBBACULL}},
	{{0x9C47D08FFB10D4B8ULL, 0xFD17B448A6855419ULL, 0x5DA4FBFC0E1108A8ULL, 0x483ADA7726A3C465ULL}}, 0};

static int ge_on_curve(const ge *a)
{
	fe y2, x3;
	fe_sqr(&y2, &a->y);
	fe_sqr(&x3, &a->x); fe_mul(&x3, &x3, &a->x); fe_add(&x3, &x3, &FE_SEVEN);
	return u256_cmp(&y2, &x3) == 0;
}
static void gej_set_ge(gej *r, const ge *a) { r->x = a->x; r->y = a->y; r->z = FE_ONE; r->inf = a->inf; }
static void gej_set_inf(gej *r) { memset(r, 0, sizeof(*r)); r->inf = 1; }

static void gej_double(gej *r, const gej *a)
{
	/* dbl-2009-l style for a = 0 */
	if (a->inf || u256_is_zero(&a->y)) { gej_set_inf(r); return; }
	fe A, B, C, D, E, F, t;
	fe_sqr(&A, &a->x);
	fe_sqr(&B, &a->y);
	fe_sqr(&C, &B);
	fe_add(&t, &a->x, &B); fe_sqr(&t, &t); fe_sub(&t, &t, &A); fe_sub(&t, &t, &C);
	fe_add(&D, &t, &t);
	fe_mul_int(&E, &A, 3);
	fe_sqr(&F, &E);
	fe z3; fe_mul(&z3, &a->y, &a->z); fe_add(&z3, &z3, &z3);
	fe x3; fe_add(&t, &D, &D); fe_sub(&x3, &F, &t);
	fe y3; fe_sub(&t, &D, &x3); fe_mul(&y3, &E, &t); fe_mul_int(&t, &C, 8); fe_sub(&y3, &y3, &t);
	r->x = x3; r->y = y3; r->z = z3; r->inf = 0;
}
static void gej_add(gej *r, const gej *a, const gej *b)
{
	if (a->inf) { *r = *b; return; }
	if (b->inf) { *r = *a; return; }
	fe z1z1, z2z2, u1, u2, s1, s2, h, rr, t;
	fe_sqr(&z1z1, &a->z); fe_sqr(&z2z2, &b->z);
	fe_mul(&u1, &a->x, &z2z2); fe_mul(&u2, &b->x, &z1z1);
	fe_mul(&s1, &a->y, &b->z); fe_mul(&s1, &s1, &z2z2);
	fe_mul(&s2, &b->y, &a->z); fe_mul(&s2, &s2, &z1z1);
	fe_sub(&h, &u2, &u1);
	fe_sub(&rr, &s2, &s1);
	if (u256_is_zero(&h)) {
		if (u256_is_zero(&rr)) gej_double(r, a); else gej_set_inf(r);
		return;
	}
	fe hh, hhh, v;
	fe_sqr(&hh, &h); fe_mul(&hhh, &hh, &h); fe_mul(&v, &u1, &hh);
	fe x3, y3, z3;
	fe_sqr(&x3, &rr); fe_sub(&x3, &x3, &hhh); fe_add(&t, &v, &v); fe_sub(&x3, &x3, &t);
	fe_sub(&t, &v, &x3); fe_mul(&y3, &rr, &t); fe_mul(&t, &s1, &hhh); fe_sub(&y3, &y3, &t);
	fe_mul(&z3, &a->z, &b->z); fe_mul(&z3, &z3, &h);
	r->x = x3; r->y = y3; r->z = z3; r->inf = 0;
}
static void gej_add_ge(gej *r, const gej *a, const ge *b)
{
	/* mixed addition (Z2 = 1): 8M + 3S, same case analysis as gej_add */
	if (b->inf) { *r = *a; return; }
	if (a->inf) { gej_set_ge(r, b); return; }
	fe z1z1, u2, s2, h, rr, t;
	fe_sqr(&z1z1, &a->z);
	fe_mul(&u2, &b->x, &z1z1);
	fe_mul(&s2, &b->y, &a->z); fe_mul(&s2, &s2, &z1z1);
	fe_sub(&h, &u2, &a->x);
	fe_sub(&rr, &s2, &a->y);
	if (u256_is_zero(&h)) {
		if (u256_is_zero(&rr)) gej_double(r, a); else gej_set_inf(r);
		return;
	}
	fe hh, hhh, v, x3, y3, z3;
	fe_sqr(&hh, &h); fe_mul(&hhh, &hh, &h); fe_mul(&v, &a->x, &hh);
	fe_sqr(&x3, &rr); fe_sub(&x3, &x3, &hhh); fe_add(&t, &v, &v); fe_sub(&x3, &x3, &t);
	fe_sub(&t, &v, &x3); fe_mul(&y3, &rr, &t); fe_mul(&t, &a->y, &hhh); fe_sub(&y3, &y3, &t);
	fe_mul(&z3, &a->z, &h);
	r->x = x3; r->y = y3; r->z = z3; r->inf = 0;
}
static void ge_set_gej(ge *r, const gej *a)
{
	if (a->inf) { memset(r, 0, sizeof(*r)); r->inf = 1; return; }
	fe zi, zi2, zi3;
	fe_inv(&zi, &a->z); fe_sqr(&zi2, &zi); fe_mul(&zi3, &zi2, &zi);
	fe_mul(&r->x, &a->x, &zi2); fe_mul(&r->y, &a->y, &zi3); r->inf = 0;
}
static void ge_neg(ge *r, const ge *a) { *r = *a; fe_neg(&r->y, &a->y); }
static void gej_neg(gej *r, const gej *a) { *r = *a; fe_neg(&r->y, &a->y); }

/* ---- static table for G: 32 windows of 8 bits, entry [w][d-1] = d * 2^(8w) * G (affine) */
static ge (*g_tbl)[255];
static volatile int g_ready;

void orc_init(void)
{
	if (g_ready) return;
#pragma omp critical(orc_init_lock)
	{
		if (!g_ready) {
			ge(*tbl)[255] = malloc(sizeof(ge) * 32 * 255);
			gej base;
			gej_set_ge(&base, &GE_G);
			for (int w = 0; w < 32; w++) {
				gej acc = base;
				for (int d = 1; d <= 255; d++) {
					ge_set_gej(&tbl[w][d - 1], &acc);
					gej_add(&acc, &acc, &base);
				}
				base = acc; /* 256 * base */
			}
			g_tbl = tbl;
			__sync_synchronize();
			g_ready = 1;
		}
	}
}

/* r = u1*G + u2*Q (either scalar may be NULL = 0).  Strauss is not needed for an oracle:
 * the G half comes from the static table, the Q half is wNAF-5 over 256 doublings. */
static void ecmult(gej *r, const sc *u1, const sc *u2, const ge *q)
{
	gej acc;
	gej_set_inf(&acc);
	if (u2 && !q->inf && !u256_is_zero(u2)) {
		/* odd multiples 1,3,..,15 of Q */
		gej tbl[8], q2, qj;
		gej_set_ge(&qj, q);
		tbl[0] = qj;
		gej_double(&q2, &qj);
		for (int i = 1; i < 8; i++) gej_add(&tbl[i], &tbl[i - 1], &q2);
		/* wNAF, width 5 */
		int naf[260];
		int len = 0;
		u256 k = *u2;
		while (!u256_is_zero(&k)) {
			int d = 0;
			if (k.d[0] & 1) {
				d = (int)(k.d[0] & 31);
				if (d >= 16) d -= 32;
				u256 dd = {{(u64)(d < 0 ? -d : d), 0, 0, 0}}, t;
				if (d < 0) u256_add(&t, &k, &dd); else u256_sub(&t, &k, &dd);
				k = t;
			}
			naf[len++] = d;
			/* k >>= 1 */
			for (int i = 0; i < 3; i++) k.d[i] = (k.d[i] >> 1) | (k.d[i + 1] << 63);
			k.d[3] >>= 1;
		}
		for (int i = len - 1; i >= 0; i--) {
			gej_double(&acc, &acc);
			int d = naf[i];
			if (d > 0) gej_add(&acc, &acc, &tbl[(d - 1) / 2]);
			else if (d < 0) { gej n; gej_neg(&n, &tbl[(-d - 1) / 2]); gej_add(&acc, &acc, &n); }
		}
	}
	if (u1) {
		orc_init();
		for (int w = 0; w < 32; w++) {
			unsigned d = (unsigned)(u1->d[w / 8] >> ((w % 8) * 8)) & 255;
			if (d) gej_add_ge(&acc, &acc, &g_tbl[w][d - 1]);
		}
	}
	*r = acc;
}

/* ------------------------------------------------------------------ SHA-256 (FIPS 180-4) */
static const uint32_t K256[64] = {
	0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5,
	0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
	0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
	0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
	0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
	0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
	0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
	0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
#define ROR(x, n) (((x) >> (n)) | ((x) << (32 - (n))))
static void sha256_block(uint32_t st[8], const uint8_t blk[64])
{
	uint32_t w[64];
	for (int i = 0; i < 16; i++)
		w[i] = ((uint32_t)blk[4 * i] << 24) | ((uint32_t)blk[4 * i + 1] << 16) | ((uint32_t)blk[4 * i + 2] << 8) | blk[4 * i + 3];
	for (int i = 16; i < 64; i++) {
		uint32_t s0 = ROR(w[i - 15], 7) ^ ROR(w[i - 15], 18) ^ (w[i - 15] >> 3);
		uint32_t s1 = ROR(w[i - 2], 17) ^ ROR(w[i - 2], 19) ^ (w[i - 2] >> 10);
		w[i] = w[i - 16] + s0 + w[i - 7] + s1;
	}
	uint32_t a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
	for (int i = 0; i < 64; i++) {
		uint32_t t1 = h + (ROR(e, 6) ^ ROR(e, 11) ^ ROR(e, 25)) + ((e & f) ^ (~e & g)) + K256[i] + w[i];
		uint32_t t2 = (ROR(a, 2) ^ ROR(a, 13) ^ ROR(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
		h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
	}
	st[0] += a; st[1] += b; st[2] += c; st[3] += d; st[4] += e; st[5] += f; st[6] += g; st[7] += h;
}
void orc_sha256(const uint8_t *data, size_t len, uint8_t out32[32])
{
	uint32_t st[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
	size_t off = 0;
	for (; off + 64 <= len; off += 64) sha256_block(st, data + off);
	uint8_t tail[128] = {0};
	size_t rem = len - off;
	memcpy(tail, data + off, rem);
	tail[rem] = 0x80;
	size_t tl = (rem + 9 <= 64) ? 64 : 128;
	u64 bits = (u64)len * 8;
	for (int i = 0; i < 8; i++) tail[tl - 1 - i] = (uint8_t)(bits >> (8 * i));
	sha256_block(st, tail);
	if (tl == 128) sha256_block(st, tail + 64);
	for (int i = 0; i < 8; i++) {
		out32[4 * i] = (uint8_t)(st[i] >> 24); out32[4 * i + 1] = (uint8_t)(st[i] >> 16);
		out32[4 * i + 2] = (uint8_t)(st[i] >> 8); out32[4 * i + 3] = (uint8_t)st[i];
	}
}
void orc_sha256d(const uint8_t *data, size_t len, uint8_t out32[32])
{
	uint8_t h[32];
	orc_sha256(data, len, h);
	orc_sha256(h, 32, out32);
}
static void tagged_hash(const char *tag, const uint8_t *msg, size_t len, uint8_t out32[32])
{
	uint8_t buf[64 + 160];
	orc_sha256((const uint8_t *)tag, strlen(tag), buf);
	memcpy(buf + 32, buf, 32);
	memcpy(buf + 64, msg, len);
	orc_sha256(buf, 64 + len, out32);
}

/* ------------------------------------------------------------------ parsing */
static int pubkey_parse_ge(ge *q, const uint8_t *pub, size_t len)
{
	q->inf = 0;
	if (len == 33 && (pub[0] == 2 || pub[0] == 3)) {
		fe c, y;
		if (!fe_from_be(&q->x, pub + 1)) return 0;
		fe_sqr(&c, &q->x); fe_mul(&c, &c, &q->x); fe_add(&c, &c, &FE_SEVEN);
		if (!fe_sqrt(&y, &c)) return 0;
		if ((int)(y.d[0] & 1) != (pub[0] & 1)) fe_neg(&y, &y);
		q->y = y;
		return 1;
	}
	if (len == 65 && (pub[0] == 4 || pub[0] == 6 || pub[0] == 7)) {
		if (!fe_from_be(&q->x, pub + 1) || !fe_from_be(&q->y, pub + 33)) return 0;
		if (pub[0] != 4 && (int)(q->y.d[0] & 1) != (pub[0] & 1)) return 0;
		return ge_on_curve(q);
	}
	return 0;
}
int orc_pubkey_parse(const uint8_t *pub, size_t len, uint8_t out64[64])
{
	ge q;
	if (!pubkey_parse_ge(&q, pub, len)) return 0;
	u256_to_be(out64, &q.x); u256_to_be(out64 + 32, &q.y);
	return 1;
}
int orc_sig_parse_compact(const uint8_t sig64[64])
{
	sc r, s;
	return !sc_from_be(&r, sig64) && !sc_from_be(&s, sig64 + 32);
}

static int der_read_len(size_t *out, const uint8_t **p, const uint8_t *end)
{
	if (*p >= end) return 0;
	unsigned b1 = *((*p)++);
	if (b1 == 0xFF) return 0;
	if (!(b1 & 0x80)) { *out = b1; return 1; }
	if (b1 == 0x80) return 0;
	size_t lenleft = b1 & 0x7F;
	if (lenleft > (size_t)(end - *p)) return 0;
	if (**p == 0) return 0;
	if (lenleft > sizeof(size_t)) return 0;
	size_t ret = 0;
	while (lenleft > 0) { ret = (ret << 8) | **p; (*p)++; lenleft--; }
	if (ret > (size_t)(end - *p)) return 0;
	if (ret < 128) return 0;
	*out = ret;
	return 1;
}
static int der_parse_integer(uint8_t out32[32], const uint8_t **p, const uint8_t *end)
{
	size_t rlen;
	int overflow = 0;
	if (*p == end || **p != 0x02) return 0;
	(*p)++;
	if (!der_read_len(&rlen, p, end)) return 0;
	if (rlen == 0 || rlen > (size_t)(end - *p)) return 0;
	if ((*p)[0] == 0x00 && rlen > 1 && ((*p)[1] & 0x80) == 0x00) return 0;
	if ((*p)[0] == 0xFF && rlen > 1 && ((*p)[1] & 0x80) == 0x80) return 0;
	if ((*p)[0] & 0x80) overflow = 1;
	const uint8_t *s = *p;
	size_t l = rlen;
	if (l > 0 && s[0] == 0) { l--; s++; }
	if (l > 32) overflow = 1;
	memset(out32, 0, 32);
	if (!overflow) {
		memcpy(out32 + 32 - l, s, l);
		sc v;
		if (sc_from_be(&v, out32)) overflow = 1;
	}
	if (overflow) memset(out32, 0, 32);
	*p += rlen;
	return 1;
}
int orc_sig_parse_der(const uint8_t *der, size_t len, uint8_t out64[64])
{
	const uint8_t *p = der, *end = der + len;
	size_t rlen;
	if (p == end || *(p++) != 0x30) return 0;
	if (!der_read_len(&rlen, &p, end)) return 0;
	if (rlen != (size_t)(end - p)) return 0;
	if (!der_parse_integer(out64, &p, end)) return 0;
	if (!der_parse_integer(out64 + 32, &p, end)) return 0;
	return p == end;
}
int orc_signature_from_der(const uint8_t *der, size_t len, uint8_t out64[64], int *sighash_type)
{
	if (len < 1) return 0;
	if (!orc_sig_parse_der(der, len - 1, out64)) return 0;
	*sighash_type = der[len - 1];
	return der[len - 1] == 0x01 || der[len - 1] == 0x83;
}

/* ------------------------------------------------------------------ verification */
static int ecdsa_verify_parsed(const uint8_t hash32[32], const sc *r, const sc *s, const ge *q)
{
	if (u256_is_zero(r) || u256_is_zero(s)) return 0;
	if (u256_cmp(s, &SC_HALF_N) > 0) return 0; /* low-S only */
	sc z, w, u1, u2;
	sc_from_be(&z, hash32);
	sc_inv(&w, s);
	sc_mul(&u1, &z, &w);
	sc_mul(&u2, r, &w);
	gej R;
	ecmult(&R, &u1, &u2, q);
	if (R.inf) return 0;
	/* x(R) mod n == r, without inversion: r*Z^2 == X, or (r+n)*Z^2 == X when r + n < p */
	fe z2, t;
	fe_sqr(&z2, &R.z);
	fe_mul(&t, r, &z2); /* r < n < p is a valid fe */
	if (u256_cmp(&t, &R.x) == 0) return 1;
	u256 rn;
	if (u256_add(&rn, r, &SC_N)) return 0;
	if (u256_cmp(&rn, &FE_P) >= 0) return 0;
	fe_mul(&t, &rn, &z2);
	return u256_cmp(&t, &R.x) == 0;
}
int orc_ecdsa_verify(const uint8_t hash32[32], const uint8_t sig64[64], const uint8_t *pub, size_t publen)
{
	sc r, s;
	ge q;
	if (sc_from_be(&r, sig64) || sc_from_be(&s, sig64 + 32)) return 0;
	if (!pubkey_parse_ge(&q, pub, publen)) return 0;
	return ecdsa_verify_parsed(hash32, &r, &s, &q);
}
int orc_schnorr_verify(const uint8_t msg32[32], const uint8_t xonly32[32], const uint8_t sig64[64])
{
	uint8_t pk33[33];
	ge pk;
	fe rx;
	sc s, e, ne;
	pk33[0] = 2;
	memcpy(pk33 + 1, xonly32, 32);
	if (!pubkey_parse_ge(&pk, pk33, 33)) return 0; /* lift_x: even y */
	if (!fe_from_be(&rx, sig64)) return 0;
	if (sc_from_be(&s, sig64 + 32)) return 0;
	uint8_t buf[96], eh[32];
	memcpy(buf, sig64, 32); memcpy(buf + 32, xonly32, 32); memcpy(buf + 64, msg32, 32);
	tagged_hash("BIP0340/challenge", buf, 96, eh);
	sc_from_be(&e, eh);
	sc_neg(&ne, &e);
	gej R;
	ecmult(&R, &s, &ne, &pk);
	if (R.inf) return 0;
	ge Ra;
	ge_set_gej(&Ra, &R);
	if (Ra.y.d[0] & 1) return 0;
	return u256_cmp(&Ra.x, &rx) == 0;
}

/* secp256k1_ecdsa_recoverable_signature_parse_compact + secp256k1_ecdsa_recover as called at common/bolt11.c:1021-1046 and
 * lightningd/signmessage.c:193 (SEC1 4.1.6): Q = (s/r)*R - (z/r)*G, R = the point with x = r (+ n if recid & 2) and
 * y parity recid & 1.  Fails like the library: r or s >= n or zero, recid outside 0..3, r + n >= p, no such point, Q = inf.
 * No low-S rule.  out33 = compressed Q. */
int orc_ecdsa_recover(const uint8_t hash32[32], const uint8_t sig64[64], int recid, uint8_t out33[33])
{
	sc r, s, z, ri, u1, u2;
	u256 x;
	uint8_t k33[33];
	ge R, Qa;
	gej Q;
	memset(out33, 0, 33);
	if (recid < 0 || recid > 3) return 0;
	if (sc_from_be(&r, sig64) || sc_from_be(&s, sig64 + 32)) return 0;
	if (u256_is_zero(&r) || u256_is_zero(&s)) return 0;
	x = r;
	if (recid & 2) {
		if (u256_add(&x, &r, &SC_N)) return 0;
		if (u256_cmp(&x, &FE_P) >= 0) return 0;
	}
	k33[0] = (uint8_t)(2 + (recid & 1));
	u256_to_be(k33 + 1, &x);
	if (!pubkey_parse_ge(&R, k33, 33)) return 0;
	sc_from_be(&z, hash32);
	sc_inv(&ri, &r);
	sc_mul(&u1, &z, &ri);
	sc_neg(&u1, &u1);
	sc_mul(&u2, &s, &ri);
	ecmult(&Q, &u1, &u2, &R);
	if (Q.inf) return 0;
	ge_set_gej(&Qa, &Q);
	out33[0] = (uint8_t)(2 + (Qa.y.d[0] & 1));
	u256_to_be(out33 + 1, &Qa.x);
	return 1;
}
void orc_ecdsa_recover_batch(size_t n, const uint8_t *hash32, const uint8_t *sig64, const uint8_t *recid, uint8_t *pub33,
			     uint8_t *ok, int nthreads)
{
	orc_init();
	long i;
#pragma omp parallel for schedule(dynamic, 16) num_threads(nthreads > 0 ? nthreads : 1)
	for (i = 0; i < (long)n; i++)
		ok[i] = (uint8_t)orc_ecdsa_recover(hash32 + 32 * i, sig64 + 64 * i, recid[i], pub33 + 33 * i);
}

/* ------------------------------------------------------------------ gossip veneer */
int orc_sigcheck_channel_announcement(const uint8_t *msg, size_t len)
{
	if (len < 260 || msg[0] != 0x01 || msg[1] != 0x00) return -1;
	size_t flen = ((size_t)msg[258] << 8) | msg[259];
	size_t koff = 260 + flen + 32 + 8;
	if (len < koff + 4 * 33) return -1;
	for (int i = 0; i < 4; i++)
		if (!orc_sig_parse_compact(msg + 2 + 64 * i)) return -1; /* wire/fromwire.c:196-198 */
	uint8_t tmp[64];
	if (!orc_pubkey_parse(msg + koff + 66, 33, tmp) || !orc_pubkey_parse(msg + koff + 99, 33, tmp))
		return -1; /* fromwire_pubkey, bitcoin/pubkey.c:102-113 */
	uint8_t h[32];
	orc_sha256d(msg + 258, len - 258, h); /* sigcheck.c:73-76 */
	for (int i = 0; i < 4; i++) /* sigcheck.c:78,87,96,105: in order, first failure wins */
		if (!orc_ecdsa_verify(h, msg + 2 + 64 * i, msg + koff + 33 * i, 33)) return i + 1;
	return 0;
}
/* BigSize (BOLT #1) as common/bigsize.c:53-104 reads it: 0 = truncated or not minimally encoded, else bytes consumed */
static size_t bigsize_read(const uint8_t *p, size_t max, uint64_t *val)
{
	if (max < 1) return 0;
	if (p[0] < 0xfd) { *val = p[0]; return 1; }
	if (p[0] == 0xfd) {
		if (max < 3) return 0;
		*val = ((uint64_t)p[1] << 8) | p[2];
		return *val < 0xfd ? 0 : 3;
	}
	if (p[0] == 0xfe) {
		if (max < 5) return 0;
		*val = ((uint64_t)p[1] << 24) | ((uint64_t)p[2] << 16) | ((uint64_t)p[3] << 8) | p[4];
		return (*val >> 16) == 0 ? 0 : 5;
	}
	if (max < 9) return 0;
	*val = 0;
	for (int i = 1; i <= 8; i++) *val = (*val << 8) | p[i];
	return (*val >> 32) == 0 ? 0 : 9;
}
/* the node_ann_tlvs stream that ends a node_announcement (wire/peer_wire.csv:367-369), by the rules of
 * fromwire_tlv (wire/tlvstream.c:144-300): types strictly increasing, BigSize minimal, length inside the message,
 * unknown even types fail, unknown odd ones are skipped; the one known record, 1 = option_will_fund, is a lease_rates
 * (peer_wire.csv:214-219: u16 u16 u16 u32 tu32 -- 10 to 14 bytes, the tu32 without a leading zero byte,
 * wire/fromwire.c:115-150) */
static int node_ann_tlvs_ok(const uint8_t *p, size_t max)
{
	int first = 1;
	uint64_t prev = 0;
	while (max > 0) {
		uint64_t type, length;
		size_t l = bigsize_read(p, max, &type);
		if (!l) return 0;
		p += l; max -= l;
		if (!first && type <= prev) return 0;
		first = 0; prev = type;
		if (type != 1 && (type & 1) == 0) return 0;
		l = bigsize_read(p, max, &length);
		if (!l) return 0;
		p += l; max -= l;
		if (length > max) return 0;
		if (type == 1) {
			if (length < 10 || length > 14) return 0;
			if (length > 10 && p[10] == 0) return 0;
		}
		p += length; max -= (size_t)length;
	}
	return 1;
}
int orc_sigcheck_channel_update(const uint8_t *msg, size_t len, const uint8_t node_id33[33])
{
	/* fromwire_channel_update reads every fixed field (peer_wire.csv:370-381: 138 bytes); trailing bytes are tolerated
	 * and signed */
	if (len < 138 || msg[0] != 0x01 || msg[1] != 0x02) return -1;
	if (!orc_sig_parse_compact(msg + 2)) return -1;
	uint8_t h[32];
	orc_sha256d(msg + 66, len - 66, h); /* sigcheck.c:30-33 */
	return orc_ecdsa_verify(h, msg + 2, node_id33, 33) ? 0 : 1;
}
int orc_sigcheck_node_announcement(const uint8_t *msg, size_t len)
{
	if (len < 68 || msg[0] != 0x01 || msg[1] != 0x01) return -1;
	if (!orc_sig_parse_compact(msg + 2)) return -1;
	size_t flen = ((size_t)msg[66] << 8) | msg[67];
	size_t off = 68 + flen + 4;
	/* node_id | rgb_color 3 | alias 32 | addrlen u16 | addresses | node_ann_tlvs (peer_wire.csv:357-367) */
	if (len < off + 33 + 3 + 32 + 2) return -1;
	size_t addrlen = ((size_t)msg[off + 68] << 8) | msg[off + 69];
	if (len < off + 70 + addrlen) return -1;
	if (!node_ann_tlvs_ok(msg + off + 70 + addrlen, len - (off + 70 + addrlen))) return -1;
	uint8_t h[32];
	orc_sha256d(msg + 66, len - 66, h); /* sigcheck.c:138-141 */
	return orc_ecdsa_verify(h, msg + 2, msg + off, 33) ? 0 : 1;
}

/* ------------------------------------------------------------------ batch drivers */
void orc_ecdsa_verify_batch(size_t n, const uint8_t *hash32, const uint8_t *sig64, const uint8_t *pub,
			    size_t publen, uint8_t *out, int nthreads)
{
	orc_init();
	long i;
#pragma omp parallel for schedule(dynamic, 16) num_threads(nthreads > 0 ? nthreads : 1)
	for (i = 0; i < (long)n; i++)
		out[i] = (uint8_t)orc_ecdsa_verify(hash32 + 32 * i, sig64 + 64 * i, pub + publen * i, publen);
}
void orc_schnorr_verify_batch(size_t n, const uint8_t *msg32, const uint8_t *xonly32, const uint8_t *sig64,
			      uint8_t *out, int nthreads)
{
	orc_init();
	long i;
#pragma omp parallel for schedule(dynamic, 16) num_threads(nthreads > 0 ? nthreads : 1)
	for (i = 0; i < (long)n; i++)
		out[i] = (uint8_t)orc_schnorr_verify(msg32 + 32 * i, xonly32 + 32 * i, sig64 + 64 * i);
}

/* per-message gossip verdicts for a concatenated batch (kind from the 2-byte type; node_ids33[i] is the signer of a
 * channel_update) */
void orc_sigcheck_gossip_batch(size_t n, const uint8_t *msgs, const uint64_t *off, const uint8_t *node_ids33, int8_t *out,
			       int nthreads)
{
	orc_init();
	long i;
#pragma omp parallel for schedule(dynamic, 16) num_threads(nthreads > 0 ? nthreads : 1)
	for (i = 0; i < (long)n; i++) {
		const uint8_t *m = msgs + off[i];
		size_t len = (size_t)(off[i + 1] - off[i]);
		int v = -1;
		if (len >= 2 && m[0] == 1 && m[1] == 0) v = orc_sigcheck_channel_announcement(m, len);
		else if (len >= 2 && m[0] == 1 && m[1] == 1) v = orc_sigcheck_node_announcement(m, len);
		else if (len >= 2 && m[0] == 1 && m[1] == 2) v = orc_sigcheck_channel_update(m, len, node_ids33 + 33 * i);
		out[i] = (int8_t)v;
	}
}

/* ------------------------------------------------------------------ signing (vector generation only) */
static int seckey_load(sc *d, const uint8_t seckey32[32])
{
	return !sc_from_be(d, seckey32) && !u256_is_zero(d);
}
int orc_pubkey_create(const uint8_t seckey32[32], uint8_t out65[65])
{
	sc d;
	gej pj;
	ge p;
	if (!seckey_load(&d, seckey32)) return 0;
	ecmult(&pj, &d, NULL, NULL);
	ge_set_gej(&p, &pj);
	out65[0] = 4;
	u256_to_be(out65 + 1, &p.x); u256_to_be(out65 + 33, &p.y);
	return 1;
}
int orc_ecdsa_sign(const uint8_t hash32[32], const uint8_t seckey32[32], const uint8_t nonce32[32], uint8_t sig64[64])
{
	sc d, k, z, r, s, t;
	gej Rj;
	ge R;
	if (!seckey_load(&d, seckey32) || !seckey_load(&k, nonce32)) return 0;
	ecmult(&Rj, &k, NULL, NULL);
	ge_set_gej(&R, &Rj);
	uint8_t xb[32];
	u256_to_be(xb, &R.x);
	sc_from_be(&r, xb);
	if (u256_is_zero(&r)) return 0;
	sc_from_be(&z, hash32);
	sc_mul(&t, &r, &d); sc_add(&t, &t, &z);
	sc_inv(&s, &k); sc_mul(&s, &s, &t);
	if (u256_is_zero(&s)) return 0;
	if (u256_cmp(&s, &SC_HALF_N) > 0) sc_neg(&s, &s);
	u256_to_be(sig64, &r); u256_to_be(sig64 + 32, &s);
	return 1;
}
int orc_schnorr_sign(const uint8_t msg32[32], const uint8_t seckey32[32], const uint8_t aux32[32], uint8_t sig64[64])
{
	sc d, k, e, s;
	gej Pj, Rj;
	ge Pa, Ra;
	uint8_t buf[96], h[32], px[32], db[32];
	if (!seckey_load(&d, seckey32)) return 0;
	ecmult(&Pj, &d, NULL, NULL);
	ge_set_gej(&Pa, &Pj);
	if (Pa.y.d[0] & 1) sc_neg(&d, &d);
	u256_to_be(px, &Pa.x);
	u256_to_be(db, &d);
	tagged_hash("BIP0340/aux", aux32, 32, h);
	for (int i = 0; i < 32; i++) buf[i] = db[i] ^ h[i];
	memcpy(buf + 32, px, 32); memcpy(buf + 64, msg32, 32);
	tagged_hash("BIP0340/nonce", buf, 96, h);
	sc_from_be(&k, h);
	if (u256_is_zero(&k)) return 0;
	ecmult(&Rj, &k, NULL, NULL);
	ge_set_gej(&Ra, &Rj);
	if (Ra.y.d[0] & 1) sc_neg(&k, &k);
	u256_to_be(sig64, &Ra.x);
	memcpy(buf, sig64, 32); memcpy(buf + 32, px, 32); memcpy(buf + 64, msg32, 32);
	tagged_hash("BIP0340/challenge", buf, 96, h);
	sc_from_be(&e, h);
	sc_mul(&s, &e, &d); sc_add(&s, &s, &k);
	u256_to_be(sig64 + 32, &s);
	(void)ge_neg;
	return 1;
}

// SHA-256 compression for one message per lane (FIPS 180-4), used for the BIP-340 challenge
// e = SHA256(SHA256(tag)||SHA256(tag)||r||pk||m) with the tag block folded into a constant
// midstate, and for the double-SHA256 of gossip message tails.
// Semantics replaced: ccan/ccan/crypto/sha256/sha256.c:87-250, bitcoin/shadouble.c:7-11,
// bip340_sighash_init() bitcoin/signature.c:389-405 (tag "BIP0340/challenge" is libsecp256k1's).
#pragma once
#include "lamd_common.h"

namespace lamd {

#define LAMD_SHA256_IV {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au, 0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u}
// state after absorbing SHA256("BIP0340/challenge") twice (one 64-byte block)
#define LAMD_BIP340_CHALLENGE_MIDSTATE {0x9cecba11u, 0x23925381u, 0x11679112u, 0xd1627e0fu, 0x97c87550u, 0x003cc765u, 0x90f61164u, 0x33e9b66au}

LAMD_HD u32 sha_ror(u32 x, int n) { return (x >> n) | (x << (32 - n)); }

#if defined(__HIP_DEVICE_COMPILE__)
__device__ __constant__ static const u32 SHA256_K[64] = {
#else
static const u32 SHA256_K[64] = {
#endif
    0x428a2f98u, 0x71374491u, 0xb5c0fbcfu, 0xe9b5dba5u, 0x3956c25bu, 0x59f111f1u, 0x923f82a4u, 0xab1c5ed5u,
    0xd807aa98u, 0x12835b01u, 0x243185beu, 0x550c7dc3u, 0x72be5d74u, 0x80deb1feu, 0x9bdc06a7u, 0xc19bf174u,
    0xe49b69c1u, 0xefbe4786u, 0x0fc19dc6u, 0x240ca1ccu, 0x2de92c6fu, 0x4a7484aau, 0x5cb0a9dcu, 0x76f988dau,
    0x983e5152u, 0xa831c66du, 0xb00327c8u, 0xbf597fc7u, 0xc6e00bf3u, 0xd5a79147u, 0x06ca6351u, 0x14292967u,
    0x27b70a85u, 0x2e1b2138u, 0x4d2c6dfcu, 0x53380d13u, 0x650a7354u, 0x766a0abbu, 0x81c2c92eu, 0x92722c85u,
    0xa2bfe8a1u, 0xa81a664bu, 0xc24b8b70u, 0xc76c51a3u, 0xd192e819u, 0xd6990624u, 0xf40e3585u, 0x106aa070u,
    0x19a4c116u, 0x1e376c08u, 0x2748774cu, 0x34b0bcb5u, 0x391c0cb3u, 0x4ed8aa4au, 0x5b9cca4fu, 0x682e6ff3u,
    0x748f82eeu, 0x78a5636fu, 0x84c87814u, 0x8cc70208u, 0x90befffau, 0xa4506cebu, 0xbef9a3f7u, 0xc67178f2u};

#if !defined(__HIP_DEVICE_COMPILE__) && defined(__x86_64__) && !defined(LAMD_NO_SHA_NI)
}  // namespace lamd
#include <cpuid.h>
#include <immintrin.h>
namespace lamd {
#define LAMD_SHA_NI 1
// Host side only: the x86 SHA extensions where the CPU has them (checked once with cpuid; every EPYC and every Xeon since Ice Lake does).  The host hashes
// where the device would be the wrong tool -- the 20 KB output list of a commitment transaction while lamd_check_commitment_signed packs its rows (one lane
// would need milliseconds, lamd_engine.hip txsig_pack), the handful of rows of a latency-path call -- and that hash sits in front of the launch: 117 us for
// the commitment row with the portable rounds fed byte by byte, 19 us with these instructions and whole-block feeding on the same core.  Same function, same words in and out; the portable
// rounds remain for other CPUs (-DLAMD_NO_SHA_NI forces them: tests/test_devmath_host.py runs both against the same vectors).
static inline bool sha256_have_ni() {
  static const int have = [] {
    unsigned a, b, c, d;
    if (!__get_cpuid_count(7, 0, &a, &b, &c, &d) || !((b >> 29) & 1u)) return 0;   // CPUID.7.0:EBX.SHA
    if (!__get_cpuid(1, &a, &b, &c, &d)) return 0;
    return (int)(((c >> 19) & 1u) && ((c >> 9) & 1u));                              // SSE4.1, SSSE3
  }();
  return have != 0;
}
__attribute__((target("sha,sse4.1,ssse3"))) static inline void sha256_compress_ni(u32 st[8], const u32 w[16]) {
  __m128i tmp = _mm_shuffle_epi32(_mm_loadu_si128((const __m128i *)st), 0xB1);            // CDAB
  __m128i s1 = _mm_shuffle_epi32(_mm_loadu_si128((const __m128i *)(st + 4)), 0x1B);       // EFGH
  __m128i s0 = _mm_alignr_epi8(tmp, s1, 8);                                               // ABEF
  s1 = _mm_blend_epi16(s1, tmp, 0xF0);                                                    // CDGH
  const __m128i save0 = s0, save1 = s1;
  __m128i m[4];
  for (int i = 0; i < 4; i++) m[i] = _mm_loadu_si128((const __m128i *)(w + 4 * i));      // the words are big-endian values already
  for (int i = 0; i < 16; i++) {
    __m128i cur;
    if (i < 4) {
      cur = m[i];
    } else {  // W[4i .. 4i+3] from the sixteen words before them
      cur = _mm_sha256msg2_epu32(_mm_add_epi32(_mm_sha256msg1_epu32(m[0], m[1]), _mm_alignr_epi8(m[3], m[2], 4)), m[3]);
      m[0] = m[1]; m[1] = m[2]; m[2] = m[3]; m[3] = cur;
    }
    __m128i t = _mm_add_epi32(cur, _mm_loadu_si128((const __m128i *)&SHA256_K[4 * i]));
    s1 = _mm_sha256rnds2_epu32(s1, s0, t);
    t = _mm_shuffle_epi32(t, 0x0E);
    s0 = _mm_sha256rnds2_epu32(s0, s1, t);
  }
  s0 = _mm_add_epi32(s0, save0);
  s1 = _mm_add_epi32(s1, save1);
  tmp = _mm_shuffle_epi32(s0, 0x1B);                                                      // FEBA
  s1 = _mm_shuffle_epi32(s1, 0xB1);                                                       // DCHG
  _mm_storeu_si128((__m128i *)st, _mm_blend_epi16(tmp, s1, 0xF0));                        // DCBA
  _mm_storeu_si128((__m128i *)(st + 4), _mm_alignr_epi8(s1, tmp, 8));                     // HGFE
}
#endif

// one compression; w[16] = the block as big-endian words (clobbered)
LAMD_HD void sha256_compress(u32 st[8], u32 w[16]) {
#if defined(LAMD_SHA_NI)
  if (sha256_have_ni()) {
    sha256_compress_ni(st, w);
    return;
  }
#endif
  u32 a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
#pragma unroll
  for (int i = 0; i < 64; i++) {
    if (i >= 16) {
      const u32 w15 = w[(i - 15) & 15], w2 = w[(i - 2) & 15];
      const u32 s0 = sha_ror(w15, 7) ^ sha_ror(w15, 18) ^ (w15 >> 3);
      const u32 s1 = sha_ror(w2, 17) ^ sha_ror(w2, 19) ^ (w2 >> 10);
      w[i & 15] = w[i & 15] + s0 + w[(i - 7) & 15] + s1;
    }
    const u32 t1 = h + (sha_ror(e, 6) ^ sha_ror(e, 11) ^ sha_ror(e, 25)) + ((e & f) ^ (~e & g)) + SHA256_K[i] + w[i & 15];
    const u32 t2 = (sha_ror(a, 2) ^ sha_ror(a, 13) ^ sha_ror(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
    h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
  }
  st[0] += a; st[1] += b; st[2] += c; st[3] += d; st[4] += e; st[5] += f; st[6] += g; st[7] += h;
}

// BIP-340 challenge hash of r32 || pk32 || msg32, each given as 8 big-endian words
// (i.e. word 0 = first four bytes).  Output digest as 8 big-endian words.
LAMD_HD void bip340_challenge(u32 out[8], const u32 r_be[8], const u32 pk_be[8], const u32 m_be[8]) {
  u32 st[8] = LAMD_BIP340_CHALLENGE_MIDSTATE;
  u32 w[16];
#pragma unroll
  for (int i = 0; i < 8; i++) { w[i] = r_be[i]; w[8 + i] = pk_be[i]; }
  sha256_compress(st, w);
#pragma unroll
  for (int i = 0; i < 8; i++) w[i] = m_be[i];
  w[8] = 0x80000000u;
#pragma unroll
  for (int i = 9; i < 15; i++) w[i] = 0;
  w[15] = 160u * 8u;  // 64 (tag block) + 96 bytes
  sha256_compress(st, w);
#pragma unroll
  for (int i = 0; i < 8; i++) out[i] = st[i];
}

}  // namespace lamd

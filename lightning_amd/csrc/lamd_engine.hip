// liblightning_amd.so: HIP kernels (gfx950) + the C-ABI engine of include/lightning_amd.h.
// One signature per wavefront lane in every kernel; the arithmetic lives in verify_core.h.
// There is deliberately no CPU verification path in this library.
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <mutex>
#include <random>
#include <string>
#include <atomic>
#include <thread>
#include <vector>

#include "../../include/lightning_amd.h"
#include "../../include/lightning_amd_debug.h"
#include "verify_core.h"
#include "fuzz.h"

using namespace lamd;

constexpr int MAX_LANES = 8;

// Wave priority (s_setprio) of the kernels that run UNDER the big table-driven ecmult launches of the other lanes: a front-end / ladder wave
// that shares a SIMD with three ecmult waves gets the issue port first, so a call's dependent chain of small launches (and its handful of
// latency-bound ladder waves) stops being stretched by work that has a whole launch to hide in.  g_prio is a bit mask set once at lamd_init
// from LAMD_PRIO (default: see lamd_init): 1 front end (init / lookup / dedupe / classify / partition), 2 cold-row ladder + its key parse,
// 4 scalar preparation, 8 key-table building, 16 BIP-340 parity stage.
__device__ u32 g_prio;
// Default 13 = front end + scalar preparation + key tables (round 6, profiles/r06_strong_scaling.txt session ac: a 1/8 gossip shard 3.48 against 3.65 ms --
// the 30 waves of 1 875 node keys' doubling chains no longer wait their turn behind three ladder waves per SIMD -- and the headline loop +0.5 %; rounds 4-5
// ran with 0: there the mask cost the chained loop 1-9 %, before the front end was fused).
#ifndef LAMD_PRIO_DEFAULT
#define LAMD_PRIO_DEFAULT 13
#endif
#define LAMD_PRIO(bit)                                     \
  do {                                                     \
    if (g_prio & (bit)) __builtin_amdgcn_s_setprio(3);     \
  } while (0)

// =====================================================================================
//                                       kernels
// =====================================================================================

// ---- static G table: one thread per entry
__global__ void __launch_bounds__(256) k_gtable_build(u32 *__restrict__ gtable, const u32 *__restrict__ bases) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= GTABLE_ENTRIES) return;
  const u32 w = (u32)(idx >> GTABLE_WINDOW_BITS), d = (u32)(idx & ((1u << GTABLE_WINDOW_BITS) - 1u));
  u32 out[GT_ENTRY_WORDS];
  if (d == 0) {
#pragma unroll
    for (int i = 0; i < GT_ENTRY_WORDS; i++) out[i] = 0;
  } else {
    u32 base[16];
#pragma unroll
    for (int i = 0; i < 16; i++) base[i] = bases[w * 16 + i];
    gtable_compute_entry(out, base, d);
  }
  uint2 *dst = reinterpret_cast<uint2 *>(gtable + idx * GT_ENTRY_WORDS);  // entries are 8-byte aligned in both layouts
#pragma unroll
  for (int i = 0; i < GT_ENTRY_WORDS / 2; i++) dst[i] = make_uint2(out[2 * i], out[2 * i + 1]);
}

// ---- ECDSA scalar preparation: each thread owns signatures tid, tid+T, ... and inverts their
// s values with one modular inversion (Montgomery's trick)
__global__ void __launch_bounds__(256) k_ecdsa_prep(size_t n, const u8 *__restrict__ hash32, const u8 *__restrict__ sig64,
                                                    prep_rec *__restrict__ recs) {
  LAMD_PRIO(4);
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t T = (size_t)gridDim.x * blockDim.x;
  ecdsa_prep_thread(tid, T, n, hash32, sig64, recs);
}

__global__ void __launch_bounds__(256) k_schnorr_prep(size_t n, const u8 *__restrict__ msg32, const u8 *__restrict__ pk32,
                                                      const u8 *__restrict__ sig64, prep_rec *__restrict__ recs) {
  LAMD_PRIO(4);
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  schnorr_prep_one(msg32 + 32 * i, pk32 + 32 * i, sig64 + 64 * i, &recs[i]);
}

constexpr int FIN_WORDS = 32;  // BIP-340 stage-1 parking space per row (Y, Z, prefix) when rows are not slot-aligned

// ---- public keys: parse / decompress / validate -> 64-byte affine words + validity byte
// (idx != nullptr: work item i handles input row idx[i]; its outputs stay at position i)
// (count != nullptr: the number of work items is min(*count, n), known only on the device)
__global__ void __launch_bounds__(256) k_keys(size_t n, const u8 *__restrict__ pub, int publen, size_t stride,
                                              const u32 *__restrict__ idx, u32 *__restrict__ qwords, u8 *__restrict__ keyok,
                                              const u32 *__restrict__ count) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || (count && i >= *count)) return;
  const size_t row = idx ? idx[i] : i;
  u32 qx[8], qy[8];
  const bool ok = parse_pubkey(pub + stride * row, publen, qx, qy);
  uint4 *dst = reinterpret_cast<uint4 *>(qwords + i * 16);
  dst[0] = make_uint4(qx[0], qx[1], qx[2], qx[3]);
  dst[1] = make_uint4(qx[4], qx[5], qx[6], qx[7]);
  dst[2] = make_uint4(qy[0], qy[1], qy[2], qy[3]);
  dst[3] = make_uint4(qy[4], qy[5], qy[6], qy[7]);
  keyok[i] = ok;
}

// ---- the keys of the cold rows (keys seen too rarely for a table; the list and its length live on the device).  A row whose key
// does not parse is decided HERE (verdict 0) and only the others are compacted onto the ladder's work list: in the configs[1]/[2]
// workloads nearly every cold row is a damaged key (1.25 % of the rows: unique by construction, so never worth a table), and a
// ladder wave with one live lane costs as much as a full one -- the ladder kernel's VALU work drops from ~2 % of the rows to the
// few hundred rows under genuinely rare keys.
__device__ __forceinline__ u32 wave_alloc(u32 *counter, bool pred, u32 weight = 0, u32 *wsum = nullptr);
__global__ void __launch_bounds__(256) k_keys_cold(size_t n, const u8 *__restrict__ pub, int publen, size_t stride, const u32 *__restrict__ idx,
                                                   const u32 *__restrict__ count, u32 *__restrict__ counter_ok, u32 *__restrict__ idx_ok,
                                                   u32 *__restrict__ qwords, u8 *__restrict__ keyok_row, u8 *__restrict__ out) {
  LAMD_PRIO(2);
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = i < n && i < *count;
  const size_t row = live ? idx[i] : 0;
  u32 qx[8], qy[8];
  const bool ok = live && parse_pubkey(pub + stride * row, publen, qx, qy);
  const u32 pos = wave_alloc(counter_ok, ok, 0, nullptr);
  if (!live) return;
  if (keyok_row) keyok_row[row] = ok;
  if (!ok) {
    out[row] = 0;
    return;
  }
  idx_ok[pos] = (u32)row;
  uint4 *dst = reinterpret_cast<uint4 *>(qwords + (size_t)pos * 16);
  dst[0] = make_uint4(qx[0], qx[1], qx[2], qx[3]);
  dst[1] = make_uint4(qx[4], qx[5], qx[6], qx[7]);
  dst[2] = make_uint4(qy[0], qy[1], qy[2], qy[3]);
  dst[3] = make_uint4(qy[4], qy[5], qy[6], qy[7]);
}

// ---- the hot kernel: R = u1*G + u2*Q and the acceptance test.  WAVES = minimum waves per SIMD the register
// allocator must leave room for (2nd __launch_bounds__ argument); the engine picks the instantiation (LAMD_ECMULT_WAVES).
template <int WAVES>
__global__ void __launch_bounds__(256, WAVES) k_ecmult(size_t n, const prep_rec *__restrict__ recs, const u32 *__restrict__ qwords,
                                                const u8 *__restrict__ keyok, const u8 *__restrict__ sig64, int mode,
                                                const u32 *__restrict__ gtable, u32 *__restrict__ slots,
                                                const u32 *__restrict__ idx, u32 *__restrict__ fin, u8 *__restrict__ keyok_row,
                                                u8 *__restrict__ out, const u32 *__restrict__ count) {
  LAMD_PRIO(2);
  // idx != nullptr (cold rows of a partitioned chunk): work item i verifies input row idx[i]; key data and the table
  // slot live at position i, the prep record / signature / verdict / BIP-340 parking space (fin) at the row
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || (count && i >= *count)) return;
  const size_t row = idx ? idx[i] : i;
  prep_rec rec;
  {
    const uint4 *src = reinterpret_cast<const uint4 *>(recs + row);
    const uint4 a = src[0], b = src[1], c = src[2], d = src[3], e = src[4];
    rec.u1[0] = a.x; rec.u1[1] = a.y; rec.u1[2] = a.z; rec.u1[3] = a.w;
    rec.u1[4] = b.x; rec.u1[5] = b.y; rec.u1[6] = b.z; rec.u1[7] = b.w;
    rec.k1[0] = c.x; rec.k1[1] = c.y; rec.k1[2] = c.z; rec.k1[3] = c.w;
    rec.k2[0] = d.x; rec.k2[1] = d.y; rec.k2[2] = d.z; rec.k2[3] = d.w;
    rec.flags = e.x;
  }
  const bool key_good = keyok ? keyok[i] != 0 : true;  // keyok == nullptr: a compacted list (k_keys_cold), every listed row has a parsed key
  if (keyok_row && keyok) keyok_row[row] = key_good;
  bool ok = (rec.flags & PREP_VALID) && key_good;
  if (ok) {  // whole waves of rejected inputs skip the ladder (s_cbranch_execz)
    u32 qx[8], qy[8];
    const uint4 *src = reinterpret_cast<const uint4 *>(qwords + i * 16);
    const uint4 a = src[0], b = src[1], c = src[2], d = src[3];
    qx[0] = a.x; qx[1] = a.y; qx[2] = a.z; qx[3] = a.w; qx[4] = b.x; qx[5] = b.y; qx[6] = b.z; qx[7] = b.w;
    qy[0] = c.x; qy[1] = c.y; qy[2] = c.z; qy[3] = c.w; qy[4] = d.x; qy[5] = d.y; qy[6] = d.z; qy[7] = d.w;
    // the ladder in its hot form (signed odd digits, bare additions: verify_core.h ecmult_lane_fast); a degenerate event -- adversarial scalars, or a
    // result at infinity -- sends this lane through the complete ladder
    bool suspect;
    const ge q = ge_from_words(qx, qy);
    gej R = ecmult_lane_fast(rec, q, slots + i * SLOT_WORDS, gtable, &suspect);
    if (__builtin_expect(suspect, 0)) R = ecmult_lane(rec, q, slots + i * SLOT_WORDS, gtable);
    u32 rw[8];
    load_words_be(rw, sig64 + 64 * row);
    if (mode == MODE_ECDSA) {
      ok = ecdsa_final(R, rw);
    } else if (mode == MODE_RECOVER) {  // 0 or SCHNORR_PENDING with the Jacobian key parked in the slot (k_recover_final)
      out[row] = recover_stage1(R, slots + i * SLOT_WORDS);
      return;
    } else {  // 0 or SCHNORR_PENDING (parity decided by k_schnorr_final[_fin])
      out[row] = schnorr_stage1(R, rw, fin ? fin + row * FIN_WORDS : slots + i * SLOT_WORDS);
      return;
    }
  }
  out[row] = ok ? 1 : 0;
}

// ---- BIP-340 stage 2: shared inversion for the y-parity test
__global__ void __launch_bounds__(256) k_schnorr_final(size_t n, u32 *__restrict__ slots, u8 *__restrict__ out) {
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t T = (size_t)gridDim.x * blockDim.x;
  schnorr_final_thread(tid, T, n, slots, out);
}

// ---- public-key recovery: prep (Montgomery batch inversion of r, like the ECDSA prep) and the shared-inversion final stage
__global__ void __launch_bounds__(256) k_recover_prep(size_t n, const u8 *__restrict__ hash32, const u8 *__restrict__ sig64,
                                                      const u8 *__restrict__ recid, prep_rec *__restrict__ recs, u8 *__restrict__ rkey33) {
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t T = (size_t)gridDim.x * blockDim.x;
  recover_prep_thread(tid, T, n, hash32, sig64, recid, recs, rkey33);
}
__global__ void __launch_bounds__(256) k_recover_final(size_t n, u32 *__restrict__ slots, u8 *__restrict__ out, u8 *__restrict__ pub33) {
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t T = (size_t)gridDim.x * blockDim.x;
  recover_final_thread(tid, T, n, slots, out, pub33);
}

// ---- gossip: per message, double-SHA256 of the signed tail and expansion into (hash, sig, key) rows
// ---- check_tx_sig batches: double-SHA256 of caller-built BIP143 preimages (bitcoin/signature.c:120-151 hashes them
// through libwally) and the sighash-type gate of bitcoin/signature.c:206-211
// (txsig_hash_one / txsig_tx_hash_one / gossip_expand_one / gossip_reduce_one: verify_core.h -- shared by these kernels and by the host-side latency paths,
// and compiled for the host by tests/devmath_host.cpp)
__global__ void __launch_bounds__(256) k_txsig_hash(size_t n, const u8 *__restrict__ pre, const u64 *__restrict__ off,
                                                    const u8 *__restrict__ sighash_type, const u8 *__restrict__ has_witness,
                                                    u8 *__restrict__ hash32, u8 *__restrict__ gate) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  gate[i] = txsig_hash_one(pre + off[i], (size_t)(off[i + 1] - off[i]), sighash_type[i], has_witness[i] != 0, hash32 + 32 * i);
}
// ---- BIP143 on the device, one (transaction, input) per lane: what bip143_sighash() (verify_core.h: the host's form, byte-wise streaming) computes,
// laid out for a wavefront.  The streaming form compiled to 72 k instructions (every shs_update site carries its own unrolled compression: 580 KB of code for
// an instruction cache of 64 KB) with the block buffer in scratch because it is indexed by a run-time byte count: 226 us for the 483 HTLC rows of a
// commitment_signed, a third of it compressions.  Here each of the four streams of a row (hashPrevouts, hashSequence, hashOutputs, the preimage) is laid
// out as a padded message in a per-lane LDS buffer -- word j of lane t at w[j * TXH_LANES + t], so a wave's accesses fall into 64 different banks whatever
// position each lane is at -- and hashed by ONE loop with ONE compression site (the second hash of the double SHA-256 is that loop's last turn).  Bytes
// from global memory are fetched sixteen at a time so that their latencies overlap.  A row whose streams do not fit the buffer never gets here: txsig_pack
// hashes it on the host (TXSIG_DEV_MAX_*).
constexpr int TXH_LANES = 64;
constexpr u32 TXH_WORDS = 240;                      // per lane: 960 bytes = 15 blocks (61 440 bytes of LDS per work-group)
constexpr u32 TXH_MAX_STREAM = 4 * TXH_WORDS - 9;   // 0x80 and the 64-bit length must fit behind the message
__device__ __forceinline__ void txh_put(u32 *w, u32 pos, u32 byte) {  // message byte `pos`; words are big-endian, LDS is little-endian
  reinterpret_cast<u8 *>(w + (size_t)(pos >> 2) * TXH_LANES)[3 - (pos & 3)] = (u8)byte;
}
__device__ __forceinline__ u32 txh_copy(u32 *w, u32 pos, const u8 *__restrict__ p, u32 n) {
#pragma unroll 1
  for (u32 k = 0; k < n; k += 16) {
    u32 b[16];
#pragma unroll
    for (u32 j = 0; j < 16; j++) b[j] = p[k + j < n ? k + j : n - 1];  // sixteen loads in flight; the index is clamped, not predicated
#pragma unroll
    for (u32 j = 0; j < 16; j++)
      if (k + j < n) txh_put(w, pos + k + j, b[j]);
  }
  return pos + n;
}
__device__ __forceinline__ u32 txh_put_be32(u32 *w, u32 pos, u32 v) {
#pragma unroll
  for (int k = 0; k < 4; k++) txh_put(w, pos + k, v >> (24 - 8 * k));
  return pos + 4;
}
__device__ __forceinline__ u32 txh_put_le(u32 *w, u32 pos, u64 v, int bytes) {
  for (int k = 0; k < bytes; k++) txh_put(w, pos + k, (u32)(v >> (8 * k)));
  return pos + bytes;
}
// false: the template is inconsistent (bip143_sighash's cases) or a stream does not fit the buffer; h = SHA256d of the preimage as 8 big-endian words
__device__ __forceinline__ bool bip143_sighash_lane(u32 *w, const tx_view &t, u32 in_idx, const u8 *__restrict__ script, u32 script_len, u64 amount,
                                                    u32 sighash_type, u32 h[8]) {
  if (in_idx >= t.n_in) return false;
  if (t.outputs_len > TXH_MAX_STREAM || (u64)t.n_in * 36 > TXH_MAX_STREAM || (u64)script_len + 165 > TXH_MAX_STREAM) return false;
  const bool acp = (sighash_type & 0x80u) != 0;
  const u32 base = sighash_type & 0x1fu;
  const bool single = base == 3, none = base == 2;
  // the outputs must parse; SINGLE needs the boundaries of output [in_idx]
  size_t at = 0, one_off = 0, one_len = 0;
  for (u32 k = 0; k < t.n_out; k++) {
    if (t.outputs_len - at < 8) return false;
    u64 sl;
    const size_t l = tx_compact_size(t.outputs + at + 8, t.outputs_len - at - 8, &sl);
    if (!l || sl > t.outputs_len - at - 8 - l) return false;
    const size_t len = 8 + l + (size_t)sl;
    if (k == in_idx) { one_off = at; one_len = len; }
    at += len;
  }
  if (at != t.outputs_len) return false;
  u32 hp[8], hs[8], ho[8];
#pragma unroll
  for (int j = 0; j < 8; j++) hp[j] = hs[j] = ho[j] = h[j] = 0;
#pragma unroll 1
  for (int ph = 0; ph < 4; ph++) {
    u32 len = 0;
    bool run = true;
    if (ph == 0) {  // hashPrevouts
      run = !acp;
      if (run)
        for (u32 i = 0; i < t.n_in; i++) len = txh_copy(w, len, t.inputs + 40 * (size_t)i, 36);
    } else if (ph == 1) {  // hashSequence
      run = !acp && !single && !none;
      if (run)
        for (u32 i = 0; i < t.n_in; i++) len = txh_copy(w, len, t.inputs + 40 * (size_t)i + 36, 4);
    } else if (ph == 2) {  // hashOutputs
      if (single) {
        run = in_idx < t.n_out;
        if (run) len = txh_copy(w, 0, t.outputs + one_off, (u32)one_len);
      } else if (none) {
        run = false;
      } else {
        len = txh_copy(w, 0, t.outputs, (u32)t.outputs_len);
      }
    } else {  // nVersion | hashPrevouts | hashSequence | outpoint | varint script | amount | nSequence | hashOutputs | nLockTime | type
      len = txh_put_le(w, len, t.version, 4);
#pragma unroll
      for (int j = 0; j < 8; j++) len = txh_put_be32(w, len, hp[j]);
#pragma unroll
      for (int j = 0; j < 8; j++) len = txh_put_be32(w, len, hs[j]);
      len = txh_copy(w, len, t.inputs + 40 * (size_t)in_idx, 36);
      if (script_len < 0xfd) len = txh_put_le(w, len, script_len, 1);
      else if (script_len <= 0xffff) { len = txh_put_le(w, len, 0xfd, 1); len = txh_put_le(w, len, script_len, 2); }
      else { len = txh_put_le(w, len, 0xfe, 1); len = txh_put_le(w, len, script_len, 4); }
      len = txh_copy(w, len, script, script_len);
      len = txh_put_le(w, len, amount, 8);
      len = txh_copy(w, len, t.inputs + 40 * (size_t)in_idx + 36, 4);
#pragma unroll
      for (int j = 0; j < 8; j++) len = txh_put_be32(w, len, ho[j]);
      len = txh_put_le(w, len, t.locktime, 4);
      len = txh_put_le(w, len, sighash_type, 4);
    }
    if (!run) continue;
    // padding: 0x80, zeros to the last two words of a block, the bit length
    u32 pos = len;
    txh_put(w, pos++, 0x80u);
    while (pos & 3) txh_put(w, pos++, 0u);
    u32 wi = pos >> 2;
    for (; (wi & 15) != 14; wi++) w[(size_t)wi * TXH_LANES] = 0;
    w[(size_t)wi * TXH_LANES] = 0;
    w[(size_t)(wi + 1) * TXH_LANES] = len * 8;
    const u32 nb = (wi + 2) >> 4;
    u32 st[8] = LAMD_SHA256_IV;
#pragma unroll 1
    for (u32 b = 0; b <= nb; b++) {  // turn nb = the second hash, over the first one's digest
      u32 x[16];
      if (b < nb) {
#pragma unroll
        for (u32 j = 0; j < 16; j++) x[j] = w[(size_t)(16 * b + j) * TXH_LANES];
      } else {
        const u32 iv[8] = LAMD_SHA256_IV;
#pragma unroll
        for (int j = 0; j < 8; j++) { x[j] = st[j]; st[j] = iv[j]; }
        x[8] = 0x80000000u;
#pragma unroll
        for (int j = 9; j < 15; j++) x[j] = 0;
        x[15] = 256;
      }
      sha256_compress(st, x);
    }
#pragma unroll
    for (int j = 0; j < 8; j++) {
      if (ph == 0) hp[j] = st[j];
      else if (ph == 1) hs[j] = st[j];
      else if (ph == 2) ho[j] = st[j];
      else h[j] = st[j];
    }
  }
  return true;
}
// hash32 must be 4-byte aligned (it is a device allocation of the context)
__global__ void __launch_bounds__(TXH_LANES) k_txsig_tx_hash(size_t n, const u32 *__restrict__ version, const u32 *__restrict__ locktime,
                                                             const u8 *__restrict__ inputs40, const u64 *__restrict__ in_off,
                                                             const u32 *__restrict__ input_num, const u64 *__restrict__ amount,
                                                             const u8 *__restrict__ outputs, const u64 *__restrict__ out_off,
                                                             const u32 *__restrict__ n_outputs, const u8 *__restrict__ scripts,
                                                             const u64 *__restrict__ script_off, const u8 *__restrict__ sighash_type,
                                                             const u8 *__restrict__ has_witness, const u8 *__restrict__ host_done,
                                                             const u8 *__restrict__ host_hash, u8 *__restrict__ hash32, u8 *__restrict__ gate) {
  __shared__ u32 lds[TXH_WORDS * TXH_LANES];
  const size_t i = (size_t)blockIdx.x * TXH_LANES + threadIdx.x;
  if (i >= n) return;
  if (host_done[i]) {  // a row with a long input / output list or script: the host hashed it (txsig_pack) -- one lane would walk 20 KB of outputs for milliseconds
    for (int b = 0; b < 32; b++) hash32[32 * i + b] = host_hash[32 * i + b];
    gate[i] = host_done[i] == 1;
    return;
  }
  const u8 t = sighash_type[i];
  bool pass = t == 1 || (t == 0x83 && has_witness[i]);  // the gate of bitcoin/signature.c:206-211 (txsig_tx_hash_one)
  u32 h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (pass) {
    tx_view tv;
    tv.version = version[i];
    tv.locktime = locktime[i];
    tv.inputs = inputs40 + 40 * in_off[i];
    tv.n_in = (u32)(in_off[i + 1] - in_off[i]);
    tv.outputs = outputs + out_off[i];
    tv.outputs_len = (size_t)(out_off[i + 1] - out_off[i]);
    tv.n_out = n_outputs[i];
    const u64 sl = script_off[i + 1] - script_off[i];
    pass = sl <= TXH_MAX_STREAM && bip143_sighash_lane(lds + threadIdx.x, tv, input_num[i], scripts + script_off[i], (u32)sl, amount[i], t, h);
  }
  u32 *out = reinterpret_cast<u32 *>(hash32) + 8 * i;
#pragma unroll
  for (int j = 0; j < 8; j++) out[j] = pass ? __builtin_bswap32(h[j]) : 0u;
  gate[i] = pass;
}
// ---- BOLT #12: merkle root + tagged signature hash of one TLV stream per lane (bolt12.h); valid[i] = the stream obeys the TLV rules
__global__ void __launch_bounds__(64) k_bolt12_hash(size_t n, const u8 *__restrict__ tlvs, const u64 *__restrict__ off, bolt12_mids mids,
                                                    u8 *__restrict__ root32, u8 *__restrict__ msg32, u8 *__restrict__ valid) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u8 root[32], msg[32];
  for (int b = 0; b < 32; b++) root[b] = msg[b] = 0;
  const bool ok = bolt12_merkle_root(tlvs + off[i], (size_t)(off[i + 1] - off[i]), mids, root);
  if (ok) bolt12_sighash(mids, root, msg);
  valid[i] = ok;
  for (int b = 0; b < 32; b++) { if (root32) root32[32 * i + b] = root[b]; msg32[32 * i + b] = msg[b]; }
}
__global__ void __launch_bounds__(256) k_apply_gate(size_t n, const u8 *__restrict__ gate, u8 *__restrict__ ok) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && !gate[i]) ok[i] = 0;
}

// ---- fee grind (verify_core.h "fee grind"): one thread prepares, one thread per candidate feerate
__global__ void __launch_bounds__(64) k_grind_setup(const u8 *__restrict__ sig64, const u8 *__restrict__ pub33,
                                                    const u8 *__restrict__ pre, u32 lead_blocks, u32 *__restrict__ slot,
                                                    const u32 *__restrict__ gtable, grind_setup *__restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  grind_setup g;
  grind_prepare(&g, sig64, pub33, pre, lead_blocks, slot, gtable);
  *out = g;
}
// candidate c = feerate min_rate + c; *best = the smallest matching c (0xFFFFFFFF: none)
__global__ void __launch_bounds__(256) k_grind(u32 ncand, u32 min_rate, u64 weight, u64 input_sat, const u8 *__restrict__ tail,
                                               u32 tail_len, u32 lead_bytes, const u8 *__restrict__ outputs, u32 outputs_len,
                                               const grind_setup *__restrict__ setup, const u32 *__restrict__ gtable,
                                               u32 *__restrict__ best) {
  const u32 c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ncand || !setup->valid) return;
  if (grind_candidate(c, min_rate, weight, input_sat, tail, tail_len, lead_bytes, outputs, outputs_len, *setup, gtable)) atomicMin(best, c);
}

// rowbase[i] = index of message i's first signature row; malformed[i] set here for framing errors
// message i = msgs[off[i], off[i + 1]) -- or, with `len` (the spans form: lamd_sigcheck_gossip_spans_device), msgs[off[i], off[i] + len[i]): the messages of a
// call need not lie back to back, nor in order
__global__ void __launch_bounds__(256) k_gossip_expand(size_t n, const u8 *__restrict__ msgs, const u64 *__restrict__ off, const u64 *__restrict__ len,
                                                       const u8 *__restrict__ node_ids, const u64 *__restrict__ rowbase,
                                                       u8 *__restrict__ hash32, u8 *__restrict__ sig64, u8 *__restrict__ pub33,
                                                       u8 *__restrict__ malformed) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const size_t row = rowbase[i];
  malformed[i] = gossip_expand_one(msgs + off[i], len ? len[i] : off[i + 1] - off[i], node_ids ? node_ids + 33 * i : nullptr, rowbase[i + 1] - row, hash32 + 32 * row, sig64 + 64 * row, pub33 + 33 * row);
}

__global__ void __launch_bounds__(256) k_gossip_reduce(size_t n, const u8 *__restrict__ msgs, const u64 *__restrict__ off,
                                                       const u64 *__restrict__ rowbase, const u8 *__restrict__ ok,
                                                       const u8 *__restrict__ keyok, const u8 *__restrict__ malformed,
                                                       int8_t *__restrict__ verdict) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const size_t row = rowbase[i];
  verdict[i] = (int8_t)gossip_reduce_one(rowbase[i + 1] - row, ok + row, keyok + row, malformed[i]);
}

// (the synthetic-workload signer kernels live in lamd_testgen.hip -> liblightning_amd_testgen.so: test / bench infrastructure,
// not part of the product library)
// ---- keyed path kernels (see verify_core.h "Keyed path")
// seeded per context (std::random_device at lamd_init): public keys come from the network, and a fixed hash would let a
// peer craft keys that all probe the same slots
LAMD_HD u64 key_hash(const u8 *p, int len, u64 seed) {
  u64 h = seed;
  for (int o = 0; o < len; o += 8) {
    u64 c = 0;
    for (int b = 0; b < 8 && o + b < len; b++) c |= (u64)p[o + b] << (8 * b);
    h = splitmix64(h ^ c);
  }
  return h;
}
// Wave-aggregated allocation from a global counter: ONE atomic per wavefront (an uncontended device atomic costs
// ~10 ns; a million lane-level atomics on one word serialise into >10 ms -- measured).  Every lane of the wave must call
// it; lanes with pred get consecutive indices.  `weight` (optional) is summed over the pred lanes into *wsum.
__device__ __forceinline__ u32 wave_alloc(u32 *counter, bool pred, u32 weight, u32 *wsum) {
  const u64 mask = __ballot(pred);
  const u32 lane = threadIdx.x & 63u;
  const u32 prefix = (u32)__popcll(mask & ((1ull << lane) - 1ull));
  if (wsum) {
    u32 w = pred ? weight : 0u;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) w += __shfl_xor(w, o, 64);
    if (lane == 0 && w) atomicAdd(wsum, w);
  }
  u32 base = 0;
  if (mask == 0) return 0;
  const int leader = __ffsll((long long)mask) - 1;
  if ((int)lane == leader) base = atomicAdd(counter, (u32)__popcll(mask));
  base = __shfl(base, leader, 64);
  return base + prefix;
}

// Block-aggregated allocation from up to four global counters at once (round 6), for 256-thread blocks in which EVERY thread calls it: the counters
// of a call's plan are single words that every wave of the call bumps -- 13 600 waves x 2-4 contended atomics were the 0.3 ms the list builders took
// for 875 k rows (profiles/r06_shard_timeline_cold.txt).  A block's four waves leave their counts in LDS, four threads do one atomic each for the
// block, every lane adds its prefix inside its wave.  pos[c] = the index lane got from counter c (meaningful where pr[c] holds).
__device__ __forceinline__ void block_alloc4(u32 *const counters[4], const bool pr[4], u32 pos[4]) {
  __shared__ u32 s_cnt[4][4], s_base[4][4];
  const u32 lane = threadIdx.x & 63u, wave = (threadIdx.x >> 6) & 3u;
  u32 pre[4];
#pragma unroll
  for (int c = 0; c < 4; c++) {
    const u64 m = __ballot(pr[c]);
    pre[c] = (u32)__popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) s_cnt[wave][c] = (u32)__popcll(m);
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    const u32 c = threadIdx.x, c0 = s_cnt[0][c], c1 = s_cnt[1][c], c2 = s_cnt[2][c], c3 = s_cnt[3][c], tot = c0 + c1 + c2 + c3;
    const u32 b = tot ? atomicAdd(counters[c], tot) : 0u;
    s_base[0][c] = b; s_base[1][c] = b + c0; s_base[2][c] = b + c0 + c1; s_base[3][c] = b + c0 + c1 + c2;
  }
  __syncthreads();
#pragma unroll
  for (int c = 0; c < 4; c++) pos[c] = s_base[wave][c] + pre[c];
}

// Key-table cache.  One entry per distinct public key that has a comb table: the serialised key bytes
// (what the callers hand over: 33 / 65 SEC1 or 32 x-only -- no parsing needed to look a key up), the comb shape, the table
// slot in the pool of that shape, and who published it when (an entry is usable by a call only if the host has SEEN the
// publishing call complete, or it ran earlier on the same lane's stream: `vis`).  meta = teeth (7 / 10; 0 = the key does
// not parse: its rows are rejected without a table) | lane << 8.
struct cache_ent {
  u32 kw[17];  // key bytes, zero padded, little-endian packed; kw[16] = byte 64 | length << 8
  u32 meta, seq, tabslot;
};
struct cache_vis { u32 seq[MAX_LANES + 1]; };
constexpr u32 ENT_NONE = 0xFFFFFFFFu;
enum { C_ENT = 0, C_USED7 = 1, C_USED10 = 2, C_WORDS = 4 };  // cache counters
// per-call counters ("plan"): everything the host used to read back to size the next launches now stays on the device;
// launches cover upper bounds and the kernels take their real extent from here
enum { P_UNIQ = 0, P_HK7 = 1, P_HK10 = 2, P_L7 = 3, P_L10 = 4, P_COLD = 5, P_HITS = 6, P_SUSPECT = 7, P_DENSE = 8, P_COLDOK = 9, P_EARLY = 10, P_TOUCHED = 11, P_G7 = 12, P_G10 = 13, P_WORDS = 16 };
constexpr u8 VERDICT_SUSPECT = 3;  // the bare-formula ecmult met Z = 0: the complete form decides (k_ecmult_keyed_careful)

LAMD_HD void key_words(u32 kw[17], const u8 *p, int len) {
#pragma unroll 1
  for (int w = 0; w < 17; w++) {
    u32 v = 0;
    for (int b = 0; b < 4; b++) {
      const int k = 4 * w + b;
      if (k < len) v |= (u32)p[k] << (8 * b);
    }
    kw[w] = v;
  }
  kw[16] |= (u32)len << 8;
}
LAMD_HD u64 key_words_hash(const u32 kw[17], u64 seed) {
  u64 h = seed;
#pragma unroll 1
  for (int w = 0; w < 16; w += 2) h = splitmix64(h ^ ((u64)kw[w] | ((u64)kw[w + 1] << 32)));
  return splitmix64(h ^ kw[16]);
}

// one thread per row: probe the cache; hits go straight onto the row list of their comb shape
__global__ void __launch_bounds__(256) k_cache_lookup(size_t n, const u8 *__restrict__ keys, int keylen, size_t stride, u64 seed,
                                                      const u32 *index, u32 mask, const cache_ent *ents, cache_vis vis,
                                                      u32 *__restrict__ row_ent, u32 *__restrict__ plan, u32 *__restrict__ list7,
                                                      u32 *__restrict__ list10, u8 *__restrict__ keyok_row, u8 *__restrict__ out,
                                                      const u8 *__restrict__ sig64, int mode) {
  LAMD_PRIO(1);
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = i < n;
  u32 found = ENT_NONE, T = 255;
  if (live) {
    u32 kw[17];
    key_words(kw, keys + stride * i, keylen);
    u32 slot = (u32)key_words_hash(kw, seed) & mask;
    for (int probe = 0; probe < 64; probe++) {
      const u32 id = index[slot];
      if (id == 0u) break;
      const cache_ent *e = ents + (id - 1u);
      const u32 seq = e->seq, meta = e->meta;
      if (seq != 0u && seq <= vis.seq[(meta >> 8) & 15u]) {
        bool same = true;
#pragma unroll 1
        for (int w = 0; w < 17; w++) same &= e->kw[w] == kw[w];
        if (same) { found = id - 1u; T = meta & 0xFFu; break; }
      }
      slot = (slot + 1u) & mask;
    }
    row_ent[i] = found;
  }
  // early reject: a row whose signature scalars are certain to fail the preparation (range, low-S) gets its verdict here and no
  // lane of an ecmult wave; its key has a table, i.e. it parsed
  const bool dead = (T == 7u || T == 10u) && sig_certain_reject(sig64 + 64 * i, mode);
  if (dead) {
    out[i] = 0;
    if (keyok_row) keyok_row[i] = 1;
    T = 254;
  }
  u32 *const ctr[4] = {&plan[P_L7], &plan[P_L10], &plan[P_HITS], &plan[P_EARLY]};
  const bool pr[4] = {T == 7u, T == 10u, found != ENT_NONE, dead};
  u32 pos[4];
  block_alloc4(ctr, pr, pos);
  const u32 p7 = pos[0], p10 = pos[1];
  if (T == 7u) list7[p7] = (u32)i;
  else if (T == 10u) list10[p10] = (u32)i;
  else if (T == 0u) {  // a key that does not parse
    out[i] = 0;
    if (keyok_row) keyok_row[i] = 0;
  }
}

// open-addressing table of row indices (+1) over the rows the cache did not know; the first row to claim a slot represents its key
__global__ void __launch_bounds__(256) k_dedupe_insert(size_t n, const u8 *__restrict__ keys, int keylen, size_t stride, u64 seed,
                                                       const u32 *__restrict__ row_ent, u32 *__restrict__ table, u32 mask,
                                                       u32 *__restrict__ rep) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (row_ent[i] != ENT_NONE) { rep[i] = ENT_NONE; return; }
  const u8 *k = keys + stride * i;
  u32 slot = (u32)key_hash(k, keylen, seed) & mask;
  for (;;) {
    // look before claiming: once a key's first row has landed, its other rows only read the slot.  (With the CAS alone every row
    // of a batch under ONE key -- a key-reuse sweep's K = 1, a hot node's gossip -- hammered the same word: 1 M serialised atomics,
    // 10 ms.)  A stale zero just falls through to the CAS.
    u32 old = __atomic_load_n(&table[slot], __ATOMIC_RELAXED);
    if (old == 0u) old = atomicCAS(&table[slot], 0u, (u32)i + 1u);
    if (old == 0u) { rep[i] = (u32)i; return; }
    const u8 *o = keys + stride * (size_t)(old - 1u);
    bool same = true;
    for (int b = 0; b < keylen; b++) same &= o[b] == k[b];
    if (same) { rep[i] = old - 1u; return; }
    slot = (slot + 1u) & mask;
  }
}
__global__ void __launch_bounds__(256) k_dedupe_number(size_t n, const u32 *__restrict__ rep, u32 *__restrict__ uid,
                                                       u32 *__restrict__ plan, u32 *__restrict__ uniq_row) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool is_rep = i < n && rep[i] == (u32)i;
  const u32 u = wave_alloc(&plan[P_UNIQ], is_rep);
  if (is_rep) {
    uid[i] = u;
    uniq_row[u] = (u32)i;
  }
}
__global__ void __launch_bounds__(256) k_dedupe_map(size_t n, const u32 *__restrict__ rep, const u32 *__restrict__ uid,
                                                    u32 *__restrict__ count) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = i < n && rep[i] != ENT_NONE;
  const u32 k = live ? uid[rep[i]] : 0u;
  // count[k] += 1, combined per wave and key: neighbouring rows often share a key (the 483 HTLC signatures of one
  // commitment), and 64 same-address atomics from one wave serialise
  const u32 lane = threadIdx.x & 63u;
  u64 todo = __ballot(live);
  while (todo) {
    const int leader = __ffsll((long long)todo) - 1;
    const u32 kk = __shfl(k, leader, 64);
    const u64 same = __ballot(live && k == kk);
    if ((int)lane == leader) atomicAdd(&count[kk], (u32)__popcll(same));
    todo &= ~same;
  }
}
// ---- the fused front end (LAMD_FUSED_FRONT=1, the default).  Every kernel of a call's front end stands in ONE dependent chain on
// the lane's main stream, and inside the pipelined loop each link of that chain waits for wave slots behind the other lanes' ecmult
// kernels: a launch that does nothing still cost 0.14 ms there (rocprofv3 trace of the cold loop, r03: 19 launches and 5 fills in
// front of the ecmult kernel, 8.9 ms of chain for 1.1 ms of VALU work).  The fused chain is six launches: k_call_init (every fill),
// k_dedupe_insert_count (insert + numbering + use counts), k_dedupe_classify, k_keys_bases_both (parse + doubling chain, both comb
// shapes in one grid), k_kc_finish_both (Gray-code chains, Z products, rescale, cache entry -- a key's chains are neighbouring lanes
// of one block), k_partition.
__global__ void __launch_bounds__(256) k_call_init(u32 *__restrict__ plan, u32 *__restrict__ table, size_t m, u32 *__restrict__ count, size_t n,
                                                   u32 *__restrict__ row_ent, u32 *__restrict__ cc) {
  LAMD_PRIO(1);
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  if (t < P_WORDS) plan[t] = 0;
  if (cc && t < C_WORDS) cc[t] = 0;
  const uint4 z = make_uint4(0, 0, 0, 0), f = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
  for (size_t i = t; i < m / 4; i += stride) reinterpret_cast<uint4 *>(table)[i] = z;   // m is a power of two >= 4
  for (size_t i = t; i < n / 4; i += stride) {
    reinterpret_cast<uint4 *>(count)[i] = z;
    if (row_ent) reinterpret_cast<uint4 *>(row_ent)[i] = f;
  }
  if (t < (n & 3)) {
    count[(n & ~(size_t)3) + t] = 0;
    if (row_ent) row_ent[(n & ~(size_t)3) + t] = 0xFFFFFFFFu;
  }
}
// insert + numbering + use counts in one pass: the row that claims a slot represents its key and takes the next place on the list
// of distinct keys; every row adds one to count[its representative's ROW] (combined per wave and key).  count and newent are
// indexed by representative row in this form (BYROW below), so no row needs its representative's number.
__global__ void __launch_bounds__(256) k_dedupe_insert_count(size_t n, const u8 *__restrict__ keys, int keylen, size_t stride, u64 seed,
                                                             const u32 *__restrict__ row_ent, u32 *__restrict__ table, u32 mask,
                                                             u32 *__restrict__ rep, u32 *__restrict__ plan, u32 *__restrict__ uniq_row,
                                                             u32 *__restrict__ count) {
  LAMD_PRIO(1);
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = i < n && row_ent[i] == ENT_NONE;
  u32 r = ENT_NONE;
  if (live) {
    const u8 *k = keys + stride * i;
    u32 slot = (u32)key_hash(k, keylen, seed) & mask;
    for (;;) {  // (look before claiming: see k_dedupe_insert)
      u32 old = __atomic_load_n(&table[slot], __ATOMIC_RELAXED);
      if (old == 0u) old = atomicCAS(&table[slot], 0u, (u32)i + 1u);
      if (old == 0u) { r = (u32)i; break; }
      const u8 *o = keys + stride * (size_t)(old - 1u);
      bool same = true;
      for (int b = 0; b < keylen; b++) same &= o[b] == k[b];
      if (same) { r = old - 1u; break; }
      slot = (slot + 1u) & mask;
    }
  }
  if (i < n) rep[i] = r;
  const bool is_rep = live && r == (u32)i;
  const u32 u = wave_alloc(&plan[P_UNIQ], is_rep);
  if (is_rep) uniq_row[u] = (u32)i;
  const u32 lane = threadIdx.x & 63u;
  u64 todo = __ballot(live);
  while (todo) {
    const int leader = __ffsll((long long)todo) - 1;
    const u32 rr = __shfl(r, leader, 64);
    const u64 same = __ballot(live && r == rr);
    if ((int)lane == leader) atomicAdd(&count[rr], (u32)__popcll(same));
    todo &= ~same;
  }
}
// A LEARNING call (the latency path met a key for the second time, see run_small) builds tables only for the keys whose fingerprint
// actually recurred -- `fps` -- not for every key its batch happens to carry (a channel_announcement's one-off bitcoin keys, the keys of
// messages that fail verification), and only while the learnt tables stay inside their budget (half of each pool: learning alone can
// then never push the bounded cache into the reset that evicts everybody's tables).  fps == nullptr: no filter.
__host__ __device__ static inline u64 small_fingerprint(u64 seed, const u8 *key, int keylen);
struct learn_filter { const u64 *fps; u32 n; const u8 *keys; int keylen; size_t stride; u64 seed; u32 budget7, budget10; };
// one thread per distinct new key: keys carried by >= thr7 rows get a 7-tooth comb, by >= thr10 rows a 10-tooth comb (table
// slot + cache entry allocated here, wave-aggregated; a full pool simply leaves the key without a table)
struct cache_caps { u32 ent, t7, t10; };
// (BYROW: count / newent are indexed by the representative's row -- the fused front end -- instead of the key's number)
template <bool BYROW>
__global__ void __launch_bounds__(256) k_dedupe_classify(size_t n, const u32 *__restrict__ count, const u32 *__restrict__ uniq_row,
                                                         u32 thr7, u32 thr10, u32 *__restrict__ plan, u32 *cc, cache_caps caps,
                                                         u32 hk7_cap, u32 hk10_cap, u32 *__restrict__ newent, u32 *__restrict__ hk7_row,
                                                         u32 *__restrict__ hk7_ent, u32 *__restrict__ hk7_slot, u32 *__restrict__ hk10_row,
                                                         u32 *__restrict__ hk10_ent, u32 *__restrict__ hk10_slot, learn_filter lf) {
  LAMD_PRIO(1);
  const size_t u = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = u < n && u < plan[P_UNIQ];
  const size_t ci = BYROW ? (live ? (size_t)uniq_row[u] : 0) : u;
  const u32 c = live ? count[ci] : 0u;
  bool listed = true;
  if (lf.fps && live) {
    const u64 fp = small_fingerprint(lf.seed, lf.keys + lf.stride * (size_t)uniq_row[u], lf.keylen);
    listed = false;
    for (u32 k = 0; k < lf.n && !listed; k++) listed = lf.fps[k] == fp;
  }
  // (the counters only grow: what this launch reads at its start is a lower bound; a batch of <= 4096 keys may overshoot a budget by that much)
  const bool room10 = cc[C_USED10] < lf.budget10, room7 = cc[C_USED7] < lf.budget7;
  const bool want10 = live && listed && room10 && c >= thr10;
  const u32 s10 = wave_alloc(&cc[C_USED10], want10);
  const bool ok10 = want10 && s10 < caps.t10;
  const bool want7 = live && listed && room7 && !ok10 && c >= (thr7 < thr10 ? thr7 : thr10);
  const u32 s7 = wave_alloc(&cc[C_USED7], want7);
  const bool ok7 = want7 && s7 < caps.t7;
  const u32 eid = wave_alloc(&cc[C_ENT], ok7 | ok10);
  const bool oke = (ok7 | ok10) && eid < caps.ent;
  const u32 j7 = wave_alloc(&plan[P_HK7], oke && ok7), j10 = wave_alloc(&plan[P_HK10], oke && ok10);
  bool placed = false;
  if (oke && ok7 && j7 < hk7_cap) { hk7_row[j7] = uniq_row[u]; hk7_ent[j7] = eid; hk7_slot[j7] = s7; placed = true; }
  if (oke && ok10 && j10 < hk10_cap) { hk10_row[j10] = uniq_row[u]; hk10_ent[j10] = eid; hk10_slot[j10] = s10; placed = true; }
  if (live) newent[ci] = placed ? eid : ENT_NONE;
}
// the new keys' cache entries (after their tables are complete) and, in cache mode, their index slots
__global__ void __launch_bounds__(256) k_cache_publish(const u32 *__restrict__ plan, int which, u32 cap, const u32 *__restrict__ hk_row,
                                                       const u32 *__restrict__ hk_ent, const u32 *__restrict__ hk_slot,
                                                       const u8 *__restrict__ keyok, const u8 *__restrict__ keys, int keylen, size_t stride,
                                                       u32 T, u32 lane, u32 seq, u64 seed, cache_ent *ents, u32 *index, u32 mask, int do_index) {
  const u32 j = blockIdx.x * blockDim.x + threadIdx.x;
  const u32 cnt = plan[which] < cap ? plan[which] : cap;
  if (j >= cnt) return;
  cache_ent e;
  key_words(e.kw, keys + stride * (size_t)hk_row[j], keylen);
  e.meta = (keyok[j] ? T : 0u) | (lane << 8);
  e.seq = seq;
  e.tabslot = hk_slot[j];
  const u32 id = hk_ent[j];
  ents[id] = e;
  if (!do_index) return;
  __threadfence();
  u32 slot = (u32)key_words_hash(e.kw, seed) & mask;
  for (;;) {  // the index is at least twice the entry capacity: a free slot exists
    if (atomicCAS(&index[slot], 0u, id + 1u) == 0u) break;
    slot = (slot + 1u) & mask;
  }
}
// The cold rows -- rows under a key that gets no table -- are known as soon as the keys are classified, long before the tables exist: their list is
// made here, right behind k_dedupe_classify, and the ladder (key parse + 128 doublings per row, the longest chain of a call) starts on the side
// stream while the main stream still builds the tables.  (Round 6: until then k_partition made this list too, AFTER the table kernels -- a gossip
// shard's ladder started 0.7 ms into a 3 ms call for no reason: profiles/r06_shard_timeline_cold.txt.)  k_partition<.., true> then leaves them alone.
__global__ void __launch_bounds__(256) k_partition_cold(size_t n, const u32 *__restrict__ rep, const u32 *__restrict__ newent, u32 *__restrict__ plan,
                                                        u32 *__restrict__ listcold) {
  LAMD_PRIO(1);
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool cold = i < n && rep[i] != ENT_NONE && newent[rep[i]] == ENT_NONE;
  const u32 pos = wave_alloc(&plan[P_COLD], cold);
  if (cold) listcold[pos] = (u32)i;
}
// rows the cache did not know: those whose key just got a table join the row list of its shape, the rest take the ladder
template <bool BYROW, bool COLD_LISTED = false>
__global__ void __launch_bounds__(256) k_partition(size_t n, u32 *__restrict__ row_ent, const u32 *__restrict__ rep, const u32 *__restrict__ uid,
                                                   const u32 *__restrict__ newent, const cache_ent *__restrict__ ents, u32 *__restrict__ plan,
                                                   u32 *__restrict__ list7, u32 *__restrict__ list10, u32 *__restrict__ listcold,
                                                   u8 *__restrict__ keyok_row, u8 *__restrict__ out, const u8 *__restrict__ sig64, int mode) {
  LAMD_PRIO(1);
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool miss = i < n && rep[i] != ENT_NONE;
  u32 T = 255;
  if (miss) {
    const u32 e = newent[BYROW ? rep[i] : uid[rep[i]]];
    if (e != ENT_NONE) {
      row_ent[i] = e;
      T = ents[e].meta & 0xFFu;
    }
  }
  const bool dead = miss && (T == 7u || T == 10u) && sig_certain_reject(sig64 + 64 * i, mode);  // early reject (see k_cache_lookup)
  if (dead) {
    out[i] = 0;
    if (keyok_row) keyok_row[i] = 1;
    T = 254;
  }
  u32 *const ctr[4] = {&plan[P_L7], &plan[P_L10], &plan[P_COLD], &plan[P_EARLY]};
  const bool pr[4] = {miss && T == 7u, miss && T == 10u, !COLD_LISTED && miss && T == 255u, dead};
  u32 pos[4];
  block_alloc4(ctr, pr, pos);
  const u32 p7 = pos[0], p10 = pos[1], pc = pos[2];
  if (!miss || dead) return;
  if (T == 7u) list7[p7] = (u32)i;
  else if (T == 10u) list10[p10] = (u32)i;
  else if (T == 0u) {
    out[i] = 0;
    if (keyok_row) keyok_row[i] = 0;
  } else if (!COLD_LISTED) listcold[pc] = (u32)i;
}
// key tables, four stages (verify_core.h "Building one key's table"): bases and prefix run one thread per key, the
// chains and the rescale one thread per (key, 16-entry chain) so that a few hundred keys still fill the chip.
// nkeys = min(plan[which], cap); key u's table lives in slot slots[u] of the pool.
template <int T>
__global__ void __launch_bounds__(256) k_kc_bases(const u32 *__restrict__ plan, int which, u32 cap, const u32 *__restrict__ qwords,
                                                  const u8 *__restrict__ keyok, u32 *__restrict__ scratch) {
  const size_t u = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const u32 nkeys = plan[which] < cap ? plan[which] : cap;
  if (u >= nkeys || !keyok[u]) return;
  u32 qx[8], qy[8];
#pragma unroll
  for (int i = 0; i < 8; i++) { qx[i] = qwords[u * 16 + i]; qy[i] = qwords[u * 16 + 8 + i]; }
  kc_bases<T>(scratch + u * kc_scratch_words(T), ge_from_words(qx, qy));
}
template <int T>
__global__ void __launch_bounds__(256) k_kc_chain_fwd(const u32 *__restrict__ plan, int which, u32 cap, const u8 *__restrict__ keyok,
                                                      u32 *__restrict__ pool, const u32 *__restrict__ slots, u32 *__restrict__ scratch) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t u = t / kc_nsub(T);
  const u32 nkeys = plan[which] < cap ? plan[which] : cap;
  if (u >= nkeys || !keyok[u]) return;
  kc_chain_fwd<T>(pool + (size_t)slots[u] * kc_stride(T), scratch + u * kc_scratch_words(T), (int)(t % kc_nsub(T)));
}
template <int T>
__global__ void __launch_bounds__(256) k_kc_prefix(const u32 *__restrict__ plan, int which, u32 cap, const u32 *__restrict__ qwords,
                                                   const u8 *__restrict__ keyok, u32 *__restrict__ pool, const u32 *__restrict__ slots,
                                                   u32 *__restrict__ scratch) {
  const size_t u = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const u32 nkeys = plan[which] < cap ? plan[which] : cap;
  if (u >= nkeys || !keyok[u]) return;
  u32 qx[8], qy[8];
#pragma unroll
  for (int i = 0; i < 8; i++) { qx[i] = qwords[u * 16 + i]; qy[i] = qwords[u * 16 + 8 + i]; }
  kc_prefix<T>(pool + (size_t)slots[u] * kc_stride(T), scratch + u * kc_scratch_words(T), ge_from_words(qx, qy));
}
template <int T>
__global__ void __launch_bounds__(256) k_kc_chain_bwd(const u32 *__restrict__ plan, int which, u32 cap, const u8 *__restrict__ keyok,
                                                      u32 *__restrict__ pool, const u32 *__restrict__ slots, const u32 *__restrict__ scratch) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t u = t / kc_nsub(T);
  const u32 nkeys = plan[which] < cap ? plan[which] : cap;
  if (u >= nkeys || !keyok[u]) return;
  kc_chain_bwd<T>(pool + (size_t)slots[u] * kc_stride(T), scratch + u * kc_scratch_words(T), (int)(t % kc_nsub(T)));
}
template <int T>
static void launch_keytables(hipStream_t st, const u32 *plan, int which, size_t cap, const u32 *qwords, const u8 *keyok, u32 *pool,
                             const u32 *slots, u32 *scratch) {
  const size_t chains = cap * kc_nsub(T);
  hipLaunchKernelGGL((k_kc_bases<T>), dim3((unsigned)((cap + 255) / 256)), dim3(256), 0, st, plan, which, (u32)cap, qwords, keyok, scratch);
  hipLaunchKernelGGL((k_kc_chain_fwd<T>), dim3((unsigned)((chains + 255) / 256)), dim3(256), 0, st, plan, which, (u32)cap, keyok, pool, slots, scratch);
  hipLaunchKernelGGL((k_kc_prefix<T>), dim3((unsigned)((cap + 255) / 256)), dim3(256), 0, st, plan, which, (u32)cap, qwords, keyok, pool, slots, scratch);
  hipLaunchKernelGGL((k_kc_chain_bwd<T>), dim3((unsigned)((chains + 255) / 256)), dim3(256), 0, st, plan, which, (u32)cap, keyok, pool, slots,
                     (const u32 *)scratch);
}

// ---- the fused front end's two table kernels (see k_call_init).  One grid covers both comb shapes: blocks [0, blocks7) work on the
// 7-tooth list, the others on the 10-tooth list.
struct kc_shape_args {
  u32 cap;                 // capacity of the list of new keys of this shape (nkeys = min(plan[which], cap))
  const u32 *hk_row;       // representative row per new key
  const u32 *hk_ent, *hk_slot;
  u32 *qwords;             // parsed key, 16 words per key
  u8 *keyok;
  u32 *scratch;
  u32 *pool;
};
template <int T>
__device__ __forceinline__ void keys_bases_body(size_t u, const u32 *__restrict__ plan, int which, const kc_shape_args &A, const u8 *__restrict__ keys,
                                                int keylen, size_t stride) {
  const u32 nkeys = plan[which] < A.cap ? plan[which] : A.cap;
  if (u >= nkeys) return;
  u32 qx[8], qy[8];
  const bool ok = parse_pubkey(keys + stride * (size_t)A.hk_row[u], keylen, qx, qy);
  uint4 *dst = reinterpret_cast<uint4 *>(A.qwords + u * 16);
  dst[0] = make_uint4(qx[0], qx[1], qx[2], qx[3]);
  dst[1] = make_uint4(qx[4], qx[5], qx[6], qx[7]);
  dst[2] = make_uint4(qy[0], qy[1], qy[2], qy[3]);
  dst[3] = make_uint4(qy[4], qy[5], qy[6], qy[7]);
  A.keyok[u] = ok;
  if (ok) kc_bases<T>(A.scratch + u * kc_scratch_words(T), ge_from_words(qx, qy));
}
__global__ void __launch_bounds__(256) k_keys_bases_both(const u32 *__restrict__ plan, unsigned blocks7, kc_shape_args A7, kc_shape_args A10,
                                                         const u8 *__restrict__ keys, int keylen, size_t stride) {
  LAMD_PRIO(8);
  if (blockIdx.x < blocks7) keys_bases_body<7>((size_t)blockIdx.x * blockDim.x + threadIdx.x, plan, P_HK7, A7, keys, keylen, stride);
  else keys_bases_body<10>((size_t)(blockIdx.x - blocks7) * blockDim.x + threadIdx.x, plan, P_HK10, A10, keys, keylen, stride);
}
// chains forward -> Z products (+ the key's cache entry) -> chains backward: the kc_nsub(T) chains of a key are neighbouring threads
// of one block (256 is a multiple of 4 and of 32), the stages talk through the key's scratch area, a block barrier between them.
struct publish_args {
  const u8 *keys;
  int keylen;
  size_t stride;
  u32 lane, seq;
  u64 seed;
  cache_ent *ents;
  u32 *index;
  u32 mask;
  int do_index;
};
template <int T>
__device__ __forceinline__ void kc_finish_body(size_t t, const u32 *__restrict__ plan, int which, const kc_shape_args &A, const publish_args &P) {
  constexpr int NS = kc_nsub(T);
  static_assert(256 % NS == 0, "a key's chains must not straddle blocks");
  const size_t u = t / NS;
  const int sub = (int)(t % NS);
  const u32 nkeys = plan[which] < A.cap ? plan[which] : A.cap;
  const bool live = u < nkeys;
  const bool ok = live && A.keyok[u];
  u32 *tab = ok ? A.pool + (size_t)A.hk_slot[u] * kc_stride(T) : nullptr;
  u32 *scr = A.scratch + u * kc_scratch_words(T);
  if (ok) kc_chain_fwd<T>(tab, scr, sub);
  // (block scope is all the stages need -- __syncthreads() carries the workgroup-scope release / acquire.  A device-scope __threadfence()
  // here writes the XCD's L2 back on every wave: the two table kernels took 3.9 ms instead of 1.9, measured in GPU session l.)
  __syncthreads();
  if (live && sub == 0) {
    if (ok) {
      u32 qx[8], qy[8];
#pragma unroll
      for (int i = 0; i < 8; i++) { qx[i] = A.qwords[u * 16 + i]; qy[i] = A.qwords[u * 16 + 8 + i]; }
      kc_prefix<T>(tab, scr, ge_from_words(qx, qy));
    }
    // the key's cache entry (k_cache_publish): a key that does not parse is entered as such
    cache_ent e;
    key_words(e.kw, P.keys + P.stride * (size_t)A.hk_row[u], P.keylen);
    e.meta = (ok ? (u32)T : 0u) | (P.lane << 8);
    e.seq = P.seq;
    e.tabslot = A.hk_slot[u];
    const u32 id = A.hk_ent[u];
    P.ents[id] = e;
    if (P.do_index) {
      __threadfence();
      u32 slot = (u32)key_words_hash(e.kw, P.seed) & P.mask;
      for (;;) {
        if (atomicCAS(&P.index[slot], 0u, id + 1u) == 0u) break;
        slot = (slot + 1u) & P.mask;
      }
    }
  }
  __syncthreads();
  if (ok) kc_chain_bwd<T>(tab, scr, sub);
}
__global__ void __launch_bounds__(256) k_kc_finish_both(const u32 *__restrict__ plan, unsigned blocks7, kc_shape_args A7, kc_shape_args A10, publish_args P) {
  LAMD_PRIO(8);
  if (blockIdx.x < blocks7) kc_finish_body<7>((size_t)blockIdx.x * blockDim.x + threadIdx.x, plan, P_HK7, A7, P);
  else kc_finish_body<10>((size_t)(blockIdx.x - blocks7) * blockDim.x + threadIdx.x, plan, P_HK10, A10, P);
}

// The tree builder in the finish kernel's place (round 6, LAMD_KC_TREE=1; verify_core.h kc_tree_affine): one lane owns KPL keys of a shape -- four 7-tooth keys
// or one 10-tooth key -- builds their entries as a doubling tree of affine additions with every level's inversions shared, and publishes their cache
// entries.  One wave per workgroup: the grid is a few hundred waves, and single-wave groups fill the SIMDs evenly.
template <int T, int KPL>
__device__ __forceinline__ void kc_tree_body(size_t t, const u32 *__restrict__ plan, int which, const kc_shape_args &A, const publish_args &P) {
  const u32 nkeys = plan[which] < A.cap ? plan[which] : A.cap;
  const size_t u0 = t * KPL;
  if (u0 >= nkeys) return;
  kc_tree_keys K;
  ge qs[KPL];
  K.n = 0;
#pragma unroll 1
  for (int k = 0; k < KPL; k++) {
    const size_t u = u0 + k;
    if (u >= nkeys) break;
    const bool ok = A.keyok[u] != 0;
    if (ok) {
      u32 qx[8], qy[8];
#pragma unroll
      for (int i = 0; i < 8; i++) { qx[i] = A.qwords[u * 16 + i]; qy[i] = A.qwords[u * 16 + 8 + i]; }
      qs[K.n] = ge_from_words(qx, qy);
      K.tab[K.n] = A.pool + (size_t)A.hk_slot[u] * kc_stride(T);
      K.scr[K.n] = A.scratch + u * kc_scratch_words(T);
      K.n++;
    }
    // the key's cache entry (as kc_finish_body): a key that does not parse is entered as such
    cache_ent e;
    key_words(e.kw, P.keys + P.stride * (size_t)A.hk_row[u], P.keylen);
    e.meta = (ok ? (u32)T : 0u) | (P.lane << 8);
    e.seq = P.seq;
    e.tabslot = A.hk_slot[u];
    const u32 id = A.hk_ent[u];
    P.ents[id] = e;
    if (P.do_index) {
      __threadfence();
      u32 slot = (u32)key_words_hash(e.kw, P.seed) & P.mask;
      for (;;) {
        if (atomicCAS(&P.index[slot], 0u, id + 1u) == 0u) break;
        slot = (slot + 1u) & P.mask;
      }
    }
  }
  if (K.n) kc_tree_affine<T>(K, qs);
}
constexpr int KC_TREE_KPL7 = 4, KC_TREE_KPL10 = 1;
__global__ void __launch_bounds__(64) k_kc_tree_both(const u32 *__restrict__ plan, unsigned blocks7, kc_shape_args A7, kc_shape_args A10, publish_args P) {
  LAMD_PRIO(8);
  if (blockIdx.x < blocks7) kc_tree_body<7, KC_TREE_KPL7>((size_t)blockIdx.x * blockDim.x + threadIdx.x, plan, P_HK7, A7, P);
  else kc_tree_body<10, KC_TREE_KPL10>((size_t)(blockIdx.x - blocks7) * blockDim.x + threadIdx.x, plan, P_HK10, A10, P);
}

// ---- rows of one key next to each other (round 4).  The row lists come out of the list builders in arrival order, so the 64 lanes of an ecmult
// wave read 64 different keys' tables: every comb entry a cache miss, 4 KB of random HBM traffic per verification (25x the algorithmic bytes).
// Three small kernels regroup the lists by key (cache entry) -- a counting sort whose histogram lives in a per-lane array over the entry ids:
//   k_group_count    every listed row takes its rank among its key's rows (one atomic per distinct key and WAVE: a batch under one key would
//                    otherwise serialise a million atomics on one word); the first row of a key notes the entry as touched
//   k_group_alloc    every touched entry gets a contiguous range of its shape's grouped list (wave-level scan + one atomic per wave) and its
//                    counter goes back to zero for the next call
//   k_group_scatter  row -> range base + rank
// After it a wave's lanes share a handful of tables (6 KB / 48 KB each, L1 / L2 resident) and HBM sees each table about once.
__device__ __forceinline__ u32 wave_alloc_range(u32 *counter, bool pred, u32 count) {
  const u32 lane = threadIdx.x & 63u;
  u32 incl = pred ? count : 0u;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const u32 up = __shfl_up(incl, o, 64);
    if (lane >= (u32)o) incl += up;
  }
  const u32 total = __shfl(incl, 63, 64);
  u32 base = 0;
  if (lane == 63u && total) base = atomicAdd(counter, total);
  base = __shfl(base, 63, 64);
  return base + incl - (pred ? count : 0u);
}
__global__ void __launch_bounds__(256) k_group_count(u32 *__restrict__ plan, const u32 *__restrict__ list7, const u32 *__restrict__ list10,
                                                     const u32 *__restrict__ row_ent, u32 *__restrict__ ent_cnt, u32 *__restrict__ rank,
                                                     u32 *__restrict__ touched) {
  LAMD_PRIO(1);
  const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x, t7 = plan[P_L7], total = t7 + plan[P_L10];
  const bool live = j < total;
  const u32 lane = threadIdx.x & 63u;
  const u32 ent = live ? row_ent[j < t7 ? list7[j] : list10[j - t7]] : ENT_NONE;
  u32 r = 0;
  u64 todo = __ballot(live);
  while (todo) {  // one round per distinct entry among the wave's rows
    const int leader = __ffsll((long long)todo) - 1;
    const u32 e = __shfl(ent, leader, 64);
    const bool mine = live && ent == e;
    const u64 same = __ballot(mine);
    u32 base = 0;
    if ((int)lane == leader) base = atomicAdd(&ent_cnt[e], (u32)__popcll(same));
    base = __shfl(base, leader, 64);
    if (mine) r = base + (u32)__popcll(same & ((1ull << lane) - 1ull));
    todo &= ~same;
  }
  if (live) rank[j] = r;
  const bool first = live && r == 0;
  const u32 p = wave_alloc(&plan[P_TOUCHED], first, 0, nullptr);
  if (first) touched[p] = ent;
}
__global__ void __launch_bounds__(256) k_group_alloc(u32 *__restrict__ plan, const u32 *__restrict__ touched, const cache_ent *__restrict__ ents,
                                                     u32 *__restrict__ ent_cnt, u32 *__restrict__ ent_base) {
  LAMD_PRIO(1);
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = t < plan[P_TOUCHED];
  const u32 ent = live ? touched[t] : 0u, c = live ? ent_cnt[ent] : 0u;
  const bool ten = live && (ents[ent].meta & 0xFFu) == 10u;
  const u32 b7 = wave_alloc_range(&plan[P_G7], live && !ten, c), b10 = wave_alloc_range(&plan[P_G10], ten, c);
  if (live) {
    ent_base[ent] = ten ? b10 : b7;
    ent_cnt[ent] = 0;
  }
}
__global__ void __launch_bounds__(256) k_group_scatter(const u32 *__restrict__ plan, const u32 *__restrict__ list7, const u32 *__restrict__ list10,
                                                       const u32 *__restrict__ row_ent, const u32 *__restrict__ ent_base, const u32 *__restrict__ rank,
                                                       u32 *__restrict__ glist7, u32 *__restrict__ glist10) {
  LAMD_PRIO(1);
  const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x, t7 = plan[P_L7], total = t7 + plan[P_L10];
  if (j >= total) return;
  const u32 i = j < t7 ? list7[j] : list10[j - t7];
  (j < t7 ? glist7 : glist10)[ent_base[row_ent[i]] + rank[j]] = i;
}
// work item j verifies one row against the comb table of its key: items [0, plan[P_L7]) are the rows of list7 (7-tooth combs),
// the plan[P_L10] items after them the rows of list10.  ONE launch covers both shapes: the lists' lengths are only known on the
// device, and a launch of its own for a list that turns out empty still has to get its blocks dispatched -- behind another lane's
// ecmult kernel that holds every wave slot that took up to 2.8 ms (rocprofv3 timeline of the streaming queue), all of it added to
// the batch's latency.  A wave that straddles the boundary runs both bodies; every other wave runs one.
// The hot form: bare addition formulas, one Z == 0 test at the end; a lane that meets it reports VERDICT_SUSPECT and the
// CAREFUL launch decides that row with the complete formulas.
// (LAMD_KEYED_THREADS: block size of the table-driven ecmult launches -- a build knob for experiments, 256 is what ships)
#ifndef LAMD_KEYED_THREADS
#define LAMD_KEYED_THREADS 256
#endif
template <bool CAREFUL, int WAVES>
__global__ void __launch_bounds__(LAMD_KEYED_THREADS, WAVES) k_ecmult_keyed(u32 *plan, const u32 *__restrict__ list7, const u32 *__restrict__ list10,
                                                      const prep_rec *__restrict__ recs, const u32 *__restrict__ row_ent,
                                                      const cache_ent *__restrict__ ents, const u32 *__restrict__ pool7,
                                                      const u32 *__restrict__ pool10, const u8 *__restrict__ sig64, int mode,
                                                      const u32 *__restrict__ gtable, u32 *__restrict__ fin, u8 *__restrict__ keyok_row,
                                                      u8 *__restrict__ out, const u32 *__restrict__ gtable5 = nullptr, int part = 0, u32 tail = 0) {
  if (CAREFUL && plan[P_SUSPECT] == 0) return;
#if defined(LAMD_G_LDS)
  extern __shared__ u32 s_g5[];
  const u32 *glds = nullptr;
  if (!CAREFUL && gtable5) {  // stage the 5-bit-window table of G into this block's LDS (104 KB)
    for (int i = threadIdx.x; i < GLDS_WORDS / 4; i += blockDim.x) reinterpret_cast<uint4 *>(s_g5)[i] = reinterpret_cast<const uint4 *>(gtable5)[i];
    __syncthreads();
    glds = s_g5;
  }
#else
  const u32 *glds = nullptr;
  (void)gtable5;
#endif
  const size_t t7 = plan[P_L7], total = t7 + plan[P_L10], stride = (size_t)gridDim.x * blockDim.x;
  // (part 1 / 2 of a launch cut in two, LAMD_ECMULT_CHAIN=2: everything but the last `tail` work items / those last items)
  const size_t cut = total > tail ? total - tail : 0, begin = part == 2 ? cut : 0, end = part == 1 ? cut : total;
#pragma unroll 1
  for (size_t j = begin + (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < end; j += stride) {
    const bool ten = j >= t7;
    const size_t i = ten ? list10[j - t7] : list7[j];
    if (CAREFUL && out[i] != VERDICT_SUSPECT) continue;
    prep_rec rec;
    {
      const uint4 *src = reinterpret_cast<const uint4 *>(recs + i);
      const uint4 a = src[0], b = src[1], c = src[2], d = src[3], e = src[4];
      rec.u1[0] = a.x; rec.u1[1] = a.y; rec.u1[2] = a.z; rec.u1[3] = a.w;
      rec.u1[4] = b.x; rec.u1[5] = b.y; rec.u1[6] = b.z; rec.u1[7] = b.w;
      rec.k1[0] = c.x; rec.k1[1] = c.y; rec.k1[2] = c.z; rec.k1[3] = c.w;
      rec.k2[0] = d.x; rec.k2[1] = d.y; rec.k2[2] = d.z; rec.k2[3] = d.w;
      rec.flags = e.x;
    }
    const size_t tabslot = ents[row_ent[i]].tabslot;
    if (!CAREFUL && keyok_row) keyok_row[i] = 1;  // rows on these lists have a parsed key (the others were rejected by lookup / partition)
    bool ok = (rec.flags & PREP_VALID) != 0;
    if (ok) {
      // the careful form returns a Jacobian point that may be flagged infinite, the hot form an XYZZ point (group.h) whose ZZ == 0 says SUSPECT
      std::conditional_t<CAREFUL, gej, gexz> R;
      bool suspect = false;
      if (ten) {
        const u32 *tab = pool10 + tabslot * kc_stride(10);
        if constexpr (CAREFUL) R = ecmult_lane_keyed<10>(rec, tab, gtable);
        else R = ecmult_lane_keyed_fast<10>(rec, tab, gtable, &suspect, glds);
      } else {
        const u32 *tab = pool7 + tabslot * kc_stride(7);
        if constexpr (CAREFUL) R = ecmult_lane_keyed<7>(rec, tab, gtable);
        else R = ecmult_lane_keyed_fast<7>(rec, tab, gtable, &suspect, glds);
      }
      if (suspect) {
        out[i] = VERDICT_SUSPECT;
        atomicAdd(&plan[P_SUSPECT], 1u);
        continue;
      }
      u32 rw[8];
      load_words_be(rw, sig64 + 64 * i);
      if (mode == MODE_ECDSA) {
        ok = ecdsa_final(R, rw);
      } else {
        out[i] = schnorr_stage1(R, rw, fin + i * FIN_WORDS);
        continue;
      }
    }
    out[i] = ok ? 1 : 0;
  }
}
// The table-driven ecmult, pairs first (verify_core.h "Pairs first"): a PERSISTENT grid -- at most as many blocks as the chip holds -- in
// which a lane owns rows tid, tid + nthreads, ... of a list and takes them in batches of <= PAIRS_BMAX rows per field inversion (1 M rows
// on 196 608 lanes: 4 or 5 rows each, one batch).  ws: PAIRS_SLOTS parking slots per lane, slot s of lane tid at
// ws[(s * nthreads + tid) * PAIRS_WS_WORDS] (a wave's 64 slots are contiguous).  Verdicts, suspects and the BIP-340 parity stage's
// input as k_ecmult_keyed<false> writes them; the CAREFUL launch of that kernel follows this one unchanged.
template <int T>
__device__ __forceinline__ void keyed_pairs_list(u32 *plan, size_t cnt, const u32 *__restrict__ list, const prep_rec *__restrict__ recs,
                                                 const u32 *__restrict__ row_ent, const cache_ent *__restrict__ ents, const u32 *__restrict__ pool,
                                                 const u8 *__restrict__ sig64, int mode, const u32 *__restrict__ gtable, u32 *__restrict__ fin,
                                                 u8 *__restrict__ keyok_row, u8 *__restrict__ out, u32 *myws, size_t nthreads, size_t tid) {
#pragma unroll 1
  for (size_t base = tid; base < cnt; base += nthreads * PAIRS_BMAX) {
    const size_t left = (cnt - base + nthreads - 1) / nthreads;  // rows of this lane from `base` on
    const int nb = left < (size_t)PAIRS_BMAX ? (int)left : PAIRS_BMAX;
    pairs_batch<T>(
        nb, gtable, myws, nthreads * PAIRS_WS_WORDS,
        [&](int b, bool first, const prep_rec **rec, const u32 **tab) {
          const size_t i = list[base + (size_t)b * nthreads];
          *rec = recs + i;
          *tab = pool + (size_t)ents[row_ent[i]].tabslot * kc_stride(T);
          if (first) {
            if (keyok_row) keyok_row[i] = 1;  // rows on these lists have a parsed key
            if (!(recs[i].flags & PREP_VALID)) {
              out[i] = 0;
              return false;
            }
          }
          return true;
        },
        [&](int b, const gej &R, bool suspect) {
          const size_t i = list[base + (size_t)b * nthreads];
          if (suspect) {
            out[i] = VERDICT_SUSPECT;
            atomicAdd(&plan[P_SUSPECT], 1u);
            return;
          }
          u32 rw[8];
          load_words_be(rw, sig64 + 64 * i);
          if (mode == MODE_ECDSA) out[i] = ecdsa_final(R, rw) ? 1 : 0;
          else out[i] = schnorr_stage1(R, rw, fin + i * FIN_WORDS);
        });
  }
}
// (one wave per workgroup: the grid is sized to fill the chip exactly once, and a free wave slot must be able to take any waiting workgroup -- with
// four-wave workgroups a quarter of them found no CU with four free slots on four SIMDs and ran as a second round: 4.0 ms instead of 3.0)
constexpr unsigned PAIRS_THREADS = 64;
template <int WAVES>
__global__ void __launch_bounds__(PAIRS_THREADS, WAVES) k_ecmult_keyed_pairs(u32 *plan, const u32 *__restrict__ list7, const u32 *__restrict__ list10,
                                                      const prep_rec *__restrict__ recs, const u32 *__restrict__ row_ent,
                                                      const cache_ent *__restrict__ ents, const u32 *__restrict__ pool7,
                                                      const u32 *__restrict__ pool10, const u8 *__restrict__ sig64, int mode,
                                                      const u32 *__restrict__ gtable, u32 *__restrict__ fin, u8 *__restrict__ keyok_row,
                                                      u8 *__restrict__ out, u32 *__restrict__ ws) {
  const size_t nthreads = (size_t)gridDim.x * blockDim.x, tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  u32 *myws = ws + tid * PAIRS_WS_WORDS;
  keyed_pairs_list<7>(plan, plan[P_L7], list7, recs, row_ent, ents, pool7, sig64, mode, gtable, fin, keyok_row, out, myws, nthreads, tid);
  keyed_pairs_list<10>(plan, plan[P_L10], list10, recs, row_ent, ents, pool10, sig64, mode, gtable, fin, keyok_row, out, myws, nthreads, tid);
}
#if defined(LAMD_PAIRS_CLOCK)
extern "C" int lamd_debug_pairs_clock(unsigned long long out[8], int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(lamd::g_pairs_clk), 64) != hipSuccess) return -1;
  if (reset) { const unsigned long long z[8] = {0, 0, 0, 0, ~0ULL, 0, 0, 0}; if (hipMemcpyToSymbol(HIP_SYMBOL(lamd::g_pairs_clk), z, 64) != hipSuccess) return -1; }
  return 0;
}
#endif
// ---- small batches with a key-table cache: one kernel probes the cache for every row and writes the three row lists straight
// away (cached 7-tooth comb / cached 10-tooth comb / ladder) -- no de-duplication, no table building, so a commitment_signed
// whose htlc key is cached costs a dozen launches instead of the two dozen of the partitioning path.  Nothing is inserted here;
// rows whose key missed but equals their neighbour's (a dense batch under a new key) are counted in plan[P_DENSE], and the host
// sends the NEXT small batch through the table-building path when that count is high.  (One kernel that also ran the comb or
// the ladder itself was tried: 248 VGPRs and 208 bytes of scratch made it slower than the separate kernels.)
__global__ void __launch_bounds__(64) k_small_lookup(size_t n, const u8 *__restrict__ keys, int keylen, size_t stride, u64 seed, const u32 *index, u32 mask,
                                                     const cache_ent *ents, cache_vis vis, u32 *__restrict__ row_ent, u32 *__restrict__ plan,
                                                     u32 *__restrict__ list7, u32 *__restrict__ list10, u32 *__restrict__ listcold,
                                                     u8 *__restrict__ keyok_row, u8 *__restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = i < n;
  u32 found = ENT_NONE, T = 255, hash = 0;
  if (live) {
    u32 kw[17];
    key_words(kw, keys + stride * i, keylen);
    hash = (u32)key_words_hash(kw, seed);
    u32 slot = hash & mask;
    for (int probe = 0; probe < 64; probe++) {
      const u32 id = index[slot];
      if (id == 0u) break;
      const cache_ent *e = ents + (id - 1u);
      const u32 seq = e->seq, meta = e->meta;
      if (seq != 0u && seq <= vis.seq[(meta >> 8) & 15u]) {
        bool same = true;
#pragma unroll 1
        for (int w = 0; w < 17; w++) same &= e->kw[w] == kw[w];
        if (same) { found = id - 1u; T = meta & 0xFFu; break; }
      }
      slot = (slot + 1u) & mask;
    }
    row_ent[i] = found;
  }
  const u32 lane = threadIdx.x & 63u;
  const u32 prev = __shfl_up(hash, 1, 64);
  const bool miss = live && found == ENT_NONE;
  const u32 p7 = wave_alloc(&plan[P_L7], T == 7u), p10 = wave_alloc(&plan[P_L10], T == 10u), pc = wave_alloc(&plan[P_COLD], miss);
  (void)wave_alloc(&plan[P_HITS], live && found != ENT_NONE);
  (void)wave_alloc(&plan[P_DENSE], miss && lane > 0 && prev == hash);
  if (T == 7u) list7[p7] = (u32)i;
  else if (T == 10u) list10[p10] = (u32)i;
  else if (miss) listcold[pc] = (u32)i;
  else if (live) {  // T == 0: a key that is known not to parse
    out[i] = 0;
    if (keyok_row) keyok_row[i] = 0;
  }
}
// ---- the latency path: up to SMALL_MAX rows in ONE launch (a grid of 64-row blocks), inputs read straight from pinned host memory, verdicts
// written straight back (host-buffer calls: run_small; small flushes of the streaming queue: lamd_flush).
// One row per LANE, one TASK per WAVE (verify_core.h "Task split"): a block of eight waves, two per SIMD of a CU.
//   phase A   wave 0: scalar preparation of its row (one division-step inversion per lane)
//             wave 1: the row's key -- probe the key-table cache (comb shape + table), or parse it and build the 8-entry ladder table
//   phase B   waves 0-3: the comb's four partial sums (or the two ladder halves); waves 4-7: three of the 11 windows of u1*G each
//   phase C   three-level merge with complete Jacobian additions (two sums per level, on different waves), acceptance test on wave 0,
//             verdict byte -> host memory, completion word
// Replaces, for such calls, H2D x 3 + a dozen launches + D2H + a stream synchronise (0.38 ms for one row) and the ~10^5-instruction
// dependent chain on one lane.  Everything uses the complete addition formulas: no suspect rows, no second pass.
struct small_part { u32 w[27]; u32 inf; };  // a Jacobian point in LDS
struct small_args {
  const u8 *a32, *sig64, *key;   // pinned host memory (device-mapped)
  int keylen, mode;
  u32 n;
  u64 seed;
  const u32 *index;              // key-table cache (nullptr: none)
  u32 mask;
  const cache_ent *ents;
  cache_vis vis;
  const u32 *pool7, *pool10, *gtable;
  u32 *slots;                    // ladder tables, SLOT_WORDS per row
  u8 *out;                       // pinned host memory: n verdict bytes
  u8 *shapes;                    // ... each row's shape (7 / 10: comb teeth, 255: ladder, 0: rejected key): statistics, and the host remembers the keys that missed
  u32 *done;                     // device memory: blocks finished (grids of more than one block)
  u32 *flag;                     // ... and the completion word (set to `ticket` last)
  u32 ticket;
  const u8 *gate;                // optional, device memory: row i verifies only if gate[i] != 0 (check_tx_sig's sighash-type gate, decided by the hashing kernel in front)
};
__device__ __forceinline__ void small_store(small_part *p, const gej &g) {
#pragma unroll
  for (int i = 0; i < 9; i++) { p->w[i] = g.x.n[i]; p->w[9 + i] = g.y.n[i]; }
  const fe z = fe_norm_weak(g.z);
#pragma unroll
  for (int i = 0; i < 9; i++) p->w[18 + i] = z.n[i];
  p->inf = g.inf;
}
__device__ __forceinline__ gej small_load(const small_part *p) {
  gej g;
#pragma unroll
  for (int i = 0; i < 9; i++) { g.x.n[i] = p->w[i]; g.y.n[i] = p->w[9 + i]; g.z.n[i] = p->w[18 + i]; }
  FE_SETMAG(g.x, 1); FE_SETMAG(g.y, 1); FE_SETMAG(g.z, 1);
  g.inf = p->inf != 0;
  return g;
}
__global__ void __launch_bounds__(512) k_small_verify(small_args A) {
  // EIGHT waves, two per SIMD: waves 0-3 compute the four comb parts (or the ladder halves), waves 4-7 three windows of u1*G each.
  // A lone wave is bound by the dependent-issue latency of its instruction stream (~8.7 cycles per instruction measured, against an
  // issue cost of ~4.3), so a second wave on the same SIMD runs in the gaps: profiles/r03_latency.txt.
  __shared__ prep_rec s_rec[64];
  __shared__ u32 s_shape[64];          // 7 / 10: comb teeth; 255: ladder; 0: rejected (key does not parse)
  __shared__ u32 s_tab[64];            // table slot in the pool of its shape
  __shared__ u32 s_zscale[64][9];      // ladder: Zg of the lane's table
  __shared__ small_part s_part[4][64]; // comb / ladder parts (on the table's isomorphic curve)
  __shared__ small_part s_g[4][64];    // parts of u1*G
  // 65 280 of the 65 536 bytes a kernel may declare statically: a field added to prep_rec / small_part must come with a smaller buffer here
  static_assert(sizeof(prep_rec) * 64 + 2 * 64 * sizeof(u32) + 64 * 9 * sizeof(u32) + 2 * 4 * 64 * sizeof(small_part) <= 65536,
                "k_small_verify: static LDS over 64 KiB");
  // (a call of more than 64 rows is a grid of such blocks: block b owns rows [64 b, 64 b + 64); the LAST block to finish -- a counter in
  // device memory -- writes the completion word)
  const u32 lane = threadIdx.x & 63u, task = threadIdx.x >> 6;
  const size_t row = (size_t)blockIdx.x * 64 + lane;
  const bool live = row < A.n;
  // ---- phase A
  if (task == 0 && live) {
    const size_t base = (size_t)blockIdx.x * 64;
    if (A.mode == MODE_ECDSA) ecdsa_prep_thread(lane, 64, lane + 1, A.a32 + 32 * base, A.sig64 + 64 * base, s_rec);   // this lane's row only
    else schnorr_prep_one(A.a32 + 32 * row, A.key + (size_t)A.keylen * row, A.sig64 + 64 * row, &s_rec[lane]);
  }
  if (task == 1 && live) {
    u32 T = 255, tabslot = 0;
    const u8 *kp = A.key + (size_t)A.keylen * row;
    if (A.index) {
      u32 kw[17];
      key_words(kw, kp, A.keylen);
      u32 slot = (u32)key_words_hash(kw, A.seed) & A.mask;
      for (int probe = 0; probe < 64; probe++) {
        const u32 id = A.index[slot];
        if (id == 0u) break;
        const cache_ent *e = A.ents + (id - 1u);
        const u32 seq = e->seq, meta = e->meta;
        if (seq != 0u && seq <= A.vis.seq[(meta >> 8) & 15u]) {
          bool same = true;
#pragma unroll 1
          for (int w = 0; w < 17; w++) same &= e->kw[w] == kw[w];
          if (same) { T = meta & 0xFFu; tabslot = e->tabslot; break; }
        }
        slot = (slot + 1u) & A.mask;
      }
    }
    if (T == 255u) {  // no table: parse the key, build the ladder's 8-entry table in the lane's slot
      u32 qx[8], qy[8];
      if (parse_pubkey(kp, A.keylen, qx, qy)) {
        const fe zg = fe_norm_weak(build_q_table(A.slots + row * SLOT_WORDS, ge_from_words(qx, qy)));
#pragma unroll
        for (int i = 0; i < 9; i++) s_zscale[lane][i] = zg.n[i];
        __threadfence_block();  // the table (global memory) is read by the ladder waves after the barrier
      } else {
        T = 0;
      }
    }
    s_shape[lane] = T;
    s_tab[lane] = tabslot;
  }
  if (task == 1 && live) A.shapes[row] = (u8)s_shape[lane];  // for lamd_get_info and the host's learning of recurring keys
  __syncthreads();
  // ---- phase B: wave t computes part t+1 of the comb (ST_H1LO .. ST_H2HI) / its ladder half, and its share of the windows of u1*G
  const u32 shape = live ? s_shape[lane] : 0u;
  const bool work = live && shape != 0u && (s_rec[lane].flags & PREP_VALID);
  if (work && task < 4) {
    prep_rec rec = s_rec[lane];
    gej part = gej_infinity();
    const int ct = (int)task + 1;
    // (ONE body for both comb shapes where a wave's lanes hold 7- and 10-tooth keys -- a commitment's funding + htlc rows: two template
    // instantiations would run one after the other; a wave that holds one shape only takes that shape's own, faster, body)
    const bool mixed = __ballot(shape == 7u) != 0 && __ballot(shape == 10u) != 0;
    if (mixed && (shape == 7u || shape == 10u))
      part = small_task_comb_rt(rec, shape == 7u ? A.pool7 + (size_t)s_tab[lane] * kc_stride(7) : A.pool10 + (size_t)s_tab[lane] * kc_stride(10), ct, (int)shape);
    else if (shape == 7u) part = small_task_comb<7>(rec, A.pool7 + (size_t)s_tab[lane] * kc_stride(7), ct);
    else if (shape == 10u) part = small_task_comb<10>(rec, A.pool10 + (size_t)s_tab[lane] * kc_stride(10), ct);
    else if (ct == ST_H1LO || ct == ST_H2LO) part = small_task_ladder(rec, A.slots + row * SLOT_WORDS, ct == ST_H2LO);
    small_store(&s_part[task][lane], part);
  } else if (work) {
    int w_lo, w_hi;
    small_g_windows((int)task - 4, false, &w_lo, &w_hi);
    small_store(&s_g[task - 4][lane], small_task_g(s_rec[lane], A.gtable, w_lo, w_hi));
  }
  __syncthreads();
  // ---- phase C: three-level merge (verify_core.h small_merge4), the two sums of a level on different waves
  if (work && task < 4) {  // level 1: wave 0: P0 + P1, wave 2: P2 + P3, wave 1: G0 + G1, wave 3: G2 + G3
    if (task == 0 || task == 2) small_store(&s_part[task][lane], gej_add_var(small_load(&s_part[task][lane]), small_load(&s_part[task + 1][lane])));
    else small_store(&s_g[task][lane], gej_add_var(small_load(&s_g[task - 1][lane]), small_load(&s_g[task][lane])));
  }
  __syncthreads();
  if (work && task < 4) {  // level 2: wave 0: S = (P01 + P23) * zscale, wave 1: G = G01 + G23
    if (task == 0) {
      gej sum = gej_add_var(small_load(&s_part[0][lane]), small_load(&s_part[2][lane]));
      if (!sum.inf) {
        fe zscale;
        if (shape == 255u) {
#pragma unroll
          for (int i = 0; i < 9; i++) zscale.n[i] = s_zscale[lane][i];
          FE_SETMAG(zscale, 1);
        } else {
          zscale = slot_load_fe(shape == 7u ? A.pool7 + (size_t)s_tab[lane] * kc_stride(7) + kc_words(7) : A.pool10 + (size_t)s_tab[lane] * kc_stride(10) + kc_words(10));
        }
        sum.z = fe_mul(fe_norm_weak(sum.z), zscale);  // back from the isomorphic curve
      }
      small_store(&s_part[0][lane], sum);
    } else if (task == 1) {
      small_store(&s_g[1][lane], gej_add_var(small_load(&s_g[1][lane]), small_load(&s_g[3][lane])));
    }
  }
  __syncthreads();
  if (task == 0 && live) {  // level 3 and the acceptance test
    bool ok = false;
    if (work) {
      const gej R = gej_add_var(small_load(&s_part[0][lane]), small_load(&s_g[1][lane]));
      u32 rw[8];
      load_words_be(rw, A.sig64 + 64 * row);
      ok = A.mode == MODE_ECDSA ? ecdsa_final(R, rw) : schnorr_accept_one(R, rw);
    }
    if (A.gate) ok &= A.gate[row] != 0;
    A.out[row] = ok ? 1 : 0;
  }
  __syncthreads();
  if (threadIdx.x == 0 && A.flag) {  // (a flush of the streaming queue has no completion word: its caller waits for the stream's event)
    __threadfence_system();  // this block's verdict and shape bytes are on their way to host memory before it counts itself done
    const bool last = gridDim.x == 1u || __hip_atomic_fetch_add(A.done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1u;
    if (last) {
      if (gridDim.x != 1u) __hip_atomic_store(A.done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // for the next launch (same stream: ordered)
      __hip_atomic_store(A.flag, A.ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

__global__ void __launch_bounds__(256) k_schnorr_final_fin(size_t n, u32 *__restrict__ fin, u8 *__restrict__ out) {
  LAMD_PRIO(16);
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t T = (size_t)gridDim.x * blockDim.x;
  schnorr_final_thread(tid, T, n, fin, out, FIN_WORDS);
}

// =====================================================================================
//                                        engine
// =====================================================================================

static constexpr size_t CHUNK_DEFAULT = (size_t)1 << 22;  // signatures per launch sequence (4 GiB of table slots at most; allocated on demand)

struct devbuf {
  void *p = nullptr;
  size_t cap = 0;
};

enum { Q_ECDSA33 = 0, Q_ECDSA65 = 1, Q_SCHNORR = 2, Q_KINDS = 3 };
#ifndef LAMD_QUEUE_SETS
#define LAMD_QUEUE_SETS 9
#endif
constexpr int QUEUE_SETS = LAMD_QUEUE_SETS;  // staging sets of the streaming queue: one open + up to QUEUE_SETS - 1 flushes in flight

// the static G table, one per device and process (lamd_init / lamd_shutdown)
struct shared_gtable { std::mutex mu; u32 *p = nullptr; int refs = 0; };
static std::mutex g_gtable_mu;                       // guards the map only: tables of different devices are built side by side (lamd_multi_init)
static std::map<int, shared_gtable> g_gtables;       // (std::map: a slot's address is stable)
static shared_gtable &gtable_slot(int device) {
  std::lock_guard<std::mutex> lk(g_gtable_mu);
  return g_gtables[device];
}

struct lamd_ctx {
  int device = 0;
  int gtable_device = -1;
  hipStream_t stream = nullptr;
  hipDeviceProp_t prop;
  u32 *gtable = nullptr;
  std::string err;
  // per-call workspaces (grown on demand, reused)
  devbuf recs, qwords, keyok, slots, vbuf;
  // keyed path: row -> cache entry, de-duplication of the rows the cache did not know, new keys per comb shape (hk7 / hk10:
  // representative row, cache entry, table slot; parsed key, validity, build scratch), row lists per shape + cold rows
  devbuf row_ent, kd_table, kd_rep, kd_uid, kd_uniq, kd_count, kd_newent, plan, kt_fin;
  devbuf hk7_row, hk7_ent, hk7_slot, hk7_qwords, hk7_keyok, hk7_scratch, hk10_row, hk10_ent, hk10_slot, hk10_qwords, hk10_keyok, hk10_scratch;
  devbuf list7, list10, listcold, listcold_ok;
  // latency path (k_small_verify): pinned, device-mapped staging for up to SMALL_MAX rows, the verdict bytes and the completion word
  u8 *h_small = nullptr;
  u8 *h_tmpl = nullptr;       // pinned: the flattened transaction templates of a lamd_check_commitment_signed / mid-size check_tx_sig call, copied to HBM from here
  size_t h_tmpl_cap = 0;
  devbuf small_done;          // block counter of k_small_verify grids (device memory)
  std::vector<u64> small_missed;   // fingerprints of keys the latency path verified without a table (MISS_SLOTS, 4-way set-associative)
  u32 prio_mask = LAMD_PRIO_DEFAULT; // LAMD_PRIO: which kernel classes raise their wave priority (g_prio)
  unsigned spin_us = 2000;         // LAMD_SPIN_US: longest busy-wait of a latency-path call before it blocks in the runtime
  std::vector<u64> learn_fps;      // the fingerprints that recurred in the call that set force_learn: only these keys get tables (learn_filter)
  devbuf learn_dev;
  bool force_learn = false;         // the next small call on this context builds tables for every key the cache misses
  u32 small_ticket = 0;
  bool small_kernel = true;   // LAMD_SMALL_KERNEL=0: such calls take the general path
  u32 *h_plan = nullptr;   // pinned read-back of the last call's plan + cache counters (statistics only: nothing waits for it)
  // Key-table cache.  The root context owns the shared one (LAMD_CACHE=1, default): entries + index + one table pool per
  // comb shape, filled by whichever lane meets a key often enough, looked up by every later call.  With LAMD_CACHE=0 every
  // lane uses its own private instance without an index, reset at the start of each call (tables live for one call).
  struct key_cache {
    bool shared = false;
    u32 cap_ent = 0, cap7 = 0, cap10 = 0, index_mask = 0;
    devbuf ents, index, pool7, pool10, counters;
  } cache_store;
  key_cache *cache = nullptr;       // what this context's calls use (lanes: the root's when shared)
  lamd_ctx *root = nullptr;         // lanes: the context that owns them
  int lane_id = 0;                  // 0 = the root context itself, 1.. = lanes
  u32 call_seq = 0;                 // root: sequence number of the last submitted keyed call
  u32 pub_seq[MAX_LANES + 1] = {};  // root: per lane (by lane_id), the sequence number of its last publishing call ...
  hipEvent_t ev_pub[MAX_LANES + 1] = {};   // ... and the event recorded after it
  u32 vis_seq[MAX_LANES + 1] = {};  // root: per lane, the newest publishing call the host has SEEN complete
  bool pub_pending[MAX_LANES + 1] = {};
  int cache_mode = 1;               // LAMD_CACHE
  size_t cache_keys = (size_t)1 << 20, cache_keys10 = (size_t)1 << 16;   // LAMD_CACHE_KEYS / LAMD_CACHE_KEYS10
  u32 cache_hwm[3] = {0, 0, 0};     // root: last counters read back (entries, 7-tooth slots, 10-tooth slots)
  u32 cache_resets = 0;
  size_t last_hits = 0, last_cold = 0, last_l7 = 0, last_l10 = 0;
  devbuf keyok_row;
  hipStream_t stream2 = nullptr;   // scalar prep runs here, concurrently with the key work on `stream`
  hipStream_t stream3 = nullptr;   // cold rows of a partitioned chunk
  hipEvent_t ev_fork = nullptr, ev_prep = nullptr, ev_cold = nullptr, ev_keys = nullptr, ev_sigs = nullptr;
  bool sigs_pending = false;  // lamd_flush sent the signature copy down another stream: the first kernel of the main stream that reads signatures waits for ev_sigs_wait
  hipEvent_t ev_sigs_wait = nullptr;   // the event that copy is followed by (the lane's own ev_sigs, or the staging set's)
  // root only: the H2D copies of the flushes, behind nothing but each other (lamd_flush).  Round 5 experiment: SEVERAL such streams, successive flushes taking
  // turns (LAMD_COPY_STREAMS, default 1: two and three measured SLOWER, 212 / 203 against 218 M/s -- profiles/r05_ab_variants.txt).  On ONE stream the three copies of a flush and the copies of the next flush follow each other with gaps of
  // 0.1-1 ms (rocprofv3 --memory-copy-trace of the cold streaming loop, profiles/r05_stream_timeline.txt): 3.0 ms of transfers took 3.5-3.9 ms, the
  // stream was busy 85-90 % of the time and set the loop's pace -- 4.25 ms per flush against the 4.05 ms the kernels need --, every call's front end
  // started the moment its keys landed and only one table-driven ecmult launch was ever in flight.  With two streams the next flush's copies run in
  // the gaps of the current one's.
  static constexpr int MAX_COPY_STREAMS = 4;
  hipStream_t copy_streams[MAX_COPY_STREAMS] = {nullptr, nullptr, nullptr, nullptr};
  int n_copy_streams = 1;
  bool copy_one = true;         // LAMD_COPY_ONE=0: a full staging set still goes down as three copies
  int copy_events = 0;          // LAMD_COPY_EVENTS: events a flush records between / behind its three copies (3: keys | signatures | hashes, 2: keys | rest, 1: all; 0: by load)
  unsigned copy_turn = 0;
  bool use_copy_stream = true;         // LAMD_COPY_STREAM=0: a flush's copies go down its lane's prep stream (the round-2 form)
  hipStream_t d2h_stream = nullptr;    // root only, LAMD_D2H_STREAM=1 (experiment): the verdict copies of every flush on a stream of their own instead of the flush's lane.
                                       // Measured (profiles/r04_ab_variants.txt): nothing on the cold streaming loop, and one more stream per engine costs the SECOND
                                       // engine of a process a quarter of its streaming rate (20 streams on 16 hardware queues) -- off
  bool use_d2h_stream = false;
  int keyed_mode = -1;           // -1 auto, 0 never, 1 whenever keys repeat at all (LAMD_KEYED)
  size_t keyed_min_rows = 8192;  // below this a batch is latency-bound: per-signature ladder
  bool early_cold = true;         // LAMD_EARLY_COLD=0: the cold rows' list is made by k_partition, after the table kernels (rounds 1-5)
  bool fused_front = true;        // LAMD_FUSED_FRONT=0: the front end of a keyed call as the 19 launches + 5 fills of round 2 (see k_call_init)
  bool kc_tree = false;           // LAMD_KC_TREE=1: key tables by the affine tree builder (k_kc_tree_both) instead of the Gray-code chains (k_kc_finish_both); read on the root context
  bool small_fused = true;        // LAMD_SMALL_FUSED=0: small batches take the partitioning path even with a cache
  bool last_small_fused = false;  // the previous small batch of this lane ran k_ecmult_small (its plan holds P_DENSE)
  size_t last_small_n = 0;
  double keyed_min_uses = 6.0;   // average signatures per distinct key that pays for a (comb) table
  double keyed_dense_uses = 48.0;  // ... and for the 10-tooth comb (512 entries per key)
  int keyed_teeth = 0;             // 0 = choose by re-use, 7 or 10 = force that comb (LAMD_KEYED_TEETH)
  int last_mode = 0;
  size_t last_n = 0;
  bool last_keyed_call = false;
  devbuf in_a, in_b, in_c, out;       // staging for the host-buffer API
  devbuf g_msgs, g_off, g_ids, g_rowbase, g_hash, g_sig, g_pub, g_malformed, g_ok, g_verdict;
  // timing
  bool timing = false;
  bool ev_recorded = false;
  int ecmult_waves = 3;
  unsigned keyed_blocks_per_cu = 0;  // LAMD_KEYED_BLOCKS_PER_CU: grid cap of the table-driven ecmult launches (0 = one thread per row: measured best -- a capped grid
                                     // leaves a tail of partly filled iterations: 3.19 ms -> 3.9 ms at 4 blocks per CU)
  u32 *gtable5 = nullptr;          // -DLAMD_G_LDS experiment: 52 x 32 entries of 5-bit windows of G, staged into LDS by the kernel
  unsigned keyed_lds_pad = 0;      // LAMD_KEYED_LDS_PAD: dynamic LDS bytes requested by the table-driven ecmult launches (occupancy limiter: 65536 = two blocks per CU)
  bool pairs = false;              // LAMD_PAIRS=1 (experiment, parity-green, measured slower: DESIGN.md 4): the table-driven ecmult of a large call pairs first (k_ecmult_keyed_pairs)
                                   // instead of one mixed addition per table entry (k_ecmult_keyed<false>)
  bool group_rows = true;          // LAMD_GROUP=0: the row lists of a large call stay in arrival order (no k_group_*)
  devbuf g_cnt, g_base, g_rank, g_touched, g_list7, g_list10;   // k_group_*: histogram / range bases over the cache's entry ids, rank per listed row, touched entries, grouped lists
  devbuf pairs_ws;                 // parking slots of k_ecmult_keyed_pairs (PAIRS_SLOTS x 48 bytes per resident lane)
  int keyed_waves = 3;             // LAMD_KEYED_WAVES: occupancy the bare-formula keyed kernels are compiled for (3: no spill; 4: a 5-dword spill, measured 60 % slower)
  size_t prep_batch = 16;  // signatures sharing one scalar inversion in the ECDSA prep (LAMD_PREP_BATCH)
  size_t prep_min_threads = 0;  // LAMD_PREP_MIN_THREADS: fewest prep threads of a large batch (0 = 256 per CU)
  u64 hash_seed = 0x243F6A8885A308D3ULL;
  size_t chunk = CHUNK_DEFAULT;  // LAMD_CHUNK_ROWS (tests force small chunks to exercise the splitting)
  hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  double last_ms[4] = {0, 0, 0, 0};
  // the dominant kernel alone: an event pair right around every large table-driven ecmult launch while timing is on (the
  // start event sits AFTER the waits for the prep / cold streams, so the interval is the launch itself, as rocprofv3 sees it);
  // read and summed per mode (ECDSA / BIP-340) by lamd_synchronize(), reset by lamd_set_timing()
  static const int MARK_SLOTS = 16;
  hipEvent_t ev_mark[MARK_SLOTS][MAX_LANES + 1] = {};  // lamd_results_mark(): one event per lane stream (+ the context's own)
  bool mark_set[MARK_SLOTS] = {};
  int mark_only[MARK_SLOTS] = {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1};  // lamd_results_mark_last(): the one lane whose event the slot holds (-1: every lane's)
  static const int KEV = 64;
  hipEvent_t kev[KEV][2] = {};
  int kev_mode[KEV] = {};
  int kev_n = 0;
  double keyed_ms_sum[2] = {0, 0};
  size_t keyed_launches[2] = {0, 0};
  // streaming queues (pinned host staging): QUEUE_SETS sets, one being filled while up to QUEUE_SETS - 1 flushed ones are
  // in flight (each on the lane picked at its flush), collected oldest first
  struct queue {
    // ONE pinned block per (set, kind), laid out for `cap` rows: keys | signatures | hashes (the order a flush sends them); h_c / h_b / h_a point
    // into it.  A flush that fills the set exactly (n == cap: a producer that reserves its batch in one go) crosses the bus as ONE copy.
    u8 *h_blk = nullptr;
    u8 *h_a = nullptr, *h_b = nullptr, *h_c = nullptr, *h_ok = nullptr;  // hash/msg, sig, key (inside h_blk); verdicts (own block)
    size_t cap = 0, n = 0;
    static size_t key_bytes_padded(size_t cap, size_t keybytes) { return (cap * keybytes + 63) & ~(size_t)63; }
    static size_t blk_bytes(size_t cap, size_t keybytes) { return key_bytes_padded(cap, keybytes) + cap * 96; }
    struct span { size_t row0; u32 ticket0; size_t count; };
    std::vector<span> tickets;  // rows [row0, row0 + count) of this queue return as verdicts [ticket0, ...) of the staging set
    // rows that were queued IN PLACE (lamd_queue_*_batch_inplace): they have a row range in the set and in its device twin like any other, but
    // their bytes stay in the caller's (registered) memory and cross the bus from there -- no host copy at all.  Ordered by row0, disjoint.
    struct foreign_span { size_t row0, count; const u8 *a, *b, *c; };
    std::vector<foreign_span> foreign;
    devbuf d_blk, d_ok;   // the device twin of h_blk (same layout), the verdicts
    hipEvent_t ev_keys = nullptr, ev_sigs = nullptr, ev_all = nullptr;  // behind the three H2D copies of a flush on the copy stream
    hipEvent_t ev_res = nullptr;     // behind the flush's last kernel on its lane: the verdict copy on the D2H stream waits for it
    bool small_flush = false;   // this flush ran as ONE k_small_verify launch over the staging rows themselves (h_ok + cap: the rows' shapes)
  };
  struct queue_set {
    queue q[Q_KINDS];
    hipEvent_t done = nullptr;
    hipEvent_t tail = nullptr;   // behind everything the flush put on its lane's main stream: `done` on the D2H stream waits for it too (a small flush has no copy)
    size_t rows = 0;  // triples queued into this set so far = the next ticket
  } qs[QUEUE_SETS];
  int q_open = 0;                 // the set being filled, -1 when every set is in flight
  int q_fifo[QUEUE_SETS] = {0};   // flushed sets, oldest first
  int q_inflight = 0;
  // Lanes: the device-pointer entry points rotate over LAMD_LANES (default 6) complete sub-contexts (own streams and
  // workspaces, the G table shared), so that the latency-bound front end of one call (key de-duplication, the count
  // read-back, table building) runs under the VALU-bound ecmult kernels of the calls before it.  A lane's `peer` is the
  // next lane; the root context (the handle the caller holds) keeps its own stream for staging, queues, generators.
  lamd_ctx *lane[MAX_LANES] = {};
  int nlanes = 0;
  lamd_ctx *peer = nullptr;
  lamd_ctx *last_lane = nullptr;
  lamd_ctx *last_chunk_lane = nullptr;  // where the last chunk of this lane's last call ran (itself or its peer)
  bool is_lane = false;
  int next_lane = 0;
  hipEvent_t ev_lane = nullptr, ev_join = nullptr;
  // root: the large table-driven ecmult launches of successive calls run one after the other (LAMD_ECMULT_CHAIN, see run_chunk)
  hipEvent_t ev_ecm[MAX_LANES + 1] = {};
  int ecm_last = -1;
  int ecm_chain = 0;
  size_t ecm_chain_min = 65536;
  u32 ecm_tail = 131072;             // LAMD_ECMULT_TAIL: work items of the low-priority second part (LAMD_ECMULT_CHAIN=2)
  bool bulk_stream = false;          // LAMD_BULK_STREAM=1 (experiment): every large table-driven ecmult launch runs on the lane's lowest-priority stream
  hipStream_t stream_lo = nullptr;   // lanes, LAMD_ECMULT_CHAIN=2: lowest-priority stream of that second part
  hipEvent_t ev_lo_go = nullptr, ev_lo_done = nullptr;
};

#define HIPCHK(ctx, call)                                                                           \
  do {                                                                                              \
    hipError_t e_ = (call);                                                                         \
    if (e_ != hipSuccess) {                                                                         \
      (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                               \
      return LAMD_ERR_HIP;                                                                          \
    }                                                                                               \
  } while (0)

static int ensure(lamd_ctx *ctx, devbuf *b, size_t bytes) {
  if (bytes <= b->cap) return LAMD_OK;
  // growing a workspace: an earlier, still running call of this context may be using the old one.  Do not rely on
  // hipFree() waiting for other streams -- drain the device first (rare: sizes settle after the first calls).
  if (b->p) {
    HIPCHK(ctx, hipDeviceSynchronize());
    HIPCHK(ctx, hipFree(b->p));
  }
  b->p = nullptr;
  b->cap = 0;
  size_t want = bytes + bytes / 4 + 4096;
  if (hipMalloc(&b->p, want) != hipSuccess) {
    want = bytes;
    hipError_t e = hipMalloc(&b->p, want);
    if (e != hipSuccess) {
      ctx->err = std::string("hipMalloc: ") + hipGetErrorString(e);
      return LAMD_ERR_NOMEM;
    }
  }
  b->cap = want;
  return LAMD_OK;
}
static void release(devbuf *b) {
  if (b->p) (void)hipFree(b->p);
  b->p = nullptr;
  b->cap = 0;
}


static inline unsigned blocks_for(size_t n) { return (unsigned)((n + 255) / 256); }
// the complete-formula kernels only see rows the bare formulas flagged (crafted scalars): a few blocks walk the list -- unless the
// lane's previous call reported many such rows (somebody is crafting them): then the launch covers the batch
static unsigned careful_grid(const lamd_ctx *ctx, size_t n) {
  const bool flood = ctx->h_plan && ((volatile const u32 *)ctx->h_plan)[P_SUSPECT] > 4096u;
  return flood || blocks_for(n) < 64u ? blocks_for(n) : 64u;
}

// blocks of a table-driven ecmult launch (the kernels walk their list with a grid stride)
static unsigned keyed_grid(const lamd_ctx *ctx, size_t n) {
  const unsigned full = (unsigned)((n + LAMD_KEYED_THREADS - 1) / LAMD_KEYED_THREADS), cap = (unsigned)ctx->prop.multiProcessorCount * ctx->keyed_blocks_per_cu;
  return ctx->keyed_blocks_per_cu && cap < full ? cap : full;
}

// ROCm maps HIP streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and streams that share a queue serialise.  The
// engine uses up to 9 streams (root + two lanes, each with a prep and a cold-row side stream); with 4 queues one lane's prep
// kernel lands behind the other lane's ecmult kernel and the lanes stop overlapping (measured: two 484-row flushes in flight
// 880 -> 1 640 batches/s, the 2 M-row step 170 -> 182 M verifies/s with >= 8 queues).  The runtime reads the variable once,
// at its first API call: exporting GPU_MAX_HW_QUEUES=16 before that is the HOST's job (documented precondition of lamd_init in
// include/lightning_amd.h -- a drop-in linked into somebody else's daemon does not edit the process environment);
// lamd_info.hw_queues_env reports what lamd_init() found.
static int hw_queues_from_env(void) {
  const char *v = getenv("GPU_MAX_HW_QUEUES");
  return v && *v ? atoi(v) : 0;
}

extern "C" const char *lamd_version(void) { return "lightning_amd 0.1 (gfx950)"; }

extern "C" const char *lamd_last_error(const lamd_ctx *ctx) { return ctx ? ctx->err.c_str() : "no context"; }

static int init_known_answers(lamd_ctx *ctx);
static int cache_alloc(lamd_ctx *owner, lamd_ctx::key_cache *kc, size_t keys7, size_t keys10, bool shared);
static int cache_reset(lamd_ctx *root);
static int cache_maybe_reset(lamd_ctx *root);

static int create_streams(lamd_ctx *ctx) {
  HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
  HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking));
  // The cold-row ladder shares the prep stream (it starts long after the lane's preparation has finished): two streams per lane
  // instead of three -- 231 against 216-223 M verifies/s in the cold loop (profiles/r03_fused_front.txt; fewer hardware queues in
  // use, see hw_queues_from_env).  LAMD_MERGE_SIDE=0 gives the ladder its own stream again.
  // NOTE for whoever queues work on stream3: by default it IS stream2.  Every stream3 launch is fenced by ev_fork (before) / ev_cold (after); a
  // launch that waited on an event recorded LATER on stream2 (ev_prep of the next chunk, say) would wait for itself.
  if (!getenv("LAMD_MERGE_SIDE") || atoi(getenv("LAMD_MERGE_SIDE")) != 0) ctx->stream3 = ctx->stream2;
  else HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->stream3, hipStreamNonBlocking));
  HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_cold, hipEventDisableTiming));
  HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_keys, hipEventDisableTiming));
  HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_sigs, hipEventDisableTiming));
  HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
  HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_prep, hipEventDisableTiming));
  HIPCHK(ctx, hipHostMalloc((void **)&ctx->h_plan, (P_WORDS + C_WORDS) * 4, hipHostMallocDefault));
  memset(ctx->h_plan, 0, (P_WORDS + C_WORDS) * 4);
  if (!ctx->is_lane)
    for (auto &e : ctx->ev_pub) HIPCHK(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
  HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_lane, hipEventDisableTiming));
  HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
  ctx->bulk_stream = getenv("LAMD_BULK_STREAM") && atoi(getenv("LAMD_BULK_STREAM")) != 0;
  if ((getenv("LAMD_ECMULT_CHAIN") && atoi(getenv("LAMD_ECMULT_CHAIN")) == 2) || ctx->bulk_stream) {  // experiments: (the tail of) a large ecmult launch on a lowest-priority stream
    int lo = 0, hi = 0;
    HIPCHK(ctx, hipDeviceGetStreamPriorityRange(&lo, &hi));   // lo = least priority (numerically greatest)
    HIPCHK(ctx, hipStreamCreateWithPriority(&ctx->stream_lo, hipStreamNonBlocking, lo));
    HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_lo_go, hipEventDisableTiming));
    HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_lo_done, hipEventDisableTiming));
  }
  for (auto &e : ctx->ev) HIPCHK(ctx, hipEventCreate(&e));
  if (!ctx->is_lane) {
    for (auto &qs : ctx->qs) {
      HIPCHK(ctx, hipEventCreateWithFlags(&qs.done, hipEventDisableTiming));
      HIPCHK(ctx, hipEventCreateWithFlags(&qs.tail, hipEventDisableTiming));
      for (auto &q : qs.q)
        for (hipEvent_t *e : {&q.ev_keys, &q.ev_sigs, &q.ev_all, &q.ev_res}) HIPCHK(ctx, hipEventCreateWithFlags(e, hipEventDisableTiming));
    }
    if (const char *w = getenv("LAMD_COPY_ONE")) ctx->copy_one = atoi(w) != 0;
    if (const char *w = getenv("LAMD_COPY_EVENTS")) ctx->copy_events = atoi(w) < 0 ? 0 : atoi(w) > 3 ? 3 : atoi(w);
    if (const char *w = getenv("LAMD_COPY_STREAMS")) ctx->n_copy_streams = atoi(w) < 1 ? 1 : atoi(w) > lamd_ctx::MAX_COPY_STREAMS ? lamd_ctx::MAX_COPY_STREAMS : atoi(w);
    // (the streams themselves are created by the first flush that needs them: an engine that is only ever handed device pointers -- bench.py's
    // resident loop, a rank of the collective path -- keeps their hardware-queue slots free for its lanes)
    if (const char *w = getenv("LAMD_D2H_STREAM")) ctx->use_d2h_stream = atoi(w) != 0;
    if (ctx->use_d2h_stream) HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->d2h_stream, hipStreamNonBlocking));
    for (auto &e : ctx->ev_ecm) HIPCHK(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    if (const char *w = getenv("LAMD_ECMULT_CHAIN")) ctx->ecm_chain = atoi(w);
    if (const char *w = getenv("LAMD_ECMULT_TAIL")) ctx->ecm_tail = (u32)atol(w);
    if (const char *w = getenv("LAMD_COPY_STREAM")) ctx->use_copy_stream = atoi(w) != 0;
  }
  return LAMD_OK;
}
static int make_lanes(lamd_ctx *root, int count) {
  for (int i = 0; i < count; i++) {
    lamd_ctx *L = new (std::nothrow) lamd_ctx();
    if (!L) return LAMD_ERR_NOMEM;
    root->lane[i] = L;
    L->is_lane = true;
    L->root = root;
    L->lane_id = i + 1;
    L->cache_mode = root->cache_mode;
    L->device = root->device;
    L->prop = root->prop;
    L->gtable = root->gtable;
    L->gtable5 = root->gtable5;
    L->ecmult_waves = root->ecmult_waves;
    L->keyed_waves = root->keyed_waves;
    L->pairs = root->pairs;
    L->group_rows = root->group_rows;
    L->keyed_lds_pad = root->keyed_lds_pad;
    L->keyed_blocks_per_cu = root->keyed_blocks_per_cu;
    L->prep_batch = root->prep_batch;
    L->prep_min_threads = root->prep_min_threads;
    L->hash_seed = root->hash_seed;
    L->chunk = root->chunk;
    L->keyed_mode = root->keyed_mode;
    L->keyed_min_rows = root->keyed_min_rows;
    L->small_fused = root->small_fused;
    L->fused_front = root->fused_front;
    L->early_cold = root->early_cold;
    L->keyed_min_uses = root->keyed_min_uses;
    L->keyed_dense_uses = root->keyed_dense_uses;
    L->keyed_teeth = root->keyed_teeth;
    const int rc = create_streams(L);
    if (rc != LAMD_OK) { root->err = L->err; return rc; }
  }
  root->nlanes = count;
  for (int i = 0; i < count; i++) root->lane[i]->peer = root->lane[(i + 1) % count];
  return LAMD_OK;
}
// the lane the next device-pointer call runs on, ordered after whatever is already queued on the context's own stream
static int pick_lane(lamd_ctx *root, lamd_ctx **out) {
  *out = root;
  const int rcc = cache_maybe_reset(root);
  if (rcc != LAMD_OK) return rcc;
  if (!root->lane[0]) return LAMD_OK;
  lamd_ctx *L = root->lane[root->next_lane];
  root->next_lane = (root->next_lane + 1) % root->nlanes;
  root->last_lane = L;
  L->timing = root->timing;
  HIPCHK(root, hipEventRecord(root->ev_lane, root->stream));
  HIPCHK(root, hipStreamWaitEvent(L->stream, root->ev_lane, 0));
  *out = L;
  return LAMD_OK;
}

extern "C" int lamd_init(lamd_ctx **out, int device) {
  if (!out) return LAMD_ERR_ARG;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return LAMD_ERR_NO_DEVICE;
  if (device < 0 || device >= count) return LAMD_ERR_ARG;
  lamd_ctx *ctx = new (std::nothrow) lamd_ctx();
  if (!ctx) return LAMD_ERR_NOMEM;
  ctx->device = device;
  *out = ctx;  // returned even on failure so the caller can read lamd_last_error(); shutdown is still valid
  HIPCHK(ctx, hipSetDevice(device));
  HIPCHK(ctx, hipGetDeviceProperties(&ctx->prop, device));
  if (strncmp(ctx->prop.gcnArchName, "gfx950", 6) != 0) {
    ctx->err = std::string("unsupported architecture ") + ctx->prop.gcnArchName + " (built for gfx950 only)";
    return LAMD_ERR_NO_DEVICE;
  }
  if (const char *w = getenv("LAMD_ECMULT_WAVES")) ctx->ecmult_waves = atoi(w);
  if (const char *w = getenv("LAMD_KEYED_WAVES")) ctx->keyed_waves = atoi(w) == 4 ? 4 : 3;
  if (const char *w = getenv("LAMD_KEYED_LDS_PAD")) ctx->keyed_lds_pad = (unsigned)atoi(w);
  if (const char *w = getenv("LAMD_PAIRS")) ctx->pairs = atoi(w) != 0;
  if (const char *w = getenv("LAMD_GROUP")) ctx->group_rows = atoi(w) != 0;
  if (const char *w = getenv("LAMD_KEYED_BLOCKS_PER_CU")) ctx->keyed_blocks_per_cu = (unsigned)atoi(w);
  if (const char *w = getenv("LAMD_SPIN_US")) ctx->spin_us = (unsigned)atoi(w);
  if (const char *w = getenv("LAMD_PRIO")) ctx->prio_mask = (u32)atoi(w);
  if (const char *w = getenv("LAMD_FUSED_FRONT")) ctx->fused_front = atoi(w) != 0;
  if (const char *w = getenv("LAMD_EARLY_COLD")) ctx->early_cold = atoi(w) != 0;
  if (const char *w = getenv("LAMD_KC_TREE")) ctx->kc_tree = atoi(w) != 0;
  if (const char *w = getenv("LAMD_PREP_BATCH")) ctx->prep_batch = atoi(w) < 1 ? 1 : (size_t)atoi(w);
  if (const char *w = getenv("LAMD_PREP_MIN_THREADS")) ctx->prep_min_threads = atol(w) < 0 ? 0 : (size_t)atol(w);
  {
    std::random_device rd;
    ctx->hash_seed = ((u64)rd() << 32) ^ (u64)rd() ^ 0x243F6A8885A308D3ULL;
  }
  if (const char *w = getenv("LAMD_CHUNK_ROWS")) ctx->chunk = (size_t)atoll(w) < 4 ? 4 : (size_t)atoll(w);
  if (const char *w = getenv("LAMD_KEYED")) ctx->keyed_mode = atoi(w);
  if (const char *w = getenv("LAMD_KEYED_MIN_USES")) ctx->keyed_min_uses = atof(w);
  if (const char *w = getenv("LAMD_KEYED_DENSE_USES")) ctx->keyed_dense_uses = atof(w);
  if (const char *w = getenv("LAMD_KEYED_TEETH")) ctx->keyed_teeth = atoi(w) == 7 ? 7 : (atoi(w) == 10 ? 10 : 0);
  if (const char *w = getenv("LAMD_KEYED_MIN_ROWS")) ctx->keyed_min_rows = (size_t)atoll(w);
  if (const char *w = getenv("LAMD_SMALL_FUSED")) ctx->small_fused = atoi(w) != 0;
  if (const char *w = getenv("LAMD_CACHE")) ctx->cache_mode = atoi(w) != 0;
  if (const char *w = getenv("LAMD_SMALL_KERNEL")) ctx->small_kernel = atoi(w) != 0;
  if (const char *w = getenv("LAMD_CACHE_KEYS")) ctx->cache_keys = (size_t)atoll(w) < 64 ? 64 : (size_t)atoll(w);
  if (const char *w = getenv("LAMD_CACHE_KEYS10")) ctx->cache_keys10 = (size_t)atoll(w) < 16 ? 16 : (size_t)atoll(w);
  int rc = create_streams(ctx);
  if (rc != LAMD_OK) return rc;
  // window bases B_w = 2^(16 w) G, computed here with the same group code the kernels use (64 doublings each)
  std::vector<u32> bases(GTABLE_WINDOWS * 16);
  {
    const u32 gx[8] = LAMD_GX, gy[8] = LAMD_GY;
    u32 cur[16];
    memcpy(cur, gx, 32);
    memcpy(cur + 8, gy, 32);
    for (int w = 0; w < GTABLE_WINDOWS; w++) {
      memcpy(&bases[w * 16], cur, 64);
      gej b = gej_from_ge(ge_from_words(cur, cur + 8));
      for (int i = 0; i < GTABLE_WINDOW_BITS; i++) b = gej_double(b);
      const fe zi = fe_inv(fe_norm_weak(b.z));
      const fe zi2 = fe_sqr(zi);
      fe_to_words(cur, fe_normalize(fe_mul(b.x, zi2)));
      fe_to_words(cur + 8, fe_normalize(fe_mul(b.y, fe_mul(zi2, zi))));
    }
  }
  HIPCHK(ctx, hipMemcpyToSymbol(HIP_SYMBOL(g_prio), &ctx->prio_mask, sizeof(u32)));
  {
    // ONE static table of G per device and process, shared by every context on that device (read-only after its build) and freed with the last of
    // them: a second engine in a process (a sidecar's reload, bench.py's cache-on engine, a test suite's two dozen) costs neither another 11 GiB nor
    // another 0.36 s
    shared_gtable &sg = gtable_slot(device);
    std::lock_guard<std::mutex> lk(sg.mu);
    if (sg.refs == 0) {
      u32 *d_bases = nullptr;
      const char *what = "hipMalloc (window bases)";
      hipError_t e = hipMalloc(&d_bases, bases.size() * 4);
      if (e == hipSuccess) { what = "hipMemcpy (window bases)"; e = hipMemcpy(d_bases, bases.data(), bases.size() * 4, hipMemcpyHostToDevice); }
      // 11 GiB: a partitioned or busy device may not have them -- say so (include/lightning_amd.h, lamd_init; one table per process and device)
      if (e == hipSuccess) { what = "hipMalloc (static G table, 11 GiB)"; e = hipMalloc(&sg.p, GTABLE_BYTES); }
      if (e == hipSuccess) {
        what = "k_gtable_build";
        hipLaunchKernelGGL(k_gtable_build, dim3(blocks_for(GTABLE_ENTRIES)), dim3(256), 0, ctx->stream, sg.p, d_bases);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
      }
      if (d_bases) (void)hipFree(d_bases);   // on every path (round 4 leaked it when a later step failed)
      if (e != hipSuccess) {
        if (sg.p) (void)hipFree(sg.p);
        sg.p = nullptr;
        (void)hipGetLastError();
        ctx->err = std::string(what) + ": " + hipGetErrorString(e);
        return e == hipErrorOutOfMemory ? LAMD_ERR_NOMEM : LAMD_ERR_HIP;
      }
    }
    sg.refs++;
    ctx->gtable = sg.p;
    ctx->gtable_device = device;
  }
#if defined(LAMD_G_LDS)
  {  // experiment: the 5-bit-window table the kernel stages into LDS (computed on the host: 1 664 entries)
    std::vector<u32> t5(GLDS_WORDS, 0);
    const u32 gx[8] = LAMD_GX, gy[8] = LAMD_GY;
    gej b = gej_from_ge(ge_from_words(gx, gy));
    for (int w = 0; w < GLDS_WINDOWS; w++) {
      const fe zi = fe_inv(fe_norm_weak(b.z));
      const fe zi2 = fe_sqr(zi);
      ge base;
      base.x = fe_normalize(fe_mul(b.x, zi2));
      base.y = fe_normalize(fe_mul(b.y, fe_mul(zi2, zi)));
      gej acc = gej_infinity();
      for (u32 d = 1; d < (1u << GLDS_BITS); d++) {
        acc = gej_add_ge(acc, base, false);
        const fe ai = fe_inv(fe_norm_weak(acc.z));
        const fe ai2 = fe_sqr(ai);
        fe_to_words(&t5[((w << GLDS_BITS) + d) * 16], fe_normalize(fe_mul(acc.x, ai2)));
        fe_to_words(&t5[((w << GLDS_BITS) + d) * 16 + 8], fe_normalize(fe_mul(acc.y, fe_mul(ai2, ai))));
      }
      for (int i = 0; i < GLDS_BITS; i++) b = gej_double(b);
    }
    HIPCHK(ctx, hipMalloc(&ctx->gtable5, GLDS_WORDS * 4));
    HIPCHK(ctx, hipMemcpy(ctx->gtable5, t5.data(), GLDS_WORDS * 4, hipMemcpyHostToDevice));
    HIPCHK(ctx, hipFuncSetAttribute((const void *)k_ecmult_keyed<false, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, GLDS_WORDS * 4));
    ctx->keyed_lds_pad = GLDS_WORDS * 4;
  }
#endif
  if (ctx->cache_mode) {  // the shared key-table cache: entries, index, one table pool per comb shape
    if ((rc = cache_alloc(ctx, &ctx->cache_store, ctx->cache_keys, ctx->cache_keys10, true)) != LAMD_OK) return rc;
    if ((rc = cache_reset(ctx)) != LAMD_OK) return rc;
    ctx->cache_resets = 0;
  }
  const char *lanes = getenv("LAMD_LANES");
  const int nl = lanes ? atoi(lanes) : 6;   // (6 since round 4: with the row-grouping stage a call's launch chain is longer; 4 / 5 / 6 / 7 / 8 lanes = 246 / 243 / 256 / 249 / 252 M verifies/s cold)
  if (nl > 1) {
    rc = make_lanes(ctx, nl > MAX_LANES ? MAX_LANES : nl);
    if (rc != LAMD_OK) return rc;
  }
  return init_known_answers(ctx);
}

extern "C" void lamd_shutdown(lamd_ctx *ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  for (auto &L : ctx->lane) {
    if (L) lamd_shutdown(L);
    L = nullptr;
  }
  for (hipStream_t cs : ctx->copy_streams)
    if (cs) (void)hipStreamSynchronize(cs);
  if (ctx->d2h_stream) (void)hipStreamSynchronize(ctx->d2h_stream);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  for (devbuf *b : {&ctx->row_ent, &ctx->kd_table, &ctx->kd_rep, &ctx->kd_uid, &ctx->kd_uniq, &ctx->kd_count, &ctx->kd_newent, &ctx->plan,
                    &ctx->kt_fin, &ctx->hk7_row, &ctx->hk7_ent, &ctx->hk7_slot, &ctx->hk7_qwords, &ctx->hk7_keyok, &ctx->hk7_scratch,
                    &ctx->hk10_row, &ctx->hk10_ent, &ctx->hk10_slot, &ctx->hk10_qwords, &ctx->hk10_keyok, &ctx->hk10_scratch, &ctx->list7,
                    &ctx->list10, &ctx->listcold, &ctx->listcold_ok, &ctx->keyok_row, &ctx->pairs_ws, &ctx->g_cnt, &ctx->g_base, &ctx->g_rank, &ctx->g_touched, &ctx->g_list7, &ctx->g_list10, &ctx->cache_store.ents, &ctx->cache_store.index, &ctx->cache_store.pool7,
                    &ctx->cache_store.pool10, &ctx->cache_store.counters})
    release(b);
  for (auto &e : ctx->ev_pub)
    if (e) (void)hipEventDestroy(e);
  if (ctx->stream2) { (void)hipStreamSynchronize(ctx->stream2); (void)hipStreamDestroy(ctx->stream2); }
  if (ctx->stream3 && ctx->stream3 != ctx->stream2) { (void)hipStreamSynchronize(ctx->stream3); (void)hipStreamDestroy(ctx->stream3); }
  if (ctx->ev_cold) (void)hipEventDestroy(ctx->ev_cold);
  if (ctx->ev_keys) (void)hipEventDestroy(ctx->ev_keys);
  if (ctx->ev_sigs) (void)hipEventDestroy(ctx->ev_sigs);
  if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
  if (ctx->ev_prep) (void)hipEventDestroy(ctx->ev_prep);
  for (devbuf *b : {&ctx->small_done, &ctx->recs, &ctx->qwords, &ctx->keyok, &ctx->slots, &ctx->vbuf, &ctx->in_a, &ctx->in_b, &ctx->in_c, &ctx->out,
                    &ctx->g_msgs, &ctx->g_off, &ctx->g_ids, &ctx->g_rowbase, &ctx->g_hash, &ctx->g_sig, &ctx->g_pub,
                    &ctx->g_malformed, &ctx->g_ok, &ctx->g_verdict})
    release(b);
  for (auto &qs : ctx->qs) {
    for (auto &q : qs.q) {
      q.h_a = q.h_b = q.h_c = nullptr;
      for (u8 **h : {&q.h_blk, &q.h_ok})
        if (*h) (void)hipHostFree(*h);
      for (devbuf *b : {&q.d_blk, &q.d_ok}) release(b);
      for (hipEvent_t e : {q.ev_keys, q.ev_sigs, q.ev_all, q.ev_res})
        if (e) (void)hipEventDestroy(e);
    }
    if (qs.done) (void)hipEventDestroy(qs.done);
    if (qs.tail) (void)hipEventDestroy(qs.tail);
  }
  for (hipStream_t cs : ctx->copy_streams)
    if (cs) (void)hipStreamDestroy(cs);
  if (ctx->d2h_stream) (void)hipStreamDestroy(ctx->d2h_stream);
  for (auto &e : ctx->ev_ecm)
    if (e) (void)hipEventDestroy(e);
  if (ctx->gtable && !ctx->is_lane) {
    shared_gtable &sg = gtable_slot(ctx->gtable_device);
    std::lock_guard<std::mutex> lk(sg.mu);
    if (sg.refs > 0 && --sg.refs == 0) {
      (void)hipFree(sg.p);
      sg.p = nullptr;
    }
    ctx->gtable = nullptr;
  }
  if (ctx->h_plan) (void)hipHostFree(ctx->h_plan);
  if (ctx->h_small) (void)hipHostFree(ctx->h_small);
  if (ctx->h_tmpl) (void)hipHostFree(ctx->h_tmpl);
  if (ctx->stream_lo) { (void)hipStreamSynchronize(ctx->stream_lo); (void)hipStreamDestroy(ctx->stream_lo); }
  for (hipEvent_t e : {ctx->ev_lo_go, ctx->ev_lo_done})
    if (e) (void)hipEventDestroy(e);
  if (ctx->ev_lane) (void)hipEventDestroy(ctx->ev_lane);
  if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
  for (auto &e : ctx->ev)
    if (e) (void)hipEventDestroy(e);
  for (auto &pair : ctx->kev)
    for (auto &e : pair)
      if (e) (void)hipEventDestroy(e);
  for (auto &slot : ctx->ev_mark)
    for (auto &e : slot)
      if (e) (void)hipEventDestroy(e);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

extern "C" void *lamd_stream(lamd_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

extern "C" int lamd_synchronize(lamd_ctx *ctx) {
  if (!ctx) return LAMD_ERR_ARG;
  for (lamd_ctx *L : ctx->lane) {
    if (!L) continue;
    const int rc = lamd_synchronize(L);
    if (rc != LAMD_OK) { ctx->err = L->err; return rc; }
  }
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->d2h_stream) HIPCHK(ctx, hipStreamSynchronize(ctx->d2h_stream));   // (root only: the verdict copies of the flushes in flight)
  if (ctx->timing && ctx->ev_recorded) {
    for (int i = 0; i < 4; i++) {
      float ms = 0;
      if (hipEventElapsedTime(&ms, ctx->ev[i], ctx->ev[i + 1]) == hipSuccess) ctx->last_ms[i] = ms;
    }
    (void)hipGetLastError();
  }
  for (int i = 0; i < ctx->kev_n; i++) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, ctx->kev[i][0], ctx->kev[i][1]) == hipSuccess) {
      ctx->keyed_ms_sum[ctx->kev_mode[i]] += ms;
      ctx->keyed_launches[ctx->kev_mode[i]]++;
    }
  }
  (void)hipGetLastError();
  ctx->kev_n = 0;
  return LAMD_OK;
}

// `stream` (a hipStream_t of the caller) waits for every verification submitted so far -- without blocking the host
extern "C" int lamd_stream_wait_results(lamd_ctx *ctx, void *stream) {
  if (!ctx) return LAMD_ERR_ARG;
  for (int i = 0; i <= MAX_LANES; i++) {
    lamd_ctx *L = i < MAX_LANES ? ctx->lane[i] : ctx;
    if (!L) continue;
    HIPCHK(ctx, hipEventRecord(L->ev_join, L->stream));
    HIPCHK(ctx, hipStreamWaitEvent((hipStream_t)stream, L->ev_join, 0));
  }
  return LAMD_OK;
}
// verification submitted from now on waits for what `stream` holds at this moment (e.g. a consumer of an earlier result buffer)
extern "C" int lamd_wait_event(lamd_ctx *ctx, void *event) {
  if (!ctx || !event) return LAMD_ERR_ARG;
  HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, (hipEvent_t)event, 0));
  return LAMD_OK;
}
extern "C" int lamd_results_mark(lamd_ctx *ctx, int slot) {
  if (!ctx || slot < 0 || slot >= lamd_ctx::MARK_SLOTS) return LAMD_ERR_ARG;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  for (int i = 0; i <= MAX_LANES; i++) {
    lamd_ctx *L = i < MAX_LANES ? ctx->lane[i] : ctx;
    hipEvent_t &e = ctx->ev_mark[slot][i];
    if (!L) continue;
    if (!e) HIPCHK(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    HIPCHK(ctx, hipEventRecord(e, L->stream));
  }
  ctx->mark_set[slot] = true;
  ctx->mark_only[slot] = -1;
  return LAMD_OK;
}
// the same for the call submitted LAST only: one event, on the lane that ran it (a call's verdict copy is the last thing on its lane, chunks on
// the peer lane joined before it).  A consumer of that call's verdicts needs no more; marking every lane costs seven event records per call
// and makes the consumer wait for whatever the other lanes were doing.
extern "C" int lamd_results_mark_last(lamd_ctx *ctx, int slot) {
  if (!ctx || slot < 0 || slot >= lamd_ctx::MARK_SLOTS) return LAMD_ERR_ARG;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  int idx = MAX_LANES;
  lamd_ctx *L = ctx;
  for (int i = 0; i < MAX_LANES; i++)
    if (ctx->lane[i] && ctx->lane[i] == ctx->last_lane) { idx = i; L = ctx->lane[i]; }
  hipEvent_t &e = ctx->ev_mark[slot][idx];
  if (!e) HIPCHK(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
  HIPCHK(ctx, hipEventRecord(e, L->stream));
  ctx->mark_set[slot] = true;
  ctx->mark_only[slot] = idx;
  return LAMD_OK;
}
extern "C" int lamd_stream_wait_mark(lamd_ctx *ctx, int slot, void *stream) {
  if (!ctx || slot < 0 || slot >= lamd_ctx::MARK_SLOTS) return LAMD_ERR_ARG;
  if (!ctx->mark_set[slot]) {
    ctx->err = "lamd_stream_wait_mark: slot was never marked";
    return LAMD_ERR_STATE;
  }
  if (ctx->mark_only[slot] >= 0) {
    HIPCHK(ctx, hipStreamWaitEvent((hipStream_t)stream, ctx->ev_mark[slot][ctx->mark_only[slot]], 0));
    return LAMD_OK;
  }
  for (int i = 0; i <= MAX_LANES; i++)
    if (ctx->ev_mark[slot][i]) HIPCHK(ctx, hipStreamWaitEvent((hipStream_t)stream, ctx->ev_mark[slot][i], 0));
  return LAMD_OK;
}

extern "C" int lamd_wait_stream(lamd_ctx *ctx, void *stream) {
  if (!ctx) return LAMD_ERR_ARG;
  HIPCHK(ctx, hipEventRecord(ctx->ev_fork, (hipStream_t)stream));
  HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_fork, 0));
  return LAMD_OK;
}

extern "C" int lamd_set_timing(lamd_ctx *ctx, int enable) {
  if (!ctx) return LAMD_ERR_ARG;
  ctx->timing = enable != 0;
  for (lamd_ctx *L : ctx->lane)
    if (L) L->timing = ctx->timing;
  // a new measurement interval: forget the launch durations summed so far (event pairs still pending are dropped)
  for (int i = 0; i <= MAX_LANES; i++) {
    lamd_ctx *L = i < MAX_LANES ? ctx->lane[i] : ctx;
    if (!L) continue;
    L->kev_n = 0;
    L->keyed_ms_sum[0] = L->keyed_ms_sum[1] = 0;
    L->keyed_launches[0] = L->keyed_launches[1] = 0;
  }
  return LAMD_OK;
}

// rows per launch sequence from the next call on (include/lightning_amd.h)
extern "C" int lamd_set_chunk_rows(lamd_ctx *ctx, size_t rows) {
  if (!ctx) return LAMD_ERR_ARG;
  const size_t c = rows == 0 ? CHUNK_DEFAULT : rows < 4096 ? 4096 : rows > CHUNK_DEFAULT ? CHUNK_DEFAULT : rows;
  ctx->chunk = c;
  for (lamd_ctx *L : ctx->lane)
    if (L) L->chunk = c;
  return LAMD_OK;
}

// scheduling of the large table-driven ecmult launches (include/lightning_amd.h); takes effect with the next call
extern "C" int lamd_set_ecmult_chain(lamd_ctx *ctx, int enable) {
  if (!ctx) return LAMD_ERR_ARG;
  ctx->ecm_chain = enable != 0;
  ctx->ecm_last = -1;
  return LAMD_OK;
}

static int get_info_of(lamd_ctx *ctx, lamd_info *info);
extern "C" int lamd_get_info(lamd_ctx *ctx, lamd_info *info) {
  if (!ctx || !info) return LAMD_ERR_ARG;
  const int rc = get_info_of(ctx->last_lane ? ctx->last_lane : ctx, info);
  info->lanes = ctx->nlanes ? ctx->nlanes : 1;
  return rc;
}
// the same for one lane (0 or 1): with two lanes, alternate calls land on alternate lanes
extern "C" int lamd_get_lane_info(lamd_ctx *ctx, int lane, lamd_info *info) {
  if (!ctx || !info || lane < 0 || lane >= (ctx->nlanes ? ctx->nlanes : 1)) return LAMD_ERR_ARG;
  const int rc = get_info_of(ctx->lane[lane] ? ctx->lane[lane] : ctx, info);
  info->lanes = ctx->nlanes ? ctx->nlanes : 1;
  return rc;
}
static int get_info_of(lamd_ctx *ctx, lamd_info *info) {
  lamd_ctx *root = ctx->root ? ctx->root : ctx;
  const lamd_ctx *self = ctx;  // the launch-duration sums are this lane's own (a sum over the lanes counts every launch once)
  if (ctx->last_chunk_lane) ctx = ctx->last_chunk_lane;
  memset(info, 0, sizeof(*info));
  info->device = ctx->device;
  info->compute_units = ctx->prop.multiProcessorCount;
  strncpy(info->arch, ctx->prop.gcnArchName, sizeof(info->arch) - 1);
  info->gtable_bytes = GTABLE_BYTES;
  info->hw_queues_env = hw_queues_from_env();
  info->queue_sets = QUEUE_SETS;
  for (int i = 0; i < 4; i++) info->last_kernel_ms[i] = ctx->last_ms[i];
  for (int i = 0; i < 2; i++) {
    info->keyed_ecmult_ms_sum[i] = self->keyed_ms_sum[i];
    info->keyed_ecmult_launches[i] = self->keyed_launches[i];
  }
  info->last_mode = ctx->last_mode;
  // the counts of the last keyed call come from the pinned copy its stream wrote at the end: exact after lamd_synchronize()
  if (ctx->last_keyed_call && ctx->h_plan) {
    const volatile u32 *h = ctx->h_plan;
    info->last_unique_keys = h[P_UNIQ];
    info->last_hot_rows = (size_t)h[P_L7] + h[P_L10];
    info->last_keyed = h[P_L10] ? 10 : (h[P_L7] ? 7 : 0);
    info->last_cache_hits = h[P_HITS];
    info->last_cold_rows = h[P_COLD];
    info->last_new_tables = (size_t)h[P_HK7] + h[P_HK10];
    info->last_suspect_rows = h[P_SUSPECT];
  }
  info->cache_enabled = root->cache_mode != 0 && root->cache_store.shared;
  info->cache_entries = root->cache_hwm[0];
  info->cache_capacity = root->cache_store.cap_ent;
  info->cache_resets = root->cache_resets;
  return LAMD_OK;
}

static void launch_prep(lamd_ctx *ctx, int mode, size_t n, const u8 *d_a, const u8 *d_sig, const u8 *d_key, prep_rec *recs) {
  if (mode == MODE_ECDSA) {
    // enough threads to fill the chip, few enough that each amortises its inversion over ~16 signatures
    size_t threads = (n + ctx->prep_batch - 1) / ctx->prep_batch;
    const size_t min_threads = ctx->prep_min_threads ? ctx->prep_min_threads : (size_t)ctx->prop.multiProcessorCount * 256;
    if (threads < min_threads) threads = n < min_threads ? n : min_threads;
    hipLaunchKernelGGL(k_ecdsa_prep, dim3(blocks_for(threads)), dim3(256), 0, ctx->stream, n, d_a, d_sig, recs);
  } else {
    hipLaunchKernelGGL(k_schnorr_prep, dim3(blocks_for(n)), dim3(256), 0, ctx->stream, n, d_a, d_key, d_sig, recs);
  }
}
static size_t final_threads(lamd_ctx *ctx, size_t n) {
  size_t threads = (n + 15) / 16;
  const size_t min_threads = (size_t)ctx->prop.multiProcessorCount * 256;
  if (threads < min_threads) threads = n < min_threads ? n : min_threads;
  return threads;
}

// Per-signature path over up to `m` work items (all rows when idx == nullptr, else the listed rows; count != nullptr: the
// real number of items is *count, on the device): keys -> ladder.  Prep records (indexed by row) must already be queued
// on stream2 / finished (ev_prep).
static int launch_direct(lamd_ctx *ctx, int mode, size_t m, const u32 *idx, const u32 *count, const prep_rec *recs, const u8 *d_sig,
                         const u8 *d_key, int keylen, size_t keystride, u32 *fin, u8 *keyok_row, u8 *d_ok, bool time_it, u32 *plan = nullptr) {
  int rc;
  if ((rc = ensure(ctx, &ctx->qwords, m * 64)) != LAMD_OK) return rc;
  if ((rc = ensure(ctx, &ctx->keyok, m)) != LAMD_OK) return rc;
  if ((rc = ensure(ctx, &ctx->slots, m * SLOT_WORDS * 4)) != LAMD_OK) return rc;
  auto kern = ctx->ecmult_waves == 2 ? k_ecmult<2> : ctx->ecmult_waves == 4 ? k_ecmult<4> : k_ecmult<3>;
  if (idx && count && plan) {
    // cold rows of a partitioned call: parse, reject the rows whose key does not parse, compact the rest (k_keys_cold)
    if ((rc = ensure(ctx, &ctx->listcold_ok, m * 4)) != LAMD_OK) return rc;
    hipLaunchKernelGGL(k_keys_cold, dim3(blocks_for(m)), dim3(256), 0, ctx->stream, m, d_key, keylen, keystride, idx, count, plan + P_COLDOK,
                       (u32 *)ctx->listcold_ok.p, (u32 *)ctx->qwords.p, keyok_row, d_ok);
    HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_prep, 0));
    hipLaunchKernelGGL(kern, dim3(blocks_for(m)), dim3(256), 0, ctx->stream, m, recs, (const u32 *)ctx->qwords.p, (const u8 *)nullptr, d_sig, mode,
                       (const u32 *)ctx->gtable, (u32 *)ctx->slots.p, (const u32 *)ctx->listcold_ok.p, fin, keyok_row, d_ok, (const u32 *)(plan + P_COLDOK));
    return LAMD_OK;
  }
  hipLaunchKernelGGL(k_keys, dim3(blocks_for(m)), dim3(256), 0, ctx->stream, m, d_key, keylen, keystride, idx, (u32 *)ctx->qwords.p,
                     (u8 *)ctx->keyok.p, count);
  if (time_it) HIPCHK(ctx, hipEventRecord(ctx->ev[2], ctx->stream));
  HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_prep, 0));
  hipLaunchKernelGGL(kern, dim3(blocks_for(m)), dim3(256), 0, ctx->stream, m, recs, (const u32 *)ctx->qwords.p, (const u8 *)ctx->keyok.p, d_sig,
                     mode, (const u32 *)ctx->gtable, (u32 *)ctx->slots.p, idx, fin, keyok_row, d_ok, count);
  return LAMD_OK;
}

// ---- key-table cache plumbing
static int cache_alloc(lamd_ctx *owner, lamd_ctx::key_cache *kc, size_t keys7, size_t keys10, bool shared) {
  int rc;
  kc->shared = shared;
  const size_t ents = keys7 + keys10;
  if (ents > kc->cap_ent || keys7 > kc->cap7 || keys10 > kc->cap10) {
    if ((rc = ensure(owner, &kc->ents, ents * sizeof(cache_ent))) != LAMD_OK) return rc;
    if ((rc = ensure(owner, &kc->pool7, keys7 * kc_stride(7) * 4)) != LAMD_OK) return rc;
    if ((rc = ensure(owner, &kc->pool10, keys10 * kc_stride(10) * 4)) != LAMD_OK) return rc;
    if ((rc = ensure(owner, &kc->counters, C_WORDS * 4)) != LAMD_OK) return rc;
    kc->cap_ent = (u32)ents;
    kc->cap7 = (u32)keys7;
    kc->cap10 = (u32)keys10;
    if (shared) {
      size_t m = 1;
      while (m < 2 * ents) m <<= 1;
      if ((rc = ensure(owner, &kc->index, m * 4)) != LAMD_OK) return rc;
      kc->index_mask = (u32)(m - 1);
    }
  }
  return LAMD_OK;
}
// empties the shared cache (every lane must be idle): entries, index and allocation counters back to zero
static int cache_reset(lamd_ctx *root) {
  lamd_ctx::key_cache *kc = &root->cache_store;
  HIPCHK(root, hipDeviceSynchronize());
  HIPCHK(root, hipMemset(kc->ents.p, 0, (size_t)kc->cap_ent * sizeof(cache_ent)));
  HIPCHK(root, hipMemset(kc->index.p, 0, ((size_t)kc->index_mask + 1) * 4));
  HIPCHK(root, hipMemset(kc->counters.p, 0, C_WORDS * 4));
  for (int l = 0; l <= MAX_LANES; l++) { root->pub_seq[l] = root->vis_seq[l] = 0; root->pub_pending[l] = false; }
  root->call_seq = 0;
  root->cache_hwm[0] = root->cache_hwm[1] = root->cache_hwm[2] = 0;
  for (lamd_ctx *L : root->lane)
    if (L && L->h_plan) memset(L->h_plan, 0, (P_WORDS + C_WORDS) * 4);
  if (root->h_plan) memset(root->h_plan, 0, (P_WORDS + C_WORDS) * 4);
  root->cache_resets++;
  return LAMD_OK;
}
extern "C" int lamd_cache_clear(lamd_ctx *ctx) {
  if (!ctx) return LAMD_ERR_ARG;
  if (ctx->cache_mode == 0 || !ctx->cache_store.shared) return LAMD_OK;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  return cache_reset(ctx);
}

// One chunk (n <= ctx->chunk rows) entirely on the context's streams; nothing in here waits for the device.
// d_key: 33/65-byte SEC1 keys or 32-byte x-only.
//  1. scalar prep for every row on stream2 (independent of the key work; joined by event before the ecmult kernels)
//  2. rows whose key has a comb table in the cache go straight onto the row list of that comb shape
//  3. the other rows are de-duplicated on the device; keys carried by enough of them get a table now (7 teeth: 38 additions
//     + 18 doublings per signature instead of the ladder's 66 + 132; 10 teeth for heavily used keys: 26 + 12) and, in
//     cache mode, an entry for the calls after this one; the remaining ("cold") rows take the per-signature ladder
//  4. table-driven ecmult per comb shape, the ladder for the cold rows on a third stream
// Every count in between (distinct keys, new tables, rows per list) stays on the device (`plan`): launches cover upper
// bounds.  keyok_out (optional, n bytes): per-row key validity for the gossip reduce.
static int run_chunk(lamd_ctx *ctx, int mode, size_t n, const u8 *d_a, const u8 *d_sig, const u8 *d_key, int keylen,
                     size_t keystride, u8 *d_ok_caller, u8 *keyok_out, bool time_it) {
  int rc;
  lamd_ctx *root = ctx->root ? ctx->root : ctx;
  if ((rc = ensure(ctx, &ctx->recs, n * sizeof(prep_rec))) != LAMD_OK) return rc;
  // The kernels of a call talk to each other through the verdict bytes (VERDICT_SUSPECT, SCHNORR_PENDING): that state lives in
  // a workspace of the lane, and the caller's buffer receives the FINAL verdicts in one copy at the end of the call.  (It used to
  // be the caller's buffer itself: two calls in flight on different lanes that were handed the same verdict buffer -- bench.py's
  // steps -- then saw each other's markers, and the shared inversion of the BIP-340 parity stage, which walks its rows twice,
  // rejected valid signatures whenever the set of pending rows changed between the passes.)
  if ((rc = ensure(ctx, &ctx->vbuf, n)) != LAMD_OK) return rc;
  u8 *d_ok = (u8 *)ctx->vbuf.p;
  prep_rec *recs = (prep_rec *)ctx->recs.p;
  if (time_it) HIPCHK(ctx, hipEventRecord(ctx->ev[0], ctx->stream));
  // stream2 must not start before the inputs (possibly still being copied on the main stream) are there
  HIPCHK(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));
  HIPCHK(ctx, hipStreamWaitEvent(ctx->stream2, ctx->ev_fork, 0));
  {
    hipStream_t main = ctx->stream;
    ctx->stream = ctx->stream2;
    launch_prep(ctx, mode, n, d_a, d_sig, d_key, recs);
    ctx->stream = main;
  }
  HIPCHK(ctx, hipEventRecord(ctx->ev_prep, ctx->stream2));

  ctx->last_mode = mode;
  ctx->last_n = n;
  const bool use_cache = root->cache_mode != 0 && root->cache_store.shared;
  // small batches are latency-bound: without a cache to consult they go straight to the ladder; with one, they are looked
  // up, and only keys that fill most of the batch (a commitment's HTLC key) are worth a table of their own
  const bool small = n < ctx->keyed_min_rows && ctx->keyed_mode <= 0;
  const bool keyed = ctx->keyed_mode != 0 && n < 0x7FFFFFFFu && (use_cache || !small);
  ctx->last_keyed_call = keyed;
  if (!keyed) {
    if (time_it) HIPCHK(ctx, hipEventRecord(ctx->ev[1], ctx->stream));
    rc = launch_direct(ctx, mode, n, nullptr, nullptr, recs, d_sig, d_key, keylen, keystride, nullptr, keyok_out, d_ok, time_it);
    if (rc != LAMD_OK) return rc;
    if (time_it) HIPCHK(ctx, hipEventRecord(ctx->ev[3], ctx->stream));
    if (mode == MODE_SCHNORR)
      hipLaunchKernelGGL(k_schnorr_final, dim3(blocks_for(final_threads(ctx, n))), dim3(256), 0, ctx->stream, n, (u32 *)ctx->slots.p, d_ok);
    HIPCHK(ctx, hipMemcpyAsync(d_ok_caller, d_ok, n, hipMemcpyDeviceToDevice, ctx->stream));
    if (time_it) {
      HIPCHK(ctx, hipEventRecord(ctx->ev[4], ctx->stream));
      ctx->ev_recorded = true;
    }
    HIPCHK(ctx, hipGetLastError());
    return LAMD_OK;
  }

  // small batch with a cache: lookup straight into the row lists (k_small_lookup), cached combs + ladder -- unless the previous
  // small batch on this lane reported a dense run of rows under a key the cache does not know: then this one takes the
  // table-building path below and publishes it
  bool latency_learn = false;  // this call builds the tables of keys the LATENCY path (k_small_verify) met for the second time
  if (small && use_cache && ctx->small_fused) {
    const u32 dense_thr = ctx->last_small_n / 8 > 32 ? (u32)(ctx->last_small_n / 8) : 32u;
    const bool learn = ctx->force_learn || (ctx->last_small_fused && ctx->h_plan && ((volatile const u32 *)ctx->h_plan)[P_DENSE] >= dense_thr);
    latency_learn = ctx->force_learn;
    ctx->force_learn = false;  // (whoever set it has already forgotten the fingerprints of this call's keys: they have tables after it)
    ctx->last_small_fused = !learn;
    ctx->last_small_n = n;
    if (!learn) {
      lamd_ctx::key_cache *kcs = &root->cache_store;
      for (devbuf *b : {&ctx->row_ent, &ctx->list7, &ctx->list10, &ctx->listcold})
        if ((rc = ensure(ctx, b, n * 4)) != LAMD_OK) return rc;
      if ((rc = ensure(ctx, &ctx->plan, P_WORDS * 4)) != LAMD_OK) return rc;
      if (mode == MODE_SCHNORR && (rc = ensure(ctx, &ctx->kt_fin, n * (size_t)FIN_WORDS * 4)) != LAMD_OK) return rc;
      u32 *plan_s = (u32 *)ctx->plan.p, *fin_s = mode == MODE_SCHNORR ? (u32 *)ctx->kt_fin.p : nullptr;
      HIPCHK(ctx, hipMemsetAsync(plan_s, 0, P_WORDS * 4, ctx->stream));
      for (int l = 0; l <= MAX_LANES; l++)
        if (root->pub_pending[l] && hipEventQuery(root->ev_pub[l]) == hipSuccess) {
          root->vis_seq[l] = root->pub_seq[l];
          root->pub_pending[l] = false;
        }
      (void)hipGetLastError();
      cache_vis vis;
      for (int l = 0; l <= MAX_LANES; l++) vis.seq[l] = root->vis_seq[l];
      vis.seq[ctx->lane_id] = root->pub_seq[ctx->lane_id];
      hipLaunchKernelGGL(k_small_lookup, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, ctx->stream, n, d_key, keylen, keystride, root->hash_seed,
                         (const u32 *)kcs->index.p, kcs->index_mask, (const cache_ent *)kcs->ents.p, vis, (u32 *)ctx->row_ent.p, plan_s, (u32 *)ctx->list7.p,
                         (u32 *)ctx->list10.p, (u32 *)ctx->listcold.p, keyok_out, d_ok);
      if (time_it) HIPCHK(ctx, hipEventRecord(ctx->ev[1], ctx->stream));
      // the ladder for the misses (parses their keys first) on the third stream, the cached combs on this one: one uncached row
      // (a commitment's funding-key signature) costs a whole ladder latency and must not sit in front of the other 483
      HIPCHK(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));
      HIPCHK(ctx, hipStreamWaitEvent(ctx->stream3, ctx->ev_fork, 0));
      {
        hipStream_t main = ctx->stream;
        ctx->stream = ctx->stream3;
        rc = launch_direct(ctx, mode, n, (const u32 *)ctx->listcold.p, (const u32 *)(plan_s + P_COLD), recs, d_sig, d_key, keylen, keystride, fin_s, keyok_out,
                           d_ok, false, plan_s);
        ctx->stream = main;
        if (rc != LAMD_OK) return rc;
        HIPCHK(ctx, hipEventRecord(ctx->ev_cold, ctx->stream3));
      }
      if (time_it) HIPCHK(ctx, hipEventRecord(ctx->ev[2], ctx->stream));
      HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_prep, 0));
      hipLaunchKernelGGL((k_ecmult_keyed<false, 3>), dim3(keyed_grid(ctx, n)), dim3(LAMD_KEYED_THREADS), 0, ctx->stream, plan_s, (const u32 *)ctx->list7.p,
                         (const u32 *)ctx->list10.p, recs, (const u32 *)ctx->row_ent.p, (const cache_ent *)kcs->ents.p, (const u32 *)kcs->pool7.p,
                         (const u32 *)kcs->pool10.p, d_sig, mode, (const u32 *)ctx->gtable, fin_s, keyok_out, d_ok);
      hipLaunchKernelGGL((k_ecmult_keyed<true, 3>), dim3(careful_grid(ctx, n)), dim3(LAMD_KEYED_THREADS), 0, ctx->stream, plan_s, (const u32 *)ctx->list7.p,
                         (const u32 *)ctx->list10.p, recs, (const u32 *)ctx->row_ent.p, (const cache_ent *)kcs->ents.p, (const u32 *)kcs->pool7.p,
                         (const u32 *)kcs->pool10.p, d_sig, mode, (const u32 *)ctx->gtable, fin_s, keyok_out, d_ok);
      HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_cold, 0));
      if (time_it) HIPCHK(ctx, hipEventRecord(ctx->ev[3], ctx->stream));
      if (mode == MODE_SCHNORR)
        hipLaunchKernelGGL(k_schnorr_final_fin, dim3(blocks_for(final_threads(ctx, n))), dim3(256), 0, ctx->stream, n, fin_s, d_ok);
      HIPCHK(ctx, hipMemcpyAsync(d_ok_caller, d_ok, n, hipMemcpyDeviceToDevice, ctx->stream));
      HIPCHK(ctx, hipMemcpyAsync(ctx->h_plan, plan_s, P_WORDS * 4, hipMemcpyDeviceToHost, ctx->stream));
      if (time_it) {
        HIPCHK(ctx, hipEventRecord(ctx->ev[4], ctx->stream));
        ctx->ev_recorded = true;
      }
      HIPCHK(ctx, hipGetLastError());
      return LAMD_OK;
    }
  }

  // thresholds: rows per key that pay for a 7-tooth / a 10-tooth comb
  u32 thr7 = ctx->keyed_mode > 0 ? 2u : (u32)(ctx->keyed_min_uses + 0.5), thr10 = (u32)(ctx->keyed_dense_uses + 0.5);
  if (ctx->keyed_teeth == 7) thr10 = 0xFFFFFFFFu;
  if (ctx->keyed_teeth == 10) { thr10 = thr7; thr7 = 0xFFFFFFFFu; }
  if (thr7 < 2) thr7 = 2;
  if (thr10 < 2) thr10 = 2;
  if (small) {
    // without a cache: only a key that carries a large share of the batch is worth a table.  With one we are here because the
    // previous small batch saw a dense run under an unknown key (a channel's htlc key): that key gets the 10-tooth comb and every
    // other key the batch misses -- the funding key, one signature per commitment but the same one every time -- a 7-tooth comb
    thr7 = use_cache && ctx->small_fused && ctx->keyed_teeth != 10 ? 1u : 0xFFFFFFFFu;
    if (ctx->keyed_teeth != 7) thr10 = (u32)(ctx->keyed_dense_uses + 0.5);
    // keys learnt for the latency path take the 10-tooth comb while its pool has room (k_dedupe_classify falls back to 7 teeth when it
    // is full): such a key is about to be met one signature per call, where the comb's dependent chain IS the latency -- 13 columns
    // instead of 19 (one cached-key call 0.18 -> 0.15 ms); the 48 KB per key are what the 2^16-slot pool is for
    if (latency_learn && ctx->keyed_teeth != 7) thr10 = 1u;
  }
  const size_t hk7_cap = thr7 == 0xFFFFFFFFu ? 1 : n / thr7 + 1, hk10_cap = thr10 == 0xFFFFFFFFu ? 1 : n / thr10 + 1;
  lamd_ctx::key_cache *kc = use_cache ? &root->cache_store : &ctx->cache_store;
  if (!use_cache && (rc = cache_alloc(ctx, kc, hk7_cap, hk10_cap, false)) != LAMD_OK) return rc;
  size_t m = 4;  // (k_call_init clears the table four words at a time)
  while (m < 2 * n) m <<= 1;
  if ((rc = ensure(ctx, &ctx->kd_table, m * 4)) != LAMD_OK) return rc;
  for (devbuf *b : {&ctx->row_ent, &ctx->kd_rep, &ctx->kd_uid, &ctx->kd_uniq, &ctx->kd_count, &ctx->kd_newent, &ctx->list7, &ctx->list10,
                    &ctx->listcold})
    if ((rc = ensure(ctx, b, n * 4)) != LAMD_OK) return rc;
  for (devbuf *b : {&ctx->hk7_row, &ctx->hk7_ent, &ctx->hk7_slot})
    if ((rc = ensure(ctx, b, hk7_cap * 4)) != LAMD_OK) return rc;
  for (devbuf *b : {&ctx->hk10_row, &ctx->hk10_ent, &ctx->hk10_slot})
    if ((rc = ensure(ctx, b, hk10_cap * 4)) != LAMD_OK) return rc;
  if ((rc = ensure(ctx, &ctx->hk7_qwords, hk7_cap * 64)) != LAMD_OK) return rc;
  if ((rc = ensure(ctx, &ctx->hk7_keyok, hk7_cap)) != LAMD_OK) return rc;
  if ((rc = ensure(ctx, &ctx->hk7_scratch, hk7_cap * kc_scratch_words(7) * 4)) != LAMD_OK) return rc;
  if ((rc = ensure(ctx, &ctx->hk10_qwords, hk10_cap * 64)) != LAMD_OK) return rc;
  if ((rc = ensure(ctx, &ctx->hk10_keyok, hk10_cap)) != LAMD_OK) return rc;
  if ((rc = ensure(ctx, &ctx->hk10_scratch, hk10_cap * kc_scratch_words(10) * 4)) != LAMD_OK) return rc;
  if ((rc = ensure(ctx, &ctx->plan, P_WORDS * 4)) != LAMD_OK) return rc;
  if (mode == MODE_SCHNORR && (rc = ensure(ctx, &ctx->kt_fin, n * (size_t)FIN_WORDS * 4)) != LAMD_OK) return rc;
  u32 *plan = (u32 *)ctx->plan.p, *cc = (u32 *)kc->counters.p;
  u32 *row_ent = (u32 *)ctx->row_ent.p, *list7 = (u32 *)ctx->list7.p, *list10 = (u32 *)ctx->list10.p, *listcold = (u32 *)ctx->listcold.p;
  u32 *fin = mode == MODE_SCHNORR ? (u32 *)ctx->kt_fin.p : nullptr;
  const cache_ent *ents = (const cache_ent *)kc->ents.p;

  u32 seq = 0;
  cache_vis vis = {};
  if (use_cache) {
    // what this call may use: entries published by calls the host has seen complete, or earlier on this lane's stream
    for (int l = 0; l <= MAX_LANES; l++)
      if (root->pub_pending[l] && hipEventQuery(root->ev_pub[l]) == hipSuccess) {
        root->vis_seq[l] = root->pub_seq[l];
        root->pub_pending[l] = false;
      }
    (void)hipGetLastError();
    for (int l = 0; l <= MAX_LANES; l++) vis.seq[l] = root->vis_seq[l];
    vis.seq[ctx->lane_id] = root->pub_seq[ctx->lane_id];
    seq = ++root->call_seq;
  }
  const cache_caps caps = {kc->cap_ent, kc->cap7, kc->cap10};
  learn_filter lf = {nullptr, 0, d_key, keylen, keystride, root->hash_seed, 0xFFFFFFFFu, 0xFFFFFFFFu};
  if (small && use_cache) {  // tables learnt from small calls stay inside half of each pool
    lf.budget7 = kc->cap7 / 2;
    lf.budget10 = kc->cap10 / 2;
    if (latency_learn && !ctx->learn_fps.empty()) {
      const size_t nf = ctx->learn_fps.size() < 4096 ? ctx->learn_fps.size() : 4096;
      if ((rc = ensure(ctx, &ctx->learn_dev, nf * 8)) != LAMD_OK) return rc;
      HIPCHK(ctx, hipMemcpyAsync(ctx->learn_dev.p, ctx->learn_fps.data(), nf * 8, hipMemcpyHostToDevice, ctx->stream));  // pageable source: staged before the call returns
      lf.fps = (const u64 *)ctx->learn_dev.p;
      lf.n = (u32)nf;
    }
  }
  bool cold_started = false;
  if (ctx->fused_front) {
    // six launches (see k_call_init)
    {
      const size_t words = m > n ? m : n;
      const unsigned ib = blocks_for(words / 4 + 1);
      hipLaunchKernelGGL(k_call_init, dim3(ib < 2048u ? ib : 2048u), dim3(256), 0, ctx->stream, plan, (u32 *)ctx->kd_table.p, m, (u32 *)ctx->kd_count.p, n,
                         use_cache ? (u32 *)nullptr : row_ent, use_cache ? (u32 *)nullptr : cc);
    }
    if (use_cache) {
      if (ctx->sigs_pending) { HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_sigs_wait, 0)); ctx->sigs_pending = false; }
      hipLaunchKernelGGL(k_cache_lookup, dim3(blocks_for(n)), dim3(256), 0, ctx->stream, n, d_key, keylen, keystride, root->hash_seed,
                         (const u32 *)kc->index.p, kc->index_mask, ents, vis, row_ent, plan, list7, list10, keyok_out, d_ok, d_sig, mode);
    }
    hipLaunchKernelGGL(k_dedupe_insert_count, dim3(blocks_for(n)), dim3(256), 0, ctx->stream, n, d_key, keylen, keystride, root->hash_seed,
                       (const u32 *)row_ent, (u32 *)ctx->kd_table.p, (u32)(m - 1), (u32 *)ctx->kd_rep.p, plan, (u32 *)ctx->kd_uniq.p, (u32 *)ctx->kd_count.p);
    hipLaunchKernelGGL((k_dedupe_classify<true>), dim3(blocks_for(n)), dim3(256), 0, ctx->stream, n, (const u32 *)ctx->kd_count.p,
                       (const u32 *)ctx->kd_uniq.p, thr7, thr10, plan, cc, caps, (u32)hk7_cap, (u32)hk10_cap, (u32 *)ctx->kd_newent.p,
                       (u32 *)ctx->hk7_row.p, (u32 *)ctx->hk7_ent.p, (u32 *)ctx->hk7_slot.p, (u32 *)ctx->hk10_row.p, (u32 *)ctx->hk10_ent.p,
                       (u32 *)ctx->hk10_slot.p, lf);
    if (time_it) HIPCHK(ctx, hipEventRecord(ctx->ev[1], ctx->stream));
    if (ctx->early_cold) {
      // the cold rows' list, and their ladder on the side stream, before the tables are built (k_partition_cold)
      hipLaunchKernelGGL(k_partition_cold, dim3(blocks_for(n)), dim3(256), 0, ctx->stream, n, (const u32 *)ctx->kd_rep.p, (const u32 *)ctx->kd_newent.p, plan, listcold);
      HIPCHK(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));
      HIPCHK(ctx, hipStreamWaitEvent(ctx->stream3, ctx->ev_fork, 0));
      hipStream_t main = ctx->stream;
      ctx->stream = ctx->stream3;
      rc = launch_direct(ctx, mode, n, (const u32 *)listcold, (const u32 *)(plan + P_COLD), recs, d_sig, d_key, keylen, keystride, fin, keyok_out, d_ok, false, plan);
      ctx->stream = main;
      if (rc != LAMD_OK) return rc;
      HIPCHK(ctx, hipEventRecord(ctx->ev_cold, ctx->stream3));
      cold_started = true;
    }
    const bool on7 = thr7 != 0xFFFFFFFFu, on10 = thr10 != 0xFFFFFFFFu;
    if (on7 || on10) {
      const kc_shape_args A7 = {(u32)(on7 ? hk7_cap : 0), (const u32 *)ctx->hk7_row.p, (const u32 *)ctx->hk7_ent.p, (const u32 *)ctx->hk7_slot.p,
                                (u32 *)ctx->hk7_qwords.p, (u8 *)ctx->hk7_keyok.p, (u32 *)ctx->hk7_scratch.p, (u32 *)kc->pool7.p};
      const kc_shape_args A10 = {(u32)(on10 ? hk10_cap : 0), (const u32 *)ctx->hk10_row.p, (const u32 *)ctx->hk10_ent.p, (const u32 *)ctx->hk10_slot.p,
                                 (u32 *)ctx->hk10_qwords.p, (u8 *)ctx->hk10_keyok.p, (u32 *)ctx->hk10_scratch.p, (u32 *)kc->pool10.p};
      const unsigned kb7 = on7 ? blocks_for(hk7_cap) : 0u, kb10 = on10 ? blocks_for(hk10_cap) : 0u;
      hipLaunchKernelGGL(k_keys_bases_both, dim3(kb7 + kb10), dim3(256), 0, ctx->stream, (const u32 *)plan, kb7, A7, A10, d_key, keylen, keystride);
      const unsigned fb7 = on7 ? blocks_for(hk7_cap * kc_nsub(7)) : 0u, fb10 = on10 ? blocks_for(hk10_cap * kc_nsub(10)) : 0u;
      const publish_args P = {d_key, keylen, keystride, (u32)ctx->lane_id, seq ? seq : 1u, root->hash_seed, (cache_ent *)kc->ents.p, (u32 *)kc->index.p,
                              kc->index_mask, (int)use_cache};
      if (root->kc_tree) {
        const unsigned tb7 = on7 ? (unsigned)((hk7_cap + 64 * KC_TREE_KPL7 - 1) / (64 * KC_TREE_KPL7)) : 0u;
        const unsigned tb10 = on10 ? (unsigned)((hk10_cap + 64 * KC_TREE_KPL10 - 1) / (64 * KC_TREE_KPL10)) : 0u;
        hipLaunchKernelGGL(k_kc_tree_both, dim3(tb7 + tb10), dim3(64), 0, ctx->stream, (const u32 *)plan, tb7, A7, A10, P);
      } else {
        hipLaunchKernelGGL(k_kc_finish_both, dim3(fb7 + fb10), dim3(256), 0, ctx->stream, (const u32 *)plan, fb7, A7, A10, P);
      }
    }
    if (use_cache) {
      HIPCHK(ctx, hipEventRecord(root->ev_pub[ctx->lane_id], ctx->stream));
      root->pub_seq[ctx->lane_id] = seq;
      root->pub_pending[ctx->lane_id] = true;
    }
    if (ctx->sigs_pending) { HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_sigs_wait, 0)); ctx->sigs_pending = false; }
    auto part = cold_started ? k_partition<true, true> : k_partition<true, false>;
    hipLaunchKernelGGL(part, dim3(blocks_for(n)), dim3(256), 0, ctx->stream, n, row_ent, (const u32 *)ctx->kd_rep.p, (const u32 *)nullptr,
                       (const u32 *)ctx->kd_newent.p, ents, plan, list7, list10, listcold, keyok_out, d_ok, d_sig, mode);
  } else {
    HIPCHK(ctx, hipMemsetAsync(plan, 0, P_WORDS * 4, ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(ctx->kd_table.p, 0, m * 4, ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(ctx->kd_count.p, 0, n * 4, ctx->stream));
    if (use_cache) {
      if (ctx->sigs_pending) { HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_sigs_wait, 0)); ctx->sigs_pending = false; }
      hipLaunchKernelGGL(k_cache_lookup, dim3(blocks_for(n)), dim3(256), 0, ctx->stream, n, d_key, keylen, keystride, root->hash_seed,
                         (const u32 *)kc->index.p, kc->index_mask, ents, vis, row_ent, plan, list7, list10, keyok_out, d_ok, d_sig, mode);
    } else {
      HIPCHK(ctx, hipMemsetAsync(row_ent, 0xFF, n * 4, ctx->stream));
      HIPCHK(ctx, hipMemsetAsync(cc, 0, C_WORDS * 4, ctx->stream));
    }
    hipLaunchKernelGGL(k_dedupe_insert, dim3(blocks_for(n)), dim3(256), 0, ctx->stream, n, d_key, keylen, keystride, root->hash_seed,
                       (const u32 *)row_ent, (u32 *)ctx->kd_table.p, (u32)(m - 1), (u32 *)ctx->kd_rep.p);
    hipLaunchKernelGGL(k_dedupe_number, dim3(blocks_for(n)), dim3(256), 0, ctx->stream, n, (const u32 *)ctx->kd_rep.p, (u32 *)ctx->kd_uid.p,
                       plan, (u32 *)ctx->kd_uniq.p);
    hipLaunchKernelGGL(k_dedupe_map, dim3(blocks_for(n)), dim3(256), 0, ctx->stream, n, (const u32 *)ctx->kd_rep.p, (const u32 *)ctx->kd_uid.p,
                       (u32 *)ctx->kd_count.p);
    hipLaunchKernelGGL((k_dedupe_classify<false>), dim3(blocks_for(n)), dim3(256), 0, ctx->stream, n, (const u32 *)ctx->kd_count.p,
                       (const u32 *)ctx->kd_uniq.p, thr7, thr10, plan, cc, caps, (u32)hk7_cap, (u32)hk10_cap, (u32 *)ctx->kd_newent.p,
                       (u32 *)ctx->hk7_row.p, (u32 *)ctx->hk7_ent.p, (u32 *)ctx->hk7_slot.p, (u32 *)ctx->hk10_row.p, (u32 *)ctx->hk10_ent.p,
                       (u32 *)ctx->hk10_slot.p, lf);
    if (time_it) HIPCHK(ctx, hipEventRecord(ctx->ev[1], ctx->stream));
    // the new keys: parse, build their tables, publish
    if (thr7 != 0xFFFFFFFFu) {
      hipLaunchKernelGGL(k_keys, dim3(blocks_for(hk7_cap)), dim3(256), 0, ctx->stream, hk7_cap, d_key, keylen, keystride, (const u32 *)ctx->hk7_row.p,
                         (u32 *)ctx->hk7_qwords.p, (u8 *)ctx->hk7_keyok.p, (const u32 *)(plan + P_HK7));
      launch_keytables<7>(ctx->stream, plan, P_HK7, hk7_cap, (const u32 *)ctx->hk7_qwords.p, (const u8 *)ctx->hk7_keyok.p, (u32 *)kc->pool7.p,
                          (const u32 *)ctx->hk7_slot.p, (u32 *)ctx->hk7_scratch.p);
      hipLaunchKernelGGL(k_cache_publish, dim3(blocks_for(hk7_cap)), dim3(256), 0, ctx->stream, (const u32 *)plan, (int)P_HK7, (u32)hk7_cap,
                         (const u32 *)ctx->hk7_row.p, (const u32 *)ctx->hk7_ent.p, (const u32 *)ctx->hk7_slot.p, (const u8 *)ctx->hk7_keyok.p, d_key,
                         keylen, keystride, 7u, (u32)ctx->lane_id, seq ? seq : 1u, root->hash_seed, (cache_ent *)kc->ents.p, (u32 *)kc->index.p,
                         kc->index_mask, (int)use_cache);
    }
    if (thr10 != 0xFFFFFFFFu) {
      hipLaunchKernelGGL(k_keys, dim3(blocks_for(hk10_cap)), dim3(256), 0, ctx->stream, hk10_cap, d_key, keylen, keystride, (const u32 *)ctx->hk10_row.p,
                         (u32 *)ctx->hk10_qwords.p, (u8 *)ctx->hk10_keyok.p, (const u32 *)(plan + P_HK10));
      launch_keytables<10>(ctx->stream, plan, P_HK10, hk10_cap, (const u32 *)ctx->hk10_qwords.p, (const u8 *)ctx->hk10_keyok.p, (u32 *)kc->pool10.p,
                           (const u32 *)ctx->hk10_slot.p, (u32 *)ctx->hk10_scratch.p);
      hipLaunchKernelGGL(k_cache_publish, dim3(blocks_for(hk10_cap)), dim3(256), 0, ctx->stream, (const u32 *)plan, (int)P_HK10, (u32)hk10_cap,
                         (const u32 *)ctx->hk10_row.p, (const u32 *)ctx->hk10_ent.p, (const u32 *)ctx->hk10_slot.p, (const u8 *)ctx->hk10_keyok.p, d_key,
                         keylen, keystride, 10u, (u32)ctx->lane_id, seq ? seq : 1u, root->hash_seed, (cache_ent *)kc->ents.p, (u32 *)kc->index.p,
                         kc->index_mask, (int)use_cache);
    }
    if (use_cache) {
      HIPCHK(ctx, hipEventRecord(root->ev_pub[ctx->lane_id], ctx->stream));
      root->pub_seq[ctx->lane_id] = seq;
      root->pub_pending[ctx->lane_id] = true;
    }
    if (ctx->sigs_pending) { HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_sigs_wait, 0)); ctx->sigs_pending = false; }
    hipLaunchKernelGGL((k_partition<false>), dim3(blocks_for(n)), dim3(256), 0, ctx->stream, n, row_ent, (const u32 *)ctx->kd_rep.p, (const u32 *)ctx->kd_uid.p,
                       (const u32 *)ctx->kd_newent.p, ents, plan, list7, list10, listcold, keyok_out, d_ok, d_sig, mode);
  }
  // cold rows (keys seen too rarely for a table) take the per-signature ladder on a third stream: usually few rows, i.e.
  // a latency-bound launch that should hide behind the table-driven kernels instead of serialising with them
  if (!cold_started) {
    HIPCHK(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));  // row lists are complete
    HIPCHK(ctx, hipStreamWaitEvent(ctx->stream3, ctx->ev_fork, 0));
    hipStream_t main = ctx->stream;
    ctx->stream = ctx->stream3;
    rc = launch_direct(ctx, mode, n, (const u32 *)listcold, (const u32 *)(plan + P_COLD), recs, d_sig, d_key, keylen, keystride, fin, keyok_out,
                       d_ok, false, plan);
    ctx->stream = main;
    if (rc != LAMD_OK) return rc;
    HIPCHK(ctx, hipEventRecord(ctx->ev_cold, ctx->stream3));
  }
  if (ctx->group_rows && n >= 4096) {
    // rows of one key next to each other (k_group_*): from here on list7 / list10 are the grouped lists
    const size_t ent_words = kc->cap_ent;
    if (ctx->g_cnt.cap < ent_words * 4) {  // the histogram must start out zeroed (k_group_alloc leaves it so)
      if ((rc = ensure(ctx, &ctx->g_cnt, ent_words * 4)) != LAMD_OK) return rc;
      HIPCHK(ctx, hipMemsetAsync(ctx->g_cnt.p, 0, ctx->g_cnt.cap, ctx->stream));
    }
    if ((rc = ensure(ctx, &ctx->g_base, ent_words * 4)) != LAMD_OK) return rc;
    for (devbuf *b : {&ctx->g_rank, &ctx->g_touched, &ctx->g_list7, &ctx->g_list10})
      if ((rc = ensure(ctx, b, n * 4)) != LAMD_OK) return rc;
    hipLaunchKernelGGL(k_group_count, dim3(blocks_for(n)), dim3(256), 0, ctx->stream, plan, (const u32 *)list7, (const u32 *)list10, (const u32 *)row_ent,
                       (u32 *)ctx->g_cnt.p, (u32 *)ctx->g_rank.p, (u32 *)ctx->g_touched.p);
    hipLaunchKernelGGL(k_group_alloc, dim3(blocks_for(n)), dim3(256), 0, ctx->stream, plan, (const u32 *)ctx->g_touched.p, ents, (u32 *)ctx->g_cnt.p,
                       (u32 *)ctx->g_base.p);
    hipLaunchKernelGGL(k_group_scatter, dim3(blocks_for(n)), dim3(256), 0, ctx->stream, (const u32 *)plan, (const u32 *)list7, (const u32 *)list10,
                       (const u32 *)row_ent, (const u32 *)ctx->g_base.p, (const u32 *)ctx->g_rank.p, (u32 *)ctx->g_list7.p, (u32 *)ctx->g_list10.p);
    list7 = (u32 *)ctx->g_list7.p;
    list10 = (u32 *)ctx->g_list10.p;
  }
  if (time_it) HIPCHK(ctx, hipEventRecord(ctx->ev[2], ctx->stream));
  HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_prep, 0));
  // (the bare-formula kernel fits 4 waves per SIMD with a spill, or 3 without: LAMD_KEYED_WAVES)
  auto fast = ctx->keyed_waves == 3 ? k_ecmult_keyed<false, 3> : k_ecmult_keyed<false, 4>;
  // LAMD_ECMULT_CHAIN=1: a large table-driven ecmult launch waits for the one submitted before it (on another lane).  The kernel
  // saturates the VALU issue port by itself (0.235 of 0.25 wave-instructions per SIMD and cycle), so two of them in flight only
  // stretch each other; what gains from running under it is the other lanes' latency-bound front end.
  const bool chain = root->ecm_chain != 0 && n >= root->ecm_chain_min;
  if (chain && root->ecm_last >= 0 && root->ecm_last != ctx->lane_id) HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, root->ev_ecm[root->ecm_last], 0));
  const bool time_kernel = time_it && ctx->kev_n < lamd_ctx::KEV;
  if (time_kernel) {
    hipEvent_t *pair = ctx->kev[ctx->kev_n];
    if (!pair[0]) {
      HIPCHK(ctx, hipEventCreate(&pair[0]));
      HIPCHK(ctx, hipEventCreate(&pair[1]));
    }
    HIPCHK(ctx, hipEventRecord(pair[0], ctx->stream));
  }
  const bool two_parts = chain && root->ecm_chain == 2 && ctx->stream_lo;
  if (two_parts) {
    // part 2 (the last ecm_tail work items) on the lane's lowest-priority stream, runnable from now on: the dispatcher gives it the wave slots
    // part 1 leaves free -- at part 1's tail, and under the head of the NEXT call's part 1, which waits for THIS part 1 only
    HIPCHK(ctx, hipEventRecord(ctx->ev_lo_go, ctx->stream));
    HIPCHK(ctx, hipStreamWaitEvent(ctx->stream_lo, ctx->ev_lo_go, 0));
    hipLaunchKernelGGL(fast, dim3((root->ecm_tail + LAMD_KEYED_THREADS - 1) / LAMD_KEYED_THREADS), dim3(LAMD_KEYED_THREADS), ctx->keyed_lds_pad, ctx->stream_lo, plan,
                       (const u32 *)list7, (const u32 *)list10, recs, (const u32 *)row_ent, ents, (const u32 *)kc->pool7.p, (const u32 *)kc->pool10.p, d_sig, mode,
                       (const u32 *)ctx->gtable, fin, keyok_out, d_ok, (const u32 *)ctx->gtable5, 2, root->ecm_tail);
    HIPCHK(ctx, hipEventRecord(ctx->ev_lo_done, ctx->stream_lo));
  }
  const bool bulk = ctx->bulk_stream && ctx->stream_lo && !two_parts && n >= root->ecm_chain_min;
  hipStream_t ks = ctx->stream;
  if (bulk) {  // the whole launch on the lowest-priority stream: the hardware dispatches the other lanes' (normal-priority) front-end blocks first
    HIPCHK(ctx, hipEventRecord(ctx->ev_lo_go, ctx->stream));
    HIPCHK(ctx, hipStreamWaitEvent(ctx->stream_lo, ctx->ev_lo_go, 0));
    ks = ctx->stream_lo;
  }
  if (ctx->pairs && ctx->keyed_waves == 3 && !two_parts && !ctx->gtable5) {
    // pairs first: a persistent grid (what the chip holds at three waves per SIMD), every lane its share of the rows in batches per inversion
    const unsigned resident = (unsigned)ctx->prop.multiProcessorCount * 3u * 4u, full = (unsigned)((n + PAIRS_THREADS - 1) / PAIRS_THREADS);
    const unsigned grid = full < resident ? full : resident;
    if ((rc = ensure(ctx, &ctx->pairs_ws, (size_t)PAIRS_SLOTS * PAIRS_WS_WORDS * 4 * grid * PAIRS_THREADS)) != LAMD_OK) return rc;
    hipLaunchKernelGGL((k_ecmult_keyed_pairs<3>), dim3(grid), dim3(PAIRS_THREADS), ctx->keyed_lds_pad, ks, plan, (const u32 *)list7, (const u32 *)list10, recs,
                       (const u32 *)row_ent, ents, (const u32 *)kc->pool7.p, (const u32 *)kc->pool10.p, d_sig, mode, (const u32 *)ctx->gtable, fin,
                       keyok_out, d_ok, (u32 *)ctx->pairs_ws.p);
  } else {
    hipLaunchKernelGGL(fast, dim3(keyed_grid(ctx, n)), dim3(LAMD_KEYED_THREADS), ctx->keyed_lds_pad, ks, plan, (const u32 *)list7, (const u32 *)list10, recs,
                       (const u32 *)row_ent, ents, (const u32 *)kc->pool7.p, (const u32 *)kc->pool10.p, d_sig, mode, (const u32 *)ctx->gtable, fin,
                       keyok_out, d_ok, (const u32 *)ctx->gtable5, two_parts ? 1 : 0, root->ecm_tail);
  }
  if (bulk) {
    HIPCHK(ctx, hipEventRecord(ctx->ev_lo_done, ctx->stream_lo));
    HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_lo_done, 0));
  }
  if (time_kernel) {
    HIPCHK(ctx, hipEventRecord(ctx->kev[ctx->kev_n][1], ctx->stream));
    ctx->kev_mode[ctx->kev_n++] = mode == MODE_SCHNORR ? 1 : 0;
  }
  if (chain) {
    HIPCHK(ctx, hipEventRecord(root->ev_ecm[ctx->lane_id], ctx->stream));
    root->ecm_last = ctx->lane_id;
  }
  if (two_parts) HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_lo_done, 0));
  // rows whose bare-formula ecmult met Z = 0 (crafted scalars, a result at infinity): the complete formulas decide
  hipLaunchKernelGGL((k_ecmult_keyed<true, 3>), dim3(careful_grid(ctx, n)), dim3(LAMD_KEYED_THREADS), 0, ctx->stream, plan, (const u32 *)list7, (const u32 *)list10, recs,
                     (const u32 *)row_ent, ents, (const u32 *)kc->pool7.p, (const u32 *)kc->pool10.p, d_sig, mode, (const u32 *)ctx->gtable, fin,
                     keyok_out, d_ok);
  HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_cold, 0));
  if (time_it) HIPCHK(ctx, hipEventRecord(ctx->ev[3], ctx->stream));
  if (mode == MODE_SCHNORR)
    hipLaunchKernelGGL(k_schnorr_final_fin, dim3(blocks_for(final_threads(ctx, n))), dim3(256), 0, ctx->stream, n, fin, d_ok);
  HIPCHK(ctx, hipMemcpyAsync(d_ok_caller, d_ok, n, hipMemcpyDeviceToDevice, ctx->stream));
  // statistics (and the cache's fill level) for whoever looks later: nothing waits for this copy
  HIPCHK(ctx, hipMemcpyAsync(ctx->h_plan, plan, P_WORDS * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(ctx->h_plan + P_WORDS, cc, C_WORDS * 4, hipMemcpyDeviceToHost, ctx->stream));
  if (time_it) {
    HIPCHK(ctx, hipEventRecord(ctx->ev[4], ctx->stream));
    ctx->ev_recorded = true;
  }
  HIPCHK(ctx, hipGetLastError());
  return LAMD_OK;
}

// the shared cache is bounded: when the last fill level read back nears a capacity, drain the device and start over
static int cache_maybe_reset(lamd_ctx *root) {
  if (root->cache_mode == 0 || !root->cache_store.shared) return LAMD_OK;
  const lamd_ctx::key_cache &kc = root->cache_store;
  u32 hw[3] = {0, 0, 0};
  for (int i = 0; i <= MAX_LANES; i++) {
    const lamd_ctx *L = i < MAX_LANES ? root->lane[i] : root;
    if (!L || !L->h_plan) continue;
    for (int k = 0; k < 3; k++) {
      const u32 v = ((volatile const u32 *)L->h_plan)[P_WORDS + k];
      if (v > hw[k]) hw[k] = v;
    }
  }
  for (int k = 0; k < 3; k++) root->cache_hwm[k] = hw[k];
  if ((double)hw[0] > 0.9 * kc.cap_ent || (double)hw[1] > 0.9 * kc.cap7 || (double)hw[2] > 0.9 * kc.cap10) return cache_reset(root);
  return LAMD_OK;
}

static int run_device(lamd_ctx *ctx, int mode, size_t n, const u8 *d_a, const u8 *d_sig, const u8 *d_key, int keylen,
                      size_t keystride, u8 *d_ok, u8 *keyok_out = nullptr) {
  HIPCHK(ctx, hipSetDevice(ctx->device));
  if (!ctx->root) ctx->last_lane = nullptr;  // a call on the context's own streams (host-buffer entry points): lamd_get_info() reports it
  bool forked = false;
  size_t k = 0;
  for (size_t o = 0; o < n; o += ctx->chunk, k++) {
    const size_t m = n - o < ctx->chunk ? n - o : ctx->chunk;
    lamd_ctx *W = (k & 1) && ctx->peer ? ctx->peer : ctx;  // odd chunks on the other lane: its front end overlaps this lane's ecmult
    W->timing = ctx->timing;
    if (W != ctx && !forked) {
      HIPCHK(ctx, hipEventRecord(ctx->ev_lane, ctx->stream));  // whatever precedes the call on this lane (staging, gossip expand)
      HIPCHK(ctx, hipStreamWaitEvent(W->stream, ctx->ev_lane, 0));
      forked = true;
    }
    const int rc = run_chunk(W, mode, m, d_a + 32 * o, d_sig + 64 * o, d_key + keystride * o, keylen, keystride, d_ok + o,
                             keyok_out ? keyok_out + o : nullptr, ctx->timing && o + ctx->chunk >= n);
    ctx->last_chunk_lane = W;
    if (rc != LAMD_OK) {
      if (W != ctx) ctx->err = W->err;
      return rc;
    }
  }
  if (forked) {  // what follows on this lane (result copies, gossip reduce) needs the other lane's chunks too
    HIPCHK(ctx, hipEventRecord(ctx->peer->ev_join, ctx->peer->stream));
    HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->peer->ev_join, 0));
  }
  return LAMD_OK;
}

extern "C" int lamd_verify_ecdsa_batch_device(lamd_ctx *ctx, size_t n, const void *d_hash32, const void *d_sig64,
                                              const void *d_pub, size_t publen, size_t pubstride, void *d_ok) {
  if (!ctx) return LAMD_ERR_ARG;
  if (n == 0) return LAMD_OK;
  if (!d_hash32 || !d_sig64 || !d_pub || !d_ok || (publen != 33 && publen != 65) || pubstride < publen) {
    ctx->err = "bad argument";
    return LAMD_ERR_ARG;
  }
  lamd_ctx *L;
  int rc = pick_lane(ctx, &L);
  if (rc != LAMD_OK) return rc;
  rc = run_device(L, MODE_ECDSA, n, (const u8 *)d_hash32, (const u8 *)d_sig64, (const u8 *)d_pub, (int)publen, pubstride, (u8 *)d_ok);
  if (rc != LAMD_OK && L != ctx) ctx->err = L->err;
  return rc;
}

extern "C" int lamd_verify_schnorr_batch_device(lamd_ctx *ctx, size_t n, const void *d_msg32, const void *d_xonly32,
                                                const void *d_sig64, void *d_ok) {
  if (!ctx) return LAMD_ERR_ARG;
  if (n == 0) return LAMD_OK;
  if (!d_msg32 || !d_xonly32 || !d_sig64 || !d_ok) {
    ctx->err = "bad argument";
    return LAMD_ERR_ARG;
  }
  lamd_ctx *L;
  int rc = pick_lane(ctx, &L);
  if (rc != LAMD_OK) return rc;
  rc = run_device(L, MODE_SCHNORR, n, (const u8 *)d_msg32, (const u8 *)d_sig64, (const u8 *)d_xonly32, 32, 32, (u8 *)d_ok);
  if (rc != LAMD_OK && L != ctx) ctx->err = L->err;
  return rc;
}

// n <= SMALL_MAX rows from host memory: one launch of k_small_verify, inputs and verdicts through pinned device-mapped memory, the
// host waits on the completion word the kernel's last instruction writes
constexpr size_t SMALL_MAX = 4096;  // rows of a call that takes the one-launch path (a grid of 64-row blocks)
constexpr size_t TXSIG_HOST_HASH_ROWS = 16;  // check_tx_sig batches up to here hash on the host (one launch); above, on the device (two launches)
constexpr size_t SMALL_OFF_SIG = SMALL_MAX * 32, SMALL_OFF_KEY = SMALL_OFF_SIG + SMALL_MAX * 64, SMALL_OFF_OUT = SMALL_OFF_KEY + SMALL_MAX * 65 + 64,
                 SMALL_OFF_SHAPES = SMALL_OFF_OUT + SMALL_MAX, SMALL_OFF_FLAG = SMALL_OFF_SHAPES + SMALL_MAX, SMALL_BYTES = SMALL_OFF_FLAG + 64;
// Keys that the latency path had to take down the ladder are remembered by fingerprint (host side, direct-mapped): the SECOND small
// call that brings such a key takes the table-building path once (every key the cache misses gets a comb and is published), and
// from then on the key is a cache hit -- a peer's node id or a channel's keys recur with every single check_signed_hash() call.
// The table is 4-way set-associative.  Rounds 3-4 had it direct-mapped with 4096 slots: two keys that share a slot evict each other's
// fingerprint on every pass and NEITHER is ever learnt -- 9 % of 400 recurring channel keys, the 0.8 ms tail of the one-commitment-per-flush
// latency (p50 0.14 ms); 65 536 direct-mapped slots still left one or two such pairs per run (the same channels slow in every pass).
constexpr size_t MISS_SLOTS = 65536, MISS_WAYS = 4;
static inline u64 *miss_bucket(std::vector<u64> &t, u64 fp) { return &t[((fp >> 1) % (MISS_SLOTS / MISS_WAYS)) * MISS_WAYS]; }
static inline bool miss_has(std::vector<u64> &t, u64 fp) {
  const u64 *b = miss_bucket(t, fp);
  return b[0] == fp || b[1] == fp || b[2] == fp || b[3] == fp;
}
static inline void miss_put(std::vector<u64> &t, u64 fp) {
  u64 *b = miss_bucket(t, fp);
  for (size_t w = 0; w < MISS_WAYS; w++)
    if (b[w] == fp) return;
  for (size_t w = 0; w < MISS_WAYS; w++)
    if (b[w] == 0) { b[w] = fp; return; }
  b[(fp >> 40) % MISS_WAYS] = fp;  // a full bucket: five waiting keys in one of 16 384 buckets
}
static inline bool miss_take(std::vector<u64> &t, u64 fp) {  // present -> removed, true
  u64 *b = miss_bucket(t, fp);
  for (size_t w = 0; w < MISS_WAYS; w++)
    if (b[w] == fp) { b[w] = 0; return true; }
  return false;
}
__host__ __device__ static inline u64 small_fingerprint(u64 seed, const u8 *key, int keylen) {
  u64 h = seed ^ 0x6D697373ull;
  for (int o = 0; o < keylen; o += 8) {
    u64 c = 0;
    for (int b = 0; b < 8 && o + b < keylen; b++) c |= (u64)key[o + b] << (8 * b);   // (little-endian word, byte by byte: host and device agree)
    h = (h ^ c) * 0x9E3779B97F4A7C15ull;
    h ^= h >> 29;
  }
  return h | 1;
}
// the learning call builds tables for the keys of its batch whose fingerprint recurred: forget exactly those fingerprints (other callers' keys
// that are still waiting for their second sight keep theirs)
// ... and hand those fingerprints to the context that will run the learning call (`dst`: the root for a host-buffer call, the lane for a
// flush): its k_dedupe_classify builds tables for exactly these keys
static void small_forget(lamd_ctx *ctx, lamd_ctx *dst, const u8 *key, size_t keystride, int keylen, size_t n) {
  dst->learn_fps.clear();
  for (size_t i = 0; i < n; i++) {
    if (i && memcmp(key + i * keystride, key + (i - 1) * keystride, keylen) == 0) continue;
    const u64 fp = small_fingerprint(ctx->hash_seed, key + i * keystride, keylen);
    if (miss_take(ctx->small_missed, fp)) dst->learn_fps.push_back(fp);
  }
}
// d_a32 / d_gate (optional, device memory): the rows' hashes were produced on the device by a kernel queued in front on ctx->stream (`a` is not
// read then) / the gate that kernel decided (check_tx_sig's sighash-type rule)
static int run_small(lamd_ctx *ctx, int mode, size_t n, const u8 *a, const u8 *sig, const u8 *key, int keylen, size_t keystride, u8 *ok,
                     const u8 *d_a32 = nullptr, const u8 *d_gate = nullptr) {
  int rc;
  if (!ctx->h_small) {
    HIPCHK(ctx, hipHostMalloc((void **)&ctx->h_small, SMALL_BYTES, hipHostMallocMapped | hipHostMallocCoherent));
    memset(ctx->h_small, 0, SMALL_BYTES);
  }
  if ((rc = ensure(ctx, &ctx->slots, ((n + 63) & ~(size_t)63) * SLOT_WORDS * 4)) != LAMD_OK) return rc;
  if (!ctx->small_done.p) {
    if ((rc = ensure(ctx, &ctx->small_done, 64)) != LAMD_OK) return rc;
    HIPCHK(ctx, hipMemsetAsync(ctx->small_done.p, 0, 64, ctx->stream));
  }
  u8 *h = ctx->h_small;
  const bool have_cache = ctx->cache_mode != 0 && ctx->cache_store.shared;
  if (have_cache) {
    if (ctx->small_missed.empty()) ctx->small_missed.assign(MISS_SLOTS, 0);
    bool learn = false;
    for (size_t i = 0; i < n && !learn; i++) {
      if (i && memcmp(key + i * keystride, key + (i - 1) * keystride, keylen) == 0) continue;  // a commitment's rows share their key
      const u64 fp = small_fingerprint(ctx->hash_seed, key + i * keystride, keylen);
      learn = miss_has(ctx->small_missed, fp);  // seen before without a table: the caller takes the learning path
    }
    if (learn) {
      small_forget(ctx, ctx, key, keystride, keylen, n);
      return 1;
    }
  }
  if (!d_a32) memcpy(h, a, n * 32);
  memcpy(h + SMALL_OFF_SIG, sig, n * 64);
  if (keystride == (size_t)keylen) memcpy(h + SMALL_OFF_KEY, key, n * keylen);
  else
    for (size_t i = 0; i < n; i++) memcpy(h + SMALL_OFF_KEY + i * keylen, key + i * keystride, keylen);
  // every verdict / shape byte is written exactly once by the kernel (0 / 1; 0 / 7 / 10 / 255): 0xEE marks "not written yet", so that the
  // host can tell whether the bytes of EVERY block have arrived when it sees the completion word (written by the last block)
  memset(h + SMALL_OFF_OUT, 0xEE, n);
  memset(h + SMALL_OFF_SHAPES, 0xEE, n);
  small_args A;
  memset(&A, 0, sizeof A);
  A.a32 = d_a32 ? d_a32 : h; A.sig64 = h + SMALL_OFF_SIG; A.key = h + SMALL_OFF_KEY;
  A.gate = d_gate;
  A.keylen = keylen; A.mode = mode; A.n = (u32)n;
  A.seed = ctx->hash_seed;
  if (ctx->cache_mode != 0 && ctx->cache_store.shared) {
    lamd_ctx::key_cache *kc = &ctx->cache_store;
    for (int l = 0; l <= MAX_LANES; l++)
      if (ctx->pub_pending[l] && hipEventQuery(ctx->ev_pub[l]) == hipSuccess) {
        ctx->vis_seq[l] = ctx->pub_seq[l];
        ctx->pub_pending[l] = false;
      }
    (void)hipGetLastError();
    for (int l = 0; l <= MAX_LANES; l++) A.vis.seq[l] = ctx->vis_seq[l];
    A.vis.seq[ctx->lane_id] = ctx->pub_seq[ctx->lane_id];
    A.index = (const u32 *)kc->index.p; A.mask = kc->index_mask; A.ents = (const cache_ent *)kc->ents.p;
    A.pool7 = (const u32 *)kc->pool7.p; A.pool10 = (const u32 *)kc->pool10.p;
  }
  A.gtable = (const u32 *)ctx->gtable;
  A.slots = (u32 *)ctx->slots.p;
  A.out = h + SMALL_OFF_OUT;
  A.shapes = h + SMALL_OFF_SHAPES;
  A.done = (u32 *)ctx->small_done.p;
  A.flag = (u32 *)(h + SMALL_OFF_FLAG);
  A.ticket = ++ctx->small_ticket ? ctx->small_ticket : ++ctx->small_ticket;
  hipLaunchKernelGGL(k_small_verify, dim3((unsigned)((n + 63) / 64)), dim3(512), 0, ctx->stream, A);
  HIPCHK(ctx, hipGetLastError());
  // spin on the completion word (the last block's last store, system scope), then make sure every block's bytes are in; fall back to
  // the stream if either does not show up
  volatile u32 *flag = (volatile u32 *)(h + SMALL_OFF_FLAG);
  auto all_in = [&] {
    for (size_t i = 0; i < n; i++)
      if (((volatile u8 *)h)[SMALL_OFF_OUT + i] == 0xEE || ((volatile u8 *)h)[SMALL_OFF_SHAPES + i] == 0xEE) return false;
    return true;
  };
  // (the kernel takes 0.14-0.9 ms; the spin is bounded by TIME -- LAMD_SPIN_US, default 2000 -- so that a daemon whose GPU is busy with other
  // lanes' work does not burn a core for tens of milliseconds: past the bound the call blocks in hipStreamSynchronize)
  bool seen = false;
  const auto t_spin = std::chrono::steady_clock::now();
  for (u32 spins = 0;; spins++) {
    if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == A.ticket && all_in()) { seen = true; break; }
#if defined(__x86_64__)
    __builtin_ia32_pause();
#endif
    if ((spins & 255u) == 255u && std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t_spin).count() > (long)ctx->spin_us) break;
  }
  if (!seen) {
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != A.ticket || !all_in()) { ctx->err = "k_small_verify: completion word / verdict bytes not written"; return LAMD_ERR_HIP; }
  }
  memcpy(ok, h + SMALL_OFF_OUT, n);
  ctx->last_mode = mode;
  ctx->last_n = n;
  u32 c[4] = {0, 0, 0, 0};  // rows per shape: 7-tooth combs, 10-tooth combs, ladder, rejected keys
  for (size_t i = 0; i < n; i++) {
    const u8 T = h[SMALL_OFF_SHAPES + i];
    c[T == 7 ? 0 : T == 10 ? 1 : T == 255 ? 2 : 3]++;
  }
  if (have_cache && c[2]) {  // remember the keys that went down the ladder
    for (size_t i = 0; i < n; i++)
      if (h[SMALL_OFF_SHAPES + i] == 255 && !(i && h[SMALL_OFF_SHAPES + i - 1] == 255 && memcmp(key + i * keystride, key + (i - 1) * keystride, keylen) == 0)) {
        const u64 fp = small_fingerprint(ctx->hash_seed, key + i * keystride, keylen);
        miss_put(ctx->small_missed, fp);
      }
  }
  {  // what lamd_get_info() reports about the last call
    u32 *hp = ctx->h_plan;
    memset(hp, 0, P_WORDS * 4);
    hp[P_L7] = c[0]; hp[P_L10] = c[1]; hp[P_COLD] = c[2]; hp[P_HITS] = c[0] + c[1] + c[3];
    ctx->last_keyed_call = true;
    ctx->last_lane = nullptr;
    ctx->last_chunk_lane = nullptr;
  }
  return LAMD_OK;
}

// does a host-buffer call of `rows` signature rows take the one-launch path?  (LAMD_KEYED=1 -- tests forcing the table-building path -- keeps
// calls of more than one block on the general path.  A cache-off engine takes it too: such an engine sends a small batch down the ladder
// either way, and the in-kernel ladder is the faster of the two -- 0.57 against 0.73 ms for a 484-row commitment, GPU session ah.)
static bool small_path(const lamd_ctx *ctx, size_t rows) {
  if (!ctx->small_kernel || rows > SMALL_MAX) return false;
  return rows <= 64 || ctx->keyed_mode <= 0;
}
static int run_host(lamd_ctx *ctx, int mode, size_t n, const u8 *a, const u8 *sig, const u8 *key, int keylen, size_t keystride,
                    u8 *ok) {
  HIPCHK(ctx, hipSetDevice(ctx->device));
  int rc;
  if ((rc = cache_maybe_reset(ctx)) != LAMD_OK) return rc;
  // (LAMD_KEYED=1 -- tests forcing the table-building path -- keeps calls of more than one block on the general path)
  if (small_path(ctx, n)) {
    rc = run_small(ctx, mode, n, a, sig, key, keylen, keystride, ok);
    if (rc != 1) return rc;
    ctx->force_learn = true;  // a key that missed before is back: this call builds and publishes the missing tables (general path below)
  }
  if ((rc = ensure(ctx, &ctx->in_a, n * 32)) != LAMD_OK) return rc;
  if ((rc = ensure(ctx, &ctx->in_b, n * 64)) != LAMD_OK) return rc;
  if ((rc = ensure(ctx, &ctx->in_c, n * keystride)) != LAMD_OK) return rc;
  if ((rc = ensure(ctx, &ctx->out, n)) != LAMD_OK) return rc;
  // the keys first: key de-duplication and table building only need them.  All three copies stay on ONE stream: the caller's
  // buffers are pageable, so the runtime stages them itself (the call blocks while it does); issuing the hash / signature copies on
  // the prep stream instead gained nothing and a 1 M-row batch came back with 5 % wrong verdicts (pageable copies in flight on two
  // streams; not root-caused) -- the pinned staging sets of the streaming queue (lamd_flush) do split their transfers.
  HIPCHK(ctx, hipMemcpyAsync(ctx->in_c.p, key, n * keystride, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(ctx->in_a.p, a, n * 32, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(ctx->in_b.p, sig, n * 64, hipMemcpyHostToDevice, ctx->stream));
  rc = run_device(ctx, mode, n, (const u8 *)ctx->in_a.p, (const u8 *)ctx->in_b.p, (const u8 *)ctx->in_c.p, keylen, keystride,
                  (u8 *)ctx->out.p);
  ctx->force_learn = false;  // (a path that does not look at the flag must not leave it set for an unrelated later call)
  if (rc != LAMD_OK) return rc;
  HIPCHK(ctx, hipMemcpyAsync(ok, ctx->out.p, n, hipMemcpyDeviceToHost, ctx->stream));
  return lamd_synchronize(ctx);
}

extern "C" int lamd_verify_ecdsa_batch(lamd_ctx *ctx, size_t n, const uint8_t *hash32, const uint8_t *sig64, const uint8_t *pub,
                                       size_t publen, size_t pubstride, uint8_t *ok) {
  if (!ctx) return LAMD_ERR_ARG;
  if (n == 0) return LAMD_OK;
  if (!hash32 || !sig64 || !pub || !ok || (publen != 33 && publen != 65) || pubstride < publen) {
    ctx->err = "bad argument";
    return LAMD_ERR_ARG;
  }
  return run_host(ctx, MODE_ECDSA, n, hash32, sig64, pub, (int)publen, pubstride, ok);
}

extern "C" int lamd_verify_schnorr_batch(lamd_ctx *ctx, size_t n, const uint8_t *msg32, const uint8_t *xonly32,
                                         const uint8_t *sig64, uint8_t *ok) {
  if (!ctx) return LAMD_ERR_ARG;
  if (n == 0) return LAMD_OK;
  if (!msg32 || !xonly32 || !sig64 || !ok) {
    ctx->err = "bad argument";
    return LAMD_ERR_ARG;
  }
  return run_host(ctx, MODE_SCHNORR, n, msg32, sig64, xonly32, 32, 32, ok);
}


// ---- known-answer check run by lamd_init: a miscompiled or misbehaving device must not hand out verdicts.
// ECDSA: the BOLT #11 example signature (reference: common/test/run-bolt11.c:465-467, key :310); BIP-340: official vector 1.
static const u8 KAT_E_HASH[32] = {0x11, 0x6f, 0xdb, 0x0f, 0x18, 0x35, 0x2c, 0x88, 0x6d, 0xeb, 0x26, 0x3f, 0x64, 0x66, 0xeb, 0x40, 0xe5, 0xe6, 0x51, 0x8b, 0x80, 0x23, 0x1a, 0x1f, 0x9d, 0xf8, 0x60, 0x88, 0xbf, 0xa4, 0x80, 0x43};
static const u8 KAT_E_SIG[64] = {0x26, 0x9f, 0xa6, 0x8a, 0x60, 0x51, 0xf2, 0x69, 0x91, 0xab, 0x50, 0xeb, 0x85, 0x14, 0x94, 0xd1, 0xa4, 0xb9, 0xc6, 0x16, 0xae, 0xee, 0x89, 0x2f, 0xf5, 0x0a, 0x14, 0x4a, 0xf4, 0x71, 0x55, 0x4a, 0x30, 0x57, 0xb2, 0xfe, 0xe4, 0x59, 0x10, 0xe2, 0x67, 0xc4, 0xef, 0x60, 0x67, 0xda, 0x10, 0x01, 0x6c, 0xf5, 0x51, 0x92, 0x37, 0xb3, 0xca, 0x1c, 0x1c, 0x20, 0x14, 0xcc, 0x1d, 0x6f, 0x69, 0xa6};
static const u8 KAT_E_PUB[33] = {0x03, 0xe7, 0x15, 0x6a, 0xe3, 0x3b, 0x0a, 0x20, 0x8d, 0x07, 0x44, 0x19, 0x91, 0x63, 0x17, 0x7e, 0x90, 0x9e, 0x80, 0x17, 0x6e, 0x55, 0xd9, 0x7a, 0x2f, 0x22, 0x1e, 0xde, 0x0f, 0x93, 0x4d, 0xd9, 0xad};
static const u8 KAT_S_MSG[32] = {0x24, 0x3f, 0x6a, 0x88, 0x85, 0xa3, 0x08, 0xd3, 0x13, 0x19, 0x8a, 0x2e, 0x03, 0x70, 0x73, 0x44, 0xa4, 0x09, 0x38, 0x22, 0x29, 0x9f, 0x31, 0xd0, 0x08, 0x2e, 0xfa, 0x98, 0xec, 0x4e, 0x6c, 0x89};
static const u8 KAT_S_PK[32] = {0xdf, 0xf1, 0xd7, 0x7f, 0x2a, 0x67, 0x1c, 0x5f, 0x36, 0x18, 0x37, 0x26, 0xdb, 0x23, 0x41, 0xbe, 0x58, 0xfe, 0xae, 0x1d, 0xa2, 0xde, 0xce, 0xd8, 0x43, 0x24, 0x0f, 0x7b, 0x50, 0x2b, 0xa6, 0x59};
static const u8 KAT_S_SIG[64] = {0x68, 0x96, 0xbd, 0x60, 0xee, 0xae, 0x29, 0x6d, 0xb4, 0x8a, 0x22, 0x9f, 0xf7, 0x1d, 0xfe, 0x07, 0x1b, 0xde, 0x41, 0x3e, 0x6d, 0x43, 0xf9, 0x17, 0xdc, 0x8d, 0xcf, 0x8c, 0x78, 0xde, 0x33, 0x41, 0x89, 0x06, 0xd1, 0x1a, 0xc9, 0x76, 0xab, 0xcc, 0xb2, 0x0b, 0x09, 0x12, 0x92, 0xbf, 0xf4, 0xea, 0x89, 0x7e, 0xfc, 0xb6, 0x39, 0xea, 0x87, 0x1c, 0xfa, 0x95, 0xf6, 0xde, 0x33, 0x9e, 0x4b, 0x0a};
static int init_known_answers(lamd_ctx *ctx) {
  u8 h[64], s[128], p[66], ok[2] = {9, 9};
  memcpy(h, KAT_E_HASH, 32); memcpy(h + 32, KAT_E_HASH, 32); h[32] ^= 1;   // second row: wrong hash
  memcpy(s, KAT_E_SIG, 64); memcpy(s + 64, KAT_E_SIG, 64);
  memcpy(p, KAT_E_PUB, 33); memcpy(p + 33, KAT_E_PUB, 33);
  int rc = lamd_verify_ecdsa_batch(ctx, 2, h, s, p, 33, 33, ok);
  if (rc != LAMD_OK) return rc;
  if (ok[0] != 1 || ok[1] != 0) { ctx->err = "device self-check failed: ECDSA known answer"; return LAMD_ERR_HIP; }
  u8 m[64], k[64];
  memcpy(m, KAT_S_MSG, 32); memcpy(m + 32, KAT_S_MSG, 32); m[40] ^= 0x80;
  memcpy(k, KAT_S_PK, 32); memcpy(k + 32, KAT_S_PK, 32);
  memcpy(s, KAT_S_SIG, 64); memcpy(s + 64, KAT_S_SIG, 64);
  ok[0] = ok[1] = 9;
  rc = lamd_verify_schnorr_batch(ctx, 2, m, k, s, ok);
  if (rc != LAMD_OK) return rc;
  if (ok[0] != 1 || ok[1] != 0) { ctx->err = "device self-check failed: BIP-340 known answer"; return LAMD_ERR_HIP; }
  return LAMD_OK;
}

// ---- single-item veneers
extern "C" int lamd_check_signed_hash(lamd_ctx *ctx, const uint8_t hash32[32], const uint8_t sig64[64], const uint8_t *pubkey,
                                      size_t publen) {
  uint8_t ok = 0;
  const int rc = lamd_verify_ecdsa_batch(ctx, 1, hash32, sig64, pubkey, publen, publen, &ok);
  return rc != LAMD_OK ? rc : ok;
}
extern "C" int lamd_check_signed_hash_nodeid(lamd_ctx *ctx, const uint8_t hash32[32], const uint8_t sig64[64],
                                             const uint8_t node_id33[33]) {
  return lamd_check_signed_hash(ctx, hash32, sig64, node_id33, 33);
}
extern "C" int lamd_check_schnorr_sig(lamd_ctx *ctx, const uint8_t hash32[32], const uint8_t pubkey33[33],
                                      const uint8_t bip340sig64[64]) {
  // bitcoin/signature.c:417-423: the (already valid) key is serialised compressed and its parity byte dropped
  if (!ctx || !pubkey33) return LAMD_ERR_ARG;
  uint8_t ok = 0;
  const int rc = lamd_verify_schnorr_batch(ctx, 1, hash32, pubkey33 + 1, bip340sig64, &ok);
  return rc != LAMD_OK ? rc : ok;
}

// ---- public-key parsing
extern "C" int lamd_pubkey_parse_batch(lamd_ctx *ctx, size_t n, const uint8_t *pub, size_t publen, size_t pubstride,
                                       uint8_t *out64, uint8_t *ok) {
  if (!ctx) return LAMD_ERR_ARG;
  if (n == 0) return LAMD_OK;
  if (!pub || !ok || (publen != 32 && publen != 33 && publen != 65) || pubstride < publen) {
    ctx->err = "bad argument";
    return LAMD_ERR_ARG;
  }
  HIPCHK(ctx, hipSetDevice(ctx->device));
  int rc;
  if ((rc = ensure(ctx, &ctx->in_c, n * pubstride)) != LAMD_OK) return rc;
  if ((rc = ensure(ctx, &ctx->qwords, n * 64)) != LAMD_OK) return rc;
  if ((rc = ensure(ctx, &ctx->keyok, n)) != LAMD_OK) return rc;
  HIPCHK(ctx, hipMemcpyAsync(ctx->in_c.p, pub, n * pubstride, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(k_keys, dim3(blocks_for(n)), dim3(256), 0, ctx->stream, n, (const u8 *)ctx->in_c.p, (int)publen, pubstride,
                     (const u32 *)nullptr, (u32 *)ctx->qwords.p, (u8 *)ctx->keyok.p, (const u32 *)nullptr);
  HIPCHK(ctx, hipGetLastError());
  HIPCHK(ctx, hipMemcpyAsync(ok, ctx->keyok.p, n, hipMemcpyDeviceToHost, ctx->stream));
  std::vector<u32> words;
  if (out64) {
    words.resize(n * 16);
    HIPCHK(ctx, hipMemcpyAsync(words.data(), ctx->qwords.p, n * 64, hipMemcpyDeviceToHost, ctx->stream));
  }
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  if (out64)
    for (size_t i = 0; i < n; i++) {
      store_words_be(out64 + 64 * i, &words[i * 16]);
      store_words_be(out64 + 64 * i + 32, &words[i * 16 + 8]);
    }
  return LAMD_OK;
}

// ---- check_tx_sig batches
extern "C" int lamd_check_tx_sig_batch(lamd_ctx *ctx, size_t n, const uint8_t *preimages, const uint64_t *off, const uint8_t *sighash_type,
                                       const uint8_t *has_witness, const uint8_t *sig64, const uint8_t *pub, size_t publen, size_t pubstride,
                                       uint8_t *ok) {
  if (!ctx) return LAMD_ERR_ARG;
  if (n == 0) return LAMD_OK;
  if (!preimages || !off || !sighash_type || !has_witness || !sig64 || !pub || !ok || (publen != 33 && publen != 65) || pubstride < publen) {
    ctx->err = "bad argument";
    return LAMD_ERR_ARG;
  }
  HIPCHK(ctx, hipSetDevice(ctx->device));
  int rc;
  // a few rows (an unmodified channeld checks ONE signature per check_tx_sig() call, channeld.c:2171,2224): gate + double SHA-256 on the
  // host -- the kernel's own inline function -- and the rows through the one-launch latency path
  if (small_path(ctx, n) && ctx->keyed_mode <= 0) {
    if ((rc = cache_maybe_reset(ctx)) != LAMD_OK) return rc;
    std::vector<u8> hs(32 * n), gate(n);
    for (size_t i = 0; i < n; i++) gate[i] = txsig_hash_one(preimages + off[i], (size_t)(off[i + 1] - off[i]), sighash_type[i], has_witness[i] != 0, &hs[32 * i]);
    rc = run_small(ctx, MODE_ECDSA, n, hs.data(), sig64, pub, (int)publen, pubstride, ok);
    if (rc == LAMD_OK) {
      for (size_t i = 0; i < n; i++)
        if (!gate[i]) ok[i] = 0;
      return LAMD_OK;
    }
    if (rc != 1) return rc;
    ctx->force_learn = true;
  }
  const size_t total = off[n] - off[0];
  std::vector<u64> rel(n + 1);
  for (size_t i = 0; i <= n; i++) rel[i] = off[i] - off[0];
  if ((rc = ensure(ctx, &ctx->g_msgs, total + 64)) != LAMD_OK) return rc;
  if ((rc = ensure(ctx, &ctx->g_off, (n + 1) * 8)) != LAMD_OK) return rc;
  if ((rc = ensure(ctx, &ctx->g_ids, 2 * n)) != LAMD_OK) return rc;
  if ((rc = ensure(ctx, &ctx->g_malformed, n)) != LAMD_OK) return rc;
  if ((rc = ensure(ctx, &ctx->in_a, n * 32)) != LAMD_OK) return rc;
  if ((rc = ensure(ctx, &ctx->in_b, n * 64)) != LAMD_OK) return rc;
  if ((rc = ensure(ctx, &ctx->in_c, n * pubstride)) != LAMD_OK) return rc;
  if ((rc = ensure(ctx, &ctx->out, n)) != LAMD_OK) return rc;
  u8 *d_types = (u8 *)ctx->g_ids.p, *d_wit = d_types + n;
  HIPCHK(ctx, hipMemcpyAsync(ctx->g_msgs.p, preimages + off[0], total, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(ctx->g_off.p, rel.data(), (n + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(d_types, sighash_type, n, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(d_wit, has_witness, n, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(ctx->in_b.p, sig64, n * 64, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(ctx->in_c.p, pub, n * pubstride, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(k_txsig_hash, dim3(blocks_for(n)), dim3(256), 0, ctx->stream, n, (const u8 *)ctx->g_msgs.p, (const u64 *)ctx->g_off.p,
                     (const u8 *)d_types, (const u8 *)d_wit, (u8 *)ctx->in_a.p, (u8 *)ctx->g_malformed.p);
  HIPCHK(ctx, hipGetLastError());
  rc = run_device(ctx, MODE_ECDSA, n, (const u8 *)ctx->in_a.p, (const u8 *)ctx->in_b.p, (const u8 *)ctx->in_c.p, (int)publen, pubstride,
                  (u8 *)ctx->out.p);
  ctx->force_learn = false;
  if (rc != LAMD_OK) return rc;
  hipLaunchKernelGGL(k_apply_gate, dim3(blocks_for(n)), dim3(256), 0, ctx->stream, n, (const u8 *)ctx->g_malformed.p, (u8 *)ctx->out.p);
  HIPCHK(ctx, hipGetLastError());
  HIPCHK(ctx, hipMemcpyAsync(ok, ctx->out.p, n, hipMemcpyDeviceToHost, ctx->stream));
  return lamd_synchronize(ctx);
}

// ---- check_tx_sig from transaction templates: the templates of a call as ONE staging blob (fixed-width columns first, 16-byte aligned, then the
// three byte strings; offsets relative to the call's first row) and the hashing kernel over it
struct txsig_blob {
  std::vector<u8> st;
  size_t o_inoff, o_outoff, o_scoff, o_amt, o_ver, o_lock, o_inum, o_nout, o_type, o_wit, o_in, o_out, o_sc, o_hdone, o_hhash, total;
};
// A row whose transaction has long lists (a commitment transaction with its 485 outputs: 20 KB under hashOutputs) or a very long script is hashed on the HOST
// while the blob is packed: SHA-256 is sequential, one lane needed ~6.5 ms for the 20 KB (measured: the whole 484-row call took that long), a host core
// ~0.1 ms; and the device's per-lane message buffer (k_txsig_tx_hash) holds 951 bytes per stream.  host_done: 0 = the device hashes the row, 1 = hashed here
// and the gate passed, 2 = hashed here and the gate refused.
constexpr size_t TXSIG_DEV_MAX_OUT = 900, TXSIG_DEV_MAX_IN = 25, TXSIG_DEV_MAX_SCRIPT = 700;  // each stream of a device row stays below TXH_MAX_STREAM
static_assert(TXSIG_DEV_MAX_OUT <= TXH_MAX_STREAM && 36 * TXSIG_DEV_MAX_IN <= TXH_MAX_STREAM && TXSIG_DEV_MAX_SCRIPT + 165 <= TXH_MAX_STREAM, "device rows must fit the lane buffer");
static void txsig_pack(txsig_blob &B, size_t n, const uint32_t *version, const uint32_t *locktime, const uint8_t *inputs40, const uint64_t *in_off,
                       const uint32_t *input_num, const uint64_t *amount_sat, const uint8_t *outputs, const uint64_t *out_off, const uint32_t *n_outputs,
                       const uint8_t *scripts, const uint64_t *script_off, const uint8_t *sighash_type, const uint8_t *has_witness) {
  const size_t nin = (size_t)(in_off[n] - in_off[0]), nout_b = (size_t)(out_off[n] - out_off[0]), nsc = (size_t)(script_off[n] - script_off[0]);
  size_t o = 0;
  auto take = [&](size_t bytes) { const size_t at = o; o = (o + bytes + 15) & ~(size_t)15; return at; };
  B.o_inoff = take((n + 1) * 8); B.o_outoff = take((n + 1) * 8); B.o_scoff = take((n + 1) * 8); B.o_amt = take(n * 8); B.o_ver = take(n * 4);
  B.o_lock = take(n * 4); B.o_inum = take(n * 4); B.o_nout = take(n * 4); B.o_type = take(n); B.o_wit = take(n); B.o_in = take(nin * 40 + 16);
  B.o_out = take(nout_b + 16); B.o_sc = take(nsc + 16); B.o_hdone = take(n); B.o_hhash = take(32 * n); B.total = o;
  B.st.assign(B.total, 0);
  std::vector<u8> &st = B.st;
  auto rel = [&](size_t at, const uint64_t *src) { for (size_t i = 0; i <= n; i++) ((u64 *)&st[at])[i] = src[i] - src[0]; };
  rel(B.o_inoff, in_off); rel(B.o_outoff, out_off); rel(B.o_scoff, script_off);
  memcpy(&st[B.o_amt], amount_sat, n * 8); memcpy(&st[B.o_ver], version, n * 4); memcpy(&st[B.o_lock], locktime, n * 4);
  memcpy(&st[B.o_inum], input_num, n * 4); memcpy(&st[B.o_nout], n_outputs, n * 4); memcpy(&st[B.o_type], sighash_type, n); memcpy(&st[B.o_wit], has_witness, n);
  memcpy(&st[B.o_in], inputs40 + 40 * in_off[0], nin * 40); memcpy(&st[B.o_out], outputs + out_off[0], nout_b); memcpy(&st[B.o_sc], scripts + script_off[0], nsc);
  for (size_t i = 0; i < n; i++)
    if ((size_t)(out_off[i + 1] - out_off[i]) > TXSIG_DEV_MAX_OUT || (size_t)(in_off[i + 1] - in_off[i]) > TXSIG_DEV_MAX_IN ||
        (size_t)(script_off[i + 1] - script_off[i]) > TXSIG_DEV_MAX_SCRIPT) {
      const bool pass = txsig_tx_hash_one(i, version, locktime, inputs40, in_off, input_num, amount_sat, outputs, out_off, n_outputs, scripts, script_off, sighash_type,
                                          has_witness, &st[B.o_hhash + 32 * i]);
      st[B.o_hdone + i] = pass ? 1 : 2;
    }
}
// BIP143 hash + gate of every row of the blob at `d` (device memory, or pinned device-mapped host memory) -> d_hash32, d_gate
static int txsig_hash_launch(lamd_ctx *ctx, size_t n, const txsig_blob &B, const u8 *d, u8 *d_hash32, u8 *d_gate) {
  hipLaunchKernelGGL(k_txsig_tx_hash, dim3((unsigned)((n + TXH_LANES - 1) / TXH_LANES)), dim3(TXH_LANES), 0, ctx->stream, n, (const u32 *)(d + B.o_ver), (const u32 *)(d + B.o_lock), d + B.o_in,
                     (const u64 *)(d + B.o_inoff), (const u32 *)(d + B.o_inum), (const u64 *)(d + B.o_amt), d + B.o_out, (const u64 *)(d + B.o_outoff),
                     (const u32 *)(d + B.o_nout), d + B.o_sc, (const u64 *)(d + B.o_scoff), d + B.o_type, d + B.o_wit, d + B.o_hdone, d + B.o_hhash, d_hash32, d_gate);
  HIPCHK(ctx, hipGetLastError());
  return LAMD_OK;
}
// the general path: templates to the device, BIP143 hashes there, the batch machinery, the gate, verdicts back
static int txsig_tx_general(lamd_ctx *ctx, size_t n, const uint32_t *version, const uint32_t *locktime, const uint8_t *inputs40, const uint64_t *in_off,
                            const uint32_t *input_num, const uint64_t *amount_sat, const uint8_t *outputs, const uint64_t *out_off, const uint32_t *n_outputs,
                            const uint8_t *scripts, const uint64_t *script_off, const uint8_t *sighash_type, const uint8_t *has_witness, const uint8_t *sig64,
                            const uint8_t *pub, size_t publen, size_t pubstride, uint8_t *ok) {
  int rc;
  txsig_blob B;
  txsig_pack(B, n, version, locktime, inputs40, in_off, input_num, amount_sat, outputs, out_off, n_outputs, scripts, script_off, sighash_type, has_witness);
  if ((rc = ensure(ctx, &ctx->g_msgs, B.total)) != LAMD_OK) return rc;
  if ((rc = ensure(ctx, &ctx->g_malformed, n)) != LAMD_OK) return rc;
  if ((rc = ensure(ctx, &ctx->in_a, n * 32)) != LAMD_OK) return rc;
  if ((rc = ensure(ctx, &ctx->in_b, n * 64)) != LAMD_OK) return rc;
  if ((rc = ensure(ctx, &ctx->in_c, n * pubstride)) != LAMD_OK) return rc;
  if ((rc = ensure(ctx, &ctx->out, n)) != LAMD_OK) return rc;
  HIPCHK(ctx, hipMemcpyAsync(ctx->g_msgs.p, B.st.data(), B.total, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(ctx->in_b.p, sig64, n * 64, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(ctx->in_c.p, pub, n * pubstride, hipMemcpyHostToDevice, ctx->stream));
  if ((rc = txsig_hash_launch(ctx, n, B, (const u8 *)ctx->g_msgs.p, (u8 *)ctx->in_a.p, (u8 *)ctx->g_malformed.p)) != LAMD_OK) return rc;
  rc = run_device(ctx, MODE_ECDSA, n, (const u8 *)ctx->in_a.p, (const u8 *)ctx->in_b.p, (const u8 *)ctx->in_c.p, (int)publen, pubstride,
                  (u8 *)ctx->out.p);
  ctx->force_learn = false;
  if (rc != LAMD_OK) return rc;
  hipLaunchKernelGGL(k_apply_gate, dim3(blocks_for(n)), dim3(256), 0, ctx->stream, n, (const u8 *)ctx->g_malformed.p, (u8 *)ctx->out.p);
  HIPCHK(ctx, hipGetLastError());
  HIPCHK(ctx, hipMemcpyAsync(ok, ctx->out.p, n, hipMemcpyDeviceToHost, ctx->stream));
  return lamd_synchronize(ctx);  // (B.st is pageable: the runtime staged it before hipMemcpyAsync returned)
}

// <= SMALL_MAX rows with the BIP143 hashes made ON THE DEVICE: the templates go into a pinned block, ONE asynchronous copy takes them to HBM, k_txsig_tx_hash
// leaves hashes and gate in device memory, k_small_verify -- queued right behind it -- takes them from there.  One small copy, two launches.  (Hashing on
// the host, which the small callers do for a handful of rows, is ~12 SHA-256 compressions per row: 1.7 ms for a 484-row commitment, 14 ms for eight of them.)
// LAMD_OK: verdicts in ok[]; 1: a key the latency path met before without a table is back -- the caller takes the batch path (force_learn is set).
static int txsig_small_device(lamd_ctx *ctx, size_t n, const uint32_t *version, const uint32_t *locktime, const uint8_t *inputs40, const uint64_t *in_off,
                              const uint32_t *input_num, const uint64_t *amount_sat, const uint8_t *outputs, const uint64_t *out_off, const uint32_t *n_outputs,
                              const uint8_t *scripts, const uint64_t *script_off, const uint8_t *sighash_type, const uint8_t *has_witness, const uint8_t *sig64,
                              const uint8_t *pub, size_t publen, size_t pubstride, uint8_t *ok) {
  int rc;
  txsig_blob B;
  txsig_pack(B, n, version, locktime, inputs40, in_off, input_num, amount_sat, outputs, out_off, n_outputs, scripts, script_off, sighash_type, has_witness);
  if (ctx->h_tmpl_cap < B.total) {
    if (ctx->h_tmpl) { HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); (void)hipHostFree(ctx->h_tmpl); ctx->h_tmpl = nullptr; ctx->h_tmpl_cap = 0; }
    const size_t want = B.total + B.total / 2 + 4096;
    HIPCHK(ctx, hipHostMalloc((void **)&ctx->h_tmpl, want, hipHostMallocMapped | hipHostMallocCoherent));
    ctx->h_tmpl_cap = want;
  }
  if ((rc = ensure(ctx, &ctx->in_a, n * 32)) != LAMD_OK) return rc;
  if ((rc = ensure(ctx, &ctx->g_malformed, n)) != LAMD_OK) return rc;
  if ((rc = ensure(ctx, &ctx->g_msgs, B.total)) != LAMD_OK) return rc;
  memcpy(ctx->h_tmpl, B.st.data(), B.total);
  // one asynchronous copy of the block (pinned -> HBM, ~250 B per row), then the hashing kernel over HBM.  The first version of this path let the kernel
  // read the templates where they lay, through the device mapping of the pinned block: 484 lanes walking ~250 bytes each byte by byte over PCIe took 6.5 ms
  // (rocprofv3, tools/commit_trace_probe.py) -- reading host memory from a kernel is fine for one 161-byte row, not for a SHA-256 input stream.
  HIPCHK(ctx, hipMemcpyAsync(ctx->g_msgs.p, ctx->h_tmpl, B.total, hipMemcpyHostToDevice, ctx->stream));
  if ((rc = txsig_hash_launch(ctx, n, B, (const u8 *)ctx->g_msgs.p, (u8 *)ctx->in_a.p, (u8 *)ctx->g_malformed.p)) != LAMD_OK) return rc;
  rc = run_small(ctx, MODE_ECDSA, n, nullptr, sig64, pub, (int)publen, pubstride, ok, (const u8 *)ctx->in_a.p, (const u8 *)ctx->g_malformed.p);
  if (rc == 1) ctx->force_learn = true;
  return rc;
}

extern "C" int lamd_check_tx_sig_tx_batch(lamd_ctx *ctx, size_t n, const uint32_t *version, const uint32_t *locktime, const uint8_t *inputs40,
                                          const uint64_t *in_off, const uint32_t *input_num, const uint64_t *amount_sat, const uint8_t *outputs,
                                          const uint64_t *out_off, const uint32_t *n_outputs, const uint8_t *scripts, const uint64_t *script_off,
                                          const uint8_t *sighash_type, const uint8_t *has_witness, const uint8_t *sig64, const uint8_t *pub,
                                          size_t publen, size_t pubstride, uint8_t *ok) {
  if (!ctx) return LAMD_ERR_ARG;
  if (n == 0) return LAMD_OK;
  if (!version || !locktime || !inputs40 || !in_off || !input_num || !amount_sat || !outputs || !out_off || !n_outputs || !scripts || !script_off ||
      !sighash_type || !has_witness || !sig64 || !pub || !ok || (publen != 33 && publen != 65) || pubstride < publen) {
    ctx->err = "bad argument";
    return LAMD_ERR_ARG;
  }
  HIPCHK(ctx, hipSetDevice(ctx->device));
  int rc;
  if ((rc = cache_maybe_reset(ctx)) != LAMD_OK) return rc;
  if (small_path(ctx, n) && ctx->keyed_mode <= 0 && n <= TXSIG_HOST_HASH_ROWS) {
    // a handful of rows: BIP143 hash on the host (the kernels' own inline function), rows through the latency path in ONE launch
    std::vector<u8> hs(32 * n), gate(n);
    for (size_t i = 0; i < n; i++)
      gate[i] = txsig_tx_hash_one(i, version, locktime, inputs40, in_off, input_num, amount_sat, outputs, out_off, n_outputs, scripts, script_off, sighash_type,
                                  has_witness, &hs[32 * i]);
    rc = run_small(ctx, MODE_ECDSA, n, hs.data(), sig64, pub, (int)publen, pubstride, ok);
    if (rc == LAMD_OK) {
      for (size_t i = 0; i < n; i++)
        if (!gate[i]) ok[i] = 0;
      return LAMD_OK;
    }
    if (rc != 1) return rc;
    ctx->force_learn = true;
  } else if (small_path(ctx, n) && ctx->keyed_mode <= 0) {  // up to 4 096 rows (several commitments at once: lamd_served's merged call): hashes on the device
    rc = txsig_small_device(ctx, n, version, locktime, inputs40, in_off, input_num, amount_sat, outputs, out_off, n_outputs, scripts, script_off, sighash_type, has_witness,
                            sig64, pub, publen, pubstride, ok);
    if (rc != 1) return rc;
  }
  return txsig_tx_general(ctx, n, version, locktime, inputs40, in_off, input_num, amount_sat, outputs, out_off, n_outputs, scripts, script_off, sighash_type,
                          has_witness, sig64, pub, publen, pubstride, ok);
}

// ---- one commitment_signed as ONE call (channeld/channeld.c:2171-2232): include/lightning_amd.h
extern "C" int lamd_check_commitment_signed(lamd_ctx *ctx, const lamd_tx_template *commit_tx, const uint8_t remote_funding33[33], const uint8_t commit_sig64[64],
                                            uint8_t commit_sighash_type, size_t n_htlc, const lamd_tx_template *htlc_txs, const uint8_t remote_htlckey33[33],
                                            const uint8_t *htlc_sigs64, const uint8_t *htlc_sighash_types, int64_t *first_bad, uint8_t *ok_rows) {
  if (!ctx) return LAMD_ERR_ARG;
  if (!commit_tx || !remote_funding33 || !commit_sig64 || !first_bad || (n_htlc && (!htlc_txs || !remote_htlckey33 || !htlc_sigs64 || !htlc_sighash_types))) {
    ctx->err = "bad argument";
    return LAMD_ERR_ARG;
  }
  *first_bad = 0;  // fails closed: an error return never leaves "all good" behind
  const size_t n = 1 + n_htlc;
  // the 1 + N (transaction, input 0 .. , signature) rows in the reference's order: row 0 the commitment transaction under the funding key,
  // row 1 + i HTLC transaction i under the htlc key
  std::vector<uint32_t> version(n), locktime(n), input_num(n), n_outputs(n);
  std::vector<uint64_t> in_off(n + 1), out_off(n + 1), sc_off(n + 1), amount(n);
  std::vector<u8> type(n), wit(n, 1), sig(64 * n), pub(33 * n);
  size_t nin = 0, nout = 0, nsc = 0;
  for (size_t i = 0; i < n; i++) {
    const lamd_tx_template *t = i ? &htlc_txs[i - 1] : commit_tx;
    if ((t->n_inputs && !t->inputs40) || (t->outputs_len && !t->outputs) || (t->script_len && !t->script)) {
      ctx->err = "bad argument: transaction template with a null array";
      return LAMD_ERR_ARG;
    }
    in_off[i] = nin; out_off[i] = nout; sc_off[i] = nsc;
    nin += t->n_inputs; nout += t->outputs_len; nsc += t->script_len;
  }
  in_off[n] = nin; out_off[n] = nout; sc_off[n] = nsc;
  std::vector<u8> inputs(40 * nin + 1), outputs(nout + 1), scripts(nsc + 1);
  for (size_t i = 0; i < n; i++) {
    const lamd_tx_template *t = i ? &htlc_txs[i - 1] : commit_tx;
    version[i] = t->version; locktime[i] = t->locktime; input_num[i] = t->input_num; n_outputs[i] = t->n_outputs; amount[i] = t->amount_sat;
    if (t->n_inputs) memcpy(&inputs[40 * in_off[i]], t->inputs40, 40 * (size_t)t->n_inputs);
    if (t->outputs_len) memcpy(&outputs[out_off[i]], t->outputs, t->outputs_len);
    if (t->script_len) memcpy(&scripts[sc_off[i]], t->script, t->script_len);
    type[i] = i ? htlc_sighash_types[i - 1] : commit_sighash_type;
    memcpy(&sig[64 * i], i ? htlc_sigs64 + 64 * (i - 1) : commit_sig64, 64);
    memcpy(&pub[33 * i], i ? remote_htlckey33 : remote_funding33, 33);
  }
  std::vector<u8> okv(n, 0);
  HIPCHK(ctx, hipSetDevice(ctx->device));
  int rc;
  if ((rc = cache_maybe_reset(ctx)) != LAMD_OK) return rc;
  bool done = false;
  if (small_path(ctx, n) && ctx->keyed_mode <= 0) {
    rc = txsig_small_device(ctx, n, version.data(), locktime.data(), inputs.data(), in_off.data(), input_num.data(), amount.data(), outputs.data(), out_off.data(),
                            n_outputs.data(), scripts.data(), sc_off.data(), type.data(), wit.data(), sig.data(), pub.data(), 33, 33, okv.data());
    if (rc == LAMD_OK) done = true;
    else if (rc != 1) return rc;   // 1: a key the latency path met before without a table is back (the channel's htlc key): built and published below
  }
  if (!done) {
    rc = txsig_tx_general(ctx, n, version.data(), locktime.data(), inputs.data(), in_off.data(), input_num.data(), amount.data(), outputs.data(), out_off.data(),
                          n_outputs.data(), scripts.data(), sc_off.data(), type.data(), wit.data(), sig.data(), pub.data(), 33, 33, okv.data());
    if (rc != LAMD_OK) return rc;
  }
  int64_t bad = -1;
  for (size_t i = 0; i < n && bad < 0; i++)
    if (!okv[i]) bad = (int64_t)i;  // 0: the commitment signature (:2171), 1 + i: htlc_sigs[i] (:2224) -- the first in the reference's order
  *first_bad = bad;
  if (ok_rows) memcpy(ok_rows, okv.data(), n);
  return LAMD_OK;
}

// ---- BOLT #12 signatures: n independent bolt12_check_signature(fields, messagename, fieldname, key, sig) calls (common/bolt12.c:80-92)
static int bolt12_hash_device(lamd_ctx *ctx, size_t n, const uint8_t *tlvs, const uint64_t *off, const char *messagename, const char *fieldname,
                              u8 *d_root, u8 *d_msg, u8 *d_valid) {
  const size_t total = (size_t)(off[n] - off[0]);
  std::vector<u64> rel(n + 1);
  for (size_t i = 0; i <= n; i++) rel[i] = off[i] - off[0];
  int rc;
  if ((rc = ensure(ctx, &ctx->g_msgs, total + 64)) != LAMD_OK) return rc;
  if ((rc = ensure(ctx, &ctx->g_off, (n + 1) * 8)) != LAMD_OK) return rc;
  bolt12_mids mids;
  const u8 leaf[6] = {'L', 'n', 'L', 'e', 'a', 'f'}, branch[8] = {'L', 'n', 'B', 'r', 'a', 'n', 'c', 'h'};
  bolt12_tag_midstate(leaf, 6, leaf, 0, mids.leaf);
  bolt12_tag_midstate(branch, 8, branch, 0, mids.branch);
  const std::string tag2 = std::string(messagename) + fieldname;   // "lightning" || messagename || fieldname (bitcoin/signature.c:389-405)
  bolt12_tag_midstate((const u8 *)"lightning", 9, (const u8 *)tag2.data(), tag2.size(), mids.sig);
  HIPCHK(ctx, hipMemcpyAsync(ctx->g_msgs.p, tlvs + off[0], total, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(ctx->g_off.p, rel.data(), (n + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));  // rel lives on this frame
  hipLaunchKernelGGL(k_bolt12_hash, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, ctx->stream, n, (const u8 *)ctx->g_msgs.p, (const u64 *)ctx->g_off.p, mids,
                     d_root, d_msg, d_valid);
  HIPCHK(ctx, hipGetLastError());
  return LAMD_OK;
}
extern "C" int lamd_bolt12_merkle_batch(lamd_ctx *ctx, size_t n, const uint8_t *tlvs, const uint64_t *off, const char *messagename,
                                        const char *fieldname, uint8_t *merkle32, uint8_t *sighash32, uint8_t *ok) {
  if (!ctx) return LAMD_ERR_ARG;
  if (n == 0) return LAMD_OK;
  if (!tlvs || !off || !messagename || !fieldname || !ok) { ctx->err = "bad argument"; return LAMD_ERR_ARG; }
  HIPCHK(ctx, hipSetDevice(ctx->device));
  int rc;
  if ((rc = ensure(ctx, &ctx->in_a, n * 32)) != LAMD_OK) return rc;
  if ((rc = ensure(ctx, &ctx->in_b, n * 64)) != LAMD_OK) return rc;
  if ((rc = ensure(ctx, &ctx->g_malformed, n)) != LAMD_OK) return rc;
  if ((rc = bolt12_hash_device(ctx, n, tlvs, off, messagename, fieldname, (u8 *)ctx->in_b.p, (u8 *)ctx->in_a.p, (u8 *)ctx->g_malformed.p)) != LAMD_OK) return rc;
  if (merkle32) HIPCHK(ctx, hipMemcpyAsync(merkle32, ctx->in_b.p, n * 32, hipMemcpyDeviceToHost, ctx->stream));
  if (sighash32) HIPCHK(ctx, hipMemcpyAsync(sighash32, ctx->in_a.p, n * 32, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(ok, ctx->g_malformed.p, n, hipMemcpyDeviceToHost, ctx->stream));
  return lamd_synchronize(ctx);
}
extern "C" int lamd_bolt12_check_signature_batch(lamd_ctx *ctx, size_t n, const uint8_t *tlvs, const uint64_t *off, const char *messagename,
                                                 const char *fieldname, const uint8_t *key33, size_t keystride, const uint8_t *sig64, uint8_t *ok) {
  if (!ctx) return LAMD_ERR_ARG;
  if (n == 0) return LAMD_OK;
  if (!tlvs || !off || !messagename || !fieldname || !key33 || keystride < 33 || !sig64 || !ok) { ctx->err = "bad argument"; return LAMD_ERR_ARG; }
  HIPCHK(ctx, hipSetDevice(ctx->device));
  int rc;
  if ((rc = cache_maybe_reset(ctx)) != LAMD_OK) return rc;
  // check_schnorr_sig serialises the key compressed and drops the parity byte (bitcoin/signature.c:417-422)
  std::vector<u8> xonly(n * 32);
  for (size_t i = 0; i < n; i++) memcpy(&xonly[32 * i], key33 + keystride * i + 1, 32);
  // a few invoices / offers (one per bolt12_check_signature() call, common/bolt12.c:80-92): merkle root and tagged hash on the host --
  // the kernel's own inline functions (bolt12.h) -- and the rows through the one-launch latency path
  if (n <= 256 && small_path(ctx, n) && ctx->keyed_mode <= 0) {
    bolt12_mids mids;
    const u8 leaf[6] = {'L', 'n', 'L', 'e', 'a', 'f'}, branch[8] = {'L', 'n', 'B', 'r', 'a', 'n', 'c', 'h'};
    bolt12_tag_midstate(leaf, 6, leaf, 0, mids.leaf);
    bolt12_tag_midstate(branch, 8, branch, 0, mids.branch);
    const std::string tag2 = std::string(messagename) + fieldname;
    bolt12_tag_midstate((const u8 *)"lightning", 9, (const u8 *)tag2.data(), tag2.size(), mids.sig);
    std::vector<u8> msg(32 * n, 0), valid(n);
    for (size_t i = 0; i < n; i++) {
      u8 root[32];
      valid[i] = bolt12_merkle_root(tlvs + off[i], (size_t)(off[i + 1] - off[i]), mids, root);
      if (valid[i]) bolt12_sighash(mids, root, &msg[32 * i]);
    }
    rc = run_small(ctx, MODE_SCHNORR, n, msg.data(), sig64, xonly.data(), 32, 32, ok);
    if (rc == LAMD_OK) {
      for (size_t i = 0; i < n; i++)
        if (!valid[i]) ok[i] = 0;
      return LAMD_OK;
    }
    if (rc != 1) return rc;
    ctx->force_learn = true;
  }
  if ((rc = ensure(ctx, &ctx->in_a, n * 32)) != LAMD_OK) return rc;
  if ((rc = ensure(ctx, &ctx->in_b, n * 64)) != LAMD_OK) return rc;
  if ((rc = ensure(ctx, &ctx->in_c, n * 32)) != LAMD_OK) return rc;
  if ((rc = ensure(ctx, &ctx->g_malformed, n)) != LAMD_OK) return rc;
  if ((rc = ensure(ctx, &ctx->out, n)) != LAMD_OK) return rc;
  if ((rc = bolt12_hash_device(ctx, n, tlvs, off, messagename, fieldname, nullptr, (u8 *)ctx->in_a.p, (u8 *)ctx->g_malformed.p)) != LAMD_OK) return rc;
  HIPCHK(ctx, hipMemcpyAsync(ctx->in_b.p, sig64, n * 64, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(ctx->in_c.p, xonly.data(), n * 32, hipMemcpyHostToDevice, ctx->stream));
  rc = run_device(ctx, MODE_SCHNORR, n, (const u8 *)ctx->in_a.p, (const u8 *)ctx->in_b.p, (const u8 *)ctx->in_c.p, 32, 32, (u8 *)ctx->out.p);
  ctx->force_learn = false;
  if (rc != LAMD_OK) return rc;
  hipLaunchKernelGGL(k_apply_gate, dim3(blocks_for(n)), dim3(256), 0, ctx->stream, n, (const u8 *)ctx->g_malformed.p, (u8 *)ctx->out.p);
  HIPCHK(ctx, hipGetLastError());
  HIPCHK(ctx, hipMemcpyAsync(ok, ctx->out.p, n, hipMemcpyDeviceToHost, ctx->stream));
  return lamd_synchronize(ctx);   // xonly lives on this frame
}

// ---- ECDSA public-key recovery (verify_core.h "ECDSA public-key recovery"): out_pub33[i] = the compressed key, ok[i] = 1, or
// a zeroed key and ok[i] = 0 where libsecp256k1's parse/recover would fail
static int recover_device(lamd_ctx *ctx, size_t n, const u8 *d_hash, const u8 *d_sig, const u8 *d_recid, u8 *d_pub33, u8 *d_ok) {
  HIPCHK(ctx, hipSetDevice(ctx->device));
  int rc;
  for (size_t o = 0; o < n; o += ctx->chunk) {
    const size_t m = n - o < ctx->chunk ? n - o : ctx->chunk;
    if ((rc = ensure(ctx, &ctx->recs, m * sizeof(prep_rec))) != LAMD_OK) return rc;
    if ((rc = ensure(ctx, &ctx->g_pub, m * 33 + 16)) != LAMD_OK) return rc;
    prep_rec *recs = (prep_rec *)ctx->recs.p;
    hipLaunchKernelGGL(k_recover_prep, dim3(blocks_for(final_threads(ctx, m))), dim3(256), 0, ctx->stream, m, d_hash + 32 * o, d_sig + 64 * o,
                       d_recid + o, recs, (u8 *)ctx->g_pub.p);
    HIPCHK(ctx, hipEventRecord(ctx->ev_prep, ctx->stream));
    rc = launch_direct(ctx, MODE_RECOVER, m, nullptr, nullptr, recs, d_sig + 64 * o, (const u8 *)ctx->g_pub.p, 33, 33, nullptr, nullptr, d_ok + o, false);
    if (rc != LAMD_OK) return rc;
    hipLaunchKernelGGL(k_recover_final, dim3(blocks_for(final_threads(ctx, m))), dim3(256), 0, ctx->stream, m, (u32 *)ctx->slots.p, d_ok + o,
                       d_pub33 + 33 * o);
    HIPCHK(ctx, hipGetLastError());
  }
  return LAMD_OK;
}
extern "C" int lamd_ecdsa_recover_batch_device(lamd_ctx *ctx, size_t n, const void *d_hash32, const void *d_sig64, const void *d_recid,
                                               void *d_pub33, void *d_ok) {
  if (!ctx) return LAMD_ERR_ARG;
  if (n == 0) return LAMD_OK;
  if (!d_hash32 || !d_sig64 || !d_recid || !d_pub33 || !d_ok) {
    ctx->err = "bad argument";
    return LAMD_ERR_ARG;
  }
  lamd_ctx *L;
  int rc = pick_lane(ctx, &L);
  if (rc != LAMD_OK) return rc;
  rc = recover_device(L, n, (const u8 *)d_hash32, (const u8 *)d_sig64, (const u8 *)d_recid, (u8 *)d_pub33, (u8 *)d_ok);
  if (rc != LAMD_OK && L != ctx) ctx->err = L->err;
  return rc;
}
extern "C" int lamd_ecdsa_recover_batch(lamd_ctx *ctx, size_t n, const uint8_t *hash32, const uint8_t *sig64, const uint8_t *recid,
                                        uint8_t *pub33, uint8_t *ok) {
  if (!ctx) return LAMD_ERR_ARG;
  if (n == 0) return LAMD_OK;
  if (!hash32 || !sig64 || !recid || !pub33 || !ok) {
    ctx->err = "bad argument";
    return LAMD_ERR_ARG;
  }
  HIPCHK(ctx, hipSetDevice(ctx->device));
  int rc;
  if ((rc = ensure(ctx, &ctx->in_a, n * 32)) != LAMD_OK) return rc;
  if ((rc = ensure(ctx, &ctx->in_b, n * 64)) != LAMD_OK) return rc;
  if ((rc = ensure(ctx, &ctx->in_c, n * 34)) != LAMD_OK) return rc;  // recid | keys out
  if ((rc = ensure(ctx, &ctx->out, n)) != LAMD_OK) return rc;
  u8 *d_recid = (u8 *)ctx->in_c.p, *d_keys = d_recid + n;
  HIPCHK(ctx, hipMemcpyAsync(ctx->in_a.p, hash32, n * 32, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(ctx->in_b.p, sig64, n * 64, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(d_recid, recid, n, hipMemcpyHostToDevice, ctx->stream));
  rc = recover_device(ctx, n, (const u8 *)ctx->in_a.p, (const u8 *)ctx->in_b.p, d_recid, d_keys, (u8 *)ctx->out.p);
  if (rc != LAMD_OK) return rc;
  HIPCHK(ctx, hipMemcpyAsync(pub33, d_keys, n * 33, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(ok, ctx->out.p, n, hipMemcpyDeviceToHost, ctx->stream));
  return lamd_synchronize(ctx);
}

// ---- fee grind: see k_grind.  Returns 1 (found; *feerate, *fee set), 0 (no candidate verifies) or an error < 0.
extern "C" int lamd_grind_htlc_tx_fee(lamd_ctx *ctx, const uint8_t *preimage, size_t preimage_len, const uint8_t *outputs,
                                      size_t outputs_len, uint64_t input_sat, uint64_t weight, uint32_t min_feerate,
                                      uint32_t max_feerate, const uint8_t sig64[64], uint8_t sighash_type, int has_witness_script,
                                      const uint8_t pubkey33[33], uint32_t *feerate, uint64_t *fee) {
  if (!ctx) return LAMD_ERR_ARG;
  if (!preimage || !outputs || !sig64 || !pubkey33 || !feerate || !fee || preimage_len < 40 + 4 || outputs_len < 9 ||
      outputs_len > (size_t)GRIND_MAX_OUTPUTS || weight == 0 || weight > ((uint64_t)1 << 31)) {
    ctx->err = "bad argument";
    return LAMD_ERR_ARG;
  }
  // the sighash-type gate of check_tx_sig (bitcoin/signature.c:206-211) is the same for every candidate
  if (!(sighash_type == 1 || (sighash_type == 0x83 && has_witness_script))) return 0;
  if (max_feerate < min_feerate) return 0;
  const size_t ncand = (size_t)max_feerate - min_feerate + 1;
  if (ncand > ((size_t)1 << 30)) {
    ctx->err = "feerate range too large";
    return LAMD_ERR_ARG;
  }
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const size_t lead = ((preimage_len - 40) / 64) * 64, tail_len = preimage_len - lead;  // hashOutputs starts in the first tail block
  // staging: preimage | outputs | sig | key | setup | best
  const size_t o_out = (preimage_len + 15) & ~(size_t)15, o_sig = (o_out + outputs_len + 15) & ~(size_t)15, o_key = o_sig + 64,
               o_setup = o_key + 48, o_best = o_setup + ((sizeof(grind_setup) + 15) & ~(size_t)15), total = o_best + 16;
  int rc;
  if ((rc = ensure(ctx, &ctx->g_msgs, total)) != LAMD_OK) return rc;
  if ((rc = ensure(ctx, &ctx->slots, (size_t)SLOT_WORDS * 4)) != LAMD_OK) return rc;
  std::vector<u8> stage(total, 0);
  memcpy(&stage[0], preimage, preimage_len);
  memcpy(&stage[o_out], outputs, outputs_len);
  memcpy(&stage[o_sig], sig64, 64);
  memcpy(&stage[o_key], pubkey33, 33);
  memset(&stage[o_best], 0xFF, 4);
  u8 *d = (u8 *)ctx->g_msgs.p;
  HIPCHK(ctx, hipMemcpyAsync(d, stage.data(), total, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(k_grind_setup, dim3(1), dim3(64), 0, ctx->stream, d + o_sig, d + o_key, d, (u32)(lead / 64), (u32 *)ctx->slots.p,
                     (const u32 *)ctx->gtable, (grind_setup *)(d + o_setup));
  hipLaunchKernelGGL(k_grind, dim3(blocks_for(ncand)), dim3(256), 0, ctx->stream, (u32)ncand, min_feerate, weight, input_sat, d + lead,
                     (u32)tail_len, (u32)lead, d + o_out, (u32)outputs_len, (const grind_setup *)(d + o_setup), (const u32 *)ctx->gtable,
                     (u32 *)(d + o_best));
  HIPCHK(ctx, hipGetLastError());
  u32 best = 0xFFFFFFFFu;
  HIPCHK(ctx, hipMemcpyAsync(&best, d + o_best, 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  if (best == 0xFFFFFFFFu) return 0;
  *feerate = min_feerate + best;
  *fee = (uint64_t)*feerate * weight / 1000;
  return 1;
}

// ---- gossip
// device core: everything resident.  The signature rows of all messages form one ECDSA batch with 33-byte keys (cut into
// chunks like any other batch: a message's rows may straddle a chunk); the reduce runs once over all messages.
static int gossip_device(lamd_ctx *ctx, size_t n, const u8 *d_msgs, const u64 *d_off, const u8 *d_ids, const u64 *d_rowbase, size_t rows,
                         int8_t *d_verdict, const u64 *d_len = nullptr) {
  int rc;
  if ((rc = ensure(ctx, &ctx->g_hash, rows * 32)) != LAMD_OK) return rc;
  if ((rc = ensure(ctx, &ctx->g_sig, rows * 64)) != LAMD_OK) return rc;
  if ((rc = ensure(ctx, &ctx->g_pub, rows * 33 + 16)) != LAMD_OK) return rc;
  if ((rc = ensure(ctx, &ctx->g_malformed, n)) != LAMD_OK) return rc;
  if ((rc = ensure(ctx, &ctx->g_ok, rows)) != LAMD_OK) return rc;
  if ((rc = ensure(ctx, &ctx->keyok_row, rows)) != LAMD_OK) return rc;
  hipLaunchKernelGGL(k_gossip_expand, dim3(blocks_for(n)), dim3(256), 0, ctx->stream, n, d_msgs, d_off, d_len, d_ids, d_rowbase, (u8 *)ctx->g_hash.p,
                     (u8 *)ctx->g_sig.p, (u8 *)ctx->g_pub.p, (u8 *)ctx->g_malformed.p);
  HIPCHK(ctx, hipGetLastError());
  rc = run_device(ctx, MODE_ECDSA, rows, (const u8 *)ctx->g_hash.p, (const u8 *)ctx->g_sig.p, (const u8 *)ctx->g_pub.p, 33, 33, (u8 *)ctx->g_ok.p,
                  (u8 *)ctx->keyok_row.p);
  if (rc != LAMD_OK) return rc;
  hipLaunchKernelGGL(k_gossip_reduce, dim3(blocks_for(n)), dim3(256), 0, ctx->stream, n, d_msgs, d_off, d_rowbase, (const u8 *)ctx->g_ok.p,
                     (const u8 *)ctx->keyok_row.p, (const u8 *)ctx->g_malformed.p, d_verdict);
  HIPCHK(ctx, hipGetLastError());
  return LAMD_OK;
}

extern "C" int lamd_sigcheck_gossip_batch_device(lamd_ctx *ctx, size_t n, const void *d_msgs, const void *d_off, const void *d_node_ids33,
                                                 const void *d_rowbase, size_t rows, void *d_verdict) {
  if (!ctx) return LAMD_ERR_ARG;
  if (n == 0) return LAMD_OK;
  if (!d_msgs || !d_off || !d_rowbase || !d_verdict || rows == 0) {
    ctx->err = "bad argument";
    return LAMD_ERR_ARG;
  }
  HIPCHK(ctx, hipSetDevice(ctx->device));
  lamd_ctx *L;
  int rc = pick_lane(ctx, &L);
  if (rc != LAMD_OK) return rc;
  // d_node_ids33 == NULL travels to the kernel as it is: a channel_update in such a batch gets verdict -1 (gossip_expand_one), nothing is
  // read through a placeholder (the host-buffer call refuses the batch with LAMD_ERR_ARG; here the message types are only known on the device)
  rc = gossip_device(L, n, (const u8 *)d_msgs, (const u64 *)d_off, (const u8 *)d_node_ids33, (const u64 *)d_rowbase, rows, (int8_t *)d_verdict);
  if (rc != LAMD_OK && L != ctx) ctx->err = L->err;
  return rc;
}

// The spans form: message i = d_msgs[d_start[i], d_start[i] + d_len[i]).  One call over any selection of a resident message blob -- a rank of a job cut
// per message kind (sharding.segment_bounds) verifies its range of the announcements and its range of the updates as ONE call with one front end.
extern "C" int lamd_sigcheck_gossip_spans_device(lamd_ctx *ctx, size_t n, const void *d_msgs, const void *d_start, const void *d_len,
                                                 const void *d_node_ids33, const void *d_rowbase, size_t rows, void *d_verdict) {
  if (!ctx) return LAMD_ERR_ARG;
  if (n == 0) return LAMD_OK;
  if (!d_msgs || !d_start || !d_len || !d_rowbase || !d_verdict || rows == 0) {
    ctx->err = "bad argument";
    return LAMD_ERR_ARG;
  }
  HIPCHK(ctx, hipSetDevice(ctx->device));
  lamd_ctx *L;
  int rc = pick_lane(ctx, &L);
  if (rc != LAMD_OK) return rc;
  rc = gossip_device(L, n, (const u8 *)d_msgs, (const u64 *)d_start, (const u8 *)d_node_ids33, (const u64 *)d_rowbase, rows, (int8_t *)d_verdict,
                     (const u64 *)d_len);
  if (rc != LAMD_OK && L != ctx) ctx->err = L->err;
  return rc;
}

extern "C" int lamd_sigcheck_gossip_batch(lamd_ctx *ctx, size_t n, const uint8_t *msgs, const uint64_t *off,
                                          const uint8_t *node_ids33, int8_t *verdict) {
  if (!ctx) return LAMD_ERR_ARG;
  if (n == 0) return LAMD_OK;
  if (!msgs || !off || !verdict) {
    ctx->err = "bad argument";
    return LAMD_ERR_ARG;
  }
  HIPCHK(ctx, hipSetDevice(ctx->device));
  // host framing pass: signature rows per message (4 for channel_announcement, 1 otherwise)
  std::vector<u64> rowbase(n + 1), rel(n + 1);
  u64 rows = 0;
  bool need_ids = false;
  for (size_t i = 0; i < n; i++) {
    rowbase[i] = rows;
    const size_t len = off[i + 1] - off[i];
    const u32 type = len >= 2 ? (((u32)msgs[off[i]] << 8) | msgs[off[i] + 1]) : 0;
    rows += type == GOSSIP_CANN ? 4 : 1;
    need_ids |= type == GOSSIP_CUPD;
  }
  rowbase[n] = rows;
  if (need_ids && !node_ids33) {
    ctx->err = "channel_update in batch but node_ids33 is NULL";
    return LAMD_ERR_ARG;
  }
  int rc;
  // a handful of messages (an unmodified gossipd checks ONE per sigcheck_*() call, gossmap_manage.c:687,924,1217): framing and the
  // double SHA-256 of the signed tail on the host -- the same inline functions the kernels run -- and the rows through the one-launch
  // latency path; the first-bad reduction on the host.  (A key that comes back without a table sends the call down the general path
  // below once: that call builds and publishes the table.)
  if (small_path(ctx, rows) && ctx->keyed_mode <= 0) {
    if ((rc = cache_maybe_reset(ctx)) != LAMD_OK) return rc;
    std::vector<u8> hs(32 * rows), sg(64 * rows), pk(33 * rows), ok(rows), bad(n);
    for (size_t i = 0; i < n; i++)
      bad[i] = gossip_expand_one(msgs + off[i], off[i + 1] - off[i], node_ids33 ? node_ids33 + 33 * i : nullptr, rowbase[i + 1] - rowbase[i], &hs[32 * rowbase[i]],
                                 &sg[64 * rowbase[i]], &pk[33 * rowbase[i]]);
    rc = run_small(ctx, MODE_ECDSA, rows, hs.data(), sg.data(), pk.data(), 33, 33, ok.data());
    if (rc == LAMD_OK) {
      const u8 *shapes = ctx->h_small + SMALL_OFF_SHAPES;
      std::vector<u8> keyok(rows);
      for (size_t r = 0; r < rows; r++) keyok[r] = shapes[r] != 0;
      for (size_t i = 0; i < n; i++) verdict[i] = (int8_t)gossip_reduce_one(rowbase[i + 1] - rowbase[i], &ok[rowbase[i]], &keyok[rowbase[i]], bad[i] != 0);
      return LAMD_OK;
    }
    if (rc != 1) return rc;
    ctx->force_learn = true;
  }
  const size_t total = off[n] - off[0];
  if ((rc = ensure(ctx, &ctx->g_msgs, total + 64)) != LAMD_OK) return rc;
  if ((rc = ensure(ctx, &ctx->g_off, (n + 1) * 8)) != LAMD_OK) return rc;
  if ((rc = ensure(ctx, &ctx->g_ids, n * 33)) != LAMD_OK) return rc;
  if ((rc = ensure(ctx, &ctx->g_rowbase, (n + 1) * 8)) != LAMD_OK) return rc;
  if ((rc = ensure(ctx, &ctx->g_verdict, n)) != LAMD_OK) return rc;
  for (size_t i = 0; i <= n; i++) rel[i] = off[i] - off[0];
  HIPCHK(ctx, hipMemcpyAsync(ctx->g_msgs.p, msgs + off[0], total, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(ctx->g_off.p, rel.data(), (n + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(ctx->g_rowbase.p, rowbase.data(), (n + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
  if (node_ids33) HIPCHK(ctx, hipMemcpyAsync(ctx->g_ids.p, node_ids33, n * 33, hipMemcpyHostToDevice, ctx->stream));
  rc = gossip_device(ctx, n, (const u8 *)ctx->g_msgs.p, (const u64 *)ctx->g_off.p, (const u8 *)ctx->g_ids.p, (const u64 *)ctx->g_rowbase.p,
                     rows, (int8_t *)ctx->g_verdict.p);
  ctx->force_learn = false;
  if (rc != LAMD_OK) return rc;
  HIPCHK(ctx, hipMemcpyAsync(verdict, ctx->g_verdict.p, n, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return LAMD_OK;
}

// ---- streaming queues
// Filling the pinned staging set is the host's share of a streamed batch (~130-160 bytes per signature); one thread copies at
// ~10 GB/s, which is less than the device verifies (cfg5: 4.84 M signatures = 775 MB per 14 ms).  Large pushes are therefore cut
// into pieces and copied by a few short-lived threads.
struct copy_job { u8 *dst; const u8 *src; size_t bytes; };
static void par_copy(const copy_job *jobs, int njobs) {
  size_t total = 0;
  for (int i = 0; i < njobs; i++) total += jobs[i].bytes;
  static const int max_threads = [] {
    const char *e = getenv("LAMD_COPY_THREADS");
    const int hw = (int)std::thread::hardware_concurrency();
    int t = e ? atoi(e) : (hw >= 8 ? 4 : (hw >= 4 ? 2 : 1));
    return t < 1 ? 1 : (t > 16 ? 16 : t);
  }();
  if (total < ((size_t)4 << 20) || max_threads == 1) {
    for (int i = 0; i < njobs; i++) memcpy(jobs[i].dst, jobs[i].src, jobs[i].bytes);
    return;
  }
  // pieces of ~total/threads bytes, walking the jobs in order
  std::vector<copy_job> pieces;
  const size_t piece = (total + max_threads - 1) / max_threads;
  for (int i = 0; i < njobs; i++)
    for (size_t o = 0; o < jobs[i].bytes; o += piece) pieces.push_back({jobs[i].dst + o, jobs[i].src + o, std::min(piece, jobs[i].bytes - o)});
  std::atomic<size_t> next{0};
  auto work = [&] {
    for (size_t k; (k = next.fetch_add(1)) < pieces.size();) memcpy(pieces[k].dst, pieces[k].src, pieces[k].bytes);
  };
  std::vector<std::thread> th;
  for (int t = 1; t < max_threads; t++) th.emplace_back(work);
  work();
  for (auto &t : th) t.join();
}
static int queue_reserve(lamd_ctx *ctx, lamd_ctx::queue &q, size_t keybytes, size_t want = 0) {
  if (q.n < q.cap && want <= q.cap) return LAMD_OK;
  size_t ncap = q.cap ? q.cap * 2 : 1024;
  if (ncap < want) ncap = want;  // a large push sizes the set in one allocation (pinned allocations are slow)
  u8 *nblk = nullptr, *nk = nullptr;
  {  // both or none: a failure (pinned memory is scarce exactly when this path is hit) must not leak the block already taken
    hipError_t e = hipHostMalloc((void **)&nblk, lamd_ctx::queue::blk_bytes(ncap, keybytes), hipHostMallocDefault);
    if (e == hipSuccess) e = hipHostMalloc((void **)&nk, 2 * ncap, hipHostMallocDefault);  // verdicts | row shapes of a small flush
    if (e != hipSuccess) {
      for (u8 *h : {nblk, nk})
        if (h) (void)hipHostFree(h);
      ctx->err = std::string("hipHostMalloc (staging set): ") + hipGetErrorString(e);
      return e == hipErrorOutOfMemory ? LAMD_ERR_NOMEM : LAMD_ERR_HIP;
    }
  }
  u8 *nc = nblk, *nb = nblk + lamd_ctx::queue::key_bytes_padded(ncap, keybytes), *na = nb + ncap * 64;
  if (q.n) {
    memcpy(na, q.h_a, q.n * 32);
    memcpy(nb, q.h_b, q.n * 64);
    memcpy(nc, q.h_c, q.n * keybytes);
  }
  for (u8 **h : {&q.h_blk, &q.h_ok})
    if (*h) (void)hipHostFree(*h);
  q.h_blk = nblk;
  q.h_a = na; q.h_b = nb; q.h_c = nc; q.h_ok = nk;
  q.cap = ncap;
  return LAMD_OK;
}
static const size_t Q_KEYBYTES[Q_KINDS] = {33, 65, 32};

static int queue_reserve_n(lamd_ctx *ctx, lamd_ctx::queue &q, size_t keybytes, size_t extra) {
  if (q.n + extra > q.cap) return queue_reserve(ctx, q, keybytes, q.n + extra);
  return LAMD_OK;
}
// room for n more triples of one kind in the open staging set: where their rows go (pinned host memory) and the ticket of the first
static int queue_take(lamd_ctx *ctx, int kind, size_t n, u8 **pa, u8 **pb, u8 **pc) {
  if (n == 0 || n > (size_t)0x3FFFFFFF) {
    ctx->err = "bad argument";
    return LAMD_ERR_ARG;
  }
  if (ctx->q_open < 0) {
    ctx->err = "queue: every staging set is in flight (collect a flush with poll/wait first)";
    return LAMD_ERR_STATE;
  }
  lamd_ctx::queue_set &set = ctx->qs[ctx->q_open];
  lamd_ctx::queue &q = set.q[kind];
  const size_t kb = Q_KEYBYTES[kind];
  // tickets are positions inside the open staging set (they restart at 0 after every flush), so they never overflow and
  // collect() needs no arithmetic across sets; a set holds fewer than 2^30 triples
  if (set.rows + n > (size_t)0x3FFFFFFF) {
    ctx->err = "queue: staging set full (flush first)";
    return LAMD_ERR_STATE;
  }
  const int rc = queue_reserve_n(ctx, q, kb, n);
  if (rc != LAMD_OK) return rc;
  *pa = q.h_a + 32 * q.n;
  *pb = q.h_b + 64 * q.n;
  *pc = q.h_c + kb * q.n;
  const size_t first = set.rows;
  if (!q.tickets.empty() && q.tickets.back().row0 + q.tickets.back().count == q.n && q.tickets.back().ticket0 + q.tickets.back().count == first)
    q.tickets.back().count += n;  // consecutive pushes of one kind: one span
  else
    q.tickets.push_back({q.n, (u32)first, n});
  q.n += n;
  set.rows += n;
  return (int)first;
}
// appends n triples (row strides 32 / 64 / keystride); returns the ticket of the first one
static int queue_push(lamd_ctx *ctx, int kind, size_t n, const u8 *a, const u8 *sig, const u8 *key, size_t keystride) {
  if (!ctx) return LAMD_ERR_ARG;
  if (!a || !sig || !key || keystride < Q_KEYBYTES[kind]) {
    ctx->err = "bad argument";
    return LAMD_ERR_ARG;
  }
  const size_t kb = Q_KEYBYTES[kind];
  u8 *da, *db, *dc;
  const int first = queue_take(ctx, kind, n, &da, &db, &dc);
  if (first < 0) return first;
  if (keystride == kb) {
    const copy_job jobs[3] = {{da, a, 32 * n}, {db, sig, 64 * n}, {dc, key, kb * n}};
    par_copy(jobs, 3);
  } else {
    const copy_job jobs[2] = {{da, a, 32 * n}, {db, sig, 64 * n}};
    par_copy(jobs, 2);
    for (size_t i = 0; i < n; i++) memcpy(dc + kb * i, key + keystride * i, kb);
  }
  return first;
}
// The producer's form: the rows are written straight into the staging set (a sidecar reads its callers' triples from a socket or
// shared memory into these pointers), so no copy from a caller-owned buffer is made at all -- the host-side copy into pinned memory
// is what bounds lamd_queue_*_batch() on large batches (~25 GB/s with four threads against 31 GB/s of triples at the device's rate).
extern "C" int lamd_queue_reserve(lamd_ctx *ctx, size_t n, size_t keylen, uint8_t **hash32, uint8_t **sig64, uint8_t **key) {
  if (!ctx) return LAMD_ERR_ARG;
  if (!hash32 || !sig64 || !key || (keylen != 33 && keylen != 65 && keylen != 32)) {
    ctx->err = "bad argument";
    return LAMD_ERR_ARG;
  }
  return queue_take(ctx, keylen == 33 ? Q_ECDSA33 : keylen == 65 ? Q_ECDSA65 : Q_SCHNORR, n, hash32, sig64, key);
}
extern "C" int lamd_queue_ecdsa(lamd_ctx *ctx, const uint8_t hash32[32], const uint8_t sig64[64], const uint8_t *pubkey,
                                size_t publen) {
  if (ctx && publen != 33 && publen != 65) {
    ctx->err = "bad key length";
    return LAMD_ERR_ARG;
  }
  return queue_push(ctx, publen == 33 ? Q_ECDSA33 : Q_ECDSA65, 1, hash32, sig64, pubkey, publen);
}
extern "C" int lamd_queue_schnorr(lamd_ctx *ctx, const uint8_t msg32[32], const uint8_t xonly32[32], const uint8_t sig64[64]) {
  return queue_push(ctx, Q_SCHNORR, 1, msg32, sig64, xonly32, 32);
}
// n triples at once (one commitment_signed: 1 + up to 483 signatures); returns the first ticket, the rest follow consecutively
extern "C" int lamd_queue_ecdsa_batch(lamd_ctx *ctx, size_t n, const uint8_t *hash32, const uint8_t *sig64, const uint8_t *pubkey,
                                      size_t publen, size_t pubstride) {
  if (ctx && publen != 33 && publen != 65) {
    ctx->err = "bad key length";
    return LAMD_ERR_ARG;
  }
  return queue_push(ctx, publen == 33 ? Q_ECDSA33 : Q_ECDSA65, n, hash32, sig64, pubkey, pubstride);
}
extern "C" int lamd_queue_schnorr_batch(lamd_ctx *ctx, size_t n, const uint8_t *msg32, const uint8_t *xonly32, const uint8_t *sig64) {
  return queue_push(ctx, Q_SCHNORR, n, msg32, sig64, xonly32, 32);
}
// The rows stay where they are: n triples in the caller's memory (keys packed, publen bytes each) get a row range in the open staging set and cross
// the bus FROM THE CALLER'S BUFFERS when the set is flushed -- the form for a host that already holds its callers' rows in pinned memory (lamd_served:
// the clients' shared blocks, registered with lamd_host_register()).  The buffers must not change until the flush that carries the rows has been
// collected (lamd_poll / lamd_wait).  Batches the latency kernel would take (<= 4 096 rows), and rows in memory the runtime does not hold pinned,
// are copied like lamd_queue_*_batch().
static int queue_push_inplace(lamd_ctx *ctx, int kind, size_t n, const u8 *a, const u8 *sig, const u8 *key) {
  if (!ctx) return LAMD_ERR_ARG;
  if (!a || !sig || !key) {
    ctx->err = "bad argument";
    return LAMD_ERR_ARG;
  }
  if (n <= SMALL_MAX || !ctx->use_copy_stream) return queue_push(ctx, kind, n, a, sig, key, Q_KEYBYTES[kind]);
  // Only memory the runtime holds pinned stays in place.  An asynchronous copy from PAGEABLE memory makes the runtime pin the pages itself for the
  // length of the transfer, and such a transient pin next to (in one page with) a registered range left the runtime unable to finish a later
  // pageable copy (tests: a process hung in an unrelated device-to-host copy, round 6) -- so rows in memory that is not registered end to end are
  // copied into the staging set like any others.
  const size_t w[3] = {32, 64, Q_KEYBYTES[kind]};
  const u8 *col[3] = {a, sig, key};
  for (int c = 0; c < 3; c++)
    for (const u8 *p : {col[c], col[c] + w[c] * n - 1}) {
      hipPointerAttribute_t at;
      if (hipPointerGetAttributes(&at, p) != hipSuccess || at.type != hipMemoryTypeHost) {
        (void)hipGetLastError();
        return queue_push(ctx, kind, n, a, sig, key, Q_KEYBYTES[kind]);
      }
    }
  u8 *da, *db, *dc;
  const int first = queue_take(ctx, kind, n, &da, &db, &dc);
  if (first < 0) return first;
  lamd_ctx::queue &q = ctx->qs[ctx->q_open].q[kind];
  q.foreign.push_back({q.n - n, n, a, sig, key});
  return first;
}
extern "C" int lamd_queue_ecdsa_batch_inplace(lamd_ctx *ctx, size_t n, const uint8_t *hash32, const uint8_t *sig64, const uint8_t *pubkey, size_t publen) {
  if (ctx && publen != 33 && publen != 65) {
    ctx->err = "bad key length";
    return LAMD_ERR_ARG;
  }
  return queue_push_inplace(ctx, publen == 33 ? Q_ECDSA33 : Q_ECDSA65, n, hash32, sig64, pubkey);
}
extern "C" int lamd_queue_schnorr_batch_inplace(lamd_ctx *ctx, size_t n, const uint8_t *msg32, const uint8_t *xonly32, const uint8_t *sig64) {
  return queue_push_inplace(ctx, Q_SCHNORR, n, msg32, sig64, xonly32);
}
// Pins a range of the caller's memory for every device (hipHostRegisterPortable), so that rows queued in place leave it by DMA.  Callable from any
// thread, also while another thread drives the context: nothing of the context is touched but its device number.  < 0: LAMD_ERR_HIP (the range stays
// usable: rows queued "in place" from memory that is not pinned are copied into the staging set).  Register whole, page-aligned blocks.
extern "C" int lamd_host_register(lamd_ctx *ctx, void *p, size_t bytes) {
  if (!ctx || !p || !bytes) return LAMD_ERR_ARG;
  if (hipSetDevice(ctx->device) != hipSuccess || hipHostRegister(p, bytes, hipHostRegisterPortable) != hipSuccess) {
    (void)hipGetLastError();
    return LAMD_ERR_HIP;
  }
  return LAMD_OK;
}
// The NUMA node a device hangs on (/sys/bus/pci/devices/<domain:bus:device.function>/numa_node), -1 when the platform does not say: where a host puts the
// threads and buffers that feed that device (lamd_served binds each device's engine thread there; bench.py binds itself).  No context needed.
extern "C" int lamd_device_numa_node(int device) {
  char bdf[64] = {0};
  if (hipDeviceGetPCIBusId(bdf, (int)sizeof bdf, device) != hipSuccess) {
    (void)hipGetLastError();
    return -1;
  }
  for (char *c = bdf; *c; c++) *c = (char)tolower((unsigned char)*c);
  const std::string path = std::string("/sys/bus/pci/devices/") + bdf + "/numa_node";
  FILE *f = fopen(path.c_str(), "r");
  if (!f) return -1;
  int node = -1;
  if (fscanf(f, "%d", &node) != 1) node = -1;
  fclose(f);
  return node;
}
extern "C" int lamd_host_unregister(lamd_ctx *ctx, void *p) {
  if (!ctx || !p) return LAMD_ERR_ARG;
  if (hipSetDevice(ctx->device) != hipSuccess || hipHostUnregister(p) != hipSuccess) {
    (void)hipGetLastError();
    return LAMD_ERR_HIP;
  }
  return LAMD_OK;
}
// Launches everything queued so far as one batch per kind (asynchronous, on the next lane) and opens the next staging set:
// queueing continues while up to QUEUE_SETS - 1 flushes are in flight.
extern "C" int lamd_flush(lamd_ctx *ctx) {
  if (!ctx) return LAMD_ERR_ARG;
  if (ctx->q_open < 0) {
    ctx->err = "flush: every staging set is in flight (collect a flush with poll/wait first)";
    return LAMD_ERR_STATE;
  }
  HIPCHK(ctx, hipSetDevice(ctx->device));
  lamd_ctx::queue_set &qs = ctx->qs[ctx->q_open];
  lamd_ctx *L;
  int rc = pick_lane(ctx, &L);
  if (rc != LAMD_OK) return rc;
  for (int kind = 0; kind < Q_KINDS; kind++) {
    lamd_ctx::queue &q = qs.q[kind];
    if (!q.n) continue;
    const size_t kb = Q_KEYBYTES[kind];
    q.small_flush = false;
    // A small flush (one commitment_signed, a handful of gossip messages) is ONE launch of the latency kernel over the pinned staging
    // rows themselves: no H2D copies, no front end, the verdict bytes written straight into the set's pinned verdict block; the caller's
    // poll / wait sees it through the flush's event as ever.  A key that was met without a table before sends the flush down the
    // general path once (that call builds and publishes the table).
    if (q.n <= SMALL_MAX && ctx->small_kernel && ctx->keyed_mode <= 0 && ctx->cache_mode != 0 && ctx->cache_store.shared) {
      if (ctx->small_missed.empty()) ctx->small_missed.assign(MISS_SLOTS, 0);
      bool learn = false;
      for (size_t i = 0; i < q.n && !learn; i++) {
        if (i && memcmp(q.h_c + i * kb, q.h_c + (i - 1) * kb, kb) == 0) continue;
        const u64 fp = small_fingerprint(ctx->hash_seed, q.h_c + i * kb, (int)kb);
        learn = miss_has(ctx->small_missed, fp);
      }
      if (learn) {
        L->force_learn = true;
        small_forget(ctx, L, q.h_c, kb, (int)kb, q.n);
      } else {
        if ((rc = ensure(L, &L->slots, ((q.n + 63) & ~(size_t)63) * SLOT_WORDS * 4)) != LAMD_OK) { ctx->err = L->err; return rc; }
        small_args A;
        memset(&A, 0, sizeof A);
        A.a32 = q.h_a; A.sig64 = q.h_b; A.key = q.h_c;
        A.keylen = (int)kb; A.mode = kind == Q_SCHNORR ? MODE_SCHNORR : MODE_ECDSA; A.n = (u32)q.n;
        A.seed = ctx->hash_seed;
        lamd_ctx::key_cache *kc = &ctx->cache_store;
        for (int l = 0; l <= MAX_LANES; l++)
          if (ctx->pub_pending[l] && hipEventQuery(ctx->ev_pub[l]) == hipSuccess) {
            ctx->vis_seq[l] = ctx->pub_seq[l];
            ctx->pub_pending[l] = false;
          }
        (void)hipGetLastError();
        for (int l = 0; l <= MAX_LANES; l++) A.vis.seq[l] = ctx->vis_seq[l];
        A.vis.seq[L->lane_id] = ctx->pub_seq[L->lane_id];
        A.index = (const u32 *)kc->index.p; A.mask = kc->index_mask; A.ents = (const cache_ent *)kc->ents.p;
        A.pool7 = (const u32 *)kc->pool7.p; A.pool10 = (const u32 *)kc->pool10.p;
        A.gtable = (const u32 *)ctx->gtable;
        A.slots = (u32 *)L->slots.p;
        A.out = q.h_ok;
        A.shapes = q.h_ok + q.cap;
        hipLaunchKernelGGL(k_small_verify, dim3((unsigned)((q.n + 63) / 64)), dim3(512), 0, L->stream, A);
        HIPCHK(ctx, hipGetLastError());
        q.small_flush = true;
        continue;
      }
    }
    if ((rc = ensure(ctx, &q.d_blk, lamd_ctx::queue::blk_bytes(q.cap, kb) + 64)) != LAMD_OK) return rc;
    if ((rc = ensure(ctx, &q.d_ok, q.n)) != LAMD_OK) return rc;
    u8 *const qd_c = (u8 *)q.d_blk.p, *const qd_b = qd_c + lamd_ctx::queue::key_bytes_padded(q.cap, kb), *const qd_a = qd_b + q.cap * 64;
    // The keys first: de-duplication and table building (main stream) only need them and start while the hashes and signatures
    // are still on the bus.  All three copies go down the lane's PREP stream back to back and the main stream waits for the
    // keys' event: a copy that itself waits for another stream's copy starts 0.5-1 ms late (rocprofv3 timeline of the pipelined
    // loop, tools/host_path_trace.py: the dependency is resolved by the runtime's host thread), a kernel that waits for a copy
    // does not.  (Round 3 tried pulling the rows over PCIe with a KERNEL on the lane's streams instead -- no SDMA queue, every
    // dependency in order on compute queues: 119-137 M verifies/s against 168-185 M/s for these copies, profiles/r03_ab_variants.txt:
    // device-initiated reads of host memory reach a fraction of the SDMA engines' 57 GB/s.  Dropped.)
    const bool split = q.n <= L->chunk;
    if (!q.foreign.empty()) {
      // rows queued in place: every run of rows -- the caller's buffers for the in-place ones, the staging set for what lies between them -- goes down
      // the copy stream into its rows of the device twin, column by column in the order and behind the events of the staged form below (keys first:
      // de-duplication and table building start on them)
      hipStream_t &csr = ctx->copy_streams[ctx->copy_turn++ % (unsigned)ctx->n_copy_streams];
      if (!csr) HIPCHK(ctx, hipStreamCreateWithFlags(&csr, hipStreamNonBlocking));
      hipStream_t cs = csr;
      auto column = [&](int col) -> hipError_t {   // 0 keys, 1 signatures, 2 hashes
        const size_t w = col == 0 ? kb : col == 1 ? 64 : 32;
        u8 *const dst = col == 0 ? qd_c : col == 1 ? qd_b : qd_a;
        const u8 *const own = col == 0 ? q.h_c : col == 1 ? q.h_b : q.h_a;
        size_t at = 0;
        hipError_t e = hipSuccess;
        for (const auto &f : q.foreign) {
          if (f.row0 > at && e == hipSuccess) e = hipMemcpyAsync(dst + at * w, own + at * w, (f.row0 - at) * w, hipMemcpyHostToDevice, cs);
          if (e == hipSuccess) e = hipMemcpyAsync(dst + f.row0 * w, col == 0 ? f.c : col == 1 ? f.b : f.a, f.count * w, hipMemcpyHostToDevice, cs);
          at = f.row0 + f.count;
        }
        if (at < q.n && e == hipSuccess) e = hipMemcpyAsync(dst + at * w, own + at * w, (q.n - at) * w, hipMemcpyHostToDevice, cs);
        return e;
      };
      const int events = ctx->copy_events ? ctx->copy_events : (ctx->q_inflight >= 2 ? 1 : 3);
      HIPCHK(ctx, column(0));
      if (events > 1) {
        HIPCHK(ctx, hipEventRecord(q.ev_keys, cs));
        HIPCHK(ctx, hipStreamWaitEvent(L->stream, q.ev_keys, 0));
      }
      HIPCHK(ctx, column(1));
      if (events >= 3) {
        HIPCHK(ctx, hipEventRecord(q.ev_sigs, cs));
        L->sigs_pending = true;
        L->ev_sigs_wait = q.ev_sigs;
      }
      HIPCHK(ctx, column(2));
      HIPCHK(ctx, hipEventRecord(q.ev_all, cs));
      if (events == 1) HIPCHK(ctx, hipStreamWaitEvent(L->stream, q.ev_all, 0));
      if (events == 2) {
        L->sigs_pending = true;
        L->ev_sigs_wait = q.ev_all;
      }
      HIPCHK(ctx, hipStreamWaitEvent(L->stream2, q.ev_all, 0));
    } else if (split && ctx->use_copy_stream) {
      // Round 3: the copies of EVERY flush go down one stream of their own, in flush order, behind nothing but each other.  On the
      // lane's prep stream they stood behind the lane's previous call, so with more flushes in flight than lanes the rows of the
      // next flush still crossed the bus only after the lane had gone idle (1.2 ms for the keys of 1 M rows before its first kernel
      // could start) -- which is why more staging sets bought nothing.  The device buffers belong to the staging set, and a set is
      // not refilled before its flush has been collected, so nothing else orders these copies.
      hipStream_t &csr = ctx->copy_streams[ctx->copy_turn++ % (unsigned)ctx->n_copy_streams];   // successive flushes (and kinds of one flush) take turns
      if (!csr) HIPCHK(ctx, hipStreamCreateWithFlags(&csr, hipStreamNonBlocking));
      hipStream_t cs = csr;
      // (LAMD_COPY_EVENTS unset: one event when other flushes are already in flight -- their kernels cover the 1.8 ms the front end now waits for the
      // signatures and hashes --, three when this flush is alone and its latency is what the caller sees)
      const int events = ctx->copy_events ? ctx->copy_events : (ctx->q_inflight >= 2 ? 1 : 3);
      if (events == 1 && q.n == q.cap && ctx->copy_one) {
        // the set is full to the row: keys | signatures | hashes are one contiguous block on both sides -- ONE copy command (every command on
        // the copy stream is followed by a gap of 0.1-1 ms while the chip is busy: profiles/r05_stream_timeline.txt)
        HIPCHK(ctx, hipMemcpyAsync(qd_c, q.h_blk, lamd_ctx::queue::blk_bytes(q.cap, kb), hipMemcpyHostToDevice, cs));
        HIPCHK(ctx, hipEventRecord(q.ev_all, cs));
        HIPCHK(ctx, hipStreamWaitEvent(L->stream, q.ev_all, 0));
        HIPCHK(ctx, hipStreamWaitEvent(L->stream2, q.ev_all, 0));
      } else if (events == 1) {
        // ONE event per flush, behind its last copy: an event record is a marker packet on the stream's compute queue, and the copy behind it
        // waits for that packet -- between the copies of a busy chip that hand-over took 0.1-1 ms each (the gaps of profiles/r05_stream_timeline.txt).
        // Three copies back to back, then the marker; the lane's front end starts once all three have landed.
        HIPCHK(ctx, hipMemcpyAsync(qd_c, q.h_c, q.n * kb, hipMemcpyHostToDevice, cs));
        HIPCHK(ctx, hipMemcpyAsync(qd_b, q.h_b, q.n * 64, hipMemcpyHostToDevice, cs));
        HIPCHK(ctx, hipMemcpyAsync(qd_a, q.h_a, q.n * 32, hipMemcpyHostToDevice, cs));
        HIPCHK(ctx, hipEventRecord(q.ev_all, cs));
        HIPCHK(ctx, hipStreamWaitEvent(L->stream, q.ev_all, 0));
        HIPCHK(ctx, hipStreamWaitEvent(L->stream2, q.ev_all, 0));
      } else {
      HIPCHK(ctx, hipMemcpyAsync(qd_c, q.h_c, q.n * kb, hipMemcpyHostToDevice, cs));
      HIPCHK(ctx, hipEventRecord(q.ev_keys, cs));
      HIPCHK(ctx, hipStreamWaitEvent(L->stream, q.ev_keys, 0));
      HIPCHK(ctx, hipMemcpyAsync(qd_b, q.h_b, q.n * 64, hipMemcpyHostToDevice, cs));
      if (events >= 3) {
        HIPCHK(ctx, hipEventRecord(q.ev_sigs, cs));
        L->sigs_pending = true;
        L->ev_sigs_wait = q.ev_sigs;
      }
      HIPCHK(ctx, hipMemcpyAsync(qd_a, q.h_a, q.n * 32, hipMemcpyHostToDevice, cs));
      HIPCHK(ctx, hipEventRecord(q.ev_all, cs));
      if (events < 3) {  // two events: keys, then everything
        L->sigs_pending = true;
        L->ev_sigs_wait = q.ev_all;
      }
      HIPCHK(ctx, hipStreamWaitEvent(L->stream2, q.ev_all, 0));  // the preparation reads all three
      }
    } else if (split) {
      HIPCHK(ctx, hipEventRecord(L->ev_fork, L->stream));  // after whatever the lane's main stream still holds
      HIPCHK(ctx, hipStreamWaitEvent(L->stream2, L->ev_fork, 0));
      HIPCHK(ctx, hipMemcpyAsync(qd_c, q.h_c, q.n * kb, hipMemcpyHostToDevice, L->stream2));
      HIPCHK(ctx, hipEventRecord(L->ev_keys, L->stream2));
      HIPCHK(ctx, hipStreamWaitEvent(L->stream, L->ev_keys, 0));
      // signatures before hashes: the row-list builders on the main stream read r and s (early reject) long before the
      // preparation needs the hashes
      HIPCHK(ctx, hipMemcpyAsync(qd_b, q.h_b, q.n * 64, hipMemcpyHostToDevice, L->stream2));
      HIPCHK(ctx, hipEventRecord(L->ev_sigs, L->stream2));
      L->sigs_pending = true;
      L->ev_sigs_wait = L->ev_sigs;
      HIPCHK(ctx, hipMemcpyAsync(qd_a, q.h_a, q.n * 32, hipMemcpyHostToDevice, L->stream2));
    } else {
      HIPCHK(ctx, hipMemcpyAsync(qd_c, q.h_c, q.n * kb, hipMemcpyHostToDevice, L->stream));
      HIPCHK(ctx, hipMemcpyAsync(qd_a, q.h_a, q.n * 32, hipMemcpyHostToDevice, L->stream));
      HIPCHK(ctx, hipMemcpyAsync(qd_b, q.h_b, q.n * 64, hipMemcpyHostToDevice, L->stream));
    }
    rc = run_device(L, kind == Q_SCHNORR ? MODE_SCHNORR : MODE_ECDSA, q.n, (const u8 *)qd_a, (const u8 *)qd_b,
                    (const u8 *)qd_c, (int)kb, kb, (u8 *)q.d_ok.p);
    L->force_learn = false;
    if (rc != LAMD_OK) {
      if (L != ctx) ctx->err = L->err;
      return rc;
    }
    if (ctx->use_d2h_stream) {
      // the verdicts leave on a stream of their own: on the lane's stream the copy (and its completion) stood between this flush's last kernel
      // and the first kernel of the lane's next call
      HIPCHK(ctx, hipEventRecord(q.ev_res, L->stream));
      HIPCHK(ctx, hipStreamWaitEvent(ctx->d2h_stream, q.ev_res, 0));
      HIPCHK(ctx, hipMemcpyAsync(q.h_ok, q.d_ok.p, q.n, hipMemcpyDeviceToHost, ctx->d2h_stream));
    } else {
      HIPCHK(ctx, hipMemcpyAsync(q.h_ok, q.d_ok.p, q.n, hipMemcpyDeviceToHost, L->stream));
    }
  }
  if (ctx->use_d2h_stream) {
    HIPCHK(ctx, hipEventRecord(qs.tail, L->stream));
    HIPCHK(ctx, hipStreamWaitEvent(ctx->d2h_stream, qs.tail, 0));
    HIPCHK(ctx, hipEventRecord(qs.done, ctx->d2h_stream));
  } else {
    HIPCHK(ctx, hipEventRecord(qs.done, L->stream));
  }
  ctx->q_fifo[ctx->q_inflight++] = ctx->q_open;
  // the next set to fill: any set that is not in flight
  ctx->q_open = -1;
  for (int sidx = 0; sidx < QUEUE_SETS && ctx->q_open < 0; sidx++) {
    bool busy = false;
    for (int k = 0; k < ctx->q_inflight; k++) busy |= ctx->q_fifo[k] == sidx;
    if (!busy) ctx->q_open = sidx;
  }
  return LAMD_OK;
}

// verdicts of the OLDEST flush, tickets in submission order
static int collect(lamd_ctx *ctx, uint8_t *ok, size_t cap, size_t *n) {
  const int sidx = ctx->q_fifo[0];
  lamd_ctx::queue_set &qs = ctx->qs[sidx];
  const size_t total = qs.rows;
  if (total > cap) {
    ctx->err = "result buffer too small";
    return LAMD_ERR_ARG;
  }
  for (int kind = 0; kind < Q_KINDS; kind++) {
    lamd_ctx::queue &q = qs.q[kind];
    for (const auto &sp : q.tickets) memcpy(ok + sp.ticket0, q.h_ok + sp.row0, sp.count);
    if (q.small_flush && !ctx->small_missed.empty()) {  // remember the keys the latency kernel had to take down the ladder (run_small does the same)
      const size_t kb = Q_KEYBYTES[kind];
      const u8 *shapes = q.h_ok + q.cap;
      for (size_t i = 0; i < q.n; i++)
        if (shapes[i] == 255 && !(i && shapes[i - 1] == 255 && memcmp(q.h_c + i * kb, q.h_c + (i - 1) * kb, kb) == 0)) {
          const u64 fp = small_fingerprint(ctx->hash_seed, q.h_c + i * kb, (int)kb);
          miss_put(ctx->small_missed, fp);
        }
    }
    q.small_flush = false;
    q.tickets.clear();
    q.foreign.clear();
    q.n = 0;
  }
  qs.rows = 0;
  if (n) *n = total;
  for (int k = 1; k < ctx->q_inflight; k++) ctx->q_fifo[k - 1] = ctx->q_fifo[k];
  ctx->q_inflight--;
  if (ctx->q_open < 0) ctx->q_open = sidx;
  return 1;
}
extern "C" int lamd_poll(lamd_ctx *ctx, uint8_t *ok, size_t cap, size_t *n) {
  if (!ctx || !ok) return LAMD_ERR_ARG;
  if (!ctx->q_inflight) {
    ctx->err = "poll before flush";
    return LAMD_ERR_STATE;
  }
  const hipError_t e = hipEventQuery(ctx->qs[ctx->q_fifo[0]].done);
  if (e == hipErrorNotReady) return 0;
  HIPCHK(ctx, e);
  return collect(ctx, ok, cap, n);
}
extern "C" int lamd_wait(lamd_ctx *ctx, uint8_t *ok, size_t cap, size_t *n) {
  if (!ctx || !ok) return LAMD_ERR_ARG;
  if (!ctx->q_inflight) {
    ctx->err = "wait before flush";
    return LAMD_ERR_STATE;
  }
  HIPCHK(ctx, hipEventSynchronize(ctx->qs[ctx->q_fifo[0]].done));
  return collect(ctx, ok, cap, n);
}

// ---- device self-test: the same inline functions evaluated on the GPU and on the host (this TU's host
// pass), stage by stage, so a miscompile / hardware difference is localised to one primitive.
constexpr int ST_LANES = 64;
constexpr int ST_WORDS = 512;  // output words per lane
struct st_in { u32 a[8], b[8]; u8 hash[32], sig[64], pub33[33], pad[3]; };

LAMD_HD void selftest_lane(const st_in &in, const u32 *gtable, u32 *slot, u32 *o) {
  int k = 0;
  const fe a = fe_from_words(in.a), b = fe_from_words(in.b);
  u32 w[8];
  fe_to_words(w, fe_normalize(fe_mul(a, b))); for (int i = 0; i < 8; i++) o[k++] = w[i];                    // 0 fe_mul
  fe_to_words(w, fe_normalize(fe_sqr(a))); for (int i = 0; i < 8; i++) o[k++] = w[i];                       // 8 fe_sqr
  fe_to_words(w, fe_normalize(fe_inv(a))); for (int i = 0; i < 8; i++) o[k++] = w[i];                       // 16 fe_inv
  fe_to_words(w, fe_normalize(fe_sqrt_candidate(a))); for (int i = 0; i < 8; i++) o[k++] = w[i];            // 24 fe_sqrt
  fe_to_words(w, fe_normalize(fe_add(fe_neg(a, 1), fe_mul_int(b, 3)))); for (int i = 0; i < 8; i++) o[k++] = w[i];  // 32 lazy add/neg
  sc x, y;
  bool of;
  x = sc_from_words(in.a, &of); y = sc_from_words(in.b, &of);
  const sc m = sc_mul(x, y); for (int i = 0; i < 8; i++) o[k++] = m.w[i];                                    // 40 sc_mul
  const sc iv = sc_inv(x); for (int i = 0; i < 8; i++) o[k++] = iv.w[i];                                     // 48 sc_inv
  glv_half h1, h2;
  glv_split(&h1, &h2, x);
  for (int i = 0; i < 4; i++) o[k++] = h1.mag[i];
  o[k++] = h1.top; o[k++] = h1.neg;
  for (int i = 0; i < 4; i++) o[k++] = h2.mag[i];
  o[k++] = h2.top; o[k++] = h2.neg;                                                                          // 56..67 glv
  u32 qx[8], qy[8];
  const bool kok = parse_pubkey(in.pub33, 33, qx, qy);
  for (int i = 0; i < 8; i++) o[k++] = qx[i];
  for (int i = 0; i < 8; i++) o[k++] = qy[i];
  o[k++] = kok;                                                                                               // 68..84 key
  prep_rec rec;
  ecdsa_prep_thread(0, 1, 1, in.hash, in.sig, &rec);
  for (int i = 0; i < 8; i++) o[k++] = rec.u1[i];
  for (int i = 0; i < 4; i++) o[k++] = rec.k1[i];
  for (int i = 0; i < 4; i++) o[k++] = rec.k2[i];
  o[k++] = rec.flags;                                                                                         // 85..101 prep
  const ge q = ge_from_words(qx, qy);
  const fe zg = build_q_table(slot, q);
  fe_to_words(w, fe_normalize(zg)); for (int i = 0; i < 8; i++) o[k++] = w[i];                              // 102 zg
  for (int e = 0; e < 8; e++)                                                                                  // 110..301 table (canonical words)
    for (int c = 0; c < 3; c++) {
      fe_to_words(w, fe_normalize(slot_load_fe(slot + e * SLOT_ENTRY_WORDS + c * TW)));
      for (int i = 0; i < 8; i++) o[k++] = w[i];
    }
  const gej R = ecmult_lane(rec, q, slot, gtable);
  fe_to_words(w, fe_normalize(R.x)); for (int i = 0; i < 8; i++) o[k++] = w[i];
  fe_to_words(w, fe_normalize(R.y)); for (int i = 0; i < 8; i++) o[k++] = w[i];
  fe_to_words(w, fe_normalize(R.z)); for (int i = 0; i < 8; i++) o[k++] = w[i];
  o[k++] = R.inf;                                                                                             // 302..326 R
  u32 rw[8];
  load_words_be(rw, in.sig);
  o[k++] = ecdsa_final(R, rw);                                                                                // 327 verdict
  while (k < ST_WORDS) o[k++] = 0;
}
__global__ void __launch_bounds__(64) k_selftest(const st_in *in, const u32 *gtable, u32 *slots, u32 *out) {
  const int i = threadIdx.x;
  selftest_lane(in[i], gtable, slots + i * SLOT_WORDS, out + i * ST_WORDS);
}

extern "C" int lamd_selftest(lamd_ctx *ctx, const uint8_t *hash32, const uint8_t *sig64, const uint8_t *pub33, char *report,
                             size_t cap) {
  if (!ctx || !hash32 || !sig64 || !pub33) return LAMD_ERR_ARG;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  std::vector<st_in> in(ST_LANES);
  u64 s = 0x1234567;
  for (int i = 0; i < ST_LANES; i++) {
    for (int j = 0; j < 8; j++) { in[i].a[j] = (u32)splitmix64(s++); in[i].b[j] = (u32)splitmix64(s++); }
    if (i == 1) for (int j = 0; j < 8; j++) in[i].a[j] = 0xFFFFFFFFu;
    memcpy(in[i].hash, hash32, 32); memcpy(in[i].sig, sig64, 64); memcpy(in[i].pub33, pub33, 33);
  }
  st_in *d_in; u32 *d_slots, *d_out;
  HIPCHK(ctx, hipMalloc(&d_in, sizeof(st_in) * ST_LANES));
  HIPCHK(ctx, hipMalloc(&d_slots, (size_t)ST_LANES * SLOT_WORDS * 4));
  HIPCHK(ctx, hipMalloc(&d_out, (size_t)ST_LANES * ST_WORDS * 4));
  HIPCHK(ctx, hipMemcpy(d_in, in.data(), sizeof(st_in) * ST_LANES, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_selftest, dim3(1), dim3(ST_LANES), 0, ctx->stream, d_in, (const u32 *)ctx->gtable, d_slots, d_out);
  HIPCHK(ctx, hipGetLastError());
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  // the host re-runs the whole lane, G additions included: it needs the device's table (a diagnostic: 11 GiB over PCIe is fine)
  std::vector<u32> got((size_t)ST_LANES * ST_WORDS);
  struct host_table {
    u32 *p = (u32 *)malloc(GTABLE_BYTES);
    ~host_table() { free(p); }
    u32 *data() { return p; }
    u32 &operator[](size_t i) { return p[i]; }
  } gt;
  if (!gt.p) { ctx->err = "selftest: no host memory for the G table copy"; return LAMD_ERR_NOMEM; }
  HIPCHK(ctx, hipMemcpy(got.data(), d_out, got.size() * 4, hipMemcpyDeviceToHost));
  HIPCHK(ctx, hipMemcpy(gt.data(), ctx->gtable, GTABLE_BYTES, hipMemcpyDeviceToHost));
  (void)hipFree(d_in); (void)hipFree(d_slots); (void)hipFree(d_out);
  // host evaluation of the same code (host gtable entries recomputed for a sample to check the build kernel)
  static const struct { int lo, hi; const char *name; } stages[] = {
      {0, 8, "fe_mul"}, {8, 16, "fe_sqr"}, {16, 24, "fe_inv"}, {24, 32, "fe_sqrt"}, {32, 40, "fe_lazy_add_neg"}, {40, 48, "sc_mul"},
      {48, 56, "sc_inv"}, {56, 68, "glv_split"}, {68, 85, "parse_pubkey"}, {85, 102, "ecdsa_prep"}, {102, 110, "table_zg"},
      {110, 302, "q_table"}, {302, 327, "ecmult_R"}, {327, 328, "ecdsa_final"}};
  int fails = 0;
  std::string rep;
  std::vector<u32> slot(SLOT_WORDS), exp(ST_WORDS);
  for (int i = 0; i < ST_LANES; i++) {
    selftest_lane(in[i], gt.data(), slot.data(), exp.data());
    for (size_t st = 0; st < sizeof(stages) / sizeof(stages[0]); st++) {
      bool bad = false;
      for (int k = stages[st].lo; k < stages[st].hi; k++) bad |= exp[k] != got[(size_t)i * ST_WORDS + k];
      if (bad) {
        fails |= 1 << st;
        if (rep.size() < 2000) rep += std::string("lane ") + std::to_string(i) + " stage " + stages[st].name + " differs; ";
      }
    }
    if (i == 0) rep += std::string("host verdict lane0=") + std::to_string(exp[327]) + " device=" + std::to_string(got[327]) + "; ";
  }
  // G table: recompute a sample of entries on the host from the bases
  {
    u32 base[16];
    const u32 gx[8] = LAMD_GX, gy[8] = LAMD_GY;
    memcpy(base, gx, 32); memcpy(base + 8, gy, 32);
    const u32 ds[] = {1, 2, 3, 255, 256, 4097, 65535};
    for (u32 d : ds) {
      u32 e[GT_ENTRY_WORDS];
      gtable_compute_entry(e, base, d);
      if (memcmp(e, &gt[(size_t)d * GT_ENTRY_WORDS], GT_ENTRY_WORDS * 4)) { fails |= 1 << 20; rep += "gtable w0 d=" + std::to_string(d) + " differs; "; }
    }
    // window 1 entry 1 must equal 65536*G = window 0 ... (2^16)G: check via doubling
    gej bb = gej_from_ge(ge_from_words(base, base + 8));
    for (int i = 0; i < GTABLE_WINDOW_BITS; i++) bb = gej_double(bb);
    const fe zi = fe_inv(fe_norm_weak(bb.z)); const fe zi2 = fe_sqr(zi);
    u32 e[GT_ENTRY_WORDS];
    slot_store_fe(e, fe_mul(bb.x, zi2));
    slot_store_fe(e + TW, fe_mul(bb.y, fe_mul(zi2, zi)));
    if (memcmp(e, &gt[(((size_t)1 << GTABLE_WINDOW_BITS) + 1) * GT_ENTRY_WORDS], GT_ENTRY_WORDS * 4)) { fails |= 1 << 21; rep += "gtable w1 d=1 differs; "; }
  }
  if (report && cap) { strncpy(report, rep.c_str(), cap - 1); report[cap - 1] = 0; }
  return fails;
}

// ---- op-level chain debugger: device runs a dependent chain of fe_sqr / fe_mul and records raw limbs in and out of
// every step; the host re-executes each step on the device's own inputs and reports the first disagreement.
constexpr int CH_ITERS = 300;
__global__ void __launch_bounds__(64) k_chain_debug(const st_in *in, u32 *out, int use_mul) {
  const int lane = threadIdx.x;
  fe a = fe_from_words(in[lane].a);
  const fe b = fe_from_words(in[lane].b);
  u32 *o = out + (size_t)lane * CH_ITERS * 18;
#pragma unroll 1
  for (int it = 0; it < CH_ITERS; it++) {
    for (int i = 0; i < 9; i++) o[it * 18 + i] = a.n[i];
    const fe r = use_mul ? fe_mul(a, b) : fe_sqr(a);
    for (int i = 0; i < 9; i++) o[it * 18 + 9 + i] = r.n[i];
    a = r;
  }
}
extern "C" int lamd_chain_debug(lamd_ctx *ctx, int use_mul, char *report, size_t cap) {
  if (!ctx) return LAMD_ERR_ARG;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  std::vector<st_in> in(ST_LANES);
  u64 s = 0x7654321;
  for (int i = 0; i < ST_LANES; i++)
    for (int j = 0; j < 8; j++) { in[i].a[j] = (u32)splitmix64(s++); in[i].b[j] = (u32)splitmix64(s++); }
  st_in *d_in; u32 *d_out;
  const size_t words = (size_t)ST_LANES * CH_ITERS * 18;
  HIPCHK(ctx, hipMalloc(&d_in, sizeof(st_in) * ST_LANES));
  HIPCHK(ctx, hipMalloc(&d_out, words * 4));
  HIPCHK(ctx, hipMemcpy(d_in, in.data(), sizeof(st_in) * ST_LANES, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_chain_debug, dim3(1), dim3(ST_LANES), 0, ctx->stream, d_in, d_out, use_mul);
  HIPCHK(ctx, hipGetLastError());
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  std::vector<u32> got(words);
  HIPCHK(ctx, hipMemcpy(got.data(), d_out, words * 4, hipMemcpyDeviceToHost));
  (void)hipFree(d_in); (void)hipFree(d_out);
  int nbad = 0;
  std::string rep;
  char buf[512];
  for (int lane = 0; lane < ST_LANES; lane++) {
    const fe b = fe_from_words(in[lane].b);
    for (int it = 0; it < CH_ITERS; it++) {
      const u32 *o = &got[((size_t)lane * CH_ITERS + it) * 18];
      fe a;
      for (int i = 0; i < 9; i++) a.n[i] = o[i];
      const fe r = use_mul ? fe_mul(a, b) : fe_sqr(a);
      bool bad = false;
      for (int i = 0; i < 9; i++) bad |= r.n[i] != o[9 + i];
      if (bad) {
        if (nbad < 3) {
          snprintf(buf, sizeof buf, "lane %d it %d in=[%x %x %x %x %x %x %x %x %x] dev=[%x %x %x %x %x %x %x %x %x] host=[%x %x %x %x %x %x %x %x %x]; ",
                   lane, it, o[0], o[1], o[2], o[3], o[4], o[5], o[6], o[7], o[8], o[9], o[10], o[11], o[12], o[13], o[14], o[15], o[16], o[17],
                   r.n[0], r.n[1], r.n[2], r.n[3], r.n[4], r.n[5], r.n[6], r.n[7], r.n[8]);
          rep += buf;
          if (use_mul) { snprintf(buf, sizeof buf, "b=[%x %x %x %x %x %x %x %x %x]; ", b.n[0], b.n[1], b.n[2], b.n[3], b.n[4], b.n[5], b.n[6], b.n[7], b.n[8]); rep += buf; }
        }
        nbad++;
      }
    }
  }
  if (report && cap) { strncpy(report, rep.c_str(), cap - 1); report[cap - 1] = 0; }
  return nbad;
}

// ---- inversion-chain debugger: every intermediate of the addition chain, device vs host
constexpr int INV_ITEMS = 24;
LAMD_HD void inv_debug_lane(const u32 aw[8], u32 *o) {
  const fe a = fe_from_words(aw);
  fe items[INV_ITEMS];
  int k = 0;
  const int ns[8] = {1, 2, 3, 5, 11, 22, 44, 88};
  for (int j = 0; j < 8; j++) items[k++] = fe_sqr_n(a, ns[j]);  // 0..7
  const fe x2 = fe_mul(fe_sqr(a), a);
  const fe x3 = fe_mul(fe_sqr(x2), a);
  const fe x6 = fe_mul(fe_sqr_n(x3, 3), x3);
  const fe x9 = fe_mul(fe_sqr_n(x6, 3), x3);
  const fe x11 = fe_mul(fe_sqr_n(x9, 2), x2);
  const fe x22 = fe_mul(fe_sqr_n(x11, 11), x11);
  const fe x44 = fe_mul(fe_sqr_n(x22, 22), x22);
  const fe x88 = fe_mul(fe_sqr_n(x44, 44), x44);
  const fe x176 = fe_mul(fe_sqr_n(x88, 88), x88);
  const fe x220 = fe_mul(fe_sqr_n(x176, 44), x44);
  const fe x223 = fe_mul(fe_sqr_n(x220, 3), x3);
  items[k++] = x2; items[k++] = x3; items[k++] = x6; items[k++] = x9; items[k++] = x11; items[k++] = x22;   // 8..13
  items[k++] = x44; items[k++] = x88; items[k++] = x176; items[k++] = x220; items[k++] = x223;              // 14..18
  const fe_chain ch = fe_pow_chain(a);
  items[k++] = ch.x2; items[k++] = ch.x22; items[k++] = ch.x223;                                             // 19..21
  items[k++] = fe_inv(a);                                                                                     // 22
  items[k++] = fe_sqrt_candidate(a);                                                                          // 23
  for (int j = 0; j < INV_ITEMS; j++) {
    u32 w[8];
    fe_to_words(w, fe_normalize(items[j]));
    for (int i = 0; i < 8; i++) o[j * 8 + i] = w[i];
  }
}
__global__ void __launch_bounds__(64) k_inv_debug(const st_in *in, u32 *out) {
  inv_debug_lane(in[threadIdx.x].a, out + threadIdx.x * INV_ITEMS * 8);
}
extern "C" int lamd_inv_debug(lamd_ctx *ctx, char *report, size_t cap) {
  if (!ctx) return LAMD_ERR_ARG;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  std::vector<st_in> in(ST_LANES);
  u64 s = 0xABCDEF;
  for (int i = 0; i < ST_LANES; i++)
    for (int j = 0; j < 8; j++) in[i].a[j] = (u32)splitmix64(s++);
  st_in *d_in; u32 *d_out;
  const size_t words = (size_t)ST_LANES * INV_ITEMS * 8;
  HIPCHK(ctx, hipMalloc(&d_in, sizeof(st_in) * ST_LANES));
  HIPCHK(ctx, hipMalloc(&d_out, words * 4));
  HIPCHK(ctx, hipMemcpy(d_in, in.data(), sizeof(st_in) * ST_LANES, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_inv_debug, dim3(1), dim3(ST_LANES), 0, ctx->stream, d_in, d_out);
  HIPCHK(ctx, hipGetLastError());
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  std::vector<u32> got(words), exp(INV_ITEMS * 8);
  HIPCHK(ctx, hipMemcpy(got.data(), d_out, words * 4, hipMemcpyDeviceToHost));
  (void)hipFree(d_in); (void)hipFree(d_out);
  int mask = 0;
  std::string rep;
  for (int lane = 0; lane < ST_LANES; lane++) {
    inv_debug_lane(in[lane].a, exp.data());
    for (int j = 0; j < INV_ITEMS; j++)
      if (memcmp(&exp[j * 8], &got[((size_t)lane * INV_ITEMS + j) * 8], 32)) {
        if (!(mask & (1 << j))) {
          rep += "item " + std::to_string(j) + " first bad at lane " + std::to_string(lane) + "; ";
          if (rep.size() < 1500) {
            char buf[400];
            const u32 *g = &got[((size_t)lane * INV_ITEMS + j) * 8], *e = &exp[j * 8], *aw = in[lane].a;
            snprintf(buf, sizeof buf, "a=%08x%08x%08x%08x%08x%08x%08x%08x dev=%08x%08x%08x%08x%08x%08x%08x%08x host=%08x%08x%08x%08x%08x%08x%08x%08x; ",
                     aw[7], aw[6], aw[5], aw[4], aw[3], aw[2], aw[1], aw[0], g[7], g[6], g[5], g[4], g[3], g[2], g[1], g[0], e[7], e[6], e[5], e[4], e[3], e[2], e[1], e[0]);
            rep += buf;
          }
        }
        mask |= 1 << j;
      }
  }
  if (report && cap) { strncpy(report, rep.c_str(), cap - 1); report[cap - 1] = 0; }
  return mask;
}

// ---- x2 debugger: fe_mul(fe_sqr(a), a) in several code shapes, raw limbs out
__global__ void __launch_bounds__(64) k_x2_debug(const st_in *in, u32 *out) {
  const int lane = threadIdx.x;
  u32 *o = out + lane * 64;
  const fe a = fe_from_words(in[lane].a);
  {  // A: fused, raw result
    const fe r = fe_mul(fe_sqr(a), a);
    for (int i = 0; i < 9; i++) o[i] = r.n[i];
  }
  {  // B: intermediate forced through an opaque register barrier
    fe s2 = fe_sqr(a);
#if defined(__HIP_DEVICE_COMPILE__)
    for (int i = 0; i < 9; i++) asm volatile("" : "+v"(s2.n[i]));
#endif
    const fe r = fe_mul(s2, a);
    for (int i = 0; i < 9; i++) o[9 + i] = s2.n[i];
    for (int i = 0; i < 9; i++) o[18 + i] = r.n[i];
  }
  {  // C: fused then normalized
    const fe r = fe_normalize(fe_mul(fe_sqr(a), a));
    for (int i = 0; i < 9; i++) o[27 + i] = r.n[i];
  }
  {  // D: a*a via fe_mul, then *a
    const fe r = fe_mul(fe_mul(a, a), a);
    for (int i = 0; i < 9; i++) o[36 + i] = r.n[i];
  }
}
extern "C" int lamd_x2_debug(lamd_ctx *ctx, char *report, size_t cap) {
  if (!ctx) return LAMD_ERR_ARG;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  std::vector<st_in> in(ST_LANES);
  u64 s = 0xABCDEF;
  for (int i = 0; i < ST_LANES; i++)
    for (int j = 0; j < 8; j++) in[i].a[j] = (u32)splitmix64(s++);
  st_in *d_in; u32 *d_out;
  HIPCHK(ctx, hipMalloc(&d_in, sizeof(st_in) * ST_LANES));
  HIPCHK(ctx, hipMalloc(&d_out, ST_LANES * 64 * 4));
  HIPCHK(ctx, hipMemcpy(d_in, in.data(), sizeof(st_in) * ST_LANES, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_x2_debug, dim3(1), dim3(ST_LANES), 0, ctx->stream, d_in, d_out);
  HIPCHK(ctx, hipGetLastError());
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  std::vector<u32> got(ST_LANES * 64);
  HIPCHK(ctx, hipMemcpy(got.data(), d_out, got.size() * 4, hipMemcpyDeviceToHost));
  (void)hipFree(d_in); (void)hipFree(d_out);
  std::string rep;
  int mask = 0;
  char buf[600];
  for (int lane = 0; lane < ST_LANES; lane++) {
    const fe a = fe_from_words(in[lane].a);
    const fe s2 = fe_sqr(a);
    const fe r = fe_mul(s2, a);
    const fe rn = fe_normalize(r);
    const fe rd = fe_mul(fe_mul(a, a), a);
    const u32 *o = &got[lane * 64];
    const fe *exps[5] = {&r, &s2, &r, &rn, &rd};
    const char *names[5] = {"A_fused_raw", "B_sqr_raw", "B_mul_raw", "C_fused_norm", "D_mulmul"};
    for (int t = 0; t < 5; t++) {
      bool bad = false;
      for (int i = 0; i < 9; i++) bad |= exps[t]->n[i] != o[t * 9 + i];
      if (bad) {
        if (!(mask & (1 << t))) {
          const u32 *g = o + t * 9;
          const u32 *e = exps[t]->n;
          snprintf(buf, sizeof buf, "%s lane %d dev=[%x %x %x %x %x %x %x %x %x] host=[%x %x %x %x %x %x %x %x %x]; ", names[t], lane, g[0], g[1], g[2], g[3], g[4],
                   g[5], g[6], g[7], g[8], e[0], e[1], e[2], e[3], e[4], e[5], e[6], e[7], e[8]);
          rep += buf;
        }
        mask |= 1 << t;
      }
    }
  }
  if (report && cap) { strncpy(report, rep.c_str(), cap - 1); report[cap - 1] = 0; }
  return mask;
}

// ---- randomised arithmetic fuzz (fuzz.h): the same inline function on the device and in this TU's host pass
__global__ void __launch_bounds__(256) k_fuzz_field(size_t lanes, int iters, u64 seed, u64 *__restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < lanes) out[i] = fuzz_lane(seed, i, iters);
}
extern "C" int lamd_fuzz_field(lamd_ctx *ctx, size_t lanes, int iters, uint64_t seed, uint64_t *ops, char *report, size_t cap) {
  if (!ctx || lanes == 0 || lanes > ((size_t)1 << 24) || iters < 1 || iters > 100000) return LAMD_ERR_ARG;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  u64 *d_out = nullptr;
  HIPCHK(ctx, hipMalloc(&d_out, lanes * 8));
  hipLaunchKernelGGL(k_fuzz_field, dim3(blocks_for(lanes)), dim3(256), 0, ctx->stream, lanes, iters, (u64)seed, d_out);
  HIPCHK(ctx, hipGetLastError());
  std::vector<u64> got(lanes);
  HIPCHK(ctx, hipMemcpyAsync(got.data(), d_out, lanes * 8, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  (void)hipFree(d_out);
  int nbad = 0;
  std::string rep;
  for (size_t i = 0; i < lanes; i++) {
    const u64 exp = fuzz_lane(seed, i, iters);
    if (exp != got[i]) {
      if (nbad < 4) rep += "lane " + std::to_string(i) + " device " + std::to_string(got[i]) + " host " + std::to_string(exp) + "; ";
      nbad++;
    }
  }
  if (ops) *ops = (uint64_t)lanes * ((uint64_t)iters * FZ_OPS_PER_ITER + FZ_OPS_TAIL);
  if (report && cap) { strncpy(report, rep.c_str(), cap - 1); report[cap - 1] = 0; }
  return nbad;
}

// ---- diagnostic peek into the engine's device work buffers (tests / debugging only)
// ---- the roofline's denominator, measured in the process that reports it (include/lightning_amd_debug.h lamd_debug_mul32_peak): a dependency-free
// stream of v_mad_u64_u32 -- eight independent 64-bit accumulators per lane, the carry-out in an SGPR pair as the multiplier's columns have it --
// on every SIMD of the chip at a given occupancy, long enough (>= min_ms per launch) for the clock governor to settle where the ecmult kernel's
// launches (3-4 ms) run.  s_memtime / s_memrealtime deltas of one wave give the counter ratio next to it.
#define LAMD_MAD8 "v_mad_u64_u32 %0, s[20:21], %8, %9, %0\n\tv_mad_u64_u32 %1, s[20:21], %8, %9, %1\n\tv_mad_u64_u32 %2, s[20:21], %8, %9, %2\n\t" \
                  "v_mad_u64_u32 %3, s[20:21], %8, %9, %3\n\tv_mad_u64_u32 %4, s[20:21], %8, %9, %4\n\tv_mad_u64_u32 %5, s[20:21], %8, %9, %5\n\t" \
                  "v_mad_u64_u32 %6, s[20:21], %8, %9, %6\n\tv_mad_u64_u32 %7, s[20:21], %8, %9, %7\n\t"
__global__ void __launch_bounds__(256) k_mul32_peak(u32 *__restrict__ out, u32 iters, u64 *__restrict__ clocks) {
  const u32 x = threadIdx.x * 2654435761u + 12345u + blockIdx.x, y = x ^ 0x9E3779B9u;
  u64 a0 = x, a1 = y, a2 = x + 1, a3 = y + 1, a4 = x + 2, a5 = y + 2, a6 = x + 3, a7 = y + 3;
  const u64 c0 = __builtin_readcyclecounter(), r0 = wall_clock64();
  for (u32 it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 16; r++)
      asm volatile(LAMD_MAD8 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y) : "s20", "s21");
  }
  const u64 c1 = __builtin_readcyclecounter(), r1 = wall_clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = (u32)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
  if (blockIdx.x == 0 && threadIdx.x == 0) { clocks[0] = c1 - c0; clocks[1] = r1 - r0; }
}
extern "C" int lamd_debug_mul32_peak(lamd_ctx *ctx, int waves_per_simd, double min_ms, int launches, double *lane_ops_per_s, double *avg_launch_ms,
                                     double *memtime_per_realtime) {
  if (!ctx || waves_per_simd < 1 || waves_per_simd > 8 || launches < 1 || !lane_ops_per_s) return LAMD_ERR_ARG;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const unsigned blocks = (unsigned)ctx->prop.multiProcessorCount * (unsigned)waves_per_simd;  // 256 threads = one wave per SIMD of a CU
  u32 *d_out = nullptr;
  u64 *d_clk = nullptr;
  HIPCHK(ctx, hipMalloc((void **)&d_out, (size_t)blocks * 256 * 4));
  if (hipMalloc((void **)&d_clk, 16) != hipSuccess) { (void)hipFree(d_out); ctx->err = "hipMalloc failed"; return LAMD_ERR_NOMEM; }
  hipEvent_t e0 = nullptr, e1 = nullptr;
  int rc = LAMD_OK;
  auto run = [&](u32 iters, float *ms) -> bool {
    if (hipEventRecord(e0, ctx->stream) != hipSuccess) return false;
    hipLaunchKernelGGL(k_mul32_peak, dim3(blocks), dim3(256), 0, ctx->stream, d_out, iters, d_clk);
    return hipEventRecord(e1, ctx->stream) == hipSuccess && hipEventSynchronize(e1) == hipSuccess && hipEventElapsedTime(ms, e0, e1) == hipSuccess;
  };
  float ms = 0;
  u32 iters = 2000;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess || !run(200, &ms) || !run(iters, &ms)) rc = LAMD_ERR_HIP;
  if (rc == LAMD_OK) {
    if (min_ms <= 0) iters = 150;  // the sub-millisecond launch of the round-1 micro-benchmark, for comparison
    else if (ms > 0) iters = (u32)((double)iters * min_ms / ms * 1.05) + 1;
    double sum = 0, ratio = 0;
    for (int l = 0; l < launches && rc == LAMD_OK; l++) {
      if (!run(iters, &ms)) { rc = LAMD_ERR_HIP; break; }
      sum += ms;
      u64 clk[2] = {0, 0};
      if (hipMemcpy(clk, d_clk, 16, hipMemcpyDeviceToHost) == hipSuccess && clk[1]) ratio += (double)clk[0] / (double)clk[1];
    }
    if (rc == LAMD_OK) {
      const double avg = sum / launches;
      *lane_ops_per_s = (double)blocks * 256.0 * (double)iters * 128.0 / (avg * 1e-3);
      if (avg_launch_ms) *avg_launch_ms = avg;
      if (memtime_per_realtime) *memtime_per_realtime = ratio / launches;
    }
  }
  if (rc != LAMD_OK) ctx->err = "lamd_debug_mul32_peak: HIP error";
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  (void)hipFree(d_out);
  (void)hipFree(d_clk);
  return rc;
}
extern "C" const void *lamd_debug_gtable(lamd_ctx *ctx) { return ctx ? (const void *)ctx->gtable : nullptr; }
extern "C" int lamd_debug_read(lamd_ctx *ctx, int which, size_t offset, size_t nbytes, void *out) {
  if (!ctx || !out) return LAMD_ERR_ARG;
  devbuf *bufs[] = {&ctx->recs, &ctx->keyok, &ctx->kd_rep, &ctx->kd_uid, &ctx->row_ent, &ctx->kd_uniq, &ctx->hk7_keyok, &ctx->hk7_qwords, &ctx->cache_store.pool7};
  if (which < 0 || which >= (int)(sizeof(bufs) / sizeof(bufs[0])) || !bufs[which]->p || offset + nbytes > bufs[which]->cap) {
    ctx->err = "debug_read: no such buffer / out of range";
    return LAMD_ERR_ARG;
  }
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  HIPCHK(ctx, hipMemcpy(out, (const u8 *)bufs[which]->p + offset, nbytes, hipMemcpyDeviceToHost));
  return LAMD_OK;
}

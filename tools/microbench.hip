// gfx950 integer-VALU microbenchmark: measures the instruction rates the secp256k1 engine is
// bounded by (P_mul32 must be MEASURED) and the throughput of candidate
// 256-bit modular-multiply formulations.  Build: hipcc --offload-arch=gfx950 -O3 tools/microbench.hip -o tools/microbench
// Output: one line per (op, waves/SIMD): shader cycles per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
typedef uint32_t u32; typedef uint64_t u64;
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %s:%d\n",hipGetErrorString(e),__FILE__,__LINE__); exit(1);} }while(0)

// ---------------------------------------------------------------- raw instruction streams
// 16 independent destination registers, 32 instructions per asm block, REP blocks per loop iteration
#define OPS_PER_BLOCK 32
#define I8(s) s(0) s(1) s(2) s(3) s(4) s(5) s(6) s(7)
#define DEF_OP(NAME, ASMLINE64, ASMLINE32, IS64)                                           \
__global__ void __launch_bounds__(256) op_##NAME(u32 *out, int iters, u64 *cyc) {          \
  u32 x = threadIdx.x * 2654435761u + 12345u, y = x ^ 0x9E3779B9u;                          \
  u64 a0 = x, a1 = y, a2 = x + 1, a3 = y + 1, a4 = x + 2, a5 = y + 2, a6 = x + 3, a7 = y + 3; \
  u32 b0 = x, b1 = y, b2 = x + 1, b3 = y + 1, b4 = x + 2, b5 = y + 2, b6 = x + 3, b7 = y + 3; \
  u64 t0 = __builtin_readcyclecounter();                                                    \
  for (int it = 0; it < iters; it++) {                                                      \
    _Pragma("unroll") for (int r = 0; r < 4; r++) {                                         \
      if (IS64) asm volatile(ASMLINE64 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y) : "vcc"); \
      else asm volatile(ASMLINE32 : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7) : "v"(x), "v"(y) : "vcc"); \
    }                                                                                       \
  }                                                                                         \
  u64 t1 = __builtin_readcyclecounter();                                                    \
  out[blockIdx.x * blockDim.x + threadIdx.x] = (u32)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) ^ b0 ^ b1 ^ b2 ^ b3 ^ b4 ^ b5 ^ b6 ^ b7; \
  if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0; \
}
#define L8(fmt) fmt(0) fmt(1) fmt(2) fmt(3) fmt(4) fmt(5) fmt(6) fmt(7)
// each line macro takes the operand index n
#define MAD64(n) "v_mad_u64_u32 %" #n ", vcc, %8, %9, %" #n "\n\t"
#define MAD64S(n) "v_mad_u64_u32 %" #n ", s[20:21], %8, %9, %" #n "\n\t"
#define LSHLADD64(n) "v_lshl_add_u64 %" #n ", %" #n ", 0, %" #n "\n\t"
#define FMA64(n) "v_fma_f64 %" #n ", %" #n ", %" #n ", %" #n "\n\t"
#define MULLO(n) "v_mul_lo_u32 %" #n ", %8, %" #n "\n\t"
#define MULHI(n) "v_mul_hi_u32 %" #n ", %8, %" #n "\n\t"
#define MUL24(n) "v_mul_u32_u24 %" #n ", %8, %" #n "\n\t"
#define MULHI24(n) "v_mul_hi_u32_u24 %" #n ", %8, %" #n "\n\t"
#define MAD24(n) "v_mad_u32_u24 %" #n ", %8, %9, %" #n "\n\t"
#define ADD32(n) "v_add_u32 %" #n ", %8, %" #n "\n\t"
#define ADD3(n) "v_add3_u32 %" #n ", %8, %9, %" #n "\n\t"
#define ADDCO(n) "v_add_co_u32 %" #n ", vcc, %8, %" #n "\n\t"
#define ADDCNOP(n) "v_addc_co_u32 %" #n ", vcc, %8, %" #n ", vcc\n\ts_nop 1\n\t"
#define XOR32(n) "v_xor_b32 %" #n ", %8, %" #n "\n\t"
#define ALIGNBIT(n) "v_alignbit_b32 %" #n ", %" #n ", %" #n ", 7\n\t"
#define MOV32(n) "v_mov_b32 %" #n ", %8\n\t"
#define FMA32(n) "v_fma_f32 %" #n ", %8, %9, %" #n "\n\t"
#define MADNOPADDC(n) "v_mad_u64_u32 %" #n ", vcc, %8, %9, %" #n "\n\ts_nop 1\n\tv_addc_co_u32 %8, vcc, 0, %8, vcc\n\t"
#define X4(m) L8(m) L8(m) L8(m) L8(m)
DEF_OP(mad_u64_u32, X4(MAD64), "", 1)
DEF_OP(mad_u64_u32_sgprcarry, X4(MAD64S), "", 1)
DEF_OP(lshl_add_u64, X4(LSHLADD64), "", 1)
DEF_OP(fma_f64, X4(FMA64), "", 1)
DEF_OP(mul_lo_u32, "", X4(MULLO), 0)
DEF_OP(mul_hi_u32, "", X4(MULHI), 0)
DEF_OP(mul_u32_u24, "", X4(MUL24), 0)
DEF_OP(mul_hi_u32_u24, "", X4(MULHI24), 0)
DEF_OP(mad_u32_u24, "", X4(MAD24), 0)
DEF_OP(add_u32, "", X4(ADD32), 0)
DEF_OP(add3_u32, "", X4(ADD3), 0)
DEF_OP(add_co_u32, "", X4(ADDCO), 0)
DEF_OP(addc_co_nop1, "", X4(ADDCNOP), 0)
DEF_OP(xor_b32, "", X4(XOR32), 0)
DEF_OP(alignbit_b32, "", X4(ALIGNBIT), 0)
DEF_OP(mov_b32, "", X4(MOV32), 0)
DEF_OP(fma_f32, "", X4(FMA32), 0)

typedef void (*opk)(u32 *, int, u64 *);
struct OpEnt { const char *name; opk fn; int per_block; };

// ---------------------------------------------------------------- candidate 256-bit mulmod formulations
// A: saturated 8x32, operand scanning, compiler-scheduled
struct MulA {
  static constexpr int NL = 8;
  __device__ static __forceinline__ void mul(u32 r[8], const u32 a[8], const u32 b[8]) {
    u32 t[16];
#pragma unroll
    for (int i = 0; i < 16; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { u32 carry = 0;
#pragma unroll
      for (int j = 0; j < 8; j++) { u64 v = (u64)a[i] * b[j] + t[i + j] + carry; t[i + j] = (u32)v; carry = (u32)(v >> 32); }
      t[i + 8] = carry; }
    reduce(r, t);
  }
  __device__ static __forceinline__ void reduce(u32 r[8], const u32 t[16]) {
    // lo + hi*(2^32+977)
    u32 m[10]; u32 c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { u64 v = (u64)t[8 + i] * 977u + c; m[i] = (u32)v; c = (u32)(v >> 32); }
    m[8] = c; m[9] = 0;
    // m += hi << 32
    u32 cc = 0;
#pragma unroll
    for (int i = 1; i < 9; i++) { u32 co; m[i] = __builtin_addc(m[i], t[7 + i], cc, &co); cc = co; }
    m[9] = cc;
    // s = lo + m[0..7]; overflow limbs e = m[8], m[9] + carry
    u32 s[8]; cc = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { u32 co; s[i] = __builtin_addc(t[i], m[i], cc, &co); cc = co; }
    u64 e = (u64)m[8] + ((u64)m[9] << 32) + cc;   // < 2^34
    // fold e*(2^32+977)
    u64 f = e * 977u;                              // < 2^44
    u32 f0 = (u32)f, f1 = (u32)(f >> 32);
    u64 g = (u64)f1 + (e & 0xffffffffu); u32 g1 = (u32)g; u32 g2 = (u32)(g >> 32) + (u32)(e >> 32);
    u32 add[8] = {f0, g1, g2, 0, 0, 0, 0, 0};
    cc = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { u32 co; r[i] = __builtin_addc(s[i], add[i], cc, &co); cc = co; }
    // final carry: + (2^32+977)
    u32 k0 = cc ? 977u : 0u, k1 = cc;
    cc = 0; u32 co;
    r[0] = __builtin_addc(r[0], k0, 0u, &co); cc = co;
    r[1] = __builtin_addc(r[1], k1, cc, &co); cc = co;
#pragma unroll
    for (int i = 2; i < 8; i++) { r[i] = __builtin_addc(r[i], 0u, cc, &co); cc = co; }
  }
};
// C: saturated 8x32, product scanning with inline-asm mad + s_nop + addc
struct MulC {
  static constexpr int NL = 8;
  __device__ static __forceinline__ void mac(u64 &acc, u32 &acc2, u32 a, u32 b) {
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\ts_nop 1\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(acc), "+v"(acc2) : "v"(a), "v"(b) : "vcc");
  }
  __device__ static __forceinline__ void mul(u32 r[8], const u32 a[8], const u32 b[8]) {
    u32 t[16]; u64 acc = 0; u32 acc2 = 0;
#pragma unroll
    for (int k = 0; k < 15; k++) {
#pragma unroll
      for (int i = 0; i < 8; i++) { int j = k - i; if (j < 0 || j > 7) continue; mac(acc, acc2, a[i], b[j]); }
      t[k] = (u32)acc; acc = (acc >> 32) | ((u64)acc2 << 32); acc2 = 0; }
    t[15] = (u32)acc;
    MulA::reduce(r, t);
  }
};
// L: lazy 10x26 (no carries during accumulation)
struct MulL {
  static constexpr int NL = 10;
  __device__ static __forceinline__ void mul(u32 r[10], const u32 a[10], const u32 b[10]) {
    const u32 M = 0x3FFFFFFu;
    u64 c[20];
#pragma unroll
    for (int k = 0; k < 19; k++) { u64 s = 0;
#pragma unroll
      for (int i = 0; i < 10; i++) { int j = k - i; if (j < 0 || j > 9) continue; s += (u64)a[i] * b[j]; }
      c[k] = s; }
    c[19] = 0;
#pragma unroll
    for (int k = 10; k < 19; k++) { c[k + 1] += c[k] >> 26; c[k] &= M; }
    c[10] += c[19] << 10; c[9] += c[19] * 0x3D10u;
#pragma unroll
    for (int k = 10; k < 19; k++) { c[k - 10] += c[k] * 0x3D10u; c[k - 9] += c[k] << 10; }
#pragma unroll
    for (int k = 0; k < 9; k++) { c[k + 1] += c[k] >> 26; c[k] &= M; }
    u64 e = c[9] >> 22; c[9] &= 0x3FFFFFu;
    c[0] += e * 977u; c[1] += e << 6;
    c[1] += c[0] >> 26; c[0] &= M; c[2] += c[1] >> 26; c[1] &= M;
#pragma unroll
    for (int k = 0; k < 10; k++) r[k] = (u32)c[k];
  }
};

template <class V> __global__ void __launch_bounds__(256) mulk(const u32 *in, u32 *out, int iters, u64 *cyc) {
  constexpr int NL = V::NL;
  int tid = blockIdx.x * blockDim.x + threadIdx.x, nt = gridDim.x * blockDim.x;
  u32 a[NL], b[NL], r[NL];
#pragma unroll
  for (int i = 0; i < NL; i++) { a[i] = in[i * nt + tid]; b[i] = in[(NL + i) * nt + tid]; }
  u64 t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
    V::mul(r, a, b);
#pragma unroll
    for (int i = 0; i < NL; i++) { a[i] = b[i]; b[i] = r[i]; }
  }
  u64 t1 = __builtin_readcyclecounter();
#pragma unroll
  for (int i = 0; i < NL; i++) out[i * nt + tid] = b[i];
  if ((threadIdx.x & 63) == 0) cyc[tid >> 6] = t1 - t0;
}

// host reference for mulmod p
typedef unsigned __int128 u128;
static void host_mulmod(u64 r[4], const u64 a[4], const u64 b[4]) {
  u64 t[8] = {0};
  for (int i = 0; i < 4; i++) { u64 c = 0; for (int j = 0; j < 4; j++) { u128 v = (u128)a[i] * b[j] + t[i + j] + c; t[i + j] = (u64)v; c = (u64)(v >> 64); } t[i + 4] = c; }
  const u64 PC = 0x1000003D1ULL; u128 acc = 0; u64 lo[4];
  for (int i = 0; i < 4; i++) { acc += (u128)t[4 + i] * PC + t[i]; lo[i] = (u64)acc; acc >>= 64; }
  u64 c = (u64)acc; acc = (u128)c * PC + lo[0]; r[0] = (u64)acc; acc >>= 64;
  for (int i = 1; i < 4; i++) { acc += lo[i]; r[i] = (u64)acc; acc >>= 64; }
  const u64 P[4] = {0xFFFFFFFEFFFFFC2FULL, ~0ULL, ~0ULL, ~0ULL};
  for (int rep = 0; rep < 2; rep++) {
    bool ge = (u64)acc != 0; if (!ge) { ge = true; for (int i = 3; i >= 0; i--) { if (r[i] < P[i]) { ge = false; break; } if (r[i] > P[i]) break; } }
    if (ge) { u64 bw = 0; for (int i = 0; i < 4; i++) { u128 v = (u128)r[i] - P[i] - bw; r[i] = (u64)v; bw = (u64)(v >> 64) & 1; } acc = 0; }
  }
}
static void canon(u64 r[4]) { const u64 P[4] = {0xFFFFFFFEFFFFFC2FULL, ~0ULL, ~0ULL, ~0ULL}; bool ge = true; for (int i = 3; i >= 0; i--) { if (r[i] < P[i]) { ge = false; break; } if (r[i] > P[i]) break; } if (ge) { u64 bw = 0; for (int i = 0; i < 4; i++) { u128 v = (u128)r[i] - P[i] - bw; r[i] = (u64)v; bw = (u64)(v >> 64) & 1; } } }

template <class V> static void run_mul(const char *name, int cus, double ghz_hint) {
  constexpr int NL = V::NL;
  for (int wps = 1; wps <= 8; wps *= 2) {
    int blocks = cus * wps, nt = blocks * 256, iters = 2000;
    std::vector<u32> hin((size_t)2 * NL * nt), hout((size_t)NL * nt);
    std::vector<u64> av((size_t)nt * 4), bv((size_t)nt * 4);
    u64 s = 88172645463325252ULL;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    for (int t = 0; t < nt; t++) {
      for (int i = 0; i < 4; i++) { av[(size_t)t * 4 + i] = rnd(); bv[(size_t)t * 4 + i] = rnd(); }
      if (NL == 8) for (int i = 0; i < 8; i++) { hin[(size_t)i * nt + t] = (u32)(av[(size_t)t * 4 + i / 2] >> (32 * (i & 1))); hin[(size_t)(8 + i) * nt + t] = (u32)(bv[(size_t)t * 4 + i / 2] >> (32 * (i & 1))); }
      else for (int i = 0; i < 10; i++) { auto get = [&](const u64 *v, int bit) { u128 w = 0; int li = bit / 64; w = v[li]; if (li + 1 < 4) w |= (u128)v[li + 1] << 64; return (u32)((w >> (bit % 64)) & 0x3FFFFFFu); };
        hin[(size_t)i * nt + t] = get(&av[(size_t)t * 4], 26 * i); hin[(size_t)(10 + i) * nt + t] = get(&bv[(size_t)t * 4], 26 * i); }
    }
    u32 *din, *dout; u64 *dcyc;
    CK(hipMalloc(&din, hin.size() * 4)); CK(hipMalloc(&dout, hout.size() * 4)); CK(hipMalloc(&dcyc, (size_t)(nt / 64) * 8));
    CK(hipMemcpy(din, hin.data(), hin.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    mulk<V><<<blocks, 256>>>(din, dout, 10, dcyc); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); mulk<V><<<blocks, 256>>>(din, dout, iters, dcyc); CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<u64> hc(nt / 64); CK(hipMemcpy(hc.data(), dcyc, hc.size() * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hout.data(), dout, hout.size() * 4, hipMemcpyDeviceToHost));
    double avg = 0; for (auto v : hc) avg += (double)v; avg /= hc.size();
    // verify a few threads
    int bad = 0;
    for (int t = 0; t < nt; t += nt / 16 + 1) {
      u64 a[4], b[4], r[4]; memcpy(a, &av[(size_t)t * 4], 32); memcpy(b, &bv[(size_t)t * 4], 32);
      for (int it = 0; it < iters; it++) { host_mulmod(r, a, b); memcpy(a, b, 32); memcpy(b, r, 32); }
      u64 g[4] = {0, 0, 0, 0};
      if (NL == 8) for (int i = 0; i < 8; i++) g[i / 2] |= (u64)hout[(size_t)i * nt + t] << (32 * (i & 1));
      else { u128 carry = 0; u64 w[5] = {0}; (void)carry; for (int i = 0; i < 10; i++) { u128 v = (u128)hout[(size_t)i * nt + t] << ((26 * i) % 64); int li = (26 * i) / 64; u128 lo = (u128)w[li] + (u64)v; w[li] = (u64)lo; u128 hi = (u128)w[li + 1] + (u64)(v >> 64) + (u64)(lo >> 64); w[li + 1] = (u64)hi; if (li + 2 < 5) w[li + 2] += (u64)(hi >> 64); }
        // w may exceed 256 bits slightly: fold
        u128 f = (u128)w[4] * 0x1000003D1ULL + w[0]; w[0] = (u64)f; f >>= 64; for (int i = 1; i < 4; i++) { f += w[i]; w[i] = (u64)f; f >>= 64; } memcpy(g, w, 32); }
      canon(g); canon(b);
      if (memcmp(g, b, 32)) bad++;
    }
    double wall_mulps = (double)nt * iters / (ms * 1e-3);
    printf("MUL %-10s waves/SIMD=%d  cyc/mul/wave=%.1f  => cyc per mul per SIMD=%.1f  wall: %.3e mul/s (%.2f ms)  verify_bad=%d\n", name, wps, avg / iters, avg / iters / wps, wall_mulps, ms, bad);
    CK(hipFree(din)); CK(hipFree(dout)); CK(hipFree(dcyc));
  }
}

int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount;
  printf("device %s CUs=%d clock=%d kHz\n", prop.gcnArchName, cus, prop.clockRate);
  OpEnt ops[] = {
    {"v_mad_u64_u32(vcc)", op_mad_u64_u32, 128}, {"v_mad_u64_u32(sgpr)", op_mad_u64_u32_sgprcarry, 128}, {"v_lshl_add_u64", op_lshl_add_u64, 128}, {"v_fma_f64", op_fma_f64, 128},
    {"v_mul_lo_u32", op_mul_lo_u32, 128}, {"v_mul_hi_u32", op_mul_hi_u32, 128}, {"v_mul_u32_u24", op_mul_u32_u24, 128}, {"v_mul_hi_u32_u24", op_mul_hi_u32_u24, 128},
    {"v_mad_u32_u24", op_mad_u32_u24, 128}, {"v_add_u32", op_add_u32, 128}, {"v_add3_u32", op_add3_u32, 128}, {"v_add_co_u32", op_add_co_u32, 128},
    {"v_addc_co+s_nop1", op_addc_co_nop1, 128}, {"v_xor_b32", op_xor_b32, 128}, {"v_alignbit_b32", op_alignbit_b32, 128}, {"v_mov_b32", op_mov_b32, 128}, {"v_fma_f32", op_fma_f32, 128},
  };
  u32 *dout; u64 *dcyc; CK(hipMalloc(&dout, (size_t)cus * 8 * 256 * 4)); CK(hipMalloc(&dcyc, (size_t)cus * 8 * 4 * 8));
  for (auto &o : ops) {
    for (int wps = 1; wps <= 8; wps *= 2) {
      int blocks = cus * wps, iters = 500;
      o.fn<<<blocks, 256>>>(dout, 10, dcyc); CK(hipDeviceSynchronize());
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      CK(hipEventRecord(e0)); o.fn<<<blocks, 256>>>(dout, iters, dcyc); CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      std::vector<u64> hc((size_t)blocks * 4); CK(hipMemcpy(hc.data(), dcyc, hc.size() * 8, hipMemcpyDeviceToHost));
      double avg = 0; for (auto v : hc) avg += (double)v; avg /= hc.size();
      double n = (double)iters * o.per_block;
      double lane_ops = (double)blocks * 256 * n / (ms * 1e-3);
      printf("OP %-22s waves/SIMD=%d  cyc/instr/wave=%.2f  cyc/instr/SIMD=%.2f  chip: %.3e lane-ops/s\n", o.name, wps, avg / n, avg / n / wps, lane_ops);
    }
  }
  run_mul<MulA>("A-rowwise", cus, 2.4);
  run_mul<MulC>("C-comba-asm", cus, 2.4);
  run_mul<MulL>("L-10x26", cus, 2.4);
  return 0;
}

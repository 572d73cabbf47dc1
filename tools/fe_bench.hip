// Field-multiplier micro-benchmark on the real headers: chains of fe_mul / fe_sqr at a chosen occupancy, plus the issue rates of
// the auxiliary instructions the multiplier's carry handling is built from.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-codegenprepare-mul24=false [-DLAMD_FE_ASM_INC='"variants/x.inc"'] tools/fe_bench.hip -o tools/fe_bench_x
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#include "../lightning_amd/csrc/group.h"
using namespace lamd;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int MODE>
__global__ void __launch_bounds__(256) k_chain(u32 *out, int iters) {
  const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  fe a, b;
  for (int i = 0; i < 9; i++) { a.n[i] = (t * 2654435761u + i * 40503u) & 0x0FFFFFFFu; b.n[i] = (t * 40503u + i * 2654435761u) & 0x0FFFFFFFu; }
  a.n[8] &= 0xFFFFFF; b.n[8] &= 0xFFFFFF;
#pragma unroll 1
  for (int it = 0; it < iters; it++) {
    if (MODE == 0) { const fe c = fe_mul(a, b); a = b; b = c; const fe d = fe_mul(a, b); a = b; b = d; }
    else if (MODE == 1) { a = fe_sqr(a); a = fe_sqr(a); }
    else if (MODE == 2) {  // the table-driven kernels' mixed addition (group.h gej_add_ge_fast): 8 multiplications + 3 squarings
      gej P;
      P.x = a; P.y = b; P.z = a; P.inf = false;
      ge Q;
      Q.x = b; Q.y = a;
      P = gej_add_ge_fast(P, Q);
      a = P.x; b = fe_norm_weak(fe_add(P.y, P.z));
    } else {               // doubling: 3 multiplications + 4 squarings
      gej P;
      P.x = a; P.y = b; P.z = a; P.inf = false;
      P = gej_double(P);
      a = P.x; b = fe_norm_weak(fe_add(P.y, P.z));
    }
  }
  u32 acc = 0;
  for (int i = 0; i < 9; i++) acc ^= a.n[i] ^ b.n[i];
  out[t] = acc;
}

#define OPK(NAME, LINE) OPKT(NAME, LINE, u64)
#define OPK32(NAME, LINE) OPKT(NAME, LINE, u32)
#define OPKT(NAME, LINE, TY)                                                                        \
  __global__ void __launch_bounds__(256) op_##NAME(u32 *out, int iters) {                           \
    u32 x = threadIdx.x * 2654435761u + 12345u, y = x ^ 0x9E3779B9u;                                \
    TY a0 = x, a1 = y, a2 = x + 1, a3 = y + 1, a4 = x + 2, a5 = y + 2, a6 = x + 3, a7 = y + 3;     \
    const u32 m29 = 0x1fffffffu;                                                                    \
    for (int it = 0; it < iters; it++) {                                                            \
      _Pragma("unroll") for (int r = 0; r < 4; r++)                                                 \
        asm volatile(LINE(0) LINE(1) LINE(2) LINE(3) LINE(4) LINE(5) LINE(6) LINE(7) LINE(0) LINE(1) LINE(2) LINE(3) LINE(4) LINE(5) LINE(6) LINE(7) \
                     LINE(0) LINE(1) LINE(2) LINE(3) LINE(4) LINE(5) LINE(6) LINE(7) LINE(0) LINE(1) LINE(2) LINE(3) LINE(4) LINE(5) LINE(6) LINE(7) \
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y), "s"(m29) : "vcc", "s20", "s21", "s22", "s23"); \
    }                                                                                               \
    out[blockIdx.x * blockDim.x + threadIdx.x] = (u32)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);     \
  }
#define L_MADV(n) "v_mad_u64_u32 %" #n ", vcc, %8, %9, %" #n "\n\t"
#define L_MADS(n) "v_mad_u64_u32 %" #n ", s[20:21], %8, %9, %" #n "\n\t"
#define L_MADI(n) "v_mad_i64_i32 %" #n ", s[20:21], %8, %9, %" #n "\n\t"
#define L_SHR64(n) "v_lshrrev_b64 %" #n ", 29, %" #n "\n\t"
#define L_ASHR64(n) "v_ashrrev_i64 %" #n ", 29, %" #n "\n\t"
#define L_ANDLIT(n) "v_and_b32 %" #n ", 0x1fffffff, %" #n "\n\t"
#define L_ANDSG(n) "v_and_b32 %" #n ", %10, %" #n "\n\t"
#define L_SHR32(n) "v_lshrrev_b32 %" #n ", 29, %" #n "\n\t"
#define L_BFE(n) "v_bfe_u32 %" #n ", %" #n ", 0, 29\n\t"
#define L_ADD(n) "v_add_u32 %" #n ", %8, %" #n "\n\t"
#define L_SUB(n) "v_sub_u32 %" #n ", %8, %" #n "\n\t"
#define L_LSHLADD(n) "v_lshl_add_u32 %" #n ", %8, 3, %" #n "\n\t"
#define L_ADD3(n) "v_add3_u32 %" #n ", %8, %9, %" #n "\n\t"
#define L_ANDOR(n) "v_and_or_b32 %" #n ", %8, %10, %" #n "\n\t"
#define L_CNDMASK(n) "v_cndmask_b32 %" #n ", %8, %" #n ", vcc\n\t"
#define L_ALIGN(n) "v_alignbit_b32 %" #n ", %8, %" #n ", 29\n\t"
OPK(madv, L_MADV) OPK(mads, L_MADS) OPK(madi, L_MADI) OPK(shr64, L_SHR64) OPK(ashr64, L_ASHR64) OPK32(andlit, L_ANDLIT) OPK32(andsg, L_ANDSG)
OPK32(shr32, L_SHR32) OPK32(bfe, L_BFE) OPK32(add, L_ADD) OPK32(sub, L_SUB) OPK32(lshladd, L_LSHLADD) OPK32(add3, L_ADD3) OPK32(andor, L_ANDOR)
OPK32(cndmask, L_CNDMASK) OPK32(align, L_ALIGN)

typedef void (*kfn)(u32 *, int);
static double run(kfn f, int blocks, int iters, u32 *dout) {
  f<<<blocks, 256>>>(dout, 4); CK(hipDeviceSynchronize());
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  double best = 1e30;
  for (int rep = 0; rep < 3; rep++) {
    CK(hipEventRecord(e0)); f<<<blocks, 256>>>(dout, iters); CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  return best * 1e-3;
}
int main(int argc, char **argv) {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  u32 *dout; CK(hipMalloc(&dout, (size_t)cus * 8 * 256 * 4));
  const bool ops = argc > 1 && argv[1][0] == 'o';
  const int W[] = {1, 2, 3, 4, 5, 6, 8};
  if (ops) {
    struct { const char *name; kfn f; } O[] = {{"v_mad_u64_u32 vcc", op_madv}, {"v_mad_u64_u32 sgpr", op_mads}, {"v_mad_i64_i32 sgpr", op_madi}, {"v_lshrrev_b64", op_shr64},
      {"v_ashrrev_i64", op_ashr64}, {"v_and_b32 literal", op_andlit}, {"v_and_b32 sgpr", op_andsg}, {"v_lshrrev_b32", op_shr32}, {"v_bfe_u32", op_bfe}, {"v_add_u32", op_add},
      {"v_sub_u32", op_sub}, {"v_lshl_add_u32", op_lshladd}, {"v_add3_u32", op_add3}, {"v_and_or_b32", op_andor}, {"v_cndmask_b32", op_cndmask}, {"v_alignbit_b32", op_align}};
    for (auto &o : O) {
      printf("OP %-20s", o.name);
      for (int w : W) { const int iters = 400; const double s = run(o.f, cus * w, iters, dout); printf("  w%d %.3e", w, (double)cus * w * 256 * iters * 128 / s); }
      printf("  lane-ops/s\n");
    }
    return 0;
  }
  const char *names[] = {"mul-chain", "sqr-chain", "madd", "double"};
  kfn fs[] = {k_chain<0>, k_chain<1>, k_chain<2>, k_chain<3>};
  const double per_iter[] = {2, 2, 1, 1};
  for (int m = 0; m < 4; m++) {
    printf("%-10s", names[m]);
    for (int w : W) { const int iters = m >= 2 ? 300 : 1500; const double s = run(fs[m], cus * w, iters, dout); printf("  w%d %.3e", w, (double)cus * w * 256 * iters * per_iter[m] / s); }
    printf(m >= 2 ? "  group operations/s\n" : "  mul(+sqr)/s\n");
  }
  return 0;
}

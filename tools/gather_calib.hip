// FETCH_SIZE calibration for the dominant kernel's access pattern (VERDICT r03 "Next" 1b; MI355X_MICROARCH.md "HBM": the counter is
// calibrated for wide streaming reads only -- "calibrate on a known byte count in your own access pattern").
// k_ecmult_keyed reads, per row, 12 random 64-byte G-table entries (64-byte aligned, 3 GiB table) and 37 random 96-byte comb entries
// (96-byte stride: two of three straddle a 128-byte line).  This tool issues exactly such gathers with a KNOWN request count over a table
// far larger than L2 + Infinity Cache:
//     gather_calib <entry_bytes 64|96> [table_GiB=3] [threads=2^22] [reads_per_thread=16]
// prints the bytes it asked for.  Run under `rocprofv3 --kernel-trace --pmc FETCH_SIZE`; factor = requested bytes / (FETCH_SIZE * 1024).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>

template <int E> __global__ void __launch_bounds__(256) k_gather(const uint4 *__restrict__ tab, size_t entries, int reads, uint32_t *out) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t x = (t + 1) * 0x9E3779B97F4A7C15ull;
  uint32_t acc = 0;
  for (int r = 0; r < reads; r++) {
    x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
    const uint4 *e = reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(tab) + (x % entries) * E);
#pragma unroll
    for (int k = 0; k < E / 16; k++) { const uint4 v = e[k]; acc += v.x ^ v.y ^ v.z ^ v.w; }
  }
  out[t] = acc;
}
__global__ void k_fill(uint4 *p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_uint4((uint32_t)i, 1, 2, 3);
}
int main(int argc, char **argv) {
  const int E = argc > 1 ? atoi(argv[1]) : 64;
  const double gib = argc > 2 ? atof(argv[2]) : 3.0;
  const size_t threads = argc > 3 ? strtoull(argv[3], 0, 0) : (1u << 22);
  const int reads = argc > 4 ? atoi(argv[4]) : 16;
  const size_t bytes = (size_t)(gib * (1ull << 30)), entries = bytes / E;
  uint4 *tab; uint32_t *out;
  if (hipMalloc(&tab, bytes) != hipSuccess || hipMalloc(&out, threads * 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, tab, bytes / 16);
  hipDeviceSynchronize();
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int rep = 0; rep < 3; rep++) {
    hipEventRecord(a, 0);
    if (E == 64) hipLaunchKernelGGL(k_gather<64>, dim3(threads / 256), dim3(256), 0, 0, tab, entries, reads, out);
    else hipLaunchKernelGGL(k_gather<96>, dim3(threads / 256), dim3(256), 0, 0, tab, entries, reads, out);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("gather E=%d table %.1f GiB: %zu threads x %d reads = %.6e bytes requested per launch, %.3f ms (%.0f GB/s)\n", E, gib, threads, reads,
           (double)threads * reads * E, ms, (double)threads * reads * E / ms / 1e6);
  }
  return 0;
}

// Shared definitions for the secp256k1 device library (compiles for gfx950 with hipcc and,
// unchanged, for the host with any C++17 compiler -- the host build exists only so that
// tests/ can drive the exact device arithmetic on the CPU; see tests/devmath_host.cpp).
#pragma once
#include <stddef.h>
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define LAMD_HD __host__ __device__ __forceinline__
#define LAMD_HD_NOINLINE __host__ __device__ __noinline__
#else
#define LAMD_HD static inline __attribute__((always_inline))
#define LAMD_HD_NOINLINE static __attribute__((noinline))
#endif

#if defined(LAMD_CHECK_MAG)
#include <assert.h>
#define LAMD_ASSERT(x) assert(x)
#else
#define LAMD_ASSERT(x) ((void)0)
#endif

namespace lamd {
typedef uint32_t u32;
typedef uint64_t u64;
typedef uint8_t u8;
}  // namespace lamd

/* lightning_amd -- diagnostics exported by liblightning_amd.so next to the product ABI (include/lightning_amd.h).
 * None of these produces a verdict; tests and bring-up tooling use them to localise a miscompile or a hardware
 * difference to one primitive (DESIGN.md "toolchain notes"). */
#ifndef LIGHTNING_AMD_DEBUG_H
#define LIGHTNING_AMD_DEBUG_H
#include "lightning_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- device self-test: evaluates every arithmetic primitive and one full ECDSA verification of
 * the given triple both on the GPU and with the same code on the host, stage by stage.
 * Returns 0 if every stage agrees, else a bit mask of disagreeing stages (report names them),
 * or < 0 on engine error.  Diagnostic only; never used to produce a verdict. */
int lamd_selftest(lamd_ctx *ctx, const uint8_t hash32[32], const uint8_t sig64[64],
		  const uint8_t pub33[33], char *report, size_t cap);

/* Diagnostic: a 300-step dependent chain of field squarings (use_mul = 0) or multiplications on
 * the device, every step re-executed on the host from the device's own input limbs.  Returns the
 * number of disagreeing steps (0 = healthy), report describes the first few. */
int lamd_chain_debug(lamd_ctx *ctx, int use_mul, char *report, size_t cap);

/* Diagnostic: every intermediate of the field inversion / square-root addition chains, device vs host. */
int lamd_inv_debug(lamd_ctx *ctx, char *report, size_t cap);

int lamd_x2_debug(lamd_ctx *ctx, char *report, size_t cap); /* diagnostic: a^3 in several code shapes */

/* Diagnostic: copy nbytes at offset of internal work buffer `which` (0 prep records, 1 per-row key validity,
 * 2 dedupe representative, 3 dedupe uid, 4 key id per row, 5 first row of each distinct key, 6 distinct-key
 * validity, 7 distinct-key affine words, 8 key tables) to host memory.  Tests only. */
int lamd_debug_read(lamd_ctx *ctx, int which, size_t offset, size_t nbytes, void *out);

/* Measurement aid for bench.py's roofline block: the chip's sustained issue rate of the 32x32->64 multiply-add (v_mad_u64_u32 with an
 * SGPR carry-out, dependency-free, eight accumulators per lane) at `waves_per_simd` (1..8) waves per SIMD over `launches` launches of at least
 * min_ms each -- the ecmult kernel runs 3 waves per SIMD in launches of 3-4 ms, and a sub-millisecond micro-benchmark sees a boost clock such
 * launches do not.  *lane_ops_per_s = lanes x multiply-adds / HIP-event time; *memtime_per_realtime = (s_memtime delta) / (s_memrealtime
 * delta, 100 MHz) of one wave.  Computes nothing a verdict depends on. */
int lamd_debug_mul32_peak(lamd_ctx *ctx, int waves_per_simd, double min_ms, int launches, double *lane_ops_per_s, double *avg_launch_ms,
			  double *memtime_per_realtime);

/* The static G table in device memory (11 windows x 2^24 affine entries): read by the test-traffic signer kernels of
 * liblightning_amd_testgen.so.  NULL without a context. */
const void *lamd_debug_gtable(lamd_ctx *ctx);

/* Randomised arithmetic fuzz ON THE DEVICE: `lanes` lanes x `iters` iterations; every iteration draws operands at the
 * magnitude limits the group law uses (limbs up to 7 x 2^29, a fifth of them exactly at the bound) and runs fe_mul in
 * every legal magnitude pairing, fe_sqr, the lazy add/neg/normalisations, gej_double and the mixed addition with live
 * values carried across iterations, folding every raw result limb into a per-lane checksum.  The host pass of the same
 * inline functions recomputes every checksum.  Returns the number of lanes whose checksums differ (0 = healthy);
 * *ops (may be NULL) = field multiplications + squarings executed on the device. */
int lamd_fuzz_field(lamd_ctx *ctx, size_t lanes, int iters, uint64_t seed, uint64_t *ops, char *report, size_t cap);

/* ---- lamd_multi_* over a caller-supplied device layer.  Everything lamd_multi does to a device goes through this table; lamd_multi_init() binds
 * it to the engine + HIP + RCCL.  Tests bind it to host memory and a CPU checker, so that the sharding, padding, threading and gather layout
 * run at 8 "devices" on a machine without a GPU (tests/c/multi_stub.c).  Not part of the drop-in boundary.  All functions return LAMD_OK or
 * a LAMD_ERR_* code; h2d / verify_* / sigcheck_gossip are called on the device's own host thread, the others on the caller's. */
typedef struct lamd_multi_backend {
	void *user;   /* passed back as the first argument (NULL: lamd_multi's own engine state -- only meaningful with lamd_multi_init()) */
	int (*dev_open)(void *user, int device, void **handle);
	void (*dev_close)(void *user, void *handle);
	void *(*dev_alloc)(void *user, void *handle, size_t bytes);
	void (*dev_free)(void *user, void *handle, void *p);
	int (*h2d)(void *user, void *handle, void *dst, const void *src, size_t bytes);   /* complete (or ordered before the device's next verification) on return */
	int (*d2h)(void *user, void *handle, void *dst, const void *src, size_t bytes);   /* after the gather; complete on return */
	int (*verify_ecdsa)(void *user, void *handle, size_t n, const void *d_hash32, const void *d_sig64, const void *d_pub, size_t publen, size_t pubstride, void *d_ok);
	int (*verify_schnorr)(void *user, void *handle, size_t n, const void *d_msg32, const void *d_xonly32, const void *d_sig64, void *d_ok);
	int (*sigcheck_gossip)(void *user, void *handle, size_t n, const void *d_msgs, const void *d_off, const void *d_node_ids33, const void *d_rowbase,
			       size_t rows, void *d_verdict);
	int (*gather_open)(void *user, void **handles, int n);
	/* every device i: d_recv[i] = d_send[0] | d_send[1] | ... (bytes each), ordered after the verifications submitted to device i */
	int (*all_gather)(void *user, void **handles, int n, void **d_send, void **d_recv, size_t bytes);
	void (*gather_close)(void *user, void **handles, int n);
	const char *(*error)(void *user);
	void *(*engine_ctx)(void *user, void *handle);   /* may be NULL */
} lamd_multi_backend;
int lamd_multi_init_backend(lamd_multi **m, const int *devices, int n_devices, const lamd_multi_backend *backend);

#ifdef __cplusplus
}
#endif
#endif /* LIGHTNING_AMD_DEBUG_H */

// The host's two SHA-256 compression paths (lightning_amd/csrc/sha256.h: the x86 SHA extensions where the CPU has them, the portable rounds otherwise; the
// latter forced by -DLAMD_NO_SHA_NI) behind the same entry points: tests/test_devmath_host.py builds this file twice and compares the two builds with each
// other, with hashlib and with the BIP143 model on long and short inputs.
#include "../../lightning_amd/csrc/verify_core.h"
using namespace lamd;
extern "C" int h_sha_ni() {
#if defined(LAMD_SHA_NI)
  return sha256_have_ni() ? 1 : 0;
#else
  return 0;
#endif
}
extern "C" void h_sha256d(const u8 *p, size_t len, u8 *out32) { sha256d_bytes(p, len, out32); }
// the streaming form, fed in pieces of `piece` bytes (the block buffer is then partly full when a long piece arrives)
extern "C" void h_sha256d_stream(const u8 *p, size_t len, size_t piece, u8 *out32) {
  sha_stream s;
  shs_init(&s);
  for (size_t o = 0; o < len; o += piece) shs_update(&s, p + o, len - o < piece ? len - o : piece);
  shs_final_double(&s, out32);
}
extern "C" int h_bip143(u32 version, u32 locktime, const u8 *inputs, u32 n_in, const u8 *outputs, size_t outputs_len, u32 n_out, u32 in_idx,
                        const u8 *script, size_t script_len, uint64_t amount, u32 sighash_type, u8 *out32) {
  tx_view t;
  t.version = version; t.locktime = locktime; t.inputs = inputs; t.n_in = n_in; t.outputs = outputs; t.outputs_len = outputs_len; t.n_out = n_out;
  return bip143_sighash(t, in_idx, script, script_len, amount, sighash_type, out32);
}

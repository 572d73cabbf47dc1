/* lightning_amd -- synthetic workload generation ON THE DEVICE: test and benchmark infrastructure, shipped as its own
 * library (liblightning_amd_testgen.so, lightning_amd/csrc/lamd_testgen.hip) so that the product library
 * (liblightning_amd.so) holds no signing code.  It reaches the engine through lamd_stream(), lamd_get_info() and
 * lamd_debug_gtable() only. */
#ifndef LIGHTNING_AMD_TESTGEN_H
#define LIGHTNING_AMD_TESTGEN_H
#include "lightning_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- synthetic workload generation ON THE DEVICE (role of devtools/mkgossip.c:131-147,235-322
 * in the reference: producing signed test traffic).  Keys and nonces are derived from the seed
 * with splitmix64; outputs are device buffers.  Not a signing API: secrets are public by
 * construction. */
/* group: rows are cut into groups of `group` consecutive rows sharing one key (the 483 HTLC
 * signatures of a commitment share remote_htlckey, channeld/channeld.c:2224-2225); 0 = every row
 * draws its key independently from the nkeys identities. */
int lamd_gen_ecdsa_device(lamd_ctx *ctx, size_t n, uint64_t seed, size_t nkeys, size_t group, size_t publen,
			  void *d_hash32, void *d_sig64, void *d_pub);
int lamd_gen_schnorr_device(lamd_ctx *ctx, size_t n, uint64_t seed, size_t nkeys, size_t group,
			    void *d_msg32, void *d_xonly32, void *d_sig64);
/* n_cann channel_announcements (432 bytes each, no features, 4 signatures, node keys drawn from
 * n_nodes identities, bitcoin keys unique) followed by n_cupd channel_updates (138 bytes) signed by
 * one of the referenced channel's nodes -- built and signed like devtools/mkgossip.c:131-147,235-322.
 * d_msgs: n_cann*432 + n_cupd*138 bytes; d_node_ids33: (n_cann+n_cupd)*33 bytes (the update's signer;
 * zero for announcements). */
int lamd_gen_gossip_device(lamd_ctx *ctx, size_t n_cann, size_t n_cupd, uint64_t seed, size_t n_nodes,
			   void *d_msgs, void *d_node_ids33);

#ifdef __cplusplus
}
#endif
#endif

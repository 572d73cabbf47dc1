// liblightning_amd_client.so -- the entry points of include/lightning_amd.h that the mirror and the gossip ingest use, forwarded to a
// lamd_served process (include/lightning_amd_served.h).  Same prototypes, same return values; the client frames bytes and verifies nothing.
// lamd_init() = connect + hand the server a shared-memory block; without a server every call fails (LAMD_ERR_NO_DEVICE at init), so a
// mirror built on this library fails closed exactly like one whose process has no GPU.
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/un.h>

#include <map>
#include <string>
#include <vector>

#include "../../include/lightning_amd.h"
#include "served_common.h"

using namespace lamd_srv;

struct flush_slot {
  uint8_t *p = nullptr;
  size_t size = 0;
  bool busy = false;
  uint64_t seq = 0;
  size_t n = 0;
};
struct lamd_ctx {
  int fd = -1;
  uint8_t *shm = nullptr;  // block 0: the synchronous calls
  size_t shm_size = 0;
  std::string err;
  // streaming: the open set (queued locally), the blocks of the flushes in flight, the replies that arrived ahead of their turn
  std::vector<uint8_t> qh, qs, qk;
  std::vector<uint64_t> runs;  // keylen << 32 | rows, in ticket order
  size_t qn = 0;
  flush_slot slot[LAMD_SRV_FLUSH_SLOTS];
  uint64_t next_seq = 1, oldest_seq = 1;
  std::map<uint64_t, lamd_srv_rep> arrived;
};

namespace {

struct section { const void *p; size_t len; };

int recv_sync_reply(lamd_ctx *c, lamd_srv_rep *rep);
// block 0 (slot 0) or the block of flush slot s - 1
int attach(lamd_ctx *c, size_t size, int slot = 0) {
  const int mfd = memfd_create("lamd_client", MFD_CLOEXEC);
  if (mfd < 0 || ftruncate(mfd, (off_t)size) != 0) {
    if (mfd >= 0) close(mfd);
    c->err = "memfd_create / ftruncate failed";
    return LAMD_ERR_NOMEM;
  }
  void *p = mmap(nullptr, size, PROT_READ | PROT_WRITE, MAP_SHARED, mfd, 0);
  if (p == MAP_FAILED) { close(mfd); c->err = "mmap failed"; return LAMD_ERR_NOMEM; }
  lamd_srv_req r;
  memset(&r, 0, sizeof r);
  r.magic = LAMD_SRV_MAGIC;
  r.op = LAMD_SRV_OP_SHM;
  r.scalar[0] = size;
  r.scalar[1] = (uint64_t)slot;
  lamd_srv_rep rep;
  const bool ok = send_with_fd(c->fd, &r, sizeof r, mfd) && recv_sync_reply(c, &rep) == LAMD_OK;
  close(mfd);
  if (!ok || rep.rc != LAMD_OK) {
    munmap(p, size);
    rep.err[sizeof rep.err - 1] = 0;
    c->err = ok ? std::string("server refused the shared block: ") + rep.err : "connection to lamd_served lost";
    return ok ? rep.rc : LAMD_ERR_STATE;
  }
  uint8_t *&dst = slot ? c->slot[slot - 1].p : c->shm;
  size_t &dsz = slot ? c->slot[slot - 1].size : c->shm_size;
  if (dst) munmap(dst, dsz);
  dst = (uint8_t *)p;
  dsz = size;
  return LAMD_OK;
}
// the reply to a synchronous request: replies to flushes (seq != 0) that arrive first are kept for lamd_poll / lamd_wait
int recv_sync_reply(lamd_ctx *c, lamd_srv_rep *rep) {
  for (;;) {
    if (!recv_all(c->fd, rep, sizeof *rep) || rep->magic != LAMD_SRV_MAGIC) return LAMD_ERR_STATE;
    if (rep->seq == 0) return LAMD_OK;
    c->arrived[rep->seq] = *rep;
  }
}

// one round trip: the input sections go into the shared block, the reply's output sections are copied to `outs`
int call(lamd_ctx *c, uint32_t op, uint64_t n, const uint64_t scalar[6], const section *in, int n_in, const section *outs, int n_out, int *engine_rc) {
  if (!c || c->fd < 0) return LAMD_ERR_ARG;
  lamd_srv_req r;
  memset(&r, 0, sizeof r);
  r.magic = LAMD_SRV_MAGIC;
  r.op = op;
  r.n = n;
  if (scalar) memcpy(r.scalar, scalar, sizeof r.scalar);
  r.n_sections = (uint32_t)n_in;
  for (int i = 0; i < n_in; i++) r.section_len[i] = in[i].len;
  size_t off[LAMD_SRV_MAX_SECTIONS];
  const size_t out_off = layout(r, off);
  size_t need = out_off + 64;
  for (int i = 0; i < n_out; i++) need += align16(outs[i].len);
  if (need > c->shm_size) {
    size_t sz = c->shm_size ? c->shm_size : (size_t)1 << 20;
    while (sz < need) sz *= 2;
    const int rc = attach(c, sz);
    if (rc != LAMD_OK) return rc;
  }
  for (int i = 0; i < n_in; i++)
    if (in[i].len) memcpy(c->shm + off[i], in[i].p, in[i].len);
  lamd_srv_rep rep;
  if (!send_all(c->fd, &r, sizeof r) || recv_sync_reply(c, &rep) != LAMD_OK) {
    c->err = "connection to lamd_served lost";
    return LAMD_ERR_STATE;
  }
  if (rep.rc < 0) {
    rep.err[sizeof rep.err - 1] = 0;
    c->err = rep.err;
    return rep.rc;
  }
  size_t o = (size_t)rep.out_offset;
  for (int i = 0; i < n_out; i++) {
    if (outs[i].p && outs[i].len) memcpy((void *)outs[i].p, c->shm + o, outs[i].len);
    o += align16(outs[i].len);
  }
  if (engine_rc) *engine_rc = rep.rc;
  return LAMD_OK;
}

}  // namespace

extern "C" {

const char *lamd_version(void) { return "lightning_amd client (lamd_served protocol 2)"; }

int lamd_init(lamd_ctx **out, int /*device: the server chose it*/) {
  if (!out) return LAMD_ERR_ARG;
  lamd_ctx *c = new lamd_ctx;
  *out = c;  // returned on failure too, so that lamd_last_error() can say why (as the engine does)
  std::string sock_path;
  if (!default_socket(&sock_path, &c->err)) return LAMD_ERR_NO_DEVICE;
  const char *path = sock_path.c_str();
  struct sockaddr_un sa;
  memset(&sa, 0, sizeof sa);
  sa.sun_family = AF_UNIX;
  if (strlen(path) >= sizeof sa.sun_path) { c->err = "LAMD_SERVED_SOCKET: path too long"; return LAMD_ERR_ARG; }
  strcpy(sa.sun_path, path);
  c->fd = socket(AF_UNIX, SOCK_STREAM | SOCK_CLOEXEC, 0);
  if (c->fd < 0 || connect(c->fd, (struct sockaddr *)&sa, sizeof sa) != 0) {
    c->err = std::string("no lamd_served at ") + path + " (" + strerror(errno) + "): there is no verification without the service";
    if (c->fd >= 0) close(c->fd);
    c->fd = -1;
    return LAMD_ERR_NO_DEVICE;
  }
  // the process that answers must be the service of THIS user (or of LAMD_SERVED_UID), and so must the owner of the socket file: anybody can
  // bind a path in a directory they can write, and the client verifies nothing itself
  uid_t peer = (uid_t)-1;
  struct stat sb;
  const uid_t want = expected_peer_uid();
  if (!peer_uid_is(c->fd, want, &peer) || stat(path, &sb) != 0 || sb.st_uid != want) {
    c->err = std::string("the server at ") + path + " does not run under uid " + std::to_string((unsigned long)want) + " (peer uid " +
             (peer == (uid_t)-1 ? std::string("unknown") : std::to_string((unsigned long)peer)) + "): refusing to trust its verdicts";
    close(c->fd);
    c->fd = -1;
    return LAMD_ERR_NO_DEVICE;
  }
  const int rc = attach(c, (size_t)1 << 20);
  if (rc != LAMD_OK) {
    close(c->fd);
    c->fd = -1;
    return rc < 0 ? rc : LAMD_ERR_STATE;
  }
  return LAMD_OK;
}
void lamd_shutdown(lamd_ctx *c) {
  if (!c) return;
  if (c->fd >= 0) close(c->fd);
  if (c->shm) munmap(c->shm, c->shm_size);
  for (flush_slot &s : c->slot)
    if (s.p) munmap(s.p, s.size);
  delete c;
}

// ---- streaming (include/lightning_amd.h "streaming"): the open set lives in this process until lamd_flush() hands it to the server
static int queue_rows(lamd_ctx *c, size_t n, const uint8_t *a32, const uint8_t *sig64, const uint8_t *key, size_t keylen, size_t keystride) {
  if (!c || c->fd < 0) return LAMD_ERR_ARG;
  if (n == 0) return (int)c->qn;
  if (!a32 || !sig64 || !key || (keylen != 32 && keylen != 33 && keylen != 65) || keystride < keylen) { c->err = "bad argument"; return LAMD_ERR_ARG; }
  if (c->qn + n >= ((size_t)1 << 30)) { c->err = "fewer than 2^30 triples per flush"; return LAMD_ERR_ARG; }
  const int first = (int)c->qn;
  c->qh.insert(c->qh.end(), a32, a32 + 32 * n);
  c->qs.insert(c->qs.end(), sig64, sig64 + 64 * n);
  if (keystride == keylen) c->qk.insert(c->qk.end(), key, key + keylen * n);
  else
    for (size_t i = 0; i < n; i++) c->qk.insert(c->qk.end(), key + keystride * i, key + keystride * i + keylen);
  if (!c->runs.empty() && (c->runs.back() >> 32) == keylen && (c->runs.back() & 0xFFFFFFFFu) + n <= 0xFFFFFFFFu) c->runs.back() += n;
  else c->runs.push_back(((uint64_t)keylen << 32) | (uint64_t)n);
  c->qn += n;
  return first;
}
int lamd_queue_ecdsa_batch(lamd_ctx *c, size_t n, const uint8_t *hash32, const uint8_t *sig64, const uint8_t *pubkey, size_t publen, size_t pubstride) {
  if (publen != 33 && publen != 65) { if (c) c->err = "publen must be 33 or 65"; return LAMD_ERR_ARG; }
  return queue_rows(c, n, hash32, sig64, pubkey, publen, pubstride);
}
int lamd_queue_schnorr_batch(lamd_ctx *c, size_t n, const uint8_t *msg32, const uint8_t *xonly32, const uint8_t *sig64) {
  return queue_rows(c, n, msg32, sig64, xonly32, 32, 32);
}
int lamd_queue_ecdsa(lamd_ctx *c, const uint8_t hash32[32], const uint8_t sig64[64], const uint8_t *pubkey, size_t publen) {
  return lamd_queue_ecdsa_batch(c, 1, hash32, sig64, pubkey, publen, publen);
}
int lamd_queue_schnorr(lamd_ctx *c, const uint8_t msg32[32], const uint8_t xonly32[32], const uint8_t sig64[64]) {
  return lamd_queue_schnorr_batch(c, 1, msg32, xonly32, sig64);
}
// the producer form: the rows are reserved in the open set and the caller writes them before lamd_flush() (pointers valid until the next queue / flush call)
int lamd_queue_reserve(lamd_ctx *c, size_t n, size_t keylen, uint8_t **hash32, uint8_t **sig64, uint8_t **key) {
  if (!c || !hash32 || !sig64 || !key || n == 0 || (keylen != 32 && keylen != 33 && keylen != 65)) { if (c) c->err = "bad argument"; return LAMD_ERR_ARG; }
  if (c->qn + n >= ((size_t)1 << 30)) { c->err = "fewer than 2^30 triples per flush"; return LAMD_ERR_ARG; }
  const int first = (int)c->qn;
  const size_t oh = c->qh.size(), os = c->qs.size(), ok_ = c->qk.size();
  c->qh.resize(oh + 32 * n);
  c->qs.resize(os + 64 * n);
  c->qk.resize(ok_ + keylen * n);
  if (!c->runs.empty() && (c->runs.back() >> 32) == keylen && (c->runs.back() & 0xFFFFFFFFu) + n <= 0xFFFFFFFFu) c->runs.back() += n;
  else c->runs.push_back(((uint64_t)keylen << 32) | (uint64_t)n);
  c->qn += n;
  *hash32 = &c->qh[oh];
  *sig64 = &c->qs[os];
  *key = &c->qk[ok_];
  return first;
}
int lamd_flush(lamd_ctx *c) {
  if (!c || c->fd < 0) return LAMD_ERR_ARG;
  if (c->qn == 0) return LAMD_OK;
  int s = -1;
  for (int i = 0; i < LAMD_SRV_FLUSH_SLOTS; i++)
    if (!c->slot[i].busy) { s = i; break; }
  if (s < 0) { c->err = "too many flushes outstanding: collect one with lamd_poll / lamd_wait"; return LAMD_ERR_STATE; }
  lamd_srv_req r;
  memset(&r, 0, sizeof r);
  r.magic = LAMD_SRV_MAGIC;
  r.op = LAMD_SRV_OP_FLUSH;
  r.n = c->qn;
  r.scalar[0] = c->next_seq;
  r.slot = (uint32_t)(s + 1);
  r.n_sections = 4;
  r.section_len[0] = 8 * c->runs.size();
  r.section_len[1] = c->qh.size();
  r.section_len[2] = c->qs.size();
  r.section_len[3] = c->qk.size();
  size_t off[LAMD_SRV_MAX_SECTIONS];
  const size_t need = layout(r, off) + 64 + align16(c->qn);
  if (need > c->slot[s].size) {
    // (the slot is free: no flush is in flight in its block, so the server lets it be replaced)
    size_t sz = c->slot[s].size ? c->slot[s].size : (size_t)1 << 20;
    while (sz < need) sz *= 2;
    const int rc = attach(c, sz, s + 1);
    if (rc != LAMD_OK) return rc < 0 ? rc : LAMD_ERR_STATE;
  }
  uint8_t *b = c->slot[s].p;
  memcpy(b + off[0], c->runs.data(), 8 * c->runs.size());
  memcpy(b + off[1], c->qh.data(), c->qh.size());
  memcpy(b + off[2], c->qs.data(), c->qs.size());
  memcpy(b + off[3], c->qk.data(), c->qk.size());
  if (!send_all(c->fd, &r, sizeof r)) { c->err = "connection to lamd_served lost"; return LAMD_ERR_STATE; }
  c->slot[s].busy = true;
  c->slot[s].seq = c->next_seq++;
  c->slot[s].n = c->qn;
  c->qh.clear(); c->qs.clear(); c->qk.clear(); c->runs.clear();
  c->qn = 0;
  return LAMD_OK;
}
// verdicts of the OLDEST outstanding flush: 1 = finished, 0 = still running (poll only), < 0 error
static int collect(lamd_ctx *c, uint8_t *ok, size_t cap, size_t *n, bool block) {
  if (!c || c->fd < 0 || !ok || !n) return LAMD_ERR_ARG;
  flush_slot *f = nullptr;
  for (flush_slot &s : c->slot)
    if (s.busy && s.seq == c->oldest_seq) f = &s;
  if (!f) { c->err = "no flush outstanding"; return LAMD_ERR_STATE; }
  if (cap < f->n) { c->err = "verdict buffer smaller than the flush"; return LAMD_ERR_ARG; }
  while (!c->arrived.count(f->seq)) {
    lamd_srv_rep rep;
    if (!block) {
      const ssize_t k = recv(c->fd, &rep, sizeof rep, MSG_PEEK | MSG_DONTWAIT);
      if (k < 0 && (errno == EAGAIN || errno == EWOULDBLOCK)) return 0;
      if (k >= 0 && (size_t)k < sizeof rep && k != 0) return 0;  // a reply is on its way
    }
    if (!recv_all(c->fd, &rep, sizeof rep) || rep.magic != LAMD_SRV_MAGIC || rep.seq == 0) { c->err = "connection to lamd_served lost"; return LAMD_ERR_STATE; }
    c->arrived[rep.seq] = rep;
  }
  const lamd_srv_rep rep = c->arrived[f->seq];
  c->arrived.erase(f->seq);
  f->busy = false;
  c->oldest_seq++;
  if (rep.rc < 0) {
    char e[sizeof rep.err];
    memcpy(e, rep.err, sizeof e);
    e[sizeof e - 1] = 0;
    c->err = e;
    return rep.rc;
  }
  if (rep.out_offset + f->n > f->size) { c->err = "reply outside the flush block"; return LAMD_ERR_STATE; }
  memcpy(ok, f->p + rep.out_offset, f->n);
  *n = f->n;
  return 1;
}
int lamd_poll(lamd_ctx *c, uint8_t *ok, size_t cap, size_t *n) { return collect(c, ok, cap, n, false); }
int lamd_wait(lamd_ctx *c, uint8_t *ok, size_t cap, size_t *n) { return collect(c, ok, cap, n, true); }
const char *lamd_last_error(const lamd_ctx *c) { return c ? c->err.c_str() : "no context"; }

int lamd_verify_ecdsa_batch(lamd_ctx *c, size_t n, const uint8_t *hash32, const uint8_t *sig64, const uint8_t *pub, size_t publen, size_t pubstride, uint8_t *ok) {
  if (!c) return LAMD_ERR_ARG;
  if (n == 0) return LAMD_OK;
  if (!hash32 || !sig64 || !pub || !ok || (publen != 33 && publen != 65) || pubstride < publen) { c->err = "bad argument"; return LAMD_ERR_ARG; }
  std::string packed;
  if (pubstride != publen) {  // the wire form is dense
    packed.resize(n * publen);
    for (size_t i = 0; i < n; i++) memcpy(&packed[i * publen], pub + i * pubstride, publen);
    pub = (const uint8_t *)packed.data();
  }
  const uint64_t sc[6] = {publen, 0, 0, 0, 0, 0};
  const section in[3] = {{hash32, 32 * n}, {sig64, 64 * n}, {pub, publen * n}}, out[1] = {{ok, n}};
  return call(c, LAMD_SRV_OP_ECDSA, n, sc, in, 3, out, 1, nullptr);
}
int lamd_verify_schnorr_batch(lamd_ctx *c, size_t n, const uint8_t *msg32, const uint8_t *xonly32, const uint8_t *sig64, uint8_t *ok) {
  if (!c) return LAMD_ERR_ARG;
  if (n == 0) return LAMD_OK;
  if (!msg32 || !xonly32 || !sig64 || !ok) { c->err = "bad argument"; return LAMD_ERR_ARG; }
  const section in[3] = {{msg32, 32 * n}, {xonly32, 32 * n}, {sig64, 64 * n}}, out[1] = {{ok, n}};
  return call(c, LAMD_SRV_OP_SCHNORR, n, nullptr, in, 3, out, 1, nullptr);
}
int lamd_check_signed_hash(lamd_ctx *c, const uint8_t hash32[32], const uint8_t sig64[64], const uint8_t *pubkey, size_t publen) {
  uint8_t ok = 0;
  const int rc = lamd_verify_ecdsa_batch(c, 1, hash32, sig64, pubkey, publen, publen, &ok);
  return rc != LAMD_OK ? rc : ok ? 1 : 0;
}
int lamd_check_signed_hash_nodeid(lamd_ctx *c, const uint8_t hash32[32], const uint8_t sig64[64], const uint8_t node_id33[33]) {
  return lamd_check_signed_hash(c, hash32, sig64, node_id33, 33);
}
int lamd_check_schnorr_sig(lamd_ctx *c, const uint8_t hash32[32], const uint8_t pubkey33[33], const uint8_t sig64[64]) {
  uint8_t ok = 0;  // bitcoin/signature.c:417-422: the compressed key loses its parity byte
  const int rc = lamd_verify_schnorr_batch(c, 1, hash32, pubkey33 + 1, sig64, &ok);
  return rc != LAMD_OK ? rc : ok ? 1 : 0;
}
int lamd_pubkey_parse_batch(lamd_ctx *c, size_t n, const uint8_t *pub, size_t publen, size_t pubstride, uint8_t *out64, uint8_t *ok) {
  if (!c) return LAMD_ERR_ARG;
  if (n == 0) return LAMD_OK;
  if (!pub || !ok || (publen != 33 && publen != 65) || pubstride < publen) { c->err = "bad argument"; return LAMD_ERR_ARG; }
  std::string packed;
  if (pubstride != publen) {
    packed.resize(n * publen);
    for (size_t i = 0; i < n; i++) memcpy(&packed[i * publen], pub + i * pubstride, publen);
    pub = (const uint8_t *)packed.data();
  }
  const uint64_t sc[6] = {publen, 0, 0, 0, 0, 0};
  const section in[1] = {{pub, publen * n}}, out[2] = {{out64, out64 ? 64 * n : 0}, {ok, n}};
  if (!out64) {  // the server always writes both sections: skip the first on the way back
    std::string xy(64 * n, 0);
    const section out2[2] = {{xy.data(), 64 * n}, {ok, n}};
    return call(c, LAMD_SRV_OP_PUBKEY_PARSE, n, sc, in, 1, out2, 2, nullptr);
  }
  return call(c, LAMD_SRV_OP_PUBKEY_PARSE, n, sc, in, 1, out, 2, nullptr);
}
int lamd_sigcheck_gossip_batch(lamd_ctx *c, size_t n, const uint8_t *msgs, const uint64_t *off, const uint8_t *node_ids33, int8_t *verdict) {
  if (!c) return LAMD_ERR_ARG;
  if (n == 0) return LAMD_OK;
  if (!msgs || !off || !verdict) { c->err = "bad argument"; return LAMD_ERR_ARG; }
  std::string rel((n + 1) * 8, 0);
  for (size_t i = 0; i <= n; i++) ((uint64_t *)&rel[0])[i] = off[i] - off[0];
  const uint64_t sc[6] = {node_ids33 ? 1u : 0u, 0, 0, 0, 0, 0};
  const section in[3] = {{msgs + off[0], (size_t)(off[n] - off[0])}, {rel.data(), rel.size()}, {node_ids33, node_ids33 ? 33 * n : 0}}, out[1] = {{verdict, n}};
  return call(c, LAMD_SRV_OP_GOSSIP, n, sc, in, 3, out, 1, nullptr);
}

// the eleven template arrays of lamd_check_tx_sig_tx_batch, offsets made relative to the first row
struct tx_arrays {
  std::string in_rel, out_rel, sc_rel;
  section s[11];
};
static void pack_tx(tx_arrays &t, size_t n, const uint32_t *version, const uint32_t *locktime, const uint8_t *inputs40, const uint64_t *in_off,
                    const uint32_t *input_num, const uint64_t *amount_sat, const uint8_t *outputs, const uint64_t *out_off, const uint32_t *n_outputs,
                    const uint8_t *scripts, const uint64_t *script_off) {
  auto rel = [&](std::string &dst, const uint64_t *src) {
    dst.assign((n + 1) * 8, 0);
    for (size_t i = 0; i <= n; i++) ((uint64_t *)&dst[0])[i] = src[i] - src[0];
  };
  rel(t.in_rel, in_off); rel(t.out_rel, out_off); rel(t.sc_rel, script_off);
  t.s[0] = {version, 4 * n}; t.s[1] = {locktime, 4 * n};
  t.s[2] = {inputs40 + 40 * in_off[0], (size_t)(40 * (in_off[n] - in_off[0]))}; t.s[3] = {t.in_rel.data(), t.in_rel.size()};
  t.s[4] = {input_num, 4 * n}; t.s[5] = {amount_sat, 8 * n};
  t.s[6] = {outputs + out_off[0], (size_t)(out_off[n] - out_off[0])}; t.s[7] = {t.out_rel.data(), t.out_rel.size()};
  t.s[8] = {n_outputs, 4 * n};
  t.s[9] = {scripts + script_off[0], (size_t)(script_off[n] - script_off[0])}; t.s[10] = {t.sc_rel.data(), t.sc_rel.size()};
}
int lamd_check_tx_sig_tx_batch(lamd_ctx *c, size_t n, const uint32_t *version, const uint32_t *locktime, const uint8_t *inputs40, const uint64_t *in_off,
                               const uint32_t *input_num, const uint64_t *amount_sat, const uint8_t *outputs, const uint64_t *out_off, const uint32_t *n_outputs,
                               const uint8_t *scripts, const uint64_t *script_off, const uint8_t *sighash_type, const uint8_t *has_witness_script,
                               const uint8_t *sig64, const uint8_t *pub, size_t publen, size_t pubstride, uint8_t *ok) {
  if (!c) return LAMD_ERR_ARG;
  if (n == 0) return LAMD_OK;
  if (!version || !locktime || !inputs40 || !in_off || !input_num || !amount_sat || !outputs || !out_off || !n_outputs || !scripts || !script_off || !sighash_type ||
      !has_witness_script || !sig64 || !pub || !ok || (publen != 33 && publen != 65) || pubstride < publen) {
    c->err = "bad argument";
    return LAMD_ERR_ARG;
  }
  tx_arrays t;
  pack_tx(t, n, version, locktime, inputs40, in_off, input_num, amount_sat, outputs, out_off, n_outputs, scripts, script_off);
  std::string packed;
  if (pubstride != publen) {
    packed.resize(n * publen);
    for (size_t i = 0; i < n; i++) memcpy(&packed[i * publen], pub + i * pubstride, publen);
    pub = (const uint8_t *)packed.data();
  }
  section in[15];
  for (int i = 0; i < 11; i++) in[i] = t.s[i];
  in[11] = {sighash_type, n}; in[12] = {has_witness_script, n}; in[13] = {sig64, 64 * n}; in[14] = {pub, publen * n};
  const uint64_t sc[6] = {publen, 0, 0, 0, 0, 0};
  const section out[1] = {{ok, n}};
  return call(c, LAMD_SRV_OP_TXSIG_TX, n, sc, in, 15, out, 1, nullptr);
}
int lamd_check_commitment_signed(lamd_ctx *c, const lamd_tx_template *commit_tx, const uint8_t remote_funding33[33], const uint8_t commit_sig64[64],
                                 uint8_t commit_sighash_type, size_t n_htlc, const lamd_tx_template *htlc_txs, const uint8_t remote_htlckey33[33],
                                 const uint8_t *htlc_sigs64, const uint8_t *htlc_sighash_types, int64_t *first_bad, uint8_t *ok_rows) {
  if (!c) return LAMD_ERR_ARG;
  if (!commit_tx || !remote_funding33 || !commit_sig64 || !first_bad || (n_htlc && (!htlc_txs || !remote_htlckey33 || !htlc_sigs64 || !htlc_sighash_types))) {
    c->err = "bad argument";
    return LAMD_ERR_ARG;
  }
  *first_bad = 0;  // fails closed
  const size_t n = 1 + n_htlc;
  std::string ver(4 * n, 0), lock(4 * n, 0), inum(4 * n, 0), nout(4 * n, 0), amt(8 * n, 0), in_off(8 * (n + 1), 0), out_off(8 * (n + 1), 0), sc_off(8 * (n + 1), 0),
      types(n, 0), sigs(64 * n, 0), ins, outs, scs;
  for (size_t i = 0; i < n; i++) {
    const lamd_tx_template *t = i ? &htlc_txs[i - 1] : commit_tx;
    if ((t->n_inputs && !t->inputs40) || (t->outputs_len && !t->outputs) || (t->script_len && !t->script)) { c->err = "bad argument: transaction template with a null array"; return LAMD_ERR_ARG; }
    ((uint32_t *)&ver[0])[i] = t->version; ((uint32_t *)&lock[0])[i] = t->locktime; ((uint32_t *)&inum[0])[i] = t->input_num; ((uint32_t *)&nout[0])[i] = t->n_outputs;
    ((uint64_t *)&amt[0])[i] = t->amount_sat;
    ((uint64_t *)&in_off[0])[i] = ins.size() / 40; ((uint64_t *)&out_off[0])[i] = outs.size(); ((uint64_t *)&sc_off[0])[i] = scs.size();
    ins.append((const char *)t->inputs40, 40 * (size_t)t->n_inputs);
    outs.append((const char *)t->outputs, (size_t)t->outputs_len);
    scs.append((const char *)t->script, (size_t)t->script_len);
    types[i] = (char)(i ? htlc_sighash_types[i - 1] : commit_sighash_type);
    memcpy(&sigs[64 * i], i ? htlc_sigs64 + 64 * (i - 1) : commit_sig64, 64);
  }
  ((uint64_t *)&in_off[0])[n] = ins.size() / 40; ((uint64_t *)&out_off[0])[n] = outs.size(); ((uint64_t *)&sc_off[0])[n] = scs.size();
  uint8_t none[33] = {0};
  const section in[15] = {{ver.data(), ver.size()}, {lock.data(), lock.size()}, {ins.data(), ins.size()}, {in_off.data(), in_off.size()}, {inum.data(), inum.size()},
                          {amt.data(), amt.size()}, {outs.data(), outs.size()}, {out_off.data(), out_off.size()}, {nout.data(), nout.size()}, {scs.data(), scs.size()},
                          {sc_off.data(), sc_off.size()}, {types.data(), n}, {sigs.data(), sigs.size()}, {remote_funding33, 33},
                          {remote_htlckey33 ? remote_htlckey33 : none, 33}};
  std::string okv(n, 0);
  int64_t fb = 0;
  const section out[2] = {{&fb, 8}, {okv.data(), n}};   // (the server lays first_bad out in 16 bytes: align16(8))
  const int rc = call(c, LAMD_SRV_OP_COMMITMENT, n, nullptr, in, 15, out, 2, nullptr);
  if (rc != LAMD_OK) return rc;
  *first_bad = fb;
  if (ok_rows) memcpy(ok_rows, okv.data(), n);
  return LAMD_OK;
}
static int bolt12(lamd_ctx *c, bool check, size_t n, const uint8_t *tlvs, const uint64_t *off, const char *messagename, const char *fieldname, const uint8_t *key33,
                  size_t keystride, const uint8_t *sig64, uint8_t *merkle32, uint8_t *sighash32, uint8_t *ok) {
  if (!c) return LAMD_ERR_ARG;
  if (n == 0) return LAMD_OK;
  if (!tlvs || !off || !messagename || !fieldname || !ok || (check && (!key33 || !sig64 || keystride < 33))) { c->err = "bad argument"; return LAMD_ERR_ARG; }
  std::string rel((n + 1) * 8, 0), keys;
  for (size_t i = 0; i <= n; i++) ((uint64_t *)&rel[0])[i] = off[i] - off[0];
  const size_t blob = (size_t)(off[n] - off[0]);
  if (check) {
    keys.resize(33 * n);
    for (size_t i = 0; i < n; i++) memcpy(&keys[33 * i], key33 + i * keystride, 33);
    const section in[6] = {{tlvs + off[0], blob}, {rel.data(), rel.size()}, {messagename, strlen(messagename) + 1}, {fieldname, strlen(fieldname) + 1},
                           {keys.data(), keys.size()}, {sig64, 64 * n}}, out[1] = {{ok, n}};
    return call(c, LAMD_SRV_OP_BOLT12_CHECK, n, nullptr, in, 6, out, 1, nullptr);
  }
  const uint64_t sc[6] = {sighash32 ? 1u : 0u, 0, 0, 0, 0, 0};
  std::string m(32 * n, 0), s(32 * n, 0);
  const section in[4] = {{tlvs + off[0], blob}, {rel.data(), rel.size()}, {messagename, strlen(messagename) + 1}, {fieldname, strlen(fieldname) + 1}},
                out[3] = {{m.data(), m.size()}, {s.data(), s.size()}, {ok, n}};
  const int rc = call(c, LAMD_SRV_OP_BOLT12_MERKLE, n, sc, in, 4, out, 3, nullptr);
  if (rc != LAMD_OK) return rc;
  if (merkle32) memcpy(merkle32, m.data(), m.size());
  if (sighash32) memcpy(sighash32, s.data(), s.size());
  return LAMD_OK;
}
int lamd_bolt12_check_signature_batch(lamd_ctx *c, size_t n, const uint8_t *tlvs, const uint64_t *off, const char *messagename, const char *fieldname,
                                      const uint8_t *key33, size_t keystride, const uint8_t *sig64, uint8_t *ok) {
  return bolt12(c, true, n, tlvs, off, messagename, fieldname, key33, keystride, sig64, nullptr, nullptr, ok);
}
int lamd_bolt12_merkle_batch(lamd_ctx *c, size_t n, const uint8_t *tlvs, const uint64_t *off, const char *messagename, const char *fieldname, uint8_t *merkle32,
                             uint8_t *sighash32, uint8_t *ok) {
  return bolt12(c, false, n, tlvs, off, messagename, fieldname, nullptr, 0, nullptr, merkle32, sighash32, ok);
}
int lamd_ecdsa_recover_batch(lamd_ctx *c, size_t n, const uint8_t *hash32, const uint8_t *sig64, const uint8_t *recid, uint8_t *pub33, uint8_t *ok) {
  if (!c) return LAMD_ERR_ARG;
  if (n == 0) return LAMD_OK;
  if (!hash32 || !sig64 || !recid || !pub33 || !ok) { c->err = "bad argument"; return LAMD_ERR_ARG; }
  const section in[3] = {{hash32, 32 * n}, {sig64, 64 * n}, {recid, n}}, out[2] = {{pub33, 33 * n}, {ok, n}};
  return call(c, LAMD_SRV_OP_RECOVER, n, nullptr, in, 3, out, 2, nullptr);
}
int lamd_grind_htlc_tx_fee(lamd_ctx *c, const uint8_t *preimage, size_t preimage_len, const uint8_t *outputs, size_t outputs_len, uint64_t input_sat, uint64_t weight,
                           uint32_t min_feerate, uint32_t max_feerate, const uint8_t sig64[64], uint8_t sighash_type, int has_witness_script, const uint8_t pubkey33[33],
                           uint32_t *feerate, uint64_t *fee) {
  if (!c) return LAMD_ERR_ARG;
  if (!preimage || !outputs || !sig64 || !pubkey33) { c->err = "bad argument"; return LAMD_ERR_ARG; }
  const uint64_t sc[6] = {input_sat, weight, min_feerate, max_feerate, sighash_type, (uint64_t)(has_witness_script != 0)};
  uint8_t res[16] = {0};
  const section in[4] = {{preimage, preimage_len}, {outputs, outputs_len}, {sig64, 64}, {pubkey33, 33}}, out[1] = {{res, 16}};
  int erc = 0;
  const int rc = call(c, LAMD_SRV_OP_GRIND, 1, sc, in, 4, out, 1, &erc);
  if (rc != LAMD_OK) return rc;
  if (erc == 1) {
    if (feerate) memcpy(feerate, res, 4);
    if (fee) memcpy(fee, res + 8, 8);
  }
  return erc;
}
/* what the server has done so far (struct lamd_srv_stats): tests and operators */
int lamd_client_server_stats(lamd_ctx *c, struct lamd_srv_stats *st) {
  if (!c || !st) return LAMD_ERR_ARG;
  const section out[1] = {{st, sizeof *st}};
  return call(c, LAMD_SRV_OP_STATS, 0, nullptr, nullptr, 0, out, 1, nullptr);
}

}  // extern "C"

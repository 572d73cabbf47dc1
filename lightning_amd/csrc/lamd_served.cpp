// lamd_served -- ONE process owns the GPU's engine context; many client processes (liblightning_amd_client.so) share it.
// Protocol and rationale: include/lightning_amd_served.h.  SURVEY.md section 7 "Process model": one channeld per channel
// (channeld/channeld.c:7019-7129), gossipd, lightningd and plugins are separate single-threaded processes that verify inline.
//
//   lamd_served [--socket PATH] [--device N | --devices A,B,..] [--engine LIB] [--max-merge ROWS] [--max-flush-rows ROWS] [--linger-us US] [--copy-flushes] [--no-numa]
//
// Threads: an acceptor; one reader per connection (blocks in recv, turns a request into a job; a synchronous job it waits for and answers, a
// flush it hands over and goes on reading); ONE engine thread PER DEVICE, the only caller of that device's context (a context is not
// thread-safe).  A job goes to the device its first key hashes to (key affinity: the rows of one channel / peer / signer keep meeting the same
// key-table cache).  An engine thread takes EVERYTHING that is queued for its device at the moment it looks: the client flushes
// (LAMD_SRV_OP_FLUSH) are queued into the engine's own staging set and flushed as ONE engine batch, up to eight of those in flight, their
// verdicts scattered back and answered when the engine reports them (lamd_poll); the ECDSA jobs of equal key length become one lamd_verify_ecdsa_batch call, the BIP-340 jobs one
// lamd_verify_schnorr_batch call, the commitment_signed validations (and check_tx_sig batches) one lamd_check_tx_sig_tx_batch call (rows copied into
// one contiguous batch, verdicts scattered back, a commitment's first_bad taken from its slice); every other operation runs by itself.
// With --linger-us the engine thread waits that long after the first job of a round for company (default 0: merge what is there).
//
// The engine is bound through dlopen (--engine, default liblightning_amd.so next to this executable): the server itself has no
// verification code, and tests bind a stub library to run the queueing / merging / scattering on a machine without a GPU.
#include <dlfcn.h>
#include <fcntl.h>
#include <sched.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/un.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/lightning_amd.h"
#include "numa_cpus.h"
#include "served_common.h"

using namespace lamd_srv;

namespace {

struct engine_api {
  void *lib = nullptr;
  decltype(&lamd_init) init = nullptr;
  decltype(&lamd_shutdown) shutdown = nullptr;
  decltype(&lamd_last_error) last_error = nullptr;
  decltype(&lamd_verify_ecdsa_batch) verify_ecdsa = nullptr;
  decltype(&lamd_verify_schnorr_batch) verify_schnorr = nullptr;
  decltype(&lamd_pubkey_parse_batch) pubkey_parse = nullptr;
  decltype(&lamd_sigcheck_gossip_batch) gossip = nullptr;
  decltype(&lamd_check_tx_sig_tx_batch) txsig_tx = nullptr;
  decltype(&lamd_check_commitment_signed) commitment = nullptr;
  decltype(&lamd_bolt12_check_signature_batch) bolt12_check = nullptr;
  decltype(&lamd_bolt12_merkle_batch) bolt12_merkle = nullptr;
  decltype(&lamd_ecdsa_recover_batch) recover = nullptr;
  decltype(&lamd_grind_htlc_tx_fee) grind = nullptr;
  decltype(&lamd_queue_ecdsa_batch) queue_ecdsa = nullptr;
  decltype(&lamd_queue_schnorr_batch) queue_schnorr = nullptr;
  decltype(&lamd_queue_ecdsa_batch_inplace) queue_ecdsa_inplace = nullptr;    // optional (an older engine library: every flush row is copied)
  decltype(&lamd_queue_schnorr_batch_inplace) queue_schnorr_inplace = nullptr;
  decltype(&lamd_host_register) host_register = nullptr;
  decltype(&lamd_host_unregister) host_unregister = nullptr;
  decltype(&lamd_device_numa_node) numa_node = nullptr;   // optional
  decltype(&lamd_flush) flush = nullptr;
  decltype(&lamd_poll) poll = nullptr;
  decltype(&lamd_wait) wait = nullptr;
};
template <typename F>
bool bind(void *lib, const char *name, F *fn) {
  *(void **)fn = dlsym(lib, name);
  if (!*fn) fprintf(stderr, "lamd_served: engine library lacks %s\n", name);
  return *fn != nullptr;
}

struct blk {
  uint8_t *p = nullptr;
  size_t size = 0;
  bool pinned = false;  // a flush block the runtime has pinned (lamd_host_register): its rows are queued IN PLACE and leave it by DMA
};
struct conn {
  int fd = -1;
  blk shm[1 + LAMD_SRV_FLUSH_SLOTS];  // 0: synchronous calls; 1..: one flush each
  std::mutex wmu;                     // replies to flushes are written by engine threads, replies to synchronous requests by the reader
  std::atomic<int> async_pending{0};  // flushes handed to an engine thread and not answered yet: the blocks must stay mapped
  std::atomic<int> slot_pending[1 + LAMD_SRV_FLUSH_SLOTS] = {};  // ... per block: a block with a flush in flight is not replaced
};
struct job {
  lamd_srv_req req;
  conn *c = nullptr;
  int slot = 0;
  size_t off[LAMD_SRV_MAX_SECTIONS];
  size_t out_off = 0;
  lamd_srv_rep rep;
  bool done = false;
  bool async = false;  // LAMD_SRV_OP_FLUSH: heap-allocated, answered and deleted by the engine thread
  // the server's OWN copies of a request's offset arrays, taken once and validated: the shared block stays writable by its client while the
  // request runs, and an offset re-read after the check could point anywhere
  std::vector<uint64_t> o_in, o_out, o_sc;
  size_t queued_rows = 0;  // flush: rows that reached the engine's staging set
};

// one engine context = one GPU = one engine thread
struct flight {  // one engine flush: the client flushes it carries, in ticket order
  std::vector<job *> parts;
  size_t rows = 0;
};
struct device {
  int id = 0;
  lamd_ctx *ctx = nullptr;
  std::mutex mu;
  std::condition_variable cv;
  std::deque<job *> Q;
  std::deque<flight> inflight;
  std::vector<uint8_t> okbuf;
  std::thread th;
  int numa = -1;          // the NUMA node the device hangs on (-1: unknown / --no-numa): its engine thread and its context's buffers live there
  cpu_set_t cpus;         // ... that node's CPUs
};

engine_api E;
std::vector<device *> g_dev;
thread_local device *D = nullptr;  // the device of the engine thread that runs the code below
#define g_ctx (D->ctx)
std::mutex donemu;
std::condition_variable donecv;
std::atomic<bool> g_quit{false};
std::mutex statmu;
lamd_srv_stats g_stats;
size_t g_max_merge = (size_t)1 << 20;
// rows of one ENGINE flush that client flushes are merged into.  Merging is for SMALL flushes (one commitment_signed = 484 rows: a hundred of them
// share one engine flush); beyond a few 10^4 rows a larger batch buys no GPU efficiency and costs pipelining -- and a staging set that has to grow to
// a size it never had is a pinned allocation of tens of milliseconds in the middle of a stream.  Measured (profiles/r06_served_stream.txt: eight
// clients streaming 31 k-row flushes): unbounded 0.25 of one in-process producer's rate, 262 144 rows 0.41, 131 072 0.69, 65 536 0.77.
size_t g_max_flush_rows = (size_t)1 << 16;
unsigned g_linger_us = 0;
// Flush rows stay in the client's block and cross the bus from there (the block is pinned when it is attached) instead of being copied into the
// engine's staging set first: the copy -- 161 bytes per row, by the one engine thread of the device and its copy helpers -- was a third of the
// service's time per streamed row (profiles/r06_served_stream.txt).  --copy-flushes switches it off (the A/B, and for a runtime that cannot pin).
bool g_inplace = true;
// Every device's engine thread -- and the thread that creates its context, while it does -- runs on the CPUs of the NUMA node the device hangs on: the
// context's pinned staging sets are first touched there and the launches come from the near socket.  --no-numa leaves the threads where the scheduler puts them.
bool g_numa = true;
const size_t ENGINE_FLUSHES_IN_FLIGHT = 8;

const uint8_t *sec(const job *j, int i) { return j->c->shm[j->slot].p + j->off[i]; }
size_t seclen(const job *j, int i) { return (size_t)j->req.section_len[i]; }
uint8_t *outp(const job *j, size_t o) { return j->c->shm[j->slot].p + j->out_off + o; }

void stat_add(uint64_t *f, uint64_t d) {
  std::lock_guard<std::mutex> lk(statmu);
  *f += d;
}
void stat_max(uint64_t *f, uint64_t v) {
  std::lock_guard<std::mutex> lk(statmu);
  if (v > *f) *f = v;
}
void fail(job *j, int rc, const char *why) {
  j->rep.rc = rc;
  snprintf(j->rep.err, sizeof j->rep.err, "%s", why);
}
void engine_error(job *j, int rc) {
  j->rep.rc = rc;
  if (rc < 0) snprintf(j->rep.err, sizeof j->rep.err, "%s", E.last_error(g_ctx));
}
// input sections present with exactly these lengths, and `out_bytes` of output fit behind them
bool shape(job *j, std::initializer_list<size_t> lens, size_t out_bytes) {
  if (j->req.n_sections != lens.size()) { fail(j, LAMD_ERR_ARG, "wrong number of sections"); return false; }
  int i = 0;
  for (size_t l : lens) {
    if (l != (size_t)-1 && seclen(j, i) != l) { fail(j, LAMD_ERR_ARG, "section length does not match n"); return false; }
    i++;
  }
  if (j->out_off + align16(out_bytes) + 64 > j->c->shm[j->slot].size) { fail(j, LAMD_ERR_ARG, "shared block too small for the reply"); return false; }
  return true;
}
const size_t ANY = (size_t)-1;

// n * w without overflow surprises (n is bounded before any use)
bool sane_n(job *j) {
  if (j->req.n > ((uint64_t)1 << 28)) { fail(j, LAMD_ERR_ARG, "n too large"); return false; }
  return true;
}

void run_merged(std::vector<job *> &js, bool schnorr) {
  // one engine call for the rows of all these requests (equal key length)
  if (js.empty()) return;
  const size_t kl = schnorr ? 32 : (size_t)js[0]->req.scalar[0];
  size_t total = 0;
  for (job *j : js) total += (size_t)j->req.n;
  stat_add(&g_stats.rows_by_device[D->id], total);
  if (js.size() == 1) {
    job *j = js[0];
    const int rc = schnorr ? E.verify_schnorr(g_ctx, (size_t)j->req.n, sec(j, 0), sec(j, 1), sec(j, 2), outp(j, 0))
                           : E.verify_ecdsa(g_ctx, (size_t)j->req.n, sec(j, 0), sec(j, 1), sec(j, 2), kl, kl, outp(j, 0));
    engine_error(j, rc);
    stat_add(&g_stats.engine_calls, 1);
    return;
  }
  std::vector<uint8_t> a(32 * total), b(64 * total), c(kl * total), ok(total);
  size_t o = 0;
  for (job *j : js) {
    const size_t n = (size_t)j->req.n;
    memcpy(&a[32 * o], sec(j, 0), 32 * n);
    if (schnorr) {
      memcpy(&c[32 * o], sec(j, 1), 32 * n);
      memcpy(&b[64 * o], sec(j, 2), 64 * n);
    } else {
      memcpy(&b[64 * o], sec(j, 1), 64 * n);
      memcpy(&c[kl * o], sec(j, 2), kl * n);
    }
    o += n;
  }
  const int rc = schnorr ? E.verify_schnorr(g_ctx, total, a.data(), c.data(), b.data(), ok.data())
                         : E.verify_ecdsa(g_ctx, total, a.data(), b.data(), c.data(), kl, kl, ok.data());
  stat_add(&g_stats.engine_calls, 1);
  stat_add(&g_stats.merged_requests, js.size());
  stat_add(&g_stats.merged_rows, total);
  stat_max(&g_stats.largest_merge_requests, js.size());
  if (rc < 0) {  // one request's rows made the merged call fail: every request again by itself, so that only the offender sees the error
    for (job *j : js) {
      std::vector<job *> one(1, j);
      run_merged(one, schnorr);
    }
    return;
  }
  o = 0;
  for (job *j : js) {
    const size_t n = (size_t)j->req.n;
    memcpy(outp(j, 0), &ok[o], n);
    engine_error(j, rc);
    o += n;
  }
}

// a TXSIG_TX / COMMITMENT request is well-formed: section lengths match n, every offset stays inside its array
bool tx_valid(job *j) {
  const lamd_srv_req &r = j->req;
  const size_t n = (size_t)r.n;
  const bool commit = r.op == LAMD_SRV_OP_COMMITMENT;
  const size_t kl = commit ? 33 : (size_t)r.scalar[0];
  if (n == 0 || (kl != 33 && kl != 65)) { fail(j, LAMD_ERR_ARG, "bad n / key length"); return false; }
  if (commit ? !shape(j, {4 * n, 4 * n, ANY, 8 * (n + 1), 4 * n, 8 * n, ANY, 8 * (n + 1), 4 * n, ANY, 8 * (n + 1), n, 64 * n, 33, 33}, 16 + n)
             : !shape(j, {4 * n, 4 * n, ANY, 8 * (n + 1), 4 * n, 8 * n, ANY, 8 * (n + 1), 4 * n, ANY, 8 * (n + 1), n, n, 64 * n, kl * n}, n))
    return false;
  j->o_in.assign((const uint64_t *)sec(j, 3), (const uint64_t *)sec(j, 3) + n + 1);
  j->o_out.assign((const uint64_t *)sec(j, 7), (const uint64_t *)sec(j, 7) + n + 1);
  j->o_sc.assign((const uint64_t *)sec(j, 10), (const uint64_t *)sec(j, 10) + n + 1);
  const uint64_t *in_off = j->o_in.data(), *out_off = j->o_out.data(), *sc_off = j->o_sc.data();
  for (size_t i = 0; i < n; i++)
    if (in_off[i + 1] < in_off[i] || in_off[i + 1] > seclen(j, 2) / 40 || out_off[i + 1] < out_off[i] || out_off[i + 1] > seclen(j, 6) ||
        sc_off[i + 1] < sc_off[i] || sc_off[i + 1] > seclen(j, 9)) {
      fail(j, LAMD_ERR_ARG, "template offsets outside their arrays");
      return false;
    }
  return true;
}

void run_one(job *j);
// The commitment_signed validations (and check_tx_sig batches under 33-byte keys) of several clients as ONE lamd_check_tx_sig_tx_batch call: the rows'
// templates are concatenated (offsets rebased), the BIP143 hashes of all of them are made on the device in one launch, the verdicts are cut back per
// request and a commitment's first_bad is the first zero of its slice -- what lamd_check_commitment_signed computes for one (channeld.c:2171-2232 order).
void run_tx_merged(std::vector<job *> &js) {
  if (js.empty()) return;
  if (js.size() == 1) { run_one(js[0]); return; }
  size_t total = 0;
  for (job *j : js) total += (size_t)j->req.n;
  stat_add(&g_stats.rows_by_device[D->id], total);
  std::vector<uint32_t> ver(total), lock(total), inum(total), nout(total);
  std::vector<uint64_t> amt(total), in_off(total + 1), out_off(total + 1), sc_off(total + 1);
  std::vector<uint8_t> type(total), wit(total), sig(64 * total), pub(33 * total), ins, outs, scs, ok(total);
  size_t o = 0;
  for (job *j : js) {
    const size_t n = (size_t)j->req.n;
    const bool commit = j->req.op == LAMD_SRV_OP_COMMITMENT;
    memcpy(&ver[o], sec(j, 0), 4 * n); memcpy(&lock[o], sec(j, 1), 4 * n); memcpy(&inum[o], sec(j, 4), 4 * n);
    memcpy(&amt[o], sec(j, 5), 8 * n); memcpy(&nout[o], sec(j, 8), 4 * n);
    const uint64_t *io = j->o_in.data(), *oo = j->o_out.data(), *so = j->o_sc.data();   // tx_valid()'s copies
    const size_t ib = ins.size() / 40, ob = outs.size(), sb = scs.size();
    for (size_t i = 0; i < n; i++) { in_off[o + i] = ib + io[i]; out_off[o + i] = ob + oo[i]; sc_off[o + i] = sb + so[i]; }
    ins.insert(ins.end(), sec(j, 2), sec(j, 2) + 40 * io[n]);
    outs.insert(outs.end(), sec(j, 6), sec(j, 6) + oo[n]);
    scs.insert(scs.end(), sec(j, 9), sec(j, 9) + so[n]);
    memcpy(&type[o], sec(j, 11), n);
    if (commit) {
      memset(&wit[o], 1, n);
      memcpy(&sig[64 * o], sec(j, 12), 64 * n);
      memcpy(&pub[33 * o], sec(j, 13), 33);
      for (size_t i = 1; i < n; i++) memcpy(&pub[33 * (o + i)], sec(j, 14), 33);
    } else {
      memcpy(&wit[o], sec(j, 12), n);
      memcpy(&sig[64 * o], sec(j, 13), 64 * n);
      memcpy(&pub[33 * o], sec(j, 14), 33 * n);
    }
    o += n;
  }
  in_off[total] = ins.size() / 40; out_off[total] = outs.size(); sc_off[total] = scs.size();
  ins.push_back(0); outs.push_back(0); scs.push_back(0);   // never empty arrays
  const int rc = E.txsig_tx(g_ctx, total, ver.data(), lock.data(), ins.data(), in_off.data(), inum.data(), amt.data(), outs.data(), out_off.data(), nout.data(), scs.data(),
                            sc_off.data(), type.data(), wit.data(), sig.data(), pub.data(), 33, 33, ok.data());
  stat_add(&g_stats.engine_calls, 1);
  stat_add(&g_stats.merged_requests, js.size());
  stat_add(&g_stats.merged_rows, total);
  stat_max(&g_stats.largest_merge_requests, js.size());
  if (rc < 0) {  // as in run_merged: only the offending client may see the error
    for (job *j : js) run_one(j);
    return;
  }
  o = 0;
  for (job *j : js) {
    const size_t n = (size_t)j->req.n;
    if (rc == LAMD_OK) {
      if (j->req.op == LAMD_SRV_OP_COMMITMENT) {
        int64_t first_bad = -1;
        for (size_t i = 0; i < n && first_bad < 0; i++)
          if (!ok[o + i]) first_bad = (int64_t)i;
        memcpy(outp(j, 0), &first_bad, 8);
        memcpy(outp(j, 16), &ok[o], n);
      } else {
        memcpy(outp(j, 0), &ok[o], n);
      }
    }
    engine_error(j, rc);
    o += n;
  }
}

void run_one(job *j) {
  const lamd_srv_req &r = j->req;
  const size_t n = (size_t)r.n;
  stat_add(&g_stats.engine_calls, 1);
  switch (r.op) {
    case LAMD_SRV_OP_PUBKEY_PARSE: {
      const size_t kl = (size_t)r.scalar[0];
      if (kl != 33 && kl != 65) { fail(j, LAMD_ERR_ARG, "publen must be 33 or 65"); return; }
      if (!shape(j, {kl * n}, align16(64 * n) + n)) return;
      engine_error(j, E.pubkey_parse(g_ctx, n, sec(j, 0), kl, kl, outp(j, 0), outp(j, align16(64 * n))));
      return;
    }
    case LAMD_SRV_OP_GOSSIP: {
      const bool ids = r.scalar[0] != 0;
      if (!shape(j, {ANY, 8 * (n + 1), ids ? 33 * n : 0}, n)) return;
      j->o_in.assign((const uint64_t *)sec(j, 1), (const uint64_t *)sec(j, 1) + n + 1);
      const uint64_t *off = j->o_in.data();
      for (size_t i = 0; i < n; i++)
        if (off[i + 1] < off[i] || off[i + 1] > seclen(j, 0)) { fail(j, LAMD_ERR_ARG, "message offsets outside the blob"); return; }
      engine_error(j, E.gossip(g_ctx, n, sec(j, 0), off, ids ? sec(j, 2) : nullptr, (int8_t *)outp(j, 0)));
      return;
    }
    case LAMD_SRV_OP_TXSIG_TX:
    case LAMD_SRV_OP_COMMITMENT: {
      // sections 0..10: version locktime inputs40 in_off input_num amount outputs out_off n_outputs scripts script_off
      const bool commit = r.op == LAMD_SRV_OP_COMMITMENT;
      const size_t kl = commit ? 33 : (size_t)r.scalar[0];
      if (!tx_valid(j)) return;
      const uint64_t *in_off = j->o_in.data(), *out_off = j->o_out.data(), *sc_off = j->o_sc.data();
      if (!commit) {
        engine_error(j, E.txsig_tx(g_ctx, n, (const uint32_t *)sec(j, 0), (const uint32_t *)sec(j, 1), sec(j, 2), in_off, (const uint32_t *)sec(j, 4),
                                   (const uint64_t *)sec(j, 5), sec(j, 6), out_off, (const uint32_t *)sec(j, 8), sec(j, 9), sc_off, sec(j, 11), sec(j, 12),
                                   sec(j, 13), sec(j, 14), kl, kl, outp(j, 0)));
        return;
      }
      std::vector<lamd_tx_template> t(n);
      const uint32_t *ver = (const uint32_t *)sec(j, 0), *lock = (const uint32_t *)sec(j, 1), *inum = (const uint32_t *)sec(j, 4), *nout = (const uint32_t *)sec(j, 8);
      const uint64_t *amt = (const uint64_t *)sec(j, 5);
      for (size_t i = 0; i < n; i++) {
        t[i].version = ver[i]; t[i].locktime = lock[i];
        t[i].inputs40 = sec(j, 2) + 40 * in_off[i]; t[i].n_inputs = (uint32_t)(in_off[i + 1] - in_off[i]);
        t[i].input_num = inum[i]; t[i].amount_sat = amt[i];
        t[i].outputs = sec(j, 6) + out_off[i]; t[i].outputs_len = out_off[i + 1] - out_off[i]; t[i].n_outputs = nout[i];
        t[i].script = sec(j, 9) + sc_off[i]; t[i].script_len = sc_off[i + 1] - sc_off[i];
      }
      int64_t first_bad = 0;
      const uint8_t *types = sec(j, 11), *sigs = sec(j, 12);
      const int rc = E.commitment(g_ctx, &t[0], sec(j, 13), sigs, types[0], n - 1, n > 1 ? &t[1] : nullptr, sec(j, 14), sigs + 64, types + 1, &first_bad, outp(j, 16));
      memcpy(outp(j, 0), &first_bad, 8);
      engine_error(j, rc);
      return;
    }
    case LAMD_SRV_OP_BOLT12_CHECK:
    case LAMD_SRV_OP_BOLT12_MERKLE: {
      const bool check = r.op == LAMD_SRV_OP_BOLT12_CHECK;
      if (check ? !shape(j, {ANY, 8 * (n + 1), ANY, ANY, 33 * n, 64 * n}, n) : !shape(j, {ANY, 8 * (n + 1), ANY, ANY}, 2 * align16(32 * n) + n)) return;
      j->o_in.assign((const uint64_t *)sec(j, 1), (const uint64_t *)sec(j, 1) + n + 1);
      const uint64_t *off = j->o_in.data();
      for (size_t i = 0; i < n; i++)
        if (off[i + 1] < off[i] || off[i + 1] > seclen(j, 0)) { fail(j, LAMD_ERR_ARG, "stream offsets outside the blob"); return; }
      if (!seclen(j, 2) || !seclen(j, 3) || sec(j, 2)[seclen(j, 2) - 1] != 0 || sec(j, 3)[seclen(j, 3) - 1] != 0) { fail(j, LAMD_ERR_ARG, "names must be NUL-terminated"); return; }
      if (check)
        engine_error(j, E.bolt12_check(g_ctx, n, sec(j, 0), off, (const char *)sec(j, 2), (const char *)sec(j, 3), sec(j, 4), 33, sec(j, 5), outp(j, 0)));
      else
        engine_error(j, E.bolt12_merkle(g_ctx, n, sec(j, 0), off, (const char *)sec(j, 2), (const char *)sec(j, 3), outp(j, 0),
                                        r.scalar[0] ? outp(j, align16(32 * n)) : nullptr, outp(j, 2 * align16(32 * n))));
      return;
    }
    case LAMD_SRV_OP_RECOVER:
      if (!shape(j, {32 * n, 64 * n, n}, align16(33 * n) + n)) return;
      engine_error(j, E.recover(g_ctx, n, sec(j, 0), sec(j, 1), sec(j, 2), outp(j, 0), outp(j, align16(33 * n))));
      return;
    case LAMD_SRV_OP_GRIND: {
      if (!shape(j, {ANY, ANY, 64, 33}, 32)) return;
      uint32_t rate = 0;
      uint64_t fee = 0;
      const int rc = E.grind(g_ctx, sec(j, 0), seclen(j, 0), sec(j, 1), seclen(j, 1), r.scalar[0], r.scalar[1], (uint32_t)r.scalar[2], (uint32_t)r.scalar[3], sec(j, 2),
                             (uint8_t)r.scalar[4], (int)r.scalar[5], sec(j, 3), &rate, &fee);
      memcpy(outp(j, 0), &rate, 4);
      memcpy(outp(j, 8), &fee, 8);
      engine_error(j, rc);
      return;
    }
    case LAMD_SRV_OP_STATS: {
      stat_add(&g_stats.engine_calls, (uint64_t)-1);
      if (!shape(j, std::initializer_list<size_t>{}, sizeof g_stats)) return;
      std::lock_guard<std::mutex> lk(statmu);
      memcpy(outp(j, 0), &g_stats, sizeof g_stats);
      j->rep.rc = LAMD_OK;
      return;
    }
    default:
      fail(j, LAMD_ERR_ARG, "unknown operation");
  }
}

// ---- streaming: the flushes of the clients become engine flushes
// a flush request is well-formed: runs of known key lengths that add up to n, sections of exactly the lengths the runs imply
bool flush_valid(job *j) {
  const lamd_srv_req &r = j->req;
  const size_t n = (size_t)r.n;
  if (!sane_n(j)) return false;
  if (n == 0 || r.n_sections != 4 || seclen(j, 0) == 0 || seclen(j, 0) % 8 != 0 || seclen(j, 0) > 8 * n) { fail(j, LAMD_ERR_ARG, "flush: bad n / runs"); return false; }
  j->o_in.assign((const uint64_t *)sec(j, 0), (const uint64_t *)sec(j, 0) + seclen(j, 0) / 8);  // the server's own copy of the run table
  size_t rows = 0, keybytes = 0;
  for (uint64_t run : j->o_in) {
    const size_t kl = (size_t)(run >> 32), cnt = (size_t)(run & 0xFFFFFFFFu);
    if ((kl != 32 && kl != 33 && kl != 65) || cnt == 0 || cnt > n) { fail(j, LAMD_ERR_ARG, "flush: run with a bad key length or count"); return false; }
    rows += cnt;
    keybytes += kl * cnt;
  }
  if (rows != n) { fail(j, LAMD_ERR_ARG, "flush: runs do not add up to n"); return false; }
  return shape(j, {seclen(j, 0), 32 * n, 64 * n, keybytes}, n);
}
void answer_async(job *j) {
  // the block is free for its client's next flush from the moment the reply can be read: count it free BEFORE the reply goes out (a client that has
  // read the reply may send the next flush on this block at once, and serve() refuses a block that still counts as carrying one)
  j->c->slot_pending[j->slot].fetch_sub(1);
  {
    // An engine thread must never sleep in a send: it serves everybody.  A connection has at most LAMD_SRV_FLUSH_SLOTS flushes outstanding (serve()
    // refuses a second flush in a block that still carries one), so their 232-byte replies always fit the socket buffer; a client that
    // manages to fill it anyway has stopped reading and loses its connection.
    std::lock_guard<std::mutex> lk(j->c->wmu);
    const ssize_t k = send(j->c->fd, &j->rep, sizeof j->rep, MSG_NOSIGNAL | MSG_DONTWAIT);
    if (k != (ssize_t)sizeof j->rep) shutdown(j->c->fd, SHUT_RDWR);
  }
  j->c->async_pending.fetch_sub(1);
  delete j;
}
// the oldest engine flush of this device: verdicts back to the clients it carried.  block = wait for it.  -> false when it is still running
bool collect_one(bool block) {
  flight &f = D->inflight.front();
  if (D->okbuf.size() < f.rows + 1) D->okbuf.resize(f.rows + 1);
  size_t n = 0;
  const int rc = block ? E.wait(g_ctx, D->okbuf.data(), D->okbuf.size(), &n) : E.poll(g_ctx, D->okbuf.data(), D->okbuf.size(), &n);
  if (rc == 0) return false;
  size_t o = 0;
  for (job *j : f.parts) {
    if (j->rep.rc == LAMD_ERR_STATE) {  // not failed while it was queued
      if (rc == 1 && n == f.rows) {
        memcpy(outp(j, 0), &D->okbuf[o], j->queued_rows);
        j->rep.rc = LAMD_OK;
      } else if (rc < 0) {
        engine_error(j, rc);
      } else {
        fail(j, LAMD_ERR_STATE, "engine flush returned another number of verdicts than were queued");
      }
    }
    o += j->queued_rows;
    answer_async(j);
  }
  D->inflight.pop_front();
  return true;
}
void submit_flushes(std::vector<job *> &fl) {
  if (fl.empty()) return;
  flight cur;
  auto launch = [&]() {
    if (cur.parts.empty()) return;
    if (cur.rows) {
      while (D->inflight.size() >= ENGINE_FLUSHES_IN_FLIGHT) collect_one(true);
      const int rc = E.flush(g_ctx);
      stat_add(&g_stats.engine_flushes, 1);
      stat_max(&g_stats.largest_engine_flush_requests, cur.parts.size());
      stat_add(&g_stats.rows_by_device[D->id], cur.rows);
      if (rc < 0)
        for (job *j : cur.parts)
          if (j->rep.rc == LAMD_ERR_STATE) engine_error(j, rc);
      if (rc < 0) {  // nothing went to the device: answer now
        for (job *j : cur.parts) answer_async(j);
        cur = flight();
        return;
      }
      D->inflight.push_back(std::move(cur));
    } else {
      for (job *j : cur.parts) answer_async(j);  // every part failed before a row was queued
    }
    cur = flight();
  };
  for (job *j : fl) {
    stat_add(&g_stats.flushes, 1);
    if (!flush_valid(j)) { answer_async(j); continue; }
    const size_t n = (size_t)j->req.n;
    stat_add(&g_stats.flush_rows, n);
    if (cur.rows && cur.rows + n > g_max_flush_rows) launch();
    size_t o = 0, ko = 0;
    for (uint64_t run : j->o_in) {
      const size_t kl = (size_t)(run >> 32), cnt = (size_t)(run & 0xFFFFFFFFu);
      // (in place: the block is the client's to write, but it carries one flush at a time and is neither replaced nor unmapped before that flush is
      // answered -- slot_pending, async_pending -- which is after the engine has handed the verdicts back)
      const bool inplace = j->c->shm[j->slot].pinned;
      const int rc = inplace ? (kl == 32 ? E.queue_schnorr_inplace(g_ctx, cnt, sec(j, 1) + 32 * o, sec(j, 3) + ko, sec(j, 2) + 64 * o)
                                         : E.queue_ecdsa_inplace(g_ctx, cnt, sec(j, 1) + 32 * o, sec(j, 2) + 64 * o, sec(j, 3) + ko, kl))
                     : kl == 32 ? E.queue_schnorr(g_ctx, cnt, sec(j, 1) + 32 * o, sec(j, 3) + ko, sec(j, 2) + 64 * o)
                                : E.queue_ecdsa(g_ctx, cnt, sec(j, 1) + 32 * o, sec(j, 2) + 64 * o, sec(j, 3) + ko, kl, kl);
      if (rc >= 0 && inplace) stat_add(&g_stats.flush_rows_in_place, cnt);
      if (rc < 0) {  // the rows queued so far stay in the set (their verdicts are dropped); this client gets the error
        engine_error(j, rc);
        break;
      }
      j->queued_rows += cnt;
      o += cnt;
      ko += kl * cnt;
    }
    cur.rows += j->queued_rows;
    cur.parts.push_back(j);
  }
  launch();
}

void run_round(std::vector<job *> &round) {
  // merge classes: ECDSA by key length, BIP-340; the rest in arrival order
  std::vector<job *> e33, e65, sch;
  size_t r33 = 0, r65 = 0, rs = 0;
  for (job *j : round) {
    const lamd_srv_req &r = j->req;
    if (r.op == LAMD_SRV_OP_ECDSA || r.op == LAMD_SRV_OP_SCHNORR) {
      const size_t n = (size_t)r.n, kl = r.op == LAMD_SRV_OP_SCHNORR ? 32 : (size_t)r.scalar[0];
      if (!sane_n(j)) continue;
      if (n == 0) { j->rep.rc = LAMD_OK; continue; }
      if (r.op == LAMD_SRV_OP_ECDSA && kl != 33 && kl != 65) { fail(j, LAMD_ERR_ARG, "publen must be 33 or 65"); continue; }
      if (r.op == LAMD_SRV_OP_ECDSA ? !shape(j, {32 * n, 64 * n, kl * n}, n) : !shape(j, {32 * n, 32 * n, 64 * n}, n)) continue;
      std::vector<job *> &cls = r.op == LAMD_SRV_OP_SCHNORR ? sch : kl == 33 ? e33 : e65;
      size_t &rows = r.op == LAMD_SRV_OP_SCHNORR ? rs : kl == 33 ? r33 : r65;
      if (rows + n > g_max_merge && !cls.empty()) {  // a merged batch stays below --max-merge rows
        run_merged(cls, r.op == LAMD_SRV_OP_SCHNORR);
        cls.clear();
        rows = 0;
      }
      cls.push_back(j);
      rows += n;
    }
  }
  run_merged(e33, false);
  run_merged(e65, false);
  run_merged(sch, true);
  // commitment_signed validations and check_tx_sig batches under 33-byte keys: one device call for all that wait
  std::vector<job *> txs;
  size_t txrows = 0;
  for (job *j : round) {
    const lamd_srv_req &r = j->req;
    const bool tx = r.op == LAMD_SRV_OP_COMMITMENT || (r.op == LAMD_SRV_OP_TXSIG_TX && r.scalar[0] == 33);
    if (!tx || !sane_n(j) || !tx_valid(j)) continue;
    if (txrows + (size_t)r.n > g_max_merge && !txs.empty()) { run_tx_merged(txs); txs.clear(); txrows = 0; }
    txs.push_back(j);
    txrows += (size_t)r.n;
  }
  run_tx_merged(txs);
  for (job *j : round) {
    const lamd_srv_req &r = j->req;
    const bool tx = r.op == LAMD_SRV_OP_COMMITMENT || (r.op == LAMD_SRV_OP_TXSIG_TX && r.scalar[0] == 33);
    if (r.op != LAMD_SRV_OP_ECDSA && r.op != LAMD_SRV_OP_SCHNORR && !tx && sane_n(j)) run_one(j);
  }
  {
    std::lock_guard<std::mutex> lk(donemu);
    for (job *j : round) j->done = true;
  }
  stat_add(&g_stats.requests, round.size());
  donecv.notify_all();
}

void engine_loop(device *dev) {
  D = dev;
  if (dev->numa >= 0) sched_setaffinity(0, sizeof dev->cpus, &dev->cpus);
  for (;;) {
    std::vector<job *> round, sync, fl;
    {
      std::unique_lock<std::mutex> lk(D->mu);
      if (D->inflight.empty()) D->cv.wait(lk, [] { return !D->Q.empty() || g_quit.load(); });
      else D->cv.wait_for(lk, std::chrono::microseconds(40), [] { return !D->Q.empty() || g_quit.load(); });  // flushes in flight: look for their verdicts in between
      if (g_quit.load() && D->Q.empty() && D->inflight.empty()) return;
      if (g_linger_us && !D->Q.empty() && D->inflight.empty()) {  // company for the first job of the round
        lk.unlock();
        std::this_thread::sleep_for(std::chrono::microseconds(g_linger_us));
        lk.lock();
      }
      round.assign(D->Q.begin(), D->Q.end());
      D->Q.clear();
    }
    for (job *j : round) (j->async ? fl : sync).push_back(j);
    submit_flushes(fl);
    if (!sync.empty()) {
      while (!D->inflight.empty()) collect_one(true);  // the context runs the synchronous calls with nothing of the streaming queue outstanding
      run_round(sync);
    }
    while (!D->inflight.empty() && collect_one(false)) {}
  }
}

// key affinity: the device a request's FIRST KEY hashes to (FNV-1a over its first 32 bytes); requests without a key go to device 0
device *pick_device(const job *j) {
  if (g_dev.size() == 1) return g_dev[0];
  const lamd_srv_req &r = j->req;
  int s = -1;
  switch (r.op) {
    case LAMD_SRV_OP_ECDSA: s = 2; break;
    case LAMD_SRV_OP_SCHNORR: s = 1; break;
    case LAMD_SRV_OP_PUBKEY_PARSE: s = 0; break;
    case LAMD_SRV_OP_GOSSIP: s = r.scalar[0] ? 2 : 0; break;
    case LAMD_SRV_OP_TXSIG_TX: s = 14; break;
    case LAMD_SRV_OP_COMMITMENT: s = 13; break;
    case LAMD_SRV_OP_BOLT12_CHECK: s = 4; break;
    case LAMD_SRV_OP_FLUSH: s = 3; break;
    default: break;
  }
  if (s < 0 || (uint32_t)s >= r.n_sections || seclen(j, s) < 32 || j->off[s] + 32 > j->c->shm[j->slot].size) return g_dev[0];
  uint64_t h = 1469598103934665603ull;
  const uint8_t *k = sec(j, s);
  for (int i = 0; i < 32; i++) h = (h ^ k[i]) * 1099511628211ull;
  return g_dev[(size_t)((h >> 17) % g_dev.size())];
}

void drop_block(blk &b) {
  if (!b.p) return;
  if (b.pinned) {
    if (E.host_unregister) E.host_unregister(g_dev[0]->ctx, b.p);
    std::lock_guard<std::mutex> lk(statmu);
    g_stats.pinned_blocks_now--;
  }
  munmap(b.p, b.size);
  b = blk();
}

void serve(conn *cp) {
  conn &c = *cp;
  const int fd = c.fd;
  {
    std::lock_guard<std::mutex> lk(statmu);
    g_stats.clients_now++;
    g_stats.clients_total++;
  }
  for (;;) {
    job *jp = new job;
    job &j = *jp;
    int newfd = -1;
    if (!recv_with_fd(fd, &j.req, sizeof j.req, &newfd)) { delete jp; break; }
    memset(&j.rep, 0, sizeof j.rep);
    j.rep.magic = LAMD_SRV_MAGIC;
    j.rep.rc = LAMD_ERR_STATE;  // a path that forgets to answer reads as a failure, never as "verified"
    j.c = &c;
    if (j.req.magic != LAMD_SRV_MAGIC) { if (newfd >= 0) close(newfd); delete jp; break; }
    auto reply = [&]() {
      std::lock_guard<std::mutex> lk(c.wmu);
      return send_all(fd, &j.rep, sizeof j.rep);
    };
    if (j.req.op == LAMD_SRV_OP_SHM) {
      const size_t sz = (size_t)j.req.scalar[0], slot = (size_t)j.req.scalar[1];
      struct stat sb;
      if (newfd < 0 || sz < 4096 || slot > LAMD_SRV_FLUSH_SLOTS || fstat(newfd, &sb) != 0 || (size_t)sb.st_size < sz) {
        if (newfd >= 0) close(newfd);
        fail(&j, LAMD_ERR_ARG, "LAMD_SRV_OP_SHM without a usable descriptor / slot");
      } else if (slot && c.slot_pending[slot].load()) {  // a flush block is replaced only while no flush is in flight in it (the client library makes sure)
        close(newfd);
        fail(&j, LAMD_ERR_STATE, "LAMD_SRV_OP_SHM on a flush slot whose flush is outstanding");
      } else {
        void *p = mmap(nullptr, sz, PROT_READ | PROT_WRITE, MAP_SHARED, newfd, 0);
        close(newfd);
        if (p == MAP_FAILED) fail(&j, LAMD_ERR_NOMEM, "mmap of the client's block failed");
        else {
          drop_block(c.shm[slot]);
          c.shm[slot].p = (uint8_t *)p;
          c.shm[slot].size = sz;
          // a flush block is pinned once, here, for every device; a runtime that refuses the mapping leaves the block to the copying form
          if (slot && g_inplace && E.host_register && E.queue_ecdsa_inplace && E.queue_schnorr_inplace && E.host_register(g_dev[0]->ctx, p, sz) == LAMD_OK) {
            c.shm[slot].pinned = true;
            stat_add(&g_stats.pinned_blocks_now, 1);
          }
          j.rep.rc = LAMD_OK;
        }
      }
      const bool ok = reply();
      delete jp;
      if (!ok) break;
      continue;
    }
    if (newfd >= 0) close(newfd);
    j.async = j.req.op == LAMD_SRV_OP_FLUSH;
    j.slot = (int)j.req.slot;
    j.rep.seq = j.async ? j.req.scalar[0] : 0;
    j.out_off = layout(j.req, j.off);
    bool slot_ok = j.req.slot <= LAMD_SRV_FLUSH_SLOTS && (j.async ? j.req.slot >= 1 && j.req.scalar[0] != 0 : j.req.slot == 0);
    const bool slot_busy = slot_ok && j.async && c.slot_pending[j.slot].load() > 0;  // one flush per block at a time: its verdicts go there
    if (!slot_ok || slot_busy || !c.shm[j.slot].p || j.out_off == (size_t)-1 || j.out_off > c.shm[j.slot].size) {
      if (!slot_ok) j.slot = 0;
      fail(&j, slot_busy ? LAMD_ERR_STATE : LAMD_ERR_ARG, slot_busy ? "this block still carries a flush" : "request does not fit its shared block");
      const bool ok = reply();
      delete jp;
      if (!ok) break;
      continue;
    }
    j.rep.out_offset = j.out_off;
    device *dev = pick_device(&j);
    if (j.async) {  // handed over: the engine thread answers and deletes it
      c.async_pending.fetch_add(1);
      c.slot_pending[j.slot].fetch_add(1);
      std::lock_guard<std::mutex> lk(dev->mu);
      dev->Q.push_back(jp);
      dev->cv.notify_one();
      continue;
    }
    {
      std::lock_guard<std::mutex> lk(dev->mu);
      dev->Q.push_back(jp);
      dev->cv.notify_one();
    }
    {
      std::unique_lock<std::mutex> lk(donemu);
      donecv.wait(lk, [&] { return j.done; });
    }
    const bool ok = reply();
    delete jp;
    if (!ok) break;
  }
  while (c.async_pending.load() > 0) std::this_thread::sleep_for(std::chrono::microseconds(200));  // engine threads still write into the blocks
  for (blk &b : c.shm) drop_block(b);
  close(fd);
  delete cp;
  std::lock_guard<std::mutex> lk(statmu);
  g_stats.clients_now--;
}

int g_listen = -1;
void on_term(int) {
  g_quit.store(true);
  if (g_listen >= 0) shutdown(g_listen, SHUT_RDWR);
}

}  // namespace

int main(int argc, char **argv) {
  std::string sock, engine, why;
  default_socket(&sock, &why);
  std::vector<int> devices;
  for (int i = 1; i < argc; i++) {
    const std::string a = argv[i];
    auto val = [&]() -> const char * { return i + 1 < argc ? argv[++i] : ""; };
    if (a == "--socket") sock = val();
    else if (a == "--device") devices.assign(1, atoi(val()));
    else if (a == "--devices") {
      devices.clear();
      for (const char *s = val(); *s;) {
        char *e = nullptr;
        devices.push_back((int)strtol(s, &e, 10));
        if (e == s) { devices.clear(); break; }
        s = *e == ',' ? e + 1 : e;
      }
    }
    else if (a == "--engine") engine = val();
    else if (a == "--max-merge") g_max_merge = (size_t)atoll(val());
    else if (a == "--max-flush-rows") g_max_flush_rows = (size_t)atoll(val());
    else if (a == "--linger-us") g_linger_us = (unsigned)atoi(val());
    else if (a == "--copy-flushes") g_inplace = false;
    else if (a == "--no-numa") g_numa = false;
    else {
      fprintf(stderr, "usage: lamd_served [--socket PATH] [--device N | --devices A,B,..] [--engine LIB] [--max-merge ROWS] [--max-flush-rows ROWS] [--linger-us US] [--copy-flushes] [--no-numa]\n");
      return 2;
    }
  }
  if (devices.empty()) devices.assign(1, 0);
  if (devices.size() > LAMD_SRV_MAX_DEVICES) { fprintf(stderr, "lamd_served: at most %d devices\n", LAMD_SRV_MAX_DEVICES); return 2; }
  if (sock.empty()) { fprintf(stderr, "lamd_served: %s\n", why.c_str()); return 2; }
  if (engine.empty()) {  // liblightning_amd.so next to this executable
    char self[4096];
    const ssize_t k = readlink("/proc/self/exe", self, sizeof self - 1);
    std::string dir = k > 0 ? std::string(self, (size_t)k) : std::string(".");
    dir = dir.substr(0, dir.find_last_of('/'));
    engine = dir + "/liblightning_amd.so";
  }
  E.lib = dlopen(engine.c_str(), RTLD_NOW | RTLD_LOCAL);
  if (!E.lib) { fprintf(stderr, "lamd_served: dlopen(%s): %s\n", engine.c_str(), dlerror()); return 1; }
  bool ok = bind(E.lib, "lamd_init", &E.init) & bind(E.lib, "lamd_shutdown", &E.shutdown) & bind(E.lib, "lamd_last_error", &E.last_error) &
            bind(E.lib, "lamd_verify_ecdsa_batch", &E.verify_ecdsa) & bind(E.lib, "lamd_verify_schnorr_batch", &E.verify_schnorr) &
            bind(E.lib, "lamd_pubkey_parse_batch", &E.pubkey_parse) & bind(E.lib, "lamd_sigcheck_gossip_batch", &E.gossip) &
            bind(E.lib, "lamd_check_tx_sig_tx_batch", &E.txsig_tx) & bind(E.lib, "lamd_check_commitment_signed", &E.commitment) &
            bind(E.lib, "lamd_bolt12_check_signature_batch", &E.bolt12_check) & bind(E.lib, "lamd_bolt12_merkle_batch", &E.bolt12_merkle) &
            bind(E.lib, "lamd_ecdsa_recover_batch", &E.recover) & bind(E.lib, "lamd_grind_htlc_tx_fee", &E.grind) &
            bind(E.lib, "lamd_queue_ecdsa_batch", &E.queue_ecdsa) & bind(E.lib, "lamd_queue_schnorr_batch", &E.queue_schnorr) &
            bind(E.lib, "lamd_flush", &E.flush) & bind(E.lib, "lamd_poll", &E.poll) & bind(E.lib, "lamd_wait", &E.wait);
  if (!ok) return 1;
  // optional: rows queued in place (silently absent in an older engine library: every flush row is copied)
  *(void **)&E.queue_ecdsa_inplace = dlsym(E.lib, "lamd_queue_ecdsa_batch_inplace");
  *(void **)&E.queue_schnorr_inplace = dlsym(E.lib, "lamd_queue_schnorr_batch_inplace");
  *(void **)&E.host_register = dlsym(E.lib, "lamd_host_register");
  *(void **)&E.host_unregister = dlsym(E.lib, "lamd_host_unregister");
  *(void **)&E.numa_node = dlsym(E.lib, "lamd_device_numa_node");
  memset(&g_stats, 0, sizeof g_stats);
  for (size_t k = 0; k < devices.size(); k++) {
    device *dev = new device;
    dev->id = (int)k;
    cpu_set_t before;
    const bool have_before = sched_getaffinity(0, sizeof before, &before) == 0;
    if (g_numa && E.numa_node) {
      const int node = E.numa_node(devices[k]);
      if (node >= 0 && lamd_node_cpus(node, &dev->cpus)) {
        dev->numa = node;
        sched_setaffinity(0, sizeof dev->cpus, &dev->cpus);   // the context's host buffers are allocated (and first touched) by this thread
      }
    }
    const int rc = E.init(&dev->ctx, devices[k]);
    if (have_before) sched_setaffinity(0, sizeof before, &before);
    if (rc != LAMD_OK) {
      fprintf(stderr, "lamd_served: lamd_init(device %d) failed (%d): %s\n", devices[k], rc, dev->ctx ? E.last_error(dev->ctx) : "no device");
      if (dev->ctx) E.shutdown(dev->ctx);
      for (device *d : g_dev) E.shutdown(d->ctx);
      return 1;  // no engine, no service: there is no CPU verification to fall back to
    }
    g_dev.push_back(dev);
  }
  g_stats.devices = g_dev.size();
  auto shutdown_all = [] { for (device *d : g_dev) E.shutdown(d->ctx); };
  g_listen = socket(AF_UNIX, SOCK_STREAM | SOCK_CLOEXEC, 0);
  struct sockaddr_un sa;
  memset(&sa, 0, sizeof sa);
  sa.sun_family = AF_UNIX;
  if (sock.size() >= sizeof sa.sun_path) { fprintf(stderr, "lamd_served: socket path too long\n"); return 1; }
  strcpy(sa.sun_path, sock.c_str());
  struct stat old;
  if (lstat(sock.c_str(), &old) == 0) {  // a stale socket of OURS is replaced; anything else at the path is somebody else's business
    if (!S_ISSOCK(old.st_mode) || old.st_uid != geteuid() || unlink(sock.c_str()) != 0) {
      fprintf(stderr, "lamd_served: %s exists and is not a socket of uid %lu: not replacing it\n", sock.c_str(), (unsigned long)geteuid());
      shutdown_all();
      return 1;
    }
  }
  const mode_t um = umask(0177);  // the socket is born 0600: no window in which another user can connect (the daemons of one lightningd run under one user)
  const bool bound = g_listen >= 0 && bind(g_listen, (struct sockaddr *)&sa, sizeof sa) == 0 && listen(g_listen, 128) == 0;
  umask(um);
  if (!bound) {
    perror("lamd_served: bind/listen");
    shutdown_all();
    return 1;
  }
  signal(SIGTERM, on_term);
  signal(SIGINT, on_term);
  signal(SIGPIPE, SIG_IGN);
  for (device *d : g_dev) d->th = std::thread(engine_loop, d);
  std::string devs;
  for (size_t k = 0; k < devices.size(); k++) devs += (k ? "," : "") + std::to_string(devices[k]);
  std::string numa_note;
  for (size_t k = 0; k < g_dev.size(); k++)
    if (g_dev[k]->numa >= 0) numa_note += (numa_note.empty() ? "; " : ", ") + std::string("device ") + std::to_string(devices[k]) + " on NUMA node " + std::to_string(g_dev[k]->numa);
  printf("lamd_served: ready on %s (device %s, engine %s%s)\n", sock.c_str(), devs.c_str(), engine.c_str(), numa_note.c_str());
  fflush(stdout);
  std::vector<std::thread> readers;
  while (!g_quit.load()) {
    const int fd = accept4(g_listen, nullptr, nullptr, SOCK_CLOEXEC);
    if (fd < 0) {
      if (errno == EINTR) continue;
      break;
    }
    uid_t peer = (uid_t)-1;
    if (!peer_uid_is(fd, expected_peer_uid(), &peer)) {  // not one of this user's daemons
      fprintf(stderr, "lamd_served: connection from uid %ld refused\n", peer == (uid_t)-1 ? -1L : (long)peer);
      close(fd);
      continue;
    }
    conn *c = new conn;
    c->fd = fd;
    readers.emplace_back(serve, c);
  }
  g_quit.store(true);
  for (device *d : g_dev) {
    d->cv.notify_all();
    d->th.join();
  }
  for (auto &t : readers) t.detach();  // blocked in recv on connections their clients still hold; the process is leaving
  close(g_listen);
  unlink(sock.c_str());
  shutdown_all();
  printf("lamd_served: %llu requests, %llu engine calls, %llu requests in merged calls (largest merge %llu); %llu flushes in %llu engine flushes\n",
         (unsigned long long)g_stats.requests, (unsigned long long)g_stats.engine_calls, (unsigned long long)g_stats.merged_requests,
         (unsigned long long)g_stats.largest_merge_requests, (unsigned long long)g_stats.flushes, (unsigned long long)g_stats.engine_flushes);
  fflush(stdout);
  _exit(0);
}

// lamd_multi_*: ONE host process driving several MI355X (include/lightning_amd.h).
//
// The reference's callers are many single-threaded daemons (one channeld per channel, channeld/channeld.c:7019-7129; one gossipd); the
// sidecar that serves them owns every GPU of the node.  Rows are independent, so the job is: cut the batch into contiguous, group-aligned
// ranges (a channel_announcement's four signatures / a commitment's 484 rows stay on one device so that a per-key table is built once),
// give every device its range -- one engine context and one host thread per device: the H2D copies of pageable caller memory are
// synchronous per thread and must run side by side --, verify, all-gather the verdict bytes on the devices (RCCL over xGMI, padded to
// the largest shard; the north star's "RCCL all-gather of the boolean result vector": every device ends with the whole vector, for
// on-device follow-up) and copy the vector to the host ONCE, from the first device.
//
// Everything that touches a device goes through a small table of functions (lamd_multi_backend, include/lightning_amd_debug.h).  The product
// binds it to the engine + HIP + RCCL (dlopen()ed at lamd_multi_init: a single-GPU process never loads librccl); tests bind it to a stub
// whose "devices" are host memory, so that the sharding, padding, threading and gather layout run at 8 devices on a machine without a GPU.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/lightning_amd.h"
#include "../../include/lightning_amd_debug.h"
#include "numa_cpus.h"

namespace {

// ---------------------------------------------------------------- the engine back end
typedef void *ncclComm_t;
struct rccl_api {
  void *lib = nullptr;
  int (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
  int (*CommDestroy)(ncclComm_t) = nullptr;
  int (*GroupStart)(void) = nullptr;
  int (*GroupEnd)(void) = nullptr;
  int (*AllGather)(const void *, void *, size_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
};
// SURVEY 8(e) words it "each GPU has its own pinned staging ring and stream".  What a device has here: its own copy stream, and TWO ways in.
// Caller memory that is page-locked goes down that stream as one asynchronous copy per array (no staging, the host thread does not wait, the
// engine's kernels wait for the copy stream on the device: lamd_wait_stream).  Pageable caller memory goes through the runtime's own staged
// hipMemcpy().  A per-device ring of pinned slots filled by the device's host thread (plus a helper thread) was built in round 5 and measured
// SLOWER than the runtime's path on one MI355X -- 96-103 against 105-113 M ECDSA-65 rows/s for a 1 M-row call, profiles/r05_ab_variants.txt: both
// copy out of pageable memory with one or two cores (12-15 GB/s each), the runtime's staging copy is the better tuned of the two and overlaps its
// own DMA -- and was deleted in round 6 rather than kept behind a switch.
struct eng_dev {
  int device = 0;
  lamd_ctx *ctx = nullptr;
  hipStream_t gstream = nullptr;  // the collective's stream on this device
  ncclComm_t comm = nullptr;
  hipStream_t cstream = nullptr;  // H2D copies of page-locked caller memory
  bool copies_pending = false;    // the copy stream holds work the next verification has to wait for
};
// caller memory that is already page-locked (hipHostMalloc / hipHostRegister) needs no staging: the copy engine reads it where it lies
bool is_pinned_host(const void *p) {
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
  return a.type == hipMemoryTypeHost;
}
struct eng_state {
  rccl_api nccl;
  std::mutex err_mu;  // the per-device worker threads report failures side by side (a dead node fails every shard at once)
  std::string err;
  bool comms = false;
  void fail(const std::string &what) {
    std::lock_guard<std::mutex> lk(err_mu);
    if (err.empty()) err = what; else if (err.find(what) == std::string::npos && err.size() < 1024) err += "; " + what;
  }
  void clear() {
    std::lock_guard<std::mutex> lk(err_mu);
    err.clear();
  }
};

int eng_open(void *user, int device, void **handle) {
  eng_state *st = (eng_state *)user;
  eng_dev *d = new eng_dev;
  d->device = device;
  const int rc = lamd_init(&d->ctx, device);  // before the communicator: the lanes take their hardware queues first (lightning_amd.h, lamd_init)
  if (rc != LAMD_OK) {
    st->fail(std::string("lamd_init(device ") + std::to_string(device) + "): " + (d->ctx ? lamd_last_error(d->ctx) : "no device"));
    if (d->ctx) lamd_shutdown(d->ctx);
    delete d;
    return rc;
  }
  if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&d->gstream, hipStreamNonBlocking) != hipSuccess) {
    st->fail("hipStreamCreate failed");
    lamd_shutdown(d->ctx);
    delete d;
    return LAMD_ERR_HIP;
  }
  if (hipStreamCreateWithFlags(&d->cstream, hipStreamNonBlocking) != hipSuccess) {
    st->fail("hipStreamCreate failed");
    *handle = d;
    return LAMD_ERR_HIP;
  }
  *handle = d;
  return LAMD_OK;
}
void eng_close(void *, void *handle) {
  eng_dev *d = (eng_dev *)handle;
  if (!d) return;
  (void)hipSetDevice(d->device);
  if (d->cstream) (void)hipStreamSynchronize(d->cstream);
  if (d->cstream) (void)hipStreamDestroy(d->cstream);
  if (d->gstream) (void)hipStreamDestroy(d->gstream);
  if (d->ctx) lamd_shutdown(d->ctx);
  delete d;
}
void *eng_alloc(void *, void *handle, size_t bytes) {
  void *p = nullptr;
  if (hipSetDevice(((eng_dev *)handle)->device) != hipSuccess || hipMalloc(&p, bytes ? bytes : 1) != hipSuccess) return nullptr;
  return p;
}
void eng_free(void *, void *handle, void *p) {
  (void)hipSetDevice(((eng_dev *)handle)->device);
  (void)hipFree(p);
}
int eng_h2d(void *user, void *handle, void *dst, const void *src, size_t bytes) {
  eng_dev *d = (eng_dev *)handle;
  if (hipSetDevice(d->device) != hipSuccess) { ((eng_state *)user)->fail("hipSetDevice failed"); return LAMD_ERR_HIP; }
  if (bytes >= (1u << 20) && is_pinned_host(src)) {  // page-locked caller memory: one asynchronous copy, no staging
    if (hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, d->cstream) != hipSuccess) {
      ((eng_state *)user)->fail("H2D from pinned caller memory failed (device " + std::to_string(d->device) + ")");
      return LAMD_ERR_HIP;
    }
    d->copies_pending = true;
    return LAMD_OK;
  }
  // synchronous: the caller's memory is pageable, the runtime stages it; the engine's (asynchronous) kernels of the chunk before run meanwhile
  if (hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice) != hipSuccess) {
    ((eng_state *)user)->fail("hipMemcpy H2D failed (device " + std::to_string(((eng_dev *)handle)->device) + ")");
    return LAMD_ERR_HIP;
  }
  return LAMD_OK;
}
int eng_d2h(void *user, void *handle, void *dst, const void *src, size_t bytes) {
  eng_dev *d = (eng_dev *)handle;
  if (hipSetDevice(d->device) != hipSuccess || hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, d->gstream) != hipSuccess ||
      hipStreamSynchronize(d->gstream) != hipSuccess) {
    ((eng_state *)user)->fail("hipMemcpy D2H failed (device " + std::to_string(d->device) + ")");
    return LAMD_ERR_HIP;
  }
  return LAMD_OK;
}
int eng_fail(void *user, void *handle, int rc) {
  if (rc != LAMD_OK) ((eng_state *)user)->fail("device " + std::to_string(((eng_dev *)handle)->device) + ": " + lamd_last_error(((eng_dev *)handle)->ctx));
  return rc;
}
// the verification about to be submitted reads what the copy stream is still carrying: a device-side edge, the host does not wait
int eng_join_copies(void *user, void *handle) {
  eng_dev *d = (eng_dev *)handle;
  if (!d->copies_pending) return LAMD_OK;
  d->copies_pending = false;
  return eng_fail(user, handle, lamd_wait_stream(d->ctx, d->cstream));
}
int eng_ecdsa(void *user, void *handle, size_t n, const void *h, const void *s, const void *p, size_t publen, size_t stride, void *ok) {
  const int rc = eng_join_copies(user, handle);
  if (rc != LAMD_OK) return rc;
  return eng_fail(user, handle, lamd_verify_ecdsa_batch_device(((eng_dev *)handle)->ctx, n, h, s, p, publen, stride, ok));
}
int eng_schnorr(void *user, void *handle, size_t n, const void *m, const void *x, const void *s, void *ok) {
  const int rc = eng_join_copies(user, handle);
  if (rc != LAMD_OK) return rc;
  return eng_fail(user, handle, lamd_verify_schnorr_batch_device(((eng_dev *)handle)->ctx, n, m, x, s, ok));
}
int eng_gossip(void *user, void *handle, size_t n, const void *msgs, const void *off, const void *ids, const void *rowbase, size_t rows, void *verdict) {
  const int rc = eng_join_copies(user, handle);
  if (rc != LAMD_OK) return rc;
  return eng_fail(user, handle, lamd_sigcheck_gossip_batch_device(((eng_dev *)handle)->ctx, n, msgs, off, ids, rowbase, rows, verdict));
}
int eng_gather_open(void *user, void **handles, int n) {
  eng_state *st = (eng_state *)user;
  rccl_api &N = st->nccl;
  N.lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
  if (!N.lib) N.lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
  if (!N.lib) { st->fail(std::string("dlopen(librccl.so): ") + dlerror()); return LAMD_ERR_HIP; }
  *(void **)&N.CommInitAll = dlsym(N.lib, "ncclCommInitAll");
  *(void **)&N.CommDestroy = dlsym(N.lib, "ncclCommDestroy");
  *(void **)&N.GroupStart = dlsym(N.lib, "ncclGroupStart");
  *(void **)&N.GroupEnd = dlsym(N.lib, "ncclGroupEnd");
  *(void **)&N.AllGather = dlsym(N.lib, "ncclAllGather");
  *(void **)&N.GetErrorString = dlsym(N.lib, "ncclGetErrorString");
  if (!N.CommInitAll || !N.CommDestroy || !N.GroupStart || !N.GroupEnd || !N.AllGather) { st->fail("librccl.so lacks an entry point"); return LAMD_ERR_HIP; }
  std::vector<int> devs(n);
  std::vector<ncclComm_t> comms(n);
  for (int i = 0; i < n; i++) devs[i] = ((eng_dev *)handles[i])->device;
  const int rc = N.CommInitAll(comms.data(), n, devs.data());
  if (rc != 0) { st->fail(std::string("ncclCommInitAll: ") + (N.GetErrorString ? N.GetErrorString(rc) : "error")); return LAMD_ERR_HIP; }
  for (int i = 0; i < n; i++) ((eng_dev *)handles[i])->comm = comms[i];
  st->comms = true;
  return LAMD_OK;
}
// every device: recv[i] = send[0] | send[1] | ... (bytes each), ordered after the verifications submitted to that device's engine
int eng_all_gather(void *user, void **handles, int n, void **send, void **recv, size_t bytes) {
  eng_state *st = (eng_state *)user;
  rccl_api &N = st->nccl;
  for (int i = 0; i < n; i++) {
    eng_dev *d = (eng_dev *)handles[i];
    if (hipSetDevice(d->device) != hipSuccess) { st->fail("hipSetDevice failed"); return LAMD_ERR_HIP; }
    const int rc = lamd_stream_wait_results(d->ctx, d->gstream);  // a device-side edge: the host does not wait for the kernels
    if (rc != LAMD_OK) return eng_fail(user, d, rc);
  }
  int rc = N.GroupStart();
  for (int i = 0; i < n && rc == 0; i++) {
    eng_dev *d = (eng_dev *)handles[i];
    (void)hipSetDevice(d->device);
    rc = N.AllGather(send[i], recv[i], bytes, 1 /* ncclUint8 */, d->comm, d->gstream);
  }
  const int rc2 = N.GroupEnd();
  if (rc != 0 || rc2 != 0) { st->fail(std::string("ncclAllGather: ") + (N.GetErrorString ? N.GetErrorString(rc ? rc : rc2) : "error")); return LAMD_ERR_HIP; }
  for (int i = 1; i < n; i++) {  // device 0's stream is drained by the D2H that follows; the others here, so that their buffers may be reused
    eng_dev *d = (eng_dev *)handles[i];
    if (hipSetDevice(d->device) != hipSuccess || hipStreamSynchronize(d->gstream) != hipSuccess) { st->fail("hipStreamSynchronize failed"); return LAMD_ERR_HIP; }
  }
  return LAMD_OK;
}
void eng_gather_close(void *user, void **handles, int n) {
  eng_state *st = (eng_state *)user;
  // whatever communicators exist (an init that failed half-way leaves some), then the library itself
  if (st->nccl.CommDestroy)
    for (int i = 0; i < n; i++)
      if (handles[i] && ((eng_dev *)handles[i])->comm) {
        (void)st->nccl.CommDestroy(((eng_dev *)handles[i])->comm);
        ((eng_dev *)handles[i])->comm = nullptr;
      }
  st->comms = false;
  if (st->nccl.lib) (void)dlclose(st->nccl.lib);
  st->nccl = rccl_api();
}
const char *eng_error(void *user) { return ((eng_state *)user)->err.c_str(); }  // read on the calling thread, after every worker has been waited for
void *eng_ctx(void *, void *handle) { return ((eng_dev *)handle)->ctx; }

// ---------------------------------------------------------------- workers
struct worker {
  std::thread th;
  std::mutex mu;
  std::condition_variable cv;
  std::function<int()> job;
  bool has_job = false, done = true, quit = false;
  int rc = LAMD_OK;
  void loop() {
    std::unique_lock<std::mutex> lk(mu);
    for (;;) {
      cv.wait(lk, [&] { return has_job || quit; });
      if (quit) return;
      std::function<int()> j = std::move(job);
      has_job = false;
      lk.unlock();
      const int r = j();
      lk.lock();
      rc = r;
      done = true;
      cv.notify_all();
    }
  }
  void post(std::function<int()> j) {
    std::lock_guard<std::mutex> lk(mu);
    job = std::move(j);
    has_job = true;
    done = false;
    cv.notify_all();
  }
  int wait() {
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, [&] { return done; });
    return rc;
  }
};
struct devbufs {
  void *p[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  size_t cap[8] = {0, 0, 0, 0, 0, 0, 0, 0};
};
enum { B_A = 0, B_B, B_C, B_SEND, B_RECV, B_OFF, B_ROWBASE, B_IDS };

}  // namespace

struct lamd_multi {
  lamd_multi_backend be;
  eng_state eng;  // the engine back end's state (unused under a custom back end)
  int n = 0;
  std::vector<int> devices;
  std::vector<void *> handle;
  std::vector<devbufs> buf;
  std::vector<worker *> w;
  std::vector<uint8_t> h_recv;
  std::string err;
  std::mutex call_mu;  // one call at a time per lamd_multi (the devices' contexts are not thread-safe)
  size_t chunk_rows = 1u << 18;  // LAMD_MULTI_CHUNK: rows per H2D + verify piece inside a shard (the copy of piece k+1 runs under the kernels of piece k)
};

extern "C" int lamd_shard_bounds(size_t n_groups, const uint32_t *group_rows, int n_shards, size_t *bounds_groups, size_t *bounds_rows) {
  if (n_shards < 1 || !bounds_groups) return LAMD_ERR_ARG;
  // contiguous ranges of whole groups, balanced by ROWS: shard k ends at the first group boundary at or past k/n_shards of the rows
  // (the rule of lightning_amd/sharding.py shard_bounds(): tests require the two to agree cut for cut)
  size_t total = 0;
  if (group_rows)
    for (size_t g = 0; g < n_groups; g++) total += group_rows[g];
  else
    total = n_groups;
  size_t g = 0, rows = 0;
  bounds_groups[0] = 0;
  if (bounds_rows) bounds_rows[0] = 0;
  for (int k = 1; k <= n_shards; k++) {
    const size_t target = k == n_shards ? total : (size_t)(((unsigned __int128)total * (unsigned)k) / (unsigned)n_shards);
    while (g < n_groups && (rows < target || g == 0)) rows += group_rows ? group_rows[g++] : (g++, (size_t)1);
    if (k == n_shards) {
      while (g < n_groups) rows += group_rows ? group_rows[g++] : (g++, (size_t)1);
    }
    bounds_groups[k] = g;
    if (bounds_rows) bounds_rows[k] = rows;
  }
  return LAMD_OK;
}

static int multi_fail(lamd_multi *m, int rc, const char *what) {
  m->err = std::string(what) + ": " + (m->be.error ? m->be.error(m->be.user) : "error");
  if (m->be.user == &m->eng) m->eng.clear();  // the next call starts with an empty report
  return rc;
}
static int ensure_buf(lamd_multi *m, int i, int which, size_t bytes) {
  devbufs &b = m->buf[i];
  if (b.cap[which] >= bytes && b.p[which]) return LAMD_OK;
  if (b.p[which]) m->be.dev_free(m->be.user, m->handle[i], b.p[which]);
  const size_t want = bytes + bytes / 4 + 256;
  b.p[which] = m->be.dev_alloc(m->be.user, m->handle[i], want);
  b.cap[which] = b.p[which] ? want : 0;
  return b.p[which] ? LAMD_OK : LAMD_ERR_NOMEM;
}

extern "C" int lamd_multi_init_backend(lamd_multi **out, const int *devices, int n_devices, const lamd_multi_backend *be) {
  if (!out || n_devices < 1 || n_devices > 64 || !be) return LAMD_ERR_ARG;
  *out = nullptr;
  lamd_multi *m = new lamd_multi;
  m->be = *be;
  if (!m->be.user) m->be.user = &m->eng;
  m->n = n_devices;
  for (int i = 0; i < n_devices; i++) m->devices.push_back(devices ? devices[i] : i);
  m->handle.assign(n_devices, nullptr);
  m->buf.assign(n_devices, devbufs());
  if (const char *e = getenv("LAMD_MULTI_CHUNK")) m->chunk_rows = (size_t)atoll(e) < 64 ? 64 : (size_t)atoll(e);
  int rc = LAMD_OK;
  // With the real engine behind it, device i's context is created and its worker thread runs on the CPUs of the NUMA node the device hangs on
  // (lamd_device_numa_node; LAMD_MULTI_NUMA=0 leaves the threads where the scheduler puts them): the worker copies the caller's rows to its device.
  std::vector<int> node(n_devices, -1);
  std::vector<cpu_set_t> cpus(n_devices);
  const bool numa = m->be.user == &m->eng && !(getenv("LAMD_MULTI_NUMA") && atoi(getenv("LAMD_MULTI_NUMA")) == 0);
  cpu_set_t before;
  const bool have_before = sched_getaffinity(0, sizeof before, &before) == 0;
  for (int i = 0; i < n_devices && rc == LAMD_OK; i++) {
    if (numa) {
      const int nd = lamd_device_numa_node(m->devices[i]);
      if (nd >= 0 && lamd_node_cpus(nd, &cpus[i])) {
        node[i] = nd;
        sched_setaffinity(0, sizeof cpus[i], &cpus[i]);
      }
    }
    rc = m->be.dev_open(m->be.user, m->devices[i], &m->handle[i]);
    if (numa && have_before) sched_setaffinity(0, sizeof before, &before);
  }
  if (rc == LAMD_OK) rc = m->be.gather_open(m->be.user, m->handle.data(), n_devices);
  if (rc != LAMD_OK) {
    *out = m;  // so that lamd_multi_last_error() can say why; the caller still calls lamd_multi_shutdown()
    multi_fail(m, rc, "lamd_multi_init");
    return rc;
  }
  for (int i = 0; i < n_devices; i++) {
    worker *wk = new worker;
    const bool bind = node[i] >= 0;
    const cpu_set_t where = cpus[i];
    wk->th = std::thread([wk, bind, where] {
      if (bind) sched_setaffinity(0, sizeof where, &where);
      wk->loop();
    });
    m->w.push_back(wk);
  }
  *out = m;
  return LAMD_OK;
}
extern "C" int lamd_multi_init(lamd_multi **out, const int *devices, int n_devices) {
  lamd_multi_backend be;
  memset(&be, 0, sizeof be);
  be.dev_open = eng_open; be.dev_close = eng_close; be.dev_alloc = eng_alloc; be.dev_free = eng_free; be.h2d = eng_h2d; be.d2h = eng_d2h;
  be.verify_ecdsa = eng_ecdsa; be.verify_schnorr = eng_schnorr; be.sigcheck_gossip = eng_gossip;
  be.gather_open = eng_gather_open; be.all_gather = eng_all_gather; be.gather_close = eng_gather_close; be.error = eng_error; be.engine_ctx = eng_ctx;
  return lamd_multi_init_backend(out, devices, n_devices, &be);
}
extern "C" void lamd_multi_shutdown(lamd_multi *m) {
  if (!m) return;
  for (worker *wk : m->w) {
    { std::lock_guard<std::mutex> lk(wk->mu); wk->quit = true; wk->cv.notify_all(); }
    wk->th.join();
    delete wk;
  }
  if (m->be.gather_close) m->be.gather_close(m->be.user, m->handle.data(), m->n);
  for (int i = 0; i < m->n; i++) {
    if (!m->handle[i]) continue;
    for (int k = 0; k < 8; k++)
      if (m->buf[i].p[k]) m->be.dev_free(m->be.user, m->handle[i], m->buf[i].p[k]);
    m->be.dev_close(m->be.user, m->handle[i]);
  }
  delete m;
}
extern "C" const char *lamd_multi_last_error(const lamd_multi *m) { return m ? m->err.c_str() : "no lamd_multi"; }
extern "C" int lamd_multi_devices(const lamd_multi *m) { return m ? m->n : 0; }
extern "C" lamd_ctx *lamd_multi_ctx(lamd_multi *m, int i) {
  if (!m || i < 0 || i >= m->n || !m->be.engine_ctx || !m->handle[i]) return nullptr;
  return (lamd_ctx *)m->be.engine_ctx(m->be.user, m->handle[i]);
}

// runs shard_job(i) on device i's thread for every device with rows, all-gathers `pad` bytes per device, copies the vector to the host
// once and scatters the shards' verdicts into out[bounds_rows[i] ..)
static int run_sharded(lamd_multi *m, const std::vector<size_t> &bounds_rows, const std::function<int(int)> &shard_job, uint8_t *out) {
  const int n = m->n;
  size_t pad = 0;
  for (int i = 0; i < n; i++) pad = std::max(pad, bounds_rows[i + 1] - bounds_rows[i]);
  pad = (pad + 15) & ~(size_t)15;
  if (pad == 0) return LAMD_OK;
  int rc = LAMD_OK;
  for (int i = 0; i < n && rc == LAMD_OK; i++) {
    rc = ensure_buf(m, i, B_SEND, pad);
    if (rc == LAMD_OK) rc = ensure_buf(m, i, B_RECV, pad * (size_t)n);
  }
  if (rc != LAMD_OK) return multi_fail(m, rc, "device allocation");
  for (int i = 0; i < n; i++) m->w[i]->post([&shard_job, i] { return shard_job(i); });
  for (int i = 0; i < n; i++) {
    const int r = m->w[i]->wait();
    if (r != LAMD_OK && rc == LAMD_OK) rc = r;
  }
  if (rc != LAMD_OK) return multi_fail(m, rc, "shard");
  std::vector<void *> send(n), recv(n);
  for (int i = 0; i < n; i++) { send[i] = m->buf[i].p[B_SEND]; recv[i] = m->buf[i].p[B_RECV]; }
  if ((rc = m->be.all_gather(m->be.user, m->handle.data(), n, send.data(), recv.data(), pad)) != LAMD_OK) return multi_fail(m, rc, "all-gather");
  m->h_recv.resize(pad * (size_t)n);
  if ((rc = m->be.d2h(m->be.user, m->handle[0], m->h_recv.data(), recv[0], pad * (size_t)n)) != LAMD_OK) return multi_fail(m, rc, "D2H");
  for (int i = 0; i < n; i++) memcpy(out + bounds_rows[i], m->h_recv.data() + pad * (size_t)i, bounds_rows[i + 1] - bounds_rows[i]);
  return LAMD_OK;
}

static int verify_rows(lamd_multi *m, int kind, size_t n, const uint8_t *a32, const uint8_t *sig64, const uint8_t *key, size_t keylen, size_t keystride,
                       size_t group, uint8_t *ok) {
  if (!m || (n && (!a32 || !sig64 || !key || !ok)) || group == 0 || (kind == 0 && keylen != 33 && keylen != 65)) return LAMD_ERR_ARG;
  if (m->w.empty()) return LAMD_ERR_STATE;
  if (n == 0) return LAMD_OK;
  std::lock_guard<std::mutex> lk(m->call_mu);
  const size_t n_groups = (n + group - 1) / group;
  std::vector<size_t> bg(m->n + 1), br(m->n + 1);
  lamd_shard_bounds(n_groups, nullptr, m->n, bg.data(), nullptr);
  for (int i = 0; i <= m->n; i++) br[i] = std::min(n, bg[i] * group);
  auto job = [&](int i) -> int {
    const size_t lo = br[i], rows = br[i + 1] - br[i];
    if (!rows) return LAMD_OK;
    int rc;
    if ((rc = ensure_buf(m, i, B_A, rows * 32)) != LAMD_OK || (rc = ensure_buf(m, i, B_B, rows * 64)) != LAMD_OK ||
        (rc = ensure_buf(m, i, B_C, rows * keylen)) != LAMD_OK)
      return rc;
    uint8_t *d_a = (uint8_t *)m->buf[i].p[B_A], *d_b = (uint8_t *)m->buf[i].p[B_B], *d_c = (uint8_t *)m->buf[i].p[B_C], *d_ok = (uint8_t *)m->buf[i].p[B_SEND];
    std::vector<uint8_t> packed;
    for (size_t o = 0; o < rows; o += m->chunk_rows) {  // piece by piece: the (synchronous) copies of piece k+1 run under the kernels of piece k
      const size_t c = std::min(m->chunk_rows, rows - o);
      if ((rc = m->be.h2d(m->be.user, m->handle[i], d_a + o * 32, a32 + (lo + o) * 32, c * 32)) != LAMD_OK) return rc;
      if ((rc = m->be.h2d(m->be.user, m->handle[i], d_b + o * 64, sig64 + (lo + o) * 64, c * 64)) != LAMD_OK) return rc;
      const uint8_t *ksrc = key + (lo + o) * keystride;
      if (keystride != keylen) {  // the device copy is dense
        packed.resize(c * keylen);
        for (size_t r = 0; r < c; r++) memcpy(&packed[r * keylen], ksrc + r * keystride, keylen);
        ksrc = packed.data();
      }
      if ((rc = m->be.h2d(m->be.user, m->handle[i], d_c + o * keylen, ksrc, c * keylen)) != LAMD_OK) return rc;
      rc = kind == 0 ? m->be.verify_ecdsa(m->be.user, m->handle[i], c, d_a + o * 32, d_b + o * 64, d_c + o * keylen, keylen, keylen, d_ok + o)
                     : m->be.verify_schnorr(m->be.user, m->handle[i], c, d_a + o * 32, d_c + o * keylen, d_b + o * 64, d_ok + o);
      if (rc != LAMD_OK) return rc;
    }
    return LAMD_OK;
  };
  return run_sharded(m, br, job, ok);
}
extern "C" int lamd_multi_verify_ecdsa_batch(lamd_multi *m, size_t n, const uint8_t *hash32, const uint8_t *sig64, const uint8_t *pub, size_t publen,
                                             size_t pubstride, size_t group_rows, uint8_t *ok) {
  return verify_rows(m, 0, n, hash32, sig64, pub, publen, pubstride, group_rows ? group_rows : 1, ok);
}
extern "C" int lamd_multi_verify_schnorr_batch(lamd_multi *m, size_t n, const uint8_t *msg32, const uint8_t *xonly32, const uint8_t *sig64,
                                               size_t group_rows, uint8_t *ok) {
  return verify_rows(m, 1, n, msg32, sig64, xonly32, 32, 32, group_rows ? group_rows : 1, ok);
}

extern "C" int lamd_multi_sigcheck_gossip_batch(lamd_multi *m, size_t n, const uint8_t *msgs, const uint64_t *off, const uint8_t *node_ids33, int8_t *verdict) {
  if (!m || (n && (!msgs || !off || !verdict))) return LAMD_ERR_ARG;
  if (m->w.empty()) return LAMD_ERR_STATE;
  if (n == 0) return LAMD_OK;
  std::lock_guard<std::mutex> lk(m->call_mu);
  // signatures per message (4 for a channel_announcement, 1 otherwise: lamd_sigcheck_gossip_batch_device): shards are cut on MESSAGE boundaries
  // and balanced by signatures; the verdict vector that is gathered has one byte per message
  // ... balanced by COST: a channel_announcement is four signatures and four key parses, two of the signatures under bitcoin keys that never recur
  // (the per-signature ladder) -- ten channel_updates' worth, measured (lightning_amd/sharding.py GOSSIP_WEIGHT_*; balanced by message count the
  // first shard of a replay that starts with its announcements would carry a third of the job)
  std::vector<uint32_t> sigs(n), wts(n);
  for (size_t i = 0; i < n; i++) {
    const uint64_t len = off[i + 1] - off[i];
    sigs[i] = (len >= 2 && msgs[off[i]] == 1 && msgs[off[i] + 1] == 0) ? 4u : 1u;
    wts[i] = sigs[i] == 4u ? 10u : 1u;
    if (!node_ids33 && len >= 2 && msgs[off[i]] == 1 && msgs[off[i] + 1] == 2) {  // as lamd_sigcheck_gossip_batch refuses it
      m->err = "channel_update in batch but node_ids33 is NULL";
      return LAMD_ERR_ARG;
    }
  }
  // A batch whose MIX changes along its length -- a gossip replay is its channel_announcements (ten updates' worth each) followed by its
  // channel_updates -- is cut kind by kind: every run of one kind (announcement / everything else) is cut into m->n ranges and device i takes
  // range i of every run, so that every device holds the same mix (with one cut over the whole replay five devices of eight get nothing but
  // announcements and the call lasts as long as the slowest kind: lightning_amd/sharding.py segment_bounds, profiles/r06_shard_timeline_cold.txt).
  // A stream that interleaves its kinds (more than MAX_RUNS runs) is uniform already and is cut once.
  constexpr size_t MAX_RUNS = 4;
  std::vector<size_t> edge{0};
  for (size_t i = 1; i < n && edge.size() <= MAX_RUNS; i++)
    if ((sigs[i] == 4u) != (sigs[i - 1] == 4u)) edge.push_back(i);
  if (edge.size() > MAX_RUNS) edge.assign(1, 0);
  edge.push_back(n);
  const size_t n_seg = edge.size() - 1;
  std::vector<std::vector<size_t>> sb(n_seg, std::vector<size_t>(m->n + 1));  // sb[s][i] .. sb[s][i + 1]: device i's messages of run s
  for (size_t s = 0; s < n_seg; s++) {
    lamd_shard_bounds(edge[s + 1] - edge[s], wts.data() + edge[s], m->n, sb[s].data(), nullptr);
    for (int i = 0; i <= m->n; i++) sb[s][i] += edge[s];
  }
  std::vector<size_t> tot(m->n + 1, 0);  // the gathered vector is in DEVICE order (each device's ranges back to back); put in job order below
  for (int i = 0; i < m->n; i++) {
    tot[i + 1] = tot[i];
    for (size_t s = 0; s < n_seg; s++) tot[i + 1] += sb[s][i + 1] - sb[s][i];
  }
  auto job = [&](int i) -> int {
    const size_t cnt = tot[i + 1] - tot[i];
    if (!cnt) return LAMD_OK;
    // the device's ranges are laid back to back in its buffers -- bytes, offsets, node ids -- and verified by ONE engine call
    uint64_t bytes = 0, rows = 0;
    for (size_t s = 0; s < n_seg; s++) bytes += off[sb[s][i + 1]] - off[sb[s][i]];
    std::vector<uint64_t> rel(cnt + 1), rowbase(cnt + 1);
    int rc;
    if ((rc = ensure_buf(m, i, B_A, bytes + 64)) != LAMD_OK || (rc = ensure_buf(m, i, B_OFF, (cnt + 1) * 8)) != LAMD_OK ||
        (rc = ensure_buf(m, i, B_ROWBASE, (cnt + 1) * 8)) != LAMD_OK || (rc = ensure_buf(m, i, B_IDS, cnt * 33)) != LAMD_OK)
      return rc;
    size_t k = 0;
    uint64_t at = 0;
    for (size_t s = 0; s < n_seg; s++) {
      const size_t lo = sb[s][i], c = sb[s][i + 1] - lo;
      if (!c) continue;
      const uint64_t base = off[lo], len = off[lo + c] - base;
      for (size_t j = 0; j < c; j++, k++) {
        rel[k] = at + (off[lo + j] - base);
        rowbase[k] = rows;
        rows += sigs[lo + j];
      }
      if ((rc = m->be.h2d(m->be.user, m->handle[i], (uint8_t *)m->buf[i].p[B_A] + at, msgs + base, len)) != LAMD_OK) return rc;
      if (node_ids33 && (rc = m->be.h2d(m->be.user, m->handle[i], (uint8_t *)m->buf[i].p[B_IDS] + (k - c) * 33, node_ids33 + lo * 33, c * 33)) != LAMD_OK) return rc;
      at += len;
    }
    rel[cnt] = at;
    rowbase[cnt] = rows;
    if ((rc = m->be.h2d(m->be.user, m->handle[i], m->buf[i].p[B_OFF], rel.data(), (cnt + 1) * 8)) != LAMD_OK) return rc;
    if ((rc = m->be.h2d(m->be.user, m->handle[i], m->buf[i].p[B_ROWBASE], rowbase.data(), (cnt + 1) * 8)) != LAMD_OK) return rc;
    return m->be.sigcheck_gossip(m->be.user, m->handle[i], cnt, m->buf[i].p[B_A], m->buf[i].p[B_OFF], node_ids33 ? m->buf[i].p[B_IDS] : nullptr,
                                 m->buf[i].p[B_ROWBASE], (size_t)rows, m->buf[i].p[B_SEND]);
  };
  if (n_seg == 1) return run_sharded(m, tot, job, (uint8_t *)verdict);
  std::vector<uint8_t> by_dev(n);
  const int rc = run_sharded(m, tot, job, by_dev.data());
  if (rc != LAMD_OK) return rc;
  for (int i = 0; i < m->n; i++) {
    size_t at = tot[i];
    for (size_t s = 0; s < n_seg; s++) {
      const size_t c = sb[s][i + 1] - sb[s][i];
      if (c) memcpy(verdict + sb[s][i], by_dev.data() + at, c);
      at += c;
    }
  }
  return LAMD_OK;
}

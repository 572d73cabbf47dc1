// Batched gossip ingest (include/lightning_amd_gossipd.h): gossipd's receive path -- gossipd/gossipd.c:172-286 and
// gossipd/gossmap_manage.c:620-1342 of the reference -- with every signature of a drained queue decided by ONE device call.
//
// Structure (deliberately not the reference's): plan() walks the queue once and decides, from framing and from the state as it
// is before the batch, which (message, signer) pairs can ever reach a sigcheck_*() call; verify() sends those to the device;
// apply_*() then run the reference's control flow per message in arrival order with verdict look-ups where the reference calls
// sigcheck_*().  The warning / trace texts are the reference's format strings (file:line cited at each).
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <algorithm>
#include <atomic>
#include <climits>
#include <sys/resource.h>
#include <thread>
#include <condition_variable>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <random>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/lightning_amd_gossipd.h"
#include "../../include/cln_shim.h"
#include "verify_core.h"  // gossip_parse_frame(): the same framing rules the device applies

namespace {

using lamd::gossip_frame;
using lamd::gossip_parse_frame;
using lamd::GOSSIP_CANN;
using lamd::GOSSIP_CUPD;
using lamd::GOSSIP_NANN;
using lamd::u32;
using lamd::u64;

// byte vectors whose resize() leaves the new bytes UNINITIALISED (every caller fills what it grows): value-initialising the gossip_store image
// as it grows was a serial memset + page-fault pass over every byte later written anyway -- half of an update run's time
template <class T>
struct noinit_alloc : std::allocator<T> {
  template <class U> struct rebind { using other = noinit_alloc<U>; };
  noinit_alloc() = default;
  template <class U> noinit_alloc(const noinit_alloc<U> &) {}
  template <class U> void construct(U *p) noexcept { ::new ((void *)p) U; }
  template <class U, class... A> void construct(U *p, A &&...a) { ::new ((void *)p) U(std::forward<A>(a)...); }
};
typedef std::vector<u8, noinit_alloc<u8>> bytes;
// the gossip_store image: a byte buffer that grows by realloc() -- for a block of this size glibc moves the PAGES (mremap), where a std::vector
// allocates anew, copies every byte and takes the page faults of the copy again (47 MB copied to make room for a batch of updates: 20 ms)
struct image_buf {
  typedef u8 value_type;
  u8 *p = nullptr;
  size_t n = 0, cap = 0;
  image_buf() = default;
  image_buf(const image_buf &) = delete;
  image_buf &operator=(const image_buf &) = delete;
  ~image_buf() { free(p); }
  u8 *data() { return p; }
  const u8 *data() const { return p; }
  const u8 *begin() const { return p; }
  size_t size() const { return n; }
  size_t capacity() const { return cap; }
  void reserve(size_t want) {
    if (want <= cap) return;
    u8 *q = (u8 *)realloc(p, want);
    if (!q) throw std::bad_alloc();
    p = q;
    cap = want;
  }
  void resize(size_t want) {  // (new bytes uninitialised: every caller fills what it grows)
    if (want > cap) reserve(want + (want >> 2) + 4096);
    n = want;
  }
  void assign(size_t count, u8 v) { resize(count); memset(p, v, count); }
};
// a message by reference: bytes of the batch arena (alive until the batch has been applied) or of a waiting list's own copy
struct mview {
  const u8 *p;
  size_t n;
  mview() : p(nullptr), n(0) {}
  mview(const u8 *p_, size_t n_) : p(p_), n(n_) {}
  mview(const bytes &b) : p(b.data()), n(b.size()) {}  // NOLINT: implicit on purpose
  const u8 *data() const { return p; }
  size_t size() const { return n; }
  const u8 &operator[](size_t i) const { return p[i]; }
  const u8 *begin() const { return p; }
  const u8 *end() const { return p + n; }
};
struct nodeid {
  u8 k[33];
  bool operator<(const nodeid &o) const { return memcmp(k, o.k, 33) < 0; }
  bool operator==(const nodeid &o) const { return memcmp(k, o.k, 33) == 0; }
};

// ---- hashing for the maps.  Keyed (the key is drawn per ingest, as gossipd seeds its siphash tables): a peer that chooses
// short_channel_ids, node ids or whole messages cannot aim at one bucket.  Equality is always decided on the full content.
static inline u64 mix64(u64 h) {
  h ^= h >> 32; h *= 0xD6E8FEB86659FD93ull;
  h ^= h >> 32; h *= 0xD6E8FEB86659FD93ull;
  h ^= h >> 32;
  return h;
}
static u64 content_hash(u64 seed, const u8 *p, size_t n) {
  u64 h = seed ^ ((u64)n * 0x9E3779B97F4A7C15ull);
  while (n >= 8) {
    u64 w;
    memcpy(&w, p, 8);
    h = (h ^ w) * 0x9E3779B97F4A7C15ull;
    h ^= h >> 29;
    p += 8; n -= 8;
  }
  u64 w = 0;
  memcpy(&w, p, n);
  h = (h ^ w) * 0x9E3779B97F4A7C15ull;
  return mix64(h);
}
struct scid_hash {
  u64 seed;
  size_t operator()(u64 x) const { return (size_t)mix64(x ^ seed); }
};
struct nodeid_hash {
  u64 seed;
  size_t operator()(const nodeid &n) const { return (size_t)content_hash(seed, n.k, 33); }
};
// K -> V for the maps every gossip message consults (channels, announcements waiting for their txout, failed txouts by short_channel_id;
// nodes by node id): an open-addressing index (linear probing, 12 bytes per slot, load <= 1/2) over entries that live in fixed blocks -- an
// entry never moves (the plan keeps chan pointers across the apply pass), an insertion allocates nothing, and the slot a key WILL probe is
// known from its hash alone, so the replay can prefetch it for the messages ahead (prefetch()).  The interface is the part of
// std::unordered_map this file uses; concurrent readers are fine, writers are not.  Traits: tag(key, seed) = the 64-bit word the index stores
// (the key itself for a short_channel_id, a seeded hash for a node id), home(tag, seed) = where probing starts, exact = "equal tags mean equal keys".
struct scid_key_traits {
  static constexpr bool exact = true;
  static u64 tag(u64 k, u64) { return k; }
  static size_t home(u64 tag, u64 seed) { return (size_t)mix64(tag ^ seed); }
};
struct nodeid_key_traits {
  static constexpr bool exact = false;
  static u64 tag(const nodeid &k, u64 seed) { return content_hash(seed, k.k, 33); }
  static size_t home(u64 tag, u64) { return (size_t)tag; }
};
template <class K, class V, class TR>
struct stable_map {
  struct entry { K first; V second; };
  static constexpr u32 BLOCK = 4096, EMPTY = 0, TOMB = 0xFFFFFFFFu;
  u64 seed;
  std::vector<u64> ikey;
  std::vector<u32> islot;          // 0 = never used, TOMB = erased, else entry index + 1
  size_t mask = 0, live = 0, filled = 0;
  std::vector<entry *> blocks;
  std::vector<u8> alive;           // per entry index
  std::vector<u32> free_list;
  u32 top = 0;                     // entry indices handed out so far

  explicit stable_map(u64 seed_) : seed(seed_) { rehash(64); }
  stable_map(const stable_map &) = delete;
  stable_map &operator=(const stable_map &) = delete;
  ~stable_map() {
    for (u32 i = 0; i < top; i++)
      if (alive[i]) at(i)->~entry();
    for (entry *b : blocks) ::operator delete((void *)b);
  }
  entry *at(u32 i) const { return blocks[i / BLOCK] + (i % BLOCK); }
  void prefetch_t(u64 tag) const { const size_t h = TR::home(tag, seed) & mask; __builtin_prefetch(&ikey[h]); __builtin_prefetch(&islot[h]); }
  void prefetch(const K &k) const { prefetch_t(TR::tag(k, seed)); }
  struct iterator {
    const stable_map *m;
    u32 i;  // entry index; m->top = end
    entry *operator->() const { return m->at(i); }
    entry &operator*() const { return *m->at(i); }
    bool operator==(const iterator &o) const { return i == o.i; }
    bool operator!=(const iterator &o) const { return i != o.i; }
    iterator &operator++() { do { i++; } while (i < m->top && !m->alive[i]); return *this; }
  };
  iterator end() const { return iterator{this, top}; }
  iterator begin() const { iterator it{this, 0}; if (top && !alive[0]) ++it; return it; }
  // index slot of k, or of the place it would be inserted at (first tombstone on the way, else the empty slot that ended the probe)
  bool locate(const K &k, u64 tag, size_t *slot) const {
    size_t tomb = (size_t)-1;
    for (size_t h = TR::home(tag, seed) & mask;; h = (h + 1) & mask) {
      const u32 s = islot[h];
      if (s == EMPTY) { *slot = tomb != (size_t)-1 ? tomb : h; return false; }
      if (s == TOMB) { if (tomb == (size_t)-1) tomb = h; continue; }
      if (ikey[h] == tag && (TR::exact || at(s - 1)->first == k)) { *slot = h; return true; }
    }
  }
  iterator find_t(const K &k, u64 tag) const {
    size_t h;
    return locate(k, tag, &h) ? iterator{this, islot[h] - 1} : end();
  }
  iterator find(const K &k) const { return find_t(k, TR::tag(k, seed)); }
  // second stage of a look-ahead: the index slot is (by now) in cache, fetch the entry it names
  void prefetch_entry_t(const K &k, u64 tag) const {
    size_t h;
    if (locate(k, tag, &h)) __builtin_prefetch(at(islot[h] - 1));
  }
  void prefetch_entry(const K &k) const { prefetch_entry_t(k, TR::tag(k, seed)); }
  size_t count(const K &k) const { return find(k).i != top; }
  size_t size() const { return live; }
  bool empty() const { return live == 0; }
  void rehash(size_t slots) {
    size_t m = 64;
    while (m < slots) m <<= 1;
    std::vector<u64> k2(m, 0);
    std::vector<u32> s2(m, EMPTY);
    const size_t mk = m - 1;
    for (size_t h = 0; h < islot.size(); h++) {
      const u32 s = islot[h];
      if (s == EMPTY || s == TOMB) continue;
      size_t g = TR::home(ikey[h], seed) & mk;
      while (s2[g] != EMPTY) g = (g + 1) & mk;
      k2[g] = ikey[h];
      s2[g] = s;
    }
    ikey.swap(k2);
    islot.swap(s2);
    mask = mk;
    filled = live;
  }
  void reserve(size_t n) { if (2 * (n + 1) > mask + 1) rehash(2 * (n + 1)); }
  template <class... A>
  std::pair<iterator, bool> emplace(const K &k, A &&...a) { return emplace_t(k, TR::tag(k, seed), std::forward<A>(a)...); }
  template <class... A>
  std::pair<iterator, bool> emplace_t(const K &k, u64 tag, A &&...a) {
    if (2 * (filled + 1) > mask + 1) rehash(live * 4 + 64);   // load <= 1/2 counting tombstones
    size_t h;
    if (locate(k, tag, &h)) return {iterator{this, islot[h] - 1}, false};
    u32 i;
    if (!free_list.empty()) { i = free_list.back(); free_list.pop_back(); }
    else {
      i = top++;
      if (i / BLOCK >= blocks.size()) blocks.push_back((entry *)::operator new(sizeof(entry) * BLOCK));
      alive.push_back(0);
    }
    ::new ((void *)at(i)) entry{k, V(std::forward<A>(a)...)};
    alive[i] = 1;
    if (islot[h] == EMPTY) filled++;
    ikey[h] = tag;
    islot[h] = i + 1;
    live++;
    return {iterator{this, i}, true};
  }
  V &operator[](const K &k) { return emplace(k).first->second; }
  size_t erase(const K &k) { return erase_t(k, TR::tag(k, seed)); }
  size_t erase_t(const K &k, u64 tag) {
    size_t h;
    if (!locate(k, tag, &h)) return 0;
    const u32 i = islot[h] - 1;
    islot[h] = TOMB;
    at(i)->~entry();
    alive[i] = 0;
    free_list.push_back(i);
    live--;
    return 1;
  }
  void erase(const iterator &it) { const K k = it->first; erase(k); }
};
// NS independent stable_maps behind the same interface, a key's shard taken from the top bits of its probe start: single-threaded code does not
// notice; a parallel pass gives every worker the keys of its own shards (worker w owns shard s when s % workers == w), so workers never write
// the same map -- and since a shard sees its keys in arrival order, the maps end up exactly as the one-by-one replay leaves them.
template <class K, class V, class TR, unsigned NS = 16>
struct sharded_map {
  typedef stable_map<K, V, TR> shard_t;
  typedef typename shard_t::entry entry;
  u64 seed;
  std::vector<std::unique_ptr<shard_t>> s;
  explicit sharded_map(u64 seed_) : seed(seed_) { for (unsigned i = 0; i < NS; i++) s.emplace_back(new shard_t(seed_)); }
  static constexpr unsigned shards() { return NS; }
  unsigned shard_t_of(u64 tag) const { return (unsigned)((u64)TR::home(tag, seed) >> 58) % NS; }
  unsigned shard_of(const K &k) const { return shard_t_of(TR::tag(k, seed)); }
  shard_t &shard(unsigned i) { return *s[i]; }
  const shard_t &shard(unsigned i) const { return *s[i]; }
  struct iterator {
    const sharded_map *m;
    u32 sh, i;  // sh == NS: end
    entry *operator->() const { return m->s[sh]->at(i); }
    entry &operator*() const { return *m->s[sh]->at(i); }
    bool operator==(const iterator &o) const { return sh == o.sh && i == o.i; }
    bool operator!=(const iterator &o) const { return !(*this == o); }
    void settle() {  // onto the next live entry at or after (sh, i)
      while (sh < NS) {
        const shard_t &t = *m->s[sh];
        while (i < t.top && !t.alive[i]) i++;
        if (i < t.top) return;
        sh++;
        i = 0;
      }
      i = 0;
    }
    iterator &operator++() { i++; settle(); return *this; }
  };
  iterator end() const { return iterator{this, NS, 0}; }
  iterator begin() const { iterator it{this, 0, 0}; it.settle(); return it; }
  iterator find(const K &k) const {
    const u64 tag = TR::tag(k, seed);
    const unsigned sh = shard_t_of(tag);
    const auto it = s[sh]->find_t(k, tag);
    return it == s[sh]->end() ? end() : iterator{this, sh, it.i};
  }
  size_t count(const K &k) const { return find(k).sh != NS; }
  size_t size() const { size_t n = 0; for (const auto &t : s) n += t->size(); return n; }
  bool empty() const { for (const auto &t : s) if (!t->empty()) return false; return true; }
  void reserve(size_t n) { for (auto &t : s) t->reserve(n / NS + n / (4 * NS) + 16); }
  void prefetch(const K &k) const { const u64 tag = TR::tag(k, seed); s[shard_t_of(tag)]->prefetch_t(tag); }
  void prefetch_entry(const K &k) const { const u64 tag = TR::tag(k, seed); s[shard_t_of(tag)]->prefetch_entry_t(k, tag); }
  template <class... A>
  std::pair<iterator, bool> emplace(const K &k, A &&...a) {
    const u64 tag = TR::tag(k, seed);
    const unsigned sh = shard_t_of(tag);
    const auto r = s[sh]->emplace_t(k, tag, std::forward<A>(a)...);
    return {iterator{this, sh, r.first.i}, r.second};
  }
  V &operator[](const K &k) { return emplace(k).first->second; }
  size_t erase(const K &k) { const u64 tag = TR::tag(k, seed); return s[shard_t_of(tag)]->erase_t(k, tag); }
  void erase(const iterator &it) { const K k = it->first; erase(k); }
};
template <class V> using scid_map = sharded_map<u64, V, scid_key_traits>;

// a (message, signer) pair by reference: the bytes live in the batch / the waiting lists for as long as a map holds the key
struct msgkey {
  const u8 *m;
  size_t len;
  const u8 *signer;  // 33 bytes or nullptr
  u64 h;
};
struct msgkey_hash {
  size_t operator()(const msgkey &k) const { return (size_t)k.h; }
};
struct msgkey_eq {
  bool operator()(const msgkey &a, const msgkey &b) const {
    return a.h == b.h && a.len == b.len && (a.signer != nullptr) == (b.signer != nullptr) && memcmp(a.m, b.m, a.len) == 0 &&
           (!a.signer || memcmp(a.signer, b.signer, 33) == 0);
  }
};
typedef std::unordered_map<msgkey, int, msgkey_hash, msgkey_eq> verdict_map;

std::string hexs(const u8 *p, size_t n) {
  static const char *d = "0123456789abcdef";
  std::string s;
  s.reserve(2 * n);
  for (size_t i = 0; i < n; i++) { s.push_back(d[p[i] >> 4]); s.push_back(d[p[i] & 15]); }
  return s;
}
std::string hexs(const mview &b) { return hexs(b.data(), b.size()); }
u64 be64(const u8 *p) { u64 v = 0; for (int i = 0; i < 8; i++) v = (v << 8) | p[i]; return v; }
u32 be32(const u8 *p) { return ((u32)p[0] << 24) | ((u32)p[1] << 16) | ((u32)p[2] << 8) | p[3]; }
u32 be16(const u8 *p) { return ((u32)p[0] << 8) | p[1]; }
// fmt_short_channel_id (bitcoin/short_channel_id.c:56-62)
std::string fmt_scid(u64 scid) {
  char b[64];
  snprintf(b, sizeof b, "%dx%dx%d", (int)(scid >> 40), (int)((scid >> 16) & 0xFFFFFF), (int)(scid & 0xFFFF));
  return b;
}
u32 scid_blocknum(u64 scid) { return (u32)(scid >> 40); }
// bitcoin/short_channel_id.h:82-87, ANNOUNCE_MIN_DEPTH 6
bool scid_depth_announceable(u64 scid, u32 height) { return (u64)scid_blocknum(scid) + 6 - 1 <= height; }
// fmt_secp256k1_ecdsa_signature prints the DER form (bitcoin/signature.c:325-335)
std::string der_hex(const u8 sig64[64]) {
  u8 out[72];
  size_t n = 2;
  for (int h = 0; h < 2; h++) {
    const u8 *v = sig64 + 32 * h;
    size_t skip = 0;
    while (skip < 31 && v[skip] == 0) skip++;
    const bool pad = v[skip] & 0x80;
    out[n++] = 0x02;
    out[n++] = (u8)(32 - skip + (pad ? 1 : 0));
    if (pad) out[n++] = 0;
    memcpy(out + n, v + skip, 32 - skip);
    n += 32 - skip;
  }
  out[0] = 0x30;
  out[1] = (u8)(n - 2);
  return hexs(out, n);
}
const u8 ORDER_N[32] = {0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFE,
                        0xBA, 0xAE, 0xDC, 0xE6, 0xAF, 0x48, 0xA0, 0x3B, 0xBF, 0xD2, 0x5E, 0x8C, 0xD0, 0x36, 0x41, 0x41};
bool sig_in_range(const u8 *sig64) { return memcmp(sig64, ORDER_N, 32) < 0 && memcmp(sig64 + 32, ORDER_N, 32) < 0; }  // wire/fromwire.c:196

// gossipd/sigcheck.c:21-26,52-113,150-157: the text for "first bad signature = which" (1-based)
std::string sigcheck_text(u32 type, int which, const mview &m) {
  const size_t off = type == GOSSIP_CANN ? 258 : 66;
  struct sha256_double h;
  sha256_double(&h, m.data() + off, m.size() - off);
  static const char *names[4] = {"Bad node_signature_1", "Bad node_signature_2", "Bad bitcoin_signature_1", "Bad bitcoin_signature_2"};
  const char *what = type == GOSSIP_CANN ? names[which - 1] : "Bad signature for";
  const u8 *sig = m.data() + 2 + (type == GOSSIP_CANN ? 64 * (which - 1) : 0);
  const char *kind = type == GOSSIP_CANN ? "channel_announcement" : (type == GOSSIP_CUPD ? "channel_update" : "node_announcement");
  return std::string(what) + " " + der_hex(sig) + " hash " + hexs(h.sha.u.u8, 32) + " on " + kind + " " + hexs(m);
}

// common/wireaddr.c:30-68,858-891: does fromwire_wireaddr_array() accept the `addresses` field?
bool wireaddrs_ok(const u8 *p, size_t len) {
  while (len) {
    const u8 type = p[0];
    p++; len--;
    size_t alen;
    switch (type) {
      case 1: alen = 4; break;
      case 2: alen = 16; break;
      case 3: alen = 10; break;
      case 4: alen = 35; break;
      case 5:
        if (len < 1) return false;
        alen = p[0];
        p++; len--;
        break;
      default: return true;  // unknown type: stop there
    }
    if (len < alen + 2) return false;
    p += alen + 2; len -= alen + 2;
  }
  return true;
}

// splits [0, n) over a few short-lived threads (the planning pass and the staging copy of a large batch: framing, range checks,
// content hashes, P2WSH script hashes -- per-message work that reads the maps but never writes them)
static unsigned host_threads() {
  static const unsigned t = [] {
    const char *e = getenv("LAMD_INGEST_THREADS");
    unsigned v = e ? (unsigned)atoi(e) : std::min(16u, std::thread::hardware_concurrency());   // (default: at most 16; LAMD_INGEST_THREADS may ask for up to 64)
    return v < 1 ? 1u : (v > 64 ? 64u : v);
  }();
  return t;
}
// A small persistent pool: the ingest's parallel passes are short (a few ms over 10^5 messages) and come in quick succession -- five per run
// of channel_updates -- so spawning and joining std::threads for each (30-50 us apiece) was a sizeable part of them (VERDICT r03 "weak" 14).
// Workers spin briefly on the generation counter before they sleep, so the passes of one batch find them awake.  One job at a time per
// pool; the planning stage that runs under an apply pass has a pool of its own (lamd_gossipd::pool_bg).
class thread_pool {
 public:
  explicit thread_pool(unsigned workers) {
    for (unsigned i = 0; i < workers; i++) th_.emplace_back([this, i] { loop(i + 1); });
  }
  ~thread_pool() {
    { std::lock_guard<std::mutex> lk(mu_); quit_ = true; gen_.fetch_add(1, std::memory_order_release); }
    cv_.notify_all();
    for (auto &t : th_) t.join();
  }
  unsigned size() const { return (unsigned)th_.size() + 1; }
  // f(t) for t in [0, n) on n threads (the caller runs t = 0); n <= size()
  void run(unsigned n, const std::function<void(unsigned)> &f) {
    if (n <= 1 || th_.empty()) { for (unsigned t = 0; t < n; t++) f(t); return; }
    job_ = &f;
    njob_ = n;
    // EVERY worker acknowledges every generation (those with id >= n run nothing): no worker can be a generation behind when the next job is
    // posted, so job_ / njob_ are stable for as long as anybody may read them
    pending_.store((unsigned)th_.size(), std::memory_order_relaxed);
    { std::lock_guard<std::mutex> lk(mu_); gen_.fetch_add(1, std::memory_order_release); }
    cv_.notify_all();
    f(0);
    for (unsigned spins = 0; pending_.load(std::memory_order_acquire) != 0; spins++) {
      if (spins < 20000) {
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
      } else {
        std::unique_lock<std::mutex> lk(mu_);
        done_cv_.wait(lk, [&] { return pending_.load(std::memory_order_acquire) == 0; });
      }
    }
  }

 private:
  void loop(unsigned id) {
    u64 seen = 0;
    for (;;) {
      u64 g = gen_.load(std::memory_order_acquire);
      for (unsigned spins = 0; g == seen && spins < 20000; spins++) {  // ~100 us awake between the passes of one batch
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
        g = gen_.load(std::memory_order_acquire);
      }
      if (g == seen) {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return gen_.load(std::memory_order_acquire) != seen; });
        g = gen_.load(std::memory_order_acquire);
      }
      seen = g;
      if (quit_) return;
      if (id < njob_) (*job_)(id);
      if (pending_.fetch_sub(1, std::memory_order_acq_rel) == 1) {
        std::lock_guard<std::mutex> lk(mu_);
        done_cv_.notify_all();
      }
    }
  }
  std::vector<std::thread> th_;
  std::mutex mu_;
  std::condition_variable cv_, done_cv_;
  std::atomic<u64> gen_{0};
  std::atomic<unsigned> pending_{0};
  const std::function<void(unsigned)> *job_ = nullptr;
  unsigned njob_ = 0;
  bool quit_ = false;
};
template <class F>
static void parallel_for(thread_pool *pool, size_t n, size_t min_per_thread, F f, unsigned max_threads = 0) {
  unsigned t = pool ? pool->size() : 1;
  if (max_threads && max_threads < t) t = max_threads;
  if (n / (min_per_thread ? min_per_thread : 1) < t) t = (unsigned)(n / (min_per_thread ? min_per_thread : 1));
  if (t <= 1) { f((size_t)0, n); return; }
  const size_t step = (n + t - 1) / t;
  pool->run(t, [&](unsigned k) { f(std::min(n, k * step), std::min(n, (k + 1) * step)); });
}

// first touch of fresh memory by ALL cores: a page fault costs 0.5-2 us (more under a hypervisor) and a flood grows the store image, the work
// buffers and the record list by hundreds of megabytes -- taken one after the other by the thread that happens to write first, the faults
// of a 2 M-message batch are several tenths of a second; page faults on different pages proceed in parallel.  The bytes are not yet part of
// any object (the caller constructs over them afterwards).
static void prefault(thread_pool *pool, void *p, size_t nbytes) {
  if (nbytes < ((size_t)4 << 20)) return;
  u8 *b = (u8 *)p;
  parallel_for(pool, nbytes >> 12, 512, [&](size_t lo, size_t hi) {
    for (size_t i = lo; i < hi; i++) ((volatile u8 *)b)[i << 12] = 0;
  });
}
// grow a vector's capacity to at least `want` elements, the new storage touched by all cores
template <class VEC>
static void reserve_prefaulted(thread_pool *pool, VEC &v, size_t want) {
  if (v.capacity() >= want) return;
  v.reserve(want + (want >> 3));
  prefault(pool, (void *)(v.data() + v.size()), (v.capacity() - v.size()) * sizeof(typename VEC::value_type));
}
// the expected scriptpubkey of a channel_announcement's funding output: always OP_0 <32 bytes> -- inline, no heap block to miss on
struct p2wsh_spk {
  u8 b[34];
  void resize(size_t) {}
  size_t size() const { return 34; }
  const u8 *data() const { return b; }
  u8 &operator[](size_t i) { return b[i]; }
};
struct pending_cannounce {  // gossmap_manage.c:36-45
  bytes msg;
  bool has_src;
  nodeid src;
  nodeid node[2];
  p2wsh_spk spk;
};
struct pending_cupdate {  // :47-62
  u64 scid;
  u8 mflags, cflags;
  u32 cltv, fee_base, fee_ppm, timestamp;
  u64 hmin, hmax;
  bytes update;
  bool has_src;
  nodeid src;
};
struct pending_nannounce {  // :64-70
  nodeid id;
  u32 timestamp;
  bytes msg;
  bool has_src;
  nodeid src;
};
struct chan {
  nodeid node[2];
  u64 cann_rec;
  bool set[2];
  u64 cupd_rec[2];
  bool dying = false;   // GOSSIP_STORE_DYING_BIT on its announcement (gossmap_chan_is_dying)
  // apply_cupd_run(): the batch index of the last channel_update of the run accepted for each direction (NONE outside a run)
  u32 run_last[2] = {0xFFFFFFFFu, 0xFFFFFFFFu};
};
struct node {
  u32 nchans;
  bool announced;
  u64 nann_rec;
  u32 run_last = 0xFFFFFFFFu;  // apply_nann_run(): the run index of the last node_announcement of the run accepted for this node (RUN_NONE outside a run)
  std::vector<u64> scids;  // its channels (gossmap_nth_chan): remove_channel() walks them
};
// one record of the store; `off` = offset of its MESSAGE in the gossip_store image (header at off - 12): what gossip_store_add()
// returns in the reference (gossipd/gossip_store.c:481-511, "by gossmap convention, offset is *after* hdr")
struct record { u32 type, timestamp; bool deleted; u64 off; u32 len; };
struct chan_dying { u64 scid; u32 deadline; u64 rec; };  // gossmap_manage.c struct chan_dying

// ---- gossip_store file format (common/gossip_store.h:15-59, gossipd/gossip_store.c:49-81)
constexpr u32 GS_DELETED = 0x8000, GS_COMPLETED = 0x2000, GS_DYING = 0x0800;
constexpr u32 WIRE_GS_CHANNEL_AMOUNT = 4101, WIRE_GS_DELETE_CHAN = 4103, WIRE_GS_CHAN_DYING = 4106, WIRE_GS_UUID = 4107;
// ccan/crc32c: CRC-32C (Castagnoli), `crc` = the CRC so far (the store seeds it with the record's timestamp)
static u32 crc32c_table[8][256];
static bool crc32c_ready = false;
static void crc32c_init() {
  for (u32 n = 0; n < 256; n++) {
    u32 c = n;
    for (int k = 0; k < 8; k++) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
    crc32c_table[0][n] = c;
  }
  for (u32 n = 0; n < 256; n++) {
    u32 c = crc32c_table[0][n];
    for (int k = 1; k < 8; k++) {
      c = crc32c_table[0][c & 0xFF] ^ (c >> 8);
      crc32c_table[k][n] = c;
    }
  }
  crc32c_ready = true;
}
static u32 crc32c_sw(u32 crc, const u8 *p, size_t len) {
  if (!crc32c_ready) crc32c_init();
  crc = ~crc;
  while (len >= 8) {
    u64 w;
    memcpy(&w, p, 8);
    w ^= crc;
    crc = crc32c_table[7][w & 0xFF] ^ crc32c_table[6][(w >> 8) & 0xFF] ^ crc32c_table[5][(w >> 16) & 0xFF] ^ crc32c_table[4][(w >> 24) & 0xFF] ^
          crc32c_table[3][(w >> 32) & 0xFF] ^ crc32c_table[2][(w >> 40) & 0xFF] ^ crc32c_table[1][(w >> 48) & 0xFF] ^ crc32c_table[0][w >> 56];
    p += 8; len -= 8;
  }
  while (len--) crc = crc32c_table[0][(crc ^ *p++) & 0xFF] ^ (crc >> 8);
  return ~crc;
}
#if defined(__x86_64__)
__attribute__((target("sse4.2"))) static u32 crc32c_hw(u32 crc, const u8 *p, size_t len) {
  u64 c = (u32)~crc;
  while (len >= 8) {
    u64 w;
    memcpy(&w, p, 8);
    c = __builtin_ia32_crc32di(c, w);
    p += 8; len -= 8;
  }
  u32 c32 = (u32)c;
  while (len--) c32 = __builtin_ia32_crc32qi(c32, *p++);
  return ~c32;
}
static u32 crc32c(u32 crc, const u8 *p, size_t len) {
  static const bool hw = __builtin_cpu_supports("sse4.2");
  return hw ? crc32c_hw(crc, p, len) : crc32c_sw(crc, p, len);
}
#else
static u32 crc32c(u32 crc, const u8 *p, size_t len) { return crc32c_sw(crc, p, len); }
#endif
void put_be16(u8 *p, u32 v) { p[0] = (u8)(v >> 8); p[1] = (u8)v; }
void put_be32(u8 *p, u32 v) { p[0] = (u8)(v >> 24); p[1] = (u8)(v >> 16); p[2] = (u8)(v >> 8); p[3] = (u8)v; }
void put_be64(u8 *p, u64 v) { for (int i = 0; i < 8; i++) p[i] = (u8)(v >> (56 - 8 * i)); }

struct queued { mview msg; bool has_src; nodeid src; };
struct qent { u64 off; u32 len; bool has_src; nodeid src; };  // a queued message inside the arena

// one message of a batch as plan() saw it
struct planned {
  u32 type;
  bool malformed;     // framing / signature range: decided on the host
  bool addrs_bad;     // node_announcement only
  int slot;           // verification slot with the signer the plan expects (-1: none)
  int keyslot;        // channel_announcement whose bitcoin keys only are parsed (-1: none)
  u64 scid;
  // filled by the parallel pass: what the serial pass needs to register the slot without touching the message again
  bool want_slot, want_keys;
  const nodeid *signer;   // the signer the plan expects (a channel's node, the relaying peer, or nullptr: the message names its own)
  u64 h;                  // vkey(message, signer).h
  u8 spk[32];             // channel_announcement: SHA256 of the 2-of-2 script (the P2WSH program)
  bool spk_set;
  pending_cannounce *pre = nullptr;   // channel_announcement the plan expects to wait for its txout: message copy, node ids and P2WSH program built by the
                                      // PARALLEL pass (three allocations and a 432-byte copy less per message in the serial replay); owned until moved from
  chan *pc;               // channel_update: the channel the plan found (nullptr: none).  chans gains and loses no entry while a batch is applied
                          // (channels appear in txout_reply, disappear in new_block / prune: none of them runs inside process()), so the
                          // pointer is what a lookup during the apply pass would return
};

}  // namespace

struct ingest_stage;
struct lamd_gossipd {
  lamd_ctx *ctx = nullptr;
  lamd_gossipd_config cfg;
  lamd_gossipd_event_fn on_event = nullptr;
  void *user = nullptr;
  lamd_gossipd_sigcheck_fn be_sig = nullptr;
  lamd_gossipd_keyparse_fn be_key = nullptr;
  void *be_user = nullptr;
  lamd_gossipd_stats st;

  const u64 seed = ((u64)std::random_device{}() << 32) ^ std::random_device{}() ^ 0x6C616D6467737064ull;
  // connectd's queue: the messages sit back to back in ONE arena (a push is an append, a batch push one memcpy)
  bytes qarena;
  std::vector<qent> queue;
  scid_map<chan> chans{seed};
  sharded_map<nodeid, node, nodeid_key_traits> nodes{seed};
  scid_map<pending_cannounce> pending_ann{seed};
  std::map<u64, pending_cannounce> early_ann;  // ordered: new_block walks early_ann by ascending scid
  std::vector<pending_cupdate> pending_cupdates, early_cupdates;
  std::vector<pending_nannounce> pending_nannounces;
  scid_map<bool> txout_failures{seed};
  std::vector<record> store;
  // the gossip_store file as gossip_store.c would hold it: version byte, (v16: the uuid record), then gossip_hdr + message per
  // record -- flags / crc / timestamp rewritten in place by del / set_timestamp / set_flag exactly as the reference pwrite()s them
  image_buf image;
  std::vector<chan_dying> dying_channels;

  // verdicts of the batch being applied: (message bytes, signer) -> verdict.  The keys refer to the bytes, they do not own them:
  // the map is emptied (drop_verdicts) before the batch / the lists it was filled from go away
  verdict_map verdicts;
  struct slotlist;
  const slotlist *cur_sl = nullptr;
  std::vector<int8_t> cur_v;

  u64 now() const { return cfg.now ? cfg.now : (u64)time(nullptr); }

  // ---- events
  void emit(lamd_gossipd_event &ev) { if (on_event) on_event(user, &ev); }
  void ev_text(int kind, bool has_peer, const nodeid *peer, const std::string &text) {
    if (!on_event) return;   // (no listener: nothing to build)
    lamd_gossipd_event ev;
    memset(&ev, 0, sizeof ev);
    ev.kind = kind;
    ev.has_peer = has_peer;
    if (has_peer) memcpy(ev.peer, peer->k, 33);
    ev.text = text.c_str();
    emit(ev);
  }
  void ev_scid(int kind, bool has_peer, const nodeid *peer, u64 scid) {
    if (!on_event) return;   // (no listener: nothing to build)
    lamd_gossipd_event ev;
    memset(&ev, 0, sizeof ev);
    ev.kind = kind;
    ev.has_peer = has_peer;
    if (has_peer) memcpy(ev.peer, peer->k, 33);
    ev.scid = scid;
    emit(ev);
  }
  void warning(bool has_peer, const nodeid *peer, const std::string &text) { ev_text(LAMD_GEV_WARNING, has_peer, peer, text); }
  // gossmap_manage.c:576-579
  void bad_gossip(bool has_peer, const nodeid *peer, const std::string &text) { if (on_event) ev_text(LAMD_GEV_TRACE, has_peer, peer, "Bad gossip order: " + text); }
  // :582-597
  void peer_warning(bool has_peer, const nodeid *peer, const std::string &text) {
    bad_gossip(has_peer, peer, text);
    if (has_peer) warning(true, peer, text);
  }
  void good_gossip(bool has_peer, const nodeid *peer) {
    if (!has_peer) return;  // gossipd.c:70-71
    if (!on_event) return;   // (no listener: nothing to build)
    lamd_gossipd_event ev;
    memset(&ev, 0, sizeof ev);
    ev.kind = LAMD_GEV_GOOD_GOSSIP;
    ev.has_peer = 1;
    memcpy(ev.peer, peer->k, 33);
    emit(ev);
  }
  // ---- store: records + the byte image of the file (append_msg, gossip_store.c:49-81)
  void store_write_event(u64 off, size_t len) {
    if (!cfg.emit_store_writes || !on_event) return;
    lamd_gossipd_event ev;
    memset(&ev, 0, sizeof ev);
    ev.kind = LAMD_GEV_STORE_WRITE;
    ev.values[0] = off;
    ev.data = image.data() + off;
    ev.len = len;
    emit(ev);
  }
  u64 store_append(u32 type, u32 timestamp, const u8 *data, size_t len) {
    const u64 hdr = image.size();
    image.resize(hdr + 12 + len);
    u8 *h = image.data() + hdr;
    put_be16(h, GS_COMPLETED);  // (the reference writes flags = 0 and then the completed bit as a one-byte pwrite: same bytes)
    put_be16(h + 2, (u32)len);
    put_be32(h + 4, crc32c(timestamp, data, len));
    put_be32(h + 8, timestamp);
    memcpy(h + 12, data, len);
    store.push_back(record{type, timestamp, false, hdr + 12, (u32)len});
    store_write_event(hdr, 12 + len);
    return store.size() - 1;
  }
  void store_init() {
    const u8 ver = cfg.store_version ? cfg.store_version : 16;  // GOSSIP_STORE_VER, gossip_store.c:24
    image.assign(1, ver);
    if ((ver & 0x1F) >= 16) {  // "v16 add uuid field" (:98): new_uuid_record, :186-197
      u8 m[34];
      put_be16(m, WIRE_GS_UUID);
      memcpy(m + 2, cfg.store_uuid, 32);
      store_append(WIRE_GS_UUID, 0, m, sizeof m);
    }
  }
  u64 store_add(u32 type, u32 timestamp, const u8 *data, size_t len) {
    const u64 idx = store_append(type, timestamp, data, len);
    if (!on_event) return idx;   // (no listener: nothing to build)
    lamd_gossipd_event ev;
    memset(&ev, 0, sizeof ev);
    ev.kind = LAMD_GEV_STORE_ADD;
    ev.index = idx;
    ev.type = type;
    ev.timestamp = timestamp;
    ev.values[0] = store[idx].off;
    ev.data = data;
    ev.len = len;
    emit(ev);
    return idx;
  }
  void store_or_flag(u64 idx, u32 flag) {  // gossip_store_set_flag, :572-593
    u8 *h = image.data() + store[idx].off - 12;
    put_be16(h, (((u32)h[0] << 8) | h[1]) | flag);
    store_write_event(store[idx].off - 12, 12);
  }
  void store_del(u64 idx) {  // gossip_store_del, :622-638: a channel_announcement takes its amount record (the next one) with it
    store[idx].deleted = true;
    store_or_flag(idx, GS_DELETED);
    if (store[idx].type == GOSSIP_CANN && idx + 1 < store.size() && store[idx + 1].type == WIRE_GS_CHANNEL_AMOUNT) {
      store[idx + 1].deleted = true;
      store_or_flag(idx + 1, GS_DELETED);
    }
    if (!on_event) return;   // (no listener: nothing to build)
    lamd_gossipd_event ev;
    memset(&ev, 0, sizeof ev);
    ev.kind = LAMD_GEV_STORE_DEL;
    ev.index = idx;
    ev.type = store[idx].type;
    ev.values[0] = store[idx].off;
    emit(ev);
  }
  void store_set_ts(u64 idx, u32 ts) {  // gossip_store_set_timestamp, :655-670: timestamp AND crc change
    store[idx].timestamp = ts;
    u8 *h = image.data() + store[idx].off - 12;
    put_be32(h + 4, crc32c(ts, image.data() + store[idx].off, store[idx].len));
    put_be32(h + 8, ts);
    store_write_event(store[idx].off - 12, 12);
    if (!on_event) return;   // (no listener: nothing to build)
    lamd_gossipd_event ev;
    memset(&ev, 0, sizeof ev);
    ev.kind = LAMD_GEV_STORE_SET_TS;
    ev.index = idx;
    ev.timestamp = ts;
    ev.values[0] = store[idx].off;
    emit(ev);
  }
  void store_set_dying(u64 idx) {
    store_or_flag(idx, GS_DYING);
    if (!on_event) return;   // (no listener: nothing to build)
    lamd_gossipd_event ev;
    memset(&ev, 0, sizeof ev);
    ev.kind = LAMD_GEV_STORE_FLAG;
    ev.index = idx;
    ev.type = store[idx].type;
    ev.values[0] = store[idx].off;
    ev.values[1] = GS_DYING;
    emit(ev);
  }
  void peer_update(bool has_peer, const nodeid *peer, u64 scid, u32 fee_base, u32 fee_ppm, u32 cltv, u64 hmin, u64 hmax) {
    if (!on_event) return;   // (no listener: nothing to build)
    lamd_gossipd_event ev;
    memset(&ev, 0, sizeof ev);
    ev.kind = LAMD_GEV_PEER_UPDATE;
    ev.has_peer = has_peer;
    if (has_peer) memcpy(ev.peer, peer->k, 33);
    ev.scid = scid;
    ev.values[0] = fee_base; ev.values[1] = fee_ppm; ev.values[2] = cltv; ev.values[3] = hmin; ev.values[4] = hmax;
    emit(ev);
  }

  // ---- verification back end
  int backend_sigcheck(size_t n, const u8 *msgs, const uint64_t *off, const u8 *ids, int8_t *verdict) {
    std::lock_guard<std::mutex> lk(be_mu);   // (an engine context serves one call at a time; the planning thread and a late verify of the apply pass may meet here)
    if (be_sig) return be_sig(be_user, n, msgs, off, ids, verdict);
    if (!ctx) return LAMD_ERR_ARG;
    return lamd_sigcheck_gossip_batch(ctx, n, msgs, off, ids, verdict);
  }
  int backend_keyparse(size_t n, const u8 *pub33, u8 *ok) {
    std::lock_guard<std::mutex> lk(be_mu);
    if (be_key) return be_key(be_user, n, pub33, ok);
    if (!ctx) return LAMD_ERR_ARG;
    return lamd_pubkey_parse_batch(ctx, n, pub33, 33, 33, nullptr, ok);
  }
  msgkey vkey(const mview &m, const nodeid *signer) const {
    msgkey k{m.data(), m.size(), signer ? signer->k : nullptr, 0};
    k.h = content_hash(seed, k.m, k.len);
    if (signer) k.h = content_hash(k.h, signer->k, 33);
    return k;
  }
  // verdict of (message, signer): the planned pairs answer from the slot list that went to the device (cur_sl / cur_v, set by
  // verify()); a pair the plan did not foresee is verified on the spot and remembered in `verdicts`
  int verdict_of(const mview &m, const nodeid *signer) {
    const msgkey k = vkey(m, signer);
    if (cur_sl) {
      const int slot = cur_sl->find(k);
      if (slot >= 0) return cur_v[slot];
    }
    auto it = verdicts.find(k);
    if (it != verdicts.end()) return it->second;
    st.late_verifies++;
    const uint64_t off[2] = {0, m.size()};
    int8_t v = -2;
    const int rc = backend_sigcheck(1, m.data(), off, signer ? signer->k : nullptr, &v);
    if (rc != LAMD_OK || v == -2) {  // a back-end failure is NOT a verdict: nothing is remembered, the caller leaves the message
      fault_rc = rc != LAMD_OK ? rc : LAMD_ERR_HIP;  // unapplied and lamd_gossipd_process() / txout_reply() return this code
      return -2;
    }
    verdicts[k] = v;
    return v;
  }
  int fault_rc = LAMD_OK;     // set by verdict_of(): an engine error during the ordered replay (never a peer-visible warning)
  bool in_process = false;    // lamd_gossipd_process() is applying a batch: callbacks must not re-enter the state-changing entry points
  bool deferred_now = false, deferred_be = false;   // set_time / set_backend called from a callback: applied when process() returns
  uint64_t d_now = 0;
  lamd_gossipd_sigcheck_fn d_be_sig = nullptr;
  lamd_gossipd_keyparse_fn d_be_key = nullptr;
  void *d_be_user = nullptr;

  struct slotlist {
    std::vector<mview> msg;
    std::vector<const nodeid *> signer;
    std::vector<u64> hs;      // vkey hash per slot
    // (message, signer) -> slot: open addressing over the precomputed content hashes, no allocation per entry (a node-based
    // unordered_map cost ~0.2 us per message of a flood here)
    std::vector<u32> tab;     // slot + 1, 0 = empty
    u32 mask = 0;
    void clear() {  // keeps the memory
      msg.clear(); signer.clear(); hs.clear();
      std::fill(tab.begin(), tab.end(), 0u);
    }
    void reserve(size_t n) {
      size_t m = 64;
      while (m < 2 * n + 2) m <<= 1;
      if (m <= tab.size()) return;
      tab.assign(m, 0);
      mask = (u32)(m - 1);
      for (size_t s = 0; s < msg.size(); s++) {
        u32 pos = (u32)hs[s] & mask;
        while (tab[pos]) pos = (pos + 1) & mask;
        tab[pos] = (u32)s + 1;
      }
    }
    int find(const msgkey &k) const {
      if (tab.empty()) return -1;
      for (u32 pos = (u32)k.h & mask;; pos = (pos + 1) & mask) {
        const u32 v = tab[pos];
        if (!v) return -1;
        const size_t s = v - 1;
        if (hs[s] == k.h && msgkey_eq()(k, msgkey{msg[s].data(), msg[s].size(), signer[s] ? signer[s]->k : nullptr, hs[s]})) return (int)s;
      }
    }
    int add(lamd_gossipd *g, const mview &m, const nodeid *signer_) { return add(&g->st.duplicates, m, signer_, g->vkey(m, signer_).h); }
    int add(uint64_t *dups, const mview &m, const nodeid *signer_, u64 h) {  // h = vkey(m, signer_).h, computed by the (parallel) plan
      const msgkey k{m.data(), m.size(), signer_ ? signer_->k : nullptr, h};
      const int have = find(k);
      if (have >= 0) { ++*dups; return have; }
      if (2 * (msg.size() + 1) + 2 > tab.size()) reserve(2 * (msg.size() + 1));
      const int s = (int)msg.size();
      msg.push_back(m);
      signer.push_back(signer_);
      hs.push_back(h);
      u32 pos = (u32)h & mask;
      while (tab[pos]) pos = (pos + 1) & mask;
      tab[pos] = (u32)s + 1;
      return s;
    }
  };
  // one device call for the signatures of `sl`.  verify_into() touches nothing of the ingest but the back end (under be_mu) and the
  // work buffers / counters it is handed: the planning stage of sub-batch k+1 runs it on its own thread while sub-batch k is applied
  struct vbufs { std::vector<uint64_t> off; bytes blob, ids; thread_pool *pool = nullptr; };
  struct vcount { uint64_t sigs = 0, msgs = 0, batches = 0, dups = 0, keyparse = 0; };
  std::mutex be_mu;
  int verify(const slotlist &sl) {
    std::vector<int8_t> v;
    vcount c;
    vf.pool = get_pool();
    const int rc = verify_into(sl, v, vf, c);
    if (rc != LAMD_OK) return rc;
    if (sl.msg.empty()) return LAMD_OK;
    cur_sl = &sl;  // until drop_verdicts(): the caller keeps `sl` alive that long
    cur_v.swap(v);
    count(c);
    return LAMD_OK;
  }
  void count(const vcount &c) {
    st.verified_sigs += c.sigs; st.verified_messages += c.msgs; st.batches += c.batches; st.duplicates += c.dups; st.keyparse_messages += c.keyparse;
  }
  int verify_into(const slotlist &sl, std::vector<int8_t> &v, vbufs &wb, vcount &cnt) {
    const size_t n = sl.msg.size();
    v.clear();
    if (!n) return LAMD_OK;
    std::vector<uint64_t> &off = wb.off;   // (work buffers kept across calls: a fresh 50 MB vector per flood is 12 000 page faults)
    bytes &vf_blob = wb.blob, &vf_ids = wb.ids;
    off.assign(n + 1, 0);
    size_t total = 0;
    bool contiguous = true, any_signer = false;
    for (size_t i = 0; i < n; i++) {
      off[i + 1] = off[i] + sl.msg[i].size();
      total += sl.msg[i].size();
      contiguous &= sl.msg[i].data() == sl.msg[0].data() + off[i];
      any_signer |= sl.signer[i] != nullptr;
    }
    // a flood's slots are its messages in arrival order, back to back in the queue's arena: the back end reads them where they lie
    const u8 *blob = sl.msg[0].data();
    if (!contiguous) {
      vf_blob.resize(total + 1);
      parallel_for(wb.pool, n, 4096, [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; i++) memcpy(&vf_blob[off[i]], sl.msg[i].data(), sl.msg[i].size());
      });
      blob = vf_blob.data();
    }
    const u8 *ids = nullptr;
    if (any_signer) {
      vf_ids.resize(33 * n);
      parallel_for(wb.pool, n, 8192, [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; i++) {
          if (sl.signer[i]) memcpy(&vf_ids[33 * i], sl.signer[i]->k, 33);
          else memset(&vf_ids[33 * i], 0, 33);
        }
      });
      ids = vf_ids.data();
    }
    for (size_t i = 0; i < n; i++) cnt.sigs += (sl.msg[i][0] == 1 && sl.msg[i][1] == 0) ? 4 : 1;
    v.assign(n, -2);
    const int rc = backend_sigcheck(n, blob, off.data(), ids, v.data());
    if (rc != LAMD_OK) return rc;
    cnt.batches++;
    cnt.msgs += n;
    return LAMD_OK;
  }
  vbufs vf;

  bool known_scid(u64 scid) const { return chans.count(scid) || pending_ann.count(scid) || early_ann.count(scid); }
  bool timestamp_reasonable(u32 ts) const {  // gossmap_manage.c:1001-1012
    const int64_t n = (int64_t)now();
    if ((int64_t)ts > n + 24 * 60 * 60) return false;
    if ((int64_t)ts < n - (int64_t)(cfg.prune_interval ? cfg.prune_interval : 1209600u)) return false;
    return true;
  }

  // ---- gossmap_manage_channel_announcement (:620-753) with the sigcheck verdict `v` (or the key-only verdict) known
  void apply_cann(const queued &q, planned &p, int key_ok) {
    const mview &m = q.msg;
    std::string err;
    do {
      if (p.malformed || (p.keyslot >= 0 && !key_ok)) { err = "Malformed channel_announcement " + hexs(m); break; }  // :648
      int v = 0;
      if (p.keyslot < 0) {
        v = p.slot >= 0 ? cur_v[p.slot] : verdict_of(m, nullptr);  // the plan's slot: no second hash of the message
        if (v == -1) { err = "Malformed channel_announcement " + hexs(m); break; }  // an invalid bitcoin key: fromwire_pubkey
        if (v == -2) return;  // engine fault: not consumed (fault_rc)
      }
      const gossip_frame f = gossip_parse_frame(m.data(), m.size());
      const size_t flen = be16(&m[258]);
      nodeid id1, id2;
      memcpy(id1.k, &m[f.keyoff], 33);
      memcpy(id2.k, &m[f.keyoff + 33], 33);
      if (!(id1 < id2)) {  // :661-665
        err = "node_id_1 must be the lesser node id! 1=" + hexs(id1.k, 33) + ", 2=" + hexs(id2.k, 33);
        break;
      }
      if (memcmp(&m[260 + flen], cfg.chain_hash, 32) != 0) return;  // :671-672
      const u64 scid = p.scid;
      if (txout_failures.count(scid)) return;                         // :679-681
      if (known_scid(scid)) return;                                   // :684-687
      if (p.keyslot >= 0) {  // the plan expected one of the drops above: verify now (never seen in practice)
        v = verdict_of(m, nullptr);
        if (v == -2) return;
      }
      if (v > 0) { err = sigcheck_text(GOSSIP_CANN, v, m); break; }   // :689-696
      pending_cannounce pca;
      if (p.pre) {  // built by the parallel planning pass
        pca = std::move(*p.pre);
        p.pre = nullptr;
      } else {
        pca.msg.assign(m.begin(), m.end());
        pca.has_src = q.has_src;
        pca.src = q.src;
        pca.node[0] = id1;
        pca.node[1] = id2;
        // scriptpubkey_p2wsh(bitcoin_redeem_2of2(key1, key2)) (:699-702; bitcoin/script.c:149-165): keys in DER order
        u8 h[32];
        if (p.spk_set) {
          memcpy(h, p.spk, 32);  // hashed by the parallel planning pass
        } else {
          const u8 *k1 = &m[f.keyoff + 66], *k2 = &m[f.keyoff + 99];
          if (memcmp(k1, k2, 33) >= 0) std::swap(k1, k2);
          u8 script[71];
          script[0] = 0x52; script[1] = 33; memcpy(script + 2, k1, 33);
          script[35] = 33; memcpy(script + 36, k2, 33);
          script[69] = 0x52; script[70] = 0xae;
          sha256_single(script, sizeof script, h);
        }
        pca.spk.resize(34);
        pca.spk[0] = 0x00; pca.spk[1] = 0x20;
        memcpy(&pca.spk[2], h, 32);
      }
      const u32 height = cfg.blockheight;
      if (!scid_depth_announceable(scid, height)) {  // :724-741
        if (height != 0 && scid_blocknum(scid) > height + 12) {
          char b[32];
          snprintf(b, sizeof b, "%u", height);
          err = "Bad gossip order: ignoring channel_announcement " + fmt_scid(scid) + " at blockheight " + b;
          break;
        }
        early_ann.emplace(scid, std::move(pca));
        return;
      }
      pending_ann.emplace(scid, std::move(pca));  // :744-750
      ev_scid(LAMD_GEV_GET_TXOUT, false, nullptr, scid);
      return;
    } while (0);
    warning(q.has_src, &q.src, err);  // gossipd.c:277-283
  }

  // ---- process_channel_update (:878-998); returns the error text ("" = none)
  // (`upd` = the message bytes: u.update for an update that waited in a list, the batch's own copy otherwise)
  std::string process_channel_update(const pending_cupdate &u, const mview &upd, int known_verdict = INT32_MIN, chan *known_chan = nullptr) {
    const int dir = u.cflags & 1;
    auto it = known_chan ? chans.end() : chans.find(u.scid);
    if (!known_chan && it == chans.end()) {
      if (txout_failures.count(u.scid)) return "";  // :901-905
      ev_scid(LAMD_GEV_QUERY_CHANNEL, u.has_src, &u.src, u.scid);
      if (on_event) bad_gossip(u.has_src, &u.src, "Unknown channel " + fmt_scid(u.scid));
      return "";
    }
    chan &c = known_chan ? *known_chan : it->second;
    const int v = known_verdict != INT32_MIN ? known_verdict : verdict_of(upd, &c.node[dir]);  // :920-926
    if (v == -2) return "";  // engine fault (fault_rc): the caller keeps the update
    if (v != 0) return sigcheck_text(GOSSIP_CUPD, 1, upd);
    if (u.mflags & 2) return "Do not set DONT_FORWARD on public channel_updates (" + fmt_scid(u.scid) + ")";  // :929-932
    if (c.set[dir]) {  // :935-946
      if (store[c.cupd_rec[dir]].timestamp >= u.timestamp) return "";
    } else if (!c.set[!dir]) {
      store_set_ts(c.cann_rec, u.timestamp);  // :950-951
    }
    const u64 rec = store_add(GOSSIP_CUPD, u.timestamp, upd.data(), upd.size());  // :955
    if (c.set[dir]) store_del(c.cupd_rec[dir]);                                             // :966-967
    c.set[dir] = true;
    c.cupd_rec[dir] = rec;
    nodeid ours;
    memcpy(ours.k, cfg.our_id, 33);
    if (c.node[!dir] == ours) peer_update(u.has_src, &u.src, u.scid, u.fee_base, u.fee_ppm, u.cltv, u.hmin, u.hmax);  // :970-980
    good_gossip(u.has_src, &u.src);
    if (on_event) {  // (the text is only built for a listener: it is a debug-level line in the reference)
      char b[16];
      snprintf(b, sizeof b, "/%d now ", dir);
      ev_text(LAMD_GEV_TRACE, u.has_src, &u.src, "Received channel_update for channel " + fmt_scid(u.scid) + b + ((u.cflags & 2) ? "DISABLED" : "ACTIVE"));
    }
    return "";
  }

  static pending_cupdate parse_cupdate(const queued &q) {
    const mview &m = q.msg;
    pending_cupdate u;
    u.scid = be64(&m[98]);
    u.timestamp = be32(&m[106]);
    u.mflags = m[110];
    u.cflags = m[111];
    u.cltv = be16(&m[112]);
    u.hmin = be64(&m[114]);
    u.fee_base = be32(&m[122]);
    u.fee_ppm = be32(&m[126]);
    u.hmax = be64(&m[130]);
    u.has_src = q.has_src;  // (u.update stays empty: only an update that has to wait keeps its own copy of the bytes)
    u.src = q.src;
    return u;
  }
  // ---- gossmap_manage_channel_update (:1014-1120)
  void apply_cupd(const queued &q, const planned &p) {
    const mview &m = q.msg;
    std::string err;
    do {
      if (p.malformed) { err = "channel_update: malformed " + hexs(m); break; }  // :1044-1046
      if (memcmp(&m[66], cfg.chain_hash, 32) != 0) return;                       // :1054-1057
      pending_cupdate u = parse_cupdate(q);
      if (!timestamp_reasonable(u.timestamp)) return;                            // :1060-1063
      if (!pending_ann.empty() && pending_ann.count(u.scid)) { u.update.assign(m.begin(), m.end()); pending_cupdates.push_back(std::move(u)); return; }  // :1066-1083
      if (!early_ann.empty() && early_ann.count(u.scid)) { u.update.assign(m.begin(), m.end()); early_cupdates.push_back(std::move(u)); return; }      // :1086-1103
      // (the plan looked the channel up when the message passed its filters; a message the plan dropped early is looked up here)
      chan *pc = p.want_slot || p.pc ? p.pc : nullptr;
      if (!pc && !p.want_slot) { auto itc = chans.find(u.scid); if (itc != chans.end()) pc = &itc->second; }
      if (!pc && q.has_src) {                                               // :1107-1116
        const int pv = (p.slot >= 0 && p.signer == &q.src) ? cur_v[p.slot] : verdict_of(m, &q.src);
        if (pv == -2) return;
        if (pv == 0) {
          peer_update(true, &q.src, u.scid, u.fee_base, u.fee_ppm, u.cltv, u.hmin, u.hmax);
          return;
        }
      }
      {  // the plan's verdict stands if the signer it expected is the one process_channel_update() will ask for
        int known = INT32_MIN;
        if (p.slot >= 0 && p.signer && pc && *p.signer == pc->node[u.cflags & 1]) known = cur_v[p.slot];
        err = process_channel_update(u, m, known, pc);
      }
    } while (0);
    if (!err.empty()) warning(q.has_src, &q.src, err);
  }

  // ---- A RUN of channel_updates for channels the map already holds, applied by all host cores (VERDICT r03 "Next" 4).
  // process_channel_update() touches only its channel (flags, record numbers, the timestamps of ITS records) plus two things global to the
  // store: where the new record lands and its record number.  So a run [a, b) of such updates is applied in four passes that give exactly
  // what the one-by-one replay gives:
  //   A  (parallel, sharded by CHANNEL: every update of a channel is seen by one thread, in arrival order)  the reference's decisions --
  //      signature verdict, DONT_FORWARD, "not newer than what we have" (:935-946, against the update accepted earlier in the run, if any),
  //      which record the accepted update supersedes, whether it is the channel's first update (set_timestamp on the announcement, :950-951);
  //   B  (serial, arrival order)  record numbers and file offsets of the accepted updates: a prefix sum;
  //   C  (parallel over the run)  the bytes: header + message at its offset (an update superseded later in the same run is written with its
  //      DELETED bit already set: the file's final content), the DELETED bit of the superseded older record, the announcement's timestamp / crc,
  //      the channel's record numbers;
  //   D  (serial, arrival order, only with a listener)  the events, each exactly as the one-by-one replay emits it.
  // A listener that asked for the file's write stream (emit_store_writes) sees intermediate states of the image: such an ingest keeps the
  // one-by-one path (run_ok()).  Everything that is not a plain update of a known channel also stays on that path.
  static constexpr u32 RUN_NONE = 0xFFFFFFFFu;
  size_t run_min = 2048;    // LAMD_INGEST_RUN_MIN: shortest run worth the four passes
  size_t sub_rows = 131072; // LAMD_INGEST_SUB: messages per sub-batch of the three-stage pipeline
  bool run_ok() const { return run_min != 0 && !(on_event && cfg.emit_store_writes); }
  bool run_member(const planned &p, const queued &q, const std::vector<int8_t> &v) const {
    return p.type == GOSSIP_CUPD && !p.malformed && p.pc && p.slot >= 0 && p.signer == &p.pc->node[q.msg[111] & 1] && v[p.slot] != -2;
  }
  // Without a listener a run also swallows INERT channel_updates -- malformed ones, and updates of channels the map does not hold (unknown or
  // on another chain, too old, too far in the future: the plan found no channel) -- whose replay changes nothing a run reads or writes: no store
  // record, no channel state; at most an entry in the lists of updates waiting for a pending announcement.  They are replayed one by one after
  // the run's passes, in arrival order among themselves.  (With a listener their warnings and traces are events whose order matters: they end
  // the run, as every other kind of message does.)  One damaged message in a hundred would otherwise cut a flood into runs too short to take.
  bool run_side(const planned &p, const std::vector<int8_t> &v) const {
    return !on_event && p.type == GOSSIP_CUPD && (p.malformed || !p.pc) && !(p.slot >= 0 && v[p.slot] == -2);
  }
  // ---- a run of plain channel_announcements (a flood's first phase): well-formed, all four signatures good, node ids in order, this chain,
  // deep enough, the waiting record already built by the planning pass.  What is left of apply_cann() for such a message is three map probes
  // and one insertion into pending_ann -- by short_channel_id shard on all cores (a shard's announcements in arrival order: the first of two
  // announcements of one channel wins, as in the replay), then one serial pass for the LAMD_GEV_GET_TXOUT events in arrival order.
  bool cann_run_member(const planned &p, const std::vector<int8_t> &v) const {
    return p.type == GOSSIP_CANN && !p.malformed && p.keyslot < 0 && p.slot >= 0 && v[p.slot] == 0 && p.pre != nullptr &&
           scid_depth_announceable(p.scid, cfg.blockheight);
  }
  // (announcements whose replay cannot change anything a run reads or writes -- malformed, an unparsable bitcoin key, a bad signature: a warning
  // to the peer and nothing else -- ride along: they are replayed in the serial pass, in arrival order among the run's events.  One damaged
  // message in a hundred would otherwise cut a flood into runs too short to take.)
  bool cann_run_side(const planned &p, const std::vector<int8_t> &v) const {
    return p.type == GOSSIP_CANN && p.keyslot < 0 && (p.malformed || (p.slot >= 0 && (v[p.slot] == -1 || v[p.slot] > 0)));
  }
  std::vector<u8> cann_took;
  void apply_cann_run(const std::vector<queued> &batch, std::vector<planned> &plan, size_t a, size_t b) {
    cann_took.assign(b - a, 0);
    const unsigned NSH = pending_ann.shards();
    for (unsigned sh = 0; sh < NSH; sh++) pending_ann.shard(sh).reserve(pending_ann.shard(sh).size() + (b - a) / NSH + (b - a) / (4 * NSH) + 16);
    const bool early_empty = early_ann.empty(), fail_empty = txout_failures.empty();
    parallel_for(get_pool(), NSH, 1, [&](size_t lo, size_t hi) {
      for (size_t i = a; i < b; i++) {
        planned &p = plan[i];
        const unsigned sh = pending_ann.shard_of(p.scid);
        if (sh < lo || sh >= hi) continue;             // (first: only the shard's owner looks at, and clears, p.pre)
        if (!cann_run_member(p, cur_v)) continue;      // rides along (cann_run_side): the serial pass replays it
        const bool drop = (!fail_empty && txout_failures.shard(sh).count(p.scid)) ||                 // :679-681
                          chans.shard(sh).count(p.scid) || (!early_empty && early_ann.count(p.scid));  // :684-687
        if (!drop && pending_ann.shard(sh).emplace(p.scid, std::move(*p.pre)).second) cann_took[i - a] = 1;   // :744-750
        p.pre = nullptr;
      }
    });
    for (size_t i = a; i < b; i++) {
      if (cann_took[i - a]) { st.run_announcements++; if (on_event) ev_scid(LAMD_GEV_GET_TXOUT, false, nullptr, plan[i].scid); }
      else if (cann_run_side(plan[i], cur_v)) apply_cann(batch[i], plan[i], 1);
    }
    st.messages += b - a;
  }
  enum : u8 { RO_DROP = 0, RO_ACCEPT = 1, RO_BADSIG = 2, RO_DONTFWD = 3, RO_SIDE = 4, RO_UNKNOWN = 5 };
  // (what pass A decides about one update.  One array PER SHARD, in the shard's arrival order: the thread that owns a channel writes only
  // its own array -- verdict bytes of neighbouring messages in one shared array ping-pong their cache lines between the cores, which made the
  // 16-thread run slower per update than the one-by-one replay)
  struct run_res { u64 prev_rec, rec, off; u32 prev_run; u8 outcome, dead, touch, pad; };
  struct run_bufs {
    std::vector<std::vector<run_res>> res;     // [shard][k]
    std::vector<u32> where;                    // run index -> k
    std::vector<u8> shard;                     // run index -> shard
    std::vector<std::vector<u32>> bucket;      // [range * T + shard] -> run indices, ascending
    std::vector<size_t> base, cnt_rec, cnt_bytes;
  };
  run_bufs rb;
  thread_pool *pool = nullptr, *pool_bg = nullptr;   // all host cores / the planning stage that runs under an apply pass (created on first use)
  thread_pool *get_pool() { if (!pool) pool = new thread_pool(host_threads() - 1); return pool; }
  thread_pool *get_pool_bg() { if (!pool_bg) pool_bg = new thread_pool(std::max(1u, host_threads() / 2) - 1); return pool_bg; }
  std::vector<queued> w_batch;
  std::vector<planned> w_plan;
  ingest_stage *w_stage = nullptr;   // three stages in flight (lamd_gossipd_process), allocated on first use
  void apply_cupd_run(const std::vector<queued> &batch, const std::vector<planned> &plan, const std::vector<int8_t> &v, size_t a, size_t b) {
    const size_t m = b - a;
    const unsigned T = std::max(1u, std::min(get_pool()->size(), (unsigned)(m / 1024 + 1)));
    rb.where.resize(m);
    rb.shard.resize(m);
    if (rb.bucket.size() < (size_t)T * T) rb.bucket.resize((size_t)T * T);
    if (rb.res.size() < T) rb.res.resize(T);
    rb.base.assign((size_t)T * T, 0);
    rb.cnt_rec.assign(T, 0);
    rb.cnt_bytes.assign(T, 0);
    const size_t step = (m + T - 1) / T;
    auto shard_of = [T](const chan *c) { return (unsigned)((mix64((u64)(uintptr_t)c) >> 32) % T); };
    thread_pool *tp = get_pool();
    auto on_threads = [&](const std::function<void(unsigned)> &f) { tp->run(T, f); };
    // ---- shard by channel, keeping arrival order inside a shard: range r of the run lists its indices per shard ...
    on_threads([&](unsigned r) {
      const size_t lo = std::min(m, r * step), hi = std::min(m, (r + 1) * step);
      for (unsigned t = 0; t < T; t++) { auto &bk = rb.bucket[(size_t)r * T + t]; bk.clear(); bk.reserve((hi - lo) / T + 16); }
      for (size_t i = lo; i < hi; i++) {
        if (!run_member(plan[a + i], batch[a + i], v)) { rb.shard[i] = 0xFF; continue; }   // an inert message inside the run (run_side)
        const unsigned t = shard_of(plan[a + i].pc);
        rb.shard[i] = (u8)t;
        rb.bucket[(size_t)r * T + t].push_back((u32)i);
      }
    });
    for (unsigned t = 0; t < T; t++) {  // ... shard t then reads ranges 0, 1, ...: where each range's share starts in the shard's array
      size_t k = 0;
      for (unsigned r = 0; r < T; r++) { rb.base[(size_t)r * T + t] = k; k += rb.bucket[(size_t)r * T + t].size(); }
      rb.res[t].resize(k);
    }
    on_threads([&](unsigned r) {
      for (unsigned t = 0; t < T; t++) {
        const auto &bk = rb.bucket[(size_t)r * T + t];
        const size_t k0 = rb.base[(size_t)r * T + t];
        for (size_t j = 0; j < bk.size(); j++) rb.where[bk[j]] = (u32)(k0 + j);
      }
    });
    // ---- pass A
    on_threads([&](unsigned t) {
      std::vector<run_res> &res = rb.res[t];
      size_t k = 0;
      for (unsigned r = 0; r < T; r++)
        for (const u32 i : rb.bucket[(size_t)r * T + t]) {
          run_res &R = res[k++];
          R = run_res{~0ull, 0, 0, RUN_NONE, RO_DROP, 0, 0, 0};
          const planned &p = plan[a + i];
          const mview &msg = batch[a + i].msg;
          chan &c = *p.pc;
          const int dir = msg[111] & 1;
          const u32 ts = be32(&msg[106]);
          if (v[p.slot] != 0) { R.outcome = RO_BADSIG; continue; }       // :920-926
          if (msg[110] & 2) { R.outcome = RO_DONTFWD; continue; }         // :929-932
          const u32 pr = c.run_last[dir];
          const bool have = pr != RUN_NONE || c.set[dir];
          if (have) {                                                      // :935-946
            const u32 cur = pr != RUN_NONE ? be32(&batch[a + pr].msg[106]) : store[c.cupd_rec[dir]].timestamp;
            if (cur >= ts) continue;   // RO_DROP
          } else if (!c.set[!dir] && c.run_last[!dir] == RUN_NONE) {
            R.touch = 1;                                                   // :950-951
          }
          R.outcome = RO_ACCEPT;
          if (pr != RUN_NONE) { R.prev_run = pr; res[rb.where[pr]].dead = 1; }
          else if (c.set[dir]) R.prev_rec = c.cupd_rec[dir];
          c.run_last[dir] = i;
        }
    });
    // ---- pass B: record numbers and file offsets = a prefix sum over arrival order (per range in parallel, the ranges' totals serially)
    on_threads([&](unsigned r) {
      const size_t lo = std::min(m, r * step), hi = std::min(m, (r + 1) * step);
      size_t nr = 0, nb = 0;
      for (size_t i = lo; i < hi; i++)
        if (rb.shard[i] != 0xFF && rb.res[rb.shard[i]][rb.where[i]].outcome == RO_ACCEPT) { nr++; nb += 12 + batch[a + i].msg.size(); }
      rb.cnt_rec[r] = nr;
      rb.cnt_bytes[r] = nb;
    });
    u64 nrec = store.size(), pos = image.size();
    std::vector<u64> rec0(T), pos0(T);
    for (unsigned r = 0; r < T; r++) { rec0[r] = nrec; pos0[r] = pos; nrec += rb.cnt_rec[r]; pos += rb.cnt_bytes[r]; }
    store.resize(nrec);
    image.resize(pos);
    // ---- pass C
    on_threads([&](unsigned r) {
      const size_t lo = std::min(m, r * step), hi = std::min(m, (r + 1) * step);
      u64 rn = rec0[r], ps = pos0[r];
      for (size_t i = lo; i < hi; i++) {
        if (rb.shard[i] == 0xFF) continue;
        run_res &R = rb.res[rb.shard[i]][rb.where[i]];
        if (R.outcome != RO_ACCEPT) continue;
        const mview &msg = batch[a + i].msg;
        chan &c = *plan[a + i].pc;
        const int dir = msg[111] & 1;
        const u32 ts = be32(&msg[106]);
        R.rec = rn++;
        R.off = ps + 12;
        ps += 12 + msg.size();
        if (R.touch) {  // gossip_store_set_timestamp on the channel_announcement: timestamp AND crc
          record &ca = store[c.cann_rec];
          ca.timestamp = ts;
          u8 *hc = image.data() + ca.off - 12;
          put_be32(hc + 4, crc32c(ts, image.data() + ca.off, ca.len));
          put_be32(hc + 8, ts);
        }
        u8 *h = image.data() + R.off - 12;
        put_be16(h, GS_COMPLETED | (R.dead ? GS_DELETED : 0u));
        put_be16(h + 2, (u32)msg.size());
        put_be32(h + 4, crc32c(ts, msg.data(), msg.size()));
        put_be32(h + 8, ts);
        memcpy(h + 12, msg.data(), msg.size());
        store[R.rec] = record{GOSSIP_CUPD, ts, R.dead != 0, R.off, (u32)msg.size()};
        if (R.prev_rec != ~0ull) {  // gossip_store_del of the record this update supersedes
          record &old = store[R.prev_rec];
          old.deleted = true;
          u8 *ho = image.data() + old.off - 12;
          put_be16(ho, (((u32)ho[0] << 8) | ho[1]) | GS_DELETED);
        }
        if (!R.dead) {  // the channel's standing update for this direction
          c.set[dir] = true;
          c.cupd_rec[dir] = R.rec;
          c.run_last[dir] = RUN_NONE;
        }
      }
    });
    st.messages += m;
    st.run_updates += m;
    if (!on_event) {  // the inert messages of the run, one by one (no listener: run_side() admits none otherwise)
      for (size_t i = 0; i < m; i++)
        if (rb.shard[i] == 0xFF) { apply_cupd(batch[a + i], plan[a + i]); st.run_updates--; }
      return;
    }
    // ---- pass D: the events of the one-by-one replay, in its order
    nodeid ours;
    memcpy(ours.k, cfg.our_id, 33);
    for (size_t i = 0; i < m; i++) {
      const run_res &R = rb.res[rb.shard[i]][rb.where[i]];
      const queued &q = batch[a + i];
      const mview &msg = q.msg;
      const chan &c = *plan[a + i].pc;
      const int dir = msg[111] & 1;
      if (R.outcome == RO_BADSIG) { warning(q.has_src, &q.src, sigcheck_text(GOSSIP_CUPD, 1, msg)); continue; }
      if (R.outcome == RO_DONTFWD) { warning(q.has_src, &q.src, "Do not set DONT_FORWARD on public channel_updates (" + fmt_scid(be64(&msg[98])) + ")"); continue; }
      if (R.outcome != RO_ACCEPT) continue;
      const pending_cupdate u = parse_cupdate(q);
      lamd_gossipd_event ev;
      if (R.touch) {
        memset(&ev, 0, sizeof ev);
        ev.kind = LAMD_GEV_STORE_SET_TS; ev.index = c.cann_rec; ev.timestamp = u.timestamp; ev.values[0] = store[c.cann_rec].off;
        emit(ev);
      }
      memset(&ev, 0, sizeof ev);
      ev.kind = LAMD_GEV_STORE_ADD; ev.index = R.rec; ev.type = GOSSIP_CUPD; ev.timestamp = u.timestamp; ev.values[0] = R.off;
      ev.data = msg.data(); ev.len = msg.size();
      emit(ev);
      const u64 old = R.prev_rec != ~0ull ? R.prev_rec : (R.prev_run != RUN_NONE ? rb.res[rb.shard[R.prev_run]][rb.where[R.prev_run]].rec : ~0ull);
      if (old != ~0ull) {
        memset(&ev, 0, sizeof ev);
        ev.kind = LAMD_GEV_STORE_DEL; ev.index = old; ev.type = GOSSIP_CUPD; ev.values[0] = store[old].off;
        emit(ev);
      }
      if (c.node[!dir] == ours) peer_update(u.has_src, &u.src, u.scid, u.fee_base, u.fee_ppm, u.cltv, u.hmin, u.hmax);
      good_gossip(u.has_src, &u.src);
      char bb[16];
      snprintf(bb, sizeof bb, "/%d now ", dir);
      ev_text(LAMD_GEV_TRACE, u.has_src, &u.src, "Received channel_update for channel " + fmt_scid(u.scid) + bb + ((u.cflags & 2) ? "DISABLED" : "ACTIVE"));
    }
  }

  // ---- a run of plain node_announcements (gossmap_manage_node_announcement, gossmap_manage.c:1162-1243, + process_node_announcement :1122-1160) on all
  // cores: well-formed, address list sound, a verification slot planned.  What the one-by-one replay does for such a message is a lookup of its node,
  // a timestamp comparison with the node's standing announcement and -- if newer -- one store record added, the old one deleted.  As for the
  // channel_update runs: the messages are sharded by NODE (a node's announcements stay in arrival order inside their shard: of two in one run the later
  // supersedes the earlier only if its timestamp is newer, as in the replay), record numbers and file offsets are a prefix sum over arrival order, the
  // record bytes are written in parallel, the events are emitted by one serial pass in the replay's order.  A message whose node the map does not hold
  // (queued behind a pending channel_announcement, or an unknown-node query) and one whose signature fails take their one-by-one path in that pass.
  bool nann_run_member(const planned &p, const std::vector<int8_t> &v) const {
    return p.type == GOSSIP_NANN && !p.malformed && !p.addrs_bad && p.slot >= 0 && v[p.slot] != -2 && v[p.slot] != -1;
  }
  std::vector<node *> rb_node;
  void apply_nann_run(const std::vector<queued> &batch, const std::vector<planned> &plan, const std::vector<int8_t> &v, size_t a, size_t b) {
    const size_t m = b - a;
    const unsigned T = std::max(1u, std::min(std::min(get_pool()->size(), (unsigned)(m / 512 + 1)), nodes.shards()));
    rb.where.resize(m);
    rb.shard.resize(m);
    rb_node.assign(m, nullptr);
    if (rb.bucket.size() < (size_t)T * T) rb.bucket.resize((size_t)T * T);
    if (rb.res.size() < T) rb.res.resize(T);
    rb.base.assign((size_t)T * T, 0);
    rb.cnt_rec.assign(T, 0);
    rb.cnt_bytes.assign(T, 0);
    const size_t step = (m + T - 1) / T;
    thread_pool *tp = get_pool();
    auto on_threads = [&](const std::function<void(unsigned)> &f) { tp->run(T, f); };
    auto id_of = [&](size_t i, nodeid *id, u32 *ts) {
      const mview &msg = batch[a + i].msg;
      const gossip_frame f = gossip_parse_frame(msg.data(), msg.size());
      memcpy(id->k, &msg[f.keyoff], 33);
      *ts = be32(&msg[f.keyoff - 4]);
    };
    // ---- shard by node id (the node map's own shards, folded onto T workers), keeping arrival order inside a shard
    on_threads([&](unsigned r) {
      const size_t lo = std::min(m, r * step), hi = std::min(m, (r + 1) * step);
      for (unsigned t = 0; t < T; t++) { auto &bk = rb.bucket[(size_t)r * T + t]; bk.clear(); bk.reserve((hi - lo) / T + 16); }
      for (size_t i = lo; i < hi; i++) {
        nodeid id;
        u32 ts;
        id_of(i, &id, &ts);
        const unsigned t = nodes.shard_of(id) % T;
        rb.shard[i] = (u8)t;
        rb.bucket[(size_t)r * T + t].push_back((u32)i);
      }
    });
    for (unsigned t = 0; t < T; t++) {
      size_t k = 0;
      for (unsigned r = 0; r < T; r++) { rb.base[(size_t)r * T + t] = k; k += rb.bucket[(size_t)r * T + t].size(); }
      rb.res[t].resize(k);
    }
    on_threads([&](unsigned r) {
      for (unsigned t = 0; t < T; t++) {
        const auto &bk = rb.bucket[(size_t)r * T + t];
        const size_t k0 = rb.base[(size_t)r * T + t];
        for (size_t j = 0; j < bk.size(); j++) rb.where[bk[j]] = (u32)(k0 + j);
      }
    });
    // ---- pass A: decisions, per shard in arrival order (:1216-1243, :1125-1126)
    on_threads([&](unsigned t) {
      std::vector<run_res> &res = rb.res[t];
      size_t k = 0;
      for (unsigned r = 0; r < T; r++)
        for (const u32 i : rb.bucket[(size_t)r * T + t]) {
          run_res &R = res[k++];
          R = run_res{~0ull, 0, 0, RUN_NONE, RO_DROP, 0, 0, 0};
          const planned &p = plan[a + i];
          if (v[p.slot] != 0) { R.outcome = RO_BADSIG; continue; }        // :1216-1219
          nodeid id;
          u32 ts;
          id_of(i, &id, &ts);
          auto it = nodes.find(id);
          if (it == nodes.end()) { R.outcome = RO_UNKNOWN; continue; }     // :1222-1238: the serial pass takes the one-by-one path
          node &n = it->second;
          rb_node[i] = &n;
          const u32 pr = n.run_last;
          if (pr != RUN_NONE || n.announced) {                             // :1125-1126
            u32 cur;
            if (pr != RUN_NONE) { nodeid pid; id_of(pr, &pid, &cur); }
            else cur = store[n.nann_rec].timestamp;
            if (cur >= ts) continue;   // RO_DROP
          }
          R.outcome = RO_ACCEPT;
          if (pr != RUN_NONE) { R.prev_run = pr; res[rb.where[pr]].dead = 1; }
          else if (n.announced) R.prev_rec = n.nann_rec;
          n.run_last = i;
        }
    });
    // ---- pass B: record numbers and file offsets = a prefix sum over arrival order
    on_threads([&](unsigned r) {
      const size_t lo = std::min(m, r * step), hi = std::min(m, (r + 1) * step);
      size_t nr = 0, nb = 0;
      for (size_t i = lo; i < hi; i++)
        if (rb.res[rb.shard[i]][rb.where[i]].outcome == RO_ACCEPT) { nr++; nb += 12 + batch[a + i].msg.size(); }
      rb.cnt_rec[r] = nr;
      rb.cnt_bytes[r] = nb;
    });
    u64 nrec = store.size(), pos = image.size();
    std::vector<u64> rec0(T), pos0(T);
    for (unsigned r = 0; r < T; r++) { rec0[r] = nrec; pos0[r] = pos; nrec += rb.cnt_rec[r]; pos += rb.cnt_bytes[r]; }
    store.resize(nrec);
    image.resize(pos);
    // ---- pass C: the records
    on_threads([&](unsigned r) {
      const size_t lo = std::min(m, r * step), hi = std::min(m, (r + 1) * step);
      u64 rn = rec0[r], ps = pos0[r];
      for (size_t i = lo; i < hi; i++) {
        run_res &R = rb.res[rb.shard[i]][rb.where[i]];
        if (R.outcome != RO_ACCEPT) continue;
        const mview &msg = batch[a + i].msg;
        node &n = *rb_node[i];
        nodeid id;
        u32 ts;
        id_of(i, &id, &ts);
        R.rec = rn++;
        R.off = ps + 12;
        ps += 12 + msg.size();
        u8 *h = image.data() + R.off - 12;
        put_be16(h, GS_COMPLETED | (R.dead ? GS_DELETED : 0u));
        put_be16(h + 2, (u32)msg.size());
        put_be32(h + 4, crc32c(ts, msg.data(), msg.size()));
        put_be32(h + 8, ts);
        memcpy(h + 12, msg.data(), msg.size());
        store[R.rec] = record{GOSSIP_NANN, ts, R.dead != 0, R.off, (u32)msg.size()};
        if (R.prev_rec != ~0ull) {  // gossip_store_del of the announcement this one supersedes (a node's records belong to its shard's worker... but
          record &old = store[R.prev_rec];  // two ranges may hold two announcements of one node: only the FIRST accepted one of the run has prev_rec set)
          old.deleted = true;
          u8 *ho = image.data() + old.off - 12;
          put_be16(ho, (((u32)ho[0] << 8) | ho[1]) | GS_DELETED);
        }
        if (!R.dead) {  // the node's standing announcement
          n.announced = true;
          n.nann_rec = R.rec;
          n.run_last = RUN_NONE;
        }
      }
    });
    st.messages += m;
    st.run_nodes += m;
    // ---- pass D: the events of the one-by-one replay, in its order; the messages that take the one-by-one path
    for (size_t i = 0; i < m; i++) {
      const run_res &R = rb.res[rb.shard[i]][rb.where[i]];
      const queued &q = batch[a + i];
      const mview &msg = q.msg;
      if (R.outcome == RO_BADSIG) { warning(q.has_src, &q.src, sigcheck_text(GOSSIP_NANN, 1, msg)); continue; }
      if (R.outcome == RO_UNKNOWN) { st.run_nodes--; apply_nann(q, plan[a + i]); continue; }
      if (R.outcome != RO_ACCEPT || !on_event) continue;
      nodeid id;
      u32 ts;
      id_of(i, &id, &ts);
      lamd_gossipd_event ev;
      memset(&ev, 0, sizeof ev);
      ev.kind = LAMD_GEV_STORE_ADD; ev.index = R.rec; ev.type = GOSSIP_NANN; ev.timestamp = ts; ev.values[0] = R.off;
      ev.data = msg.data(); ev.len = msg.size();
      emit(ev);
      const u64 old = R.prev_rec != ~0ull ? R.prev_rec : (R.prev_run != RUN_NONE ? rb.res[rb.shard[R.prev_run]][rb.where[R.prev_run]].rec : ~0ull);
      if (old != ~0ull) {
        memset(&ev, 0, sizeof ev);
        ev.kind = LAMD_GEV_STORE_DEL; ev.index = old; ev.type = GOSSIP_NANN; ev.values[0] = store[old].off;
        emit(ev);
      }
      good_gossip(q.has_src, &q.src);
      ev_text(LAMD_GEV_TRACE, q.has_src, &q.src, "Received node_announcement for node " + hexs(id.k, 33));
    }
  }

  // ---- process_node_announcement (:1122-1160)
  void process_node_announcement(node &n, u32 timestamp, const nodeid &id, const mview &m, bool has_src, const nodeid *src) {
    if (n.announced && store[n.nann_rec].timestamp >= timestamp) return;
    const u64 rec = store_add(GOSSIP_NANN, timestamp, m.data(), m.size());
    if (n.announced) store_del(n.nann_rec);
    n.announced = true;
    n.nann_rec = rec;
    good_gossip(has_src, src);
    if (on_event) ev_text(LAMD_GEV_TRACE, has_src, src, "Received node_announcement for node " + hexs(id.k, 33));
  }
  void unknown_node(bool has_src, const nodeid *src, const nodeid &id) {  // :1231-1238
    if (!on_event) return;   // (no listener: nothing to build)
    lamd_gossipd_event ev;
    memset(&ev, 0, sizeof ev);
    ev.kind = LAMD_GEV_QUERY_NODE;
    ev.has_peer = has_src;
    if (has_src) memcpy(ev.peer, src->k, 33);
    ev.data = id.k;
    ev.len = 33;
    emit(ev);
    if (on_event) bad_gossip(has_src, src, "node_announcement: unknown node " + hexs(id.k, 33));
  }
  // ---- gossmap_manage_node_announcement (:1162-1243)
  void apply_nann(const queued &q, const planned &p) {
    const mview &m = q.msg;
    std::string err;
    do {
      if (p.malformed) { err = "node_announcement: malformed " + hexs(m); break; }  // :1197-1198
      if (p.addrs_bad) { err = "node_announcement: malformed wireaddrs  in " + hexs(m); break; }  // :1210-1213 (tal_hex(NULL) is "")
      const int v = p.slot >= 0 ? cur_v[p.slot] : verdict_of(m, nullptr);
      if (v == -2) return;
      if (v == -1) { err = "node_announcement: malformed " + hexs(m); break; }
      if (v != 0) { err = sigcheck_text(GOSSIP_NANN, 1, m); break; }  // :1216-1219
      const gossip_frame f = gossip_parse_frame(m.data(), m.size());
      nodeid id;
      memcpy(id.k, &m[f.keyoff], 33);
      const u32 timestamp = be32(&m[f.keyoff - 4]);
      auto it = nodes.find(id);
      if (it == nodes.end()) {
        if (!pending_ann.empty() || !early_ann.empty()) {  // :1224-1230
          pending_nannounce pn;
          pn.id = id; pn.timestamp = timestamp; pn.msg.assign(m.begin(), m.end()); pn.has_src = q.has_src; pn.src = q.src;
          pending_nannounces.push_back(std::move(pn));
          return;
        }
        unknown_node(q.has_src, &q.src, id);
        return;
      }
      process_node_announcement(it->second, timestamp, id, m, q.has_src, &q.src);
      return;
    } while (0);
    warning(q.has_src, &q.src, err);
  }

  // ---- reprocess_queued_msgs (:1284-1342): the signatures of every waiting channel_update whose channel now exists go to
  // the device as one batch first
  void drop_verdicts() {  // (clear() walks the whole bucket array of a map that once held a large batch)
    cur_sl = nullptr;
    cur_v.clear();
    if (!verdicts.empty() || verdicts.bucket_count() > 64) verdict_map().swap(verdicts);
  }
  int reprocess_queued_msgs() {
    const bool pending_empty = pending_ann.empty(), early_empty = early_ann.empty();
    if (!pending_empty && !early_empty) return LAMD_OK;
    if (pending_cupdates.empty() && early_cupdates.empty() && pending_nannounces.empty()) return LAMD_OK;  // nothing is waiting
    slotlist sl;
    auto plan_list = [&](const std::vector<pending_cupdate> &l) {
      for (const pending_cupdate &u : l) {
        auto it = chans.find(u.scid);
        if (it != chans.end()) sl.add(this, u.update, &it->second.node[u.cflags & 1]);
      }
    };
    drop_verdicts();
    if (pending_empty) plan_list(pending_cupdates);
    if (early_empty) plan_list(early_cupdates);
    const int rc = verify(sl);
    if (rc != LAMD_OK) return rc;
    auto process_pending = [&](const pending_cupdate &u) {  // :1245-1268
      if (fault_rc != LAMD_OK) { pending_cupdates.push_back(u); return; }  // after an engine fault the rest keeps waiting, in order
      const std::string err = process_channel_update(u, u.update);
      if (fault_rc != LAMD_OK) { pending_cupdates.push_back(u); return; }
      if (!err.empty()) peer_warning(u.has_src, &u.src, "channel_update: " + err);
    };
    // (the lists outlive the verdict map: its keys refer to their messages)
    std::vector<pending_cupdate> lp, le;
    std::vector<pending_nannounce> ln;
    if (pending_empty) {
      lp.swap(pending_cupdates);
      for (const pending_cupdate &u : lp) process_pending(u);
    }
    if (early_empty) {
      le.swap(early_cupdates);
      for (pending_cupdate &u : le) {
        if (pending_ann.count(u.scid)) { pending_cupdates.push_back(u); continue; }
        process_pending(u);
      }
    }
    if (early_empty && pending_empty && fault_rc == LAMD_OK) {
      ln.swap(pending_nannounces);
      for (const pending_nannounce &pn : ln) {
        auto it = nodes.find(pn.id);
        if (it == nodes.end()) { unknown_node(pn.has_src, &pn.src, pn.id); continue; }
        process_node_announcement(it->second, pn.timestamp, pn.id, pn.msg, pn.has_src, &pn.src);
      }
    }
    drop_verdicts();
    const int frc = fault_rc;
    fault_rc = LAMD_OK;
    return frc;
  }

  // ---- remove_channel (gossmap_manage.c:296-375) and what leads to it: pruning (:398-470), spent / dying channels (:1369-1497)
  bool channel_already_dying(u64 scid) const {
    for (const chan_dying &d : dying_channels)
      if (d.scid == scid) return true;
    return false;
  }
  // any_cannounce_preceeds_offset (:267-286): does another, not dying, channel of the node sit before `off` in the store?
  bool any_cannounce_precedes(const node &n, u64 exclude_scid, u64 off) const {
    for (u64 sc : n.scids) {
      if (sc == exclude_scid) continue;
      const chan &c = chans.find(sc)->second;
      if (store[c.cann_rec].off > off) continue;
      if (c.dying) continue;
      return true;
    }
    return false;
  }
  bool all_node_channels_dying(const node &n, u64 ignore_scid) const {  // :289-298
    for (u64 sc : n.scids)
      if (sc != ignore_scid && !chans.find(sc)->second.dying) return false;
    return true;
  }
  void remove_channel(u64 scid) {
    auto it = chans.find(scid);
    if (it == chans.end()) return;
    const chan c = it->second;
    txout_failures[scid] = true;   // :309 suppress any now-obsolete updates / announcements
    pending_ann.erase(scid);       // :312-313
    early_ann.erase(scid);
    u8 tomb[10];
    put_be16(tomb, WIRE_GS_DELETE_CHAN);
    put_be64(tomb + 2, scid);
    store_add(WIRE_GS_DELETE_CHAN, 0, tomb, sizeof tomb);  // :316-318
    store_del(c.cann_rec);                                  // :321
    for (int dir = 0; dir < 2; dir++)
      if (c.set[dir]) store_del(c.cupd_rec[dir]);
    for (int dir = 0; dir < 2; dir++) {                     // :328-373 node_announcements that should no longer be there
      if (dir == 1 && c.node[1] == c.node[0]) continue;
      auto nit = nodes.find(c.node[dir]);
      if (nit == nodes.end()) continue;
      node &n = nit->second;
      if (!n.announced) continue;
      if (n.nchans == 1) {  // last channel: delete the node_announcement
        store_del(n.nann_rec);
        n.announced = false;
        continue;
      }
      u64 rec;
      if (store[c.cann_rec].off < store[n.nann_rec].off && !any_cannounce_precedes(n, scid, store[n.nann_rec].off)) {
        // this was the last channel_announcement in front of the node_announcement: delete and re-add it to keep the order
        const record r = store[n.nann_rec];
        const bytes copy(image.begin() + r.off, image.begin() + r.off + r.len);
        store_del(n.nann_rec);
        rec = store_add(GOSSIP_NANN, r.timestamp, copy.data(), copy.size());
        n.nann_rec = rec;
      } else {
        if (c.dying) continue;
        rec = n.nann_rec;
      }
      if (all_node_channels_dying(n, scid)) store_set_dying(rec);
    }
    // what the reference's next gossmap refresh does: the channel is gone, and so is a node left without channels
    for (int dir = 0; dir < 2; dir++) {
      if (dir == 1 && c.node[1] == c.node[0]) continue;
      auto nit = nodes.find(c.node[dir]);
      if (nit == nodes.end()) continue;
      node &n = nit->second;
      n.scids.erase(std::remove(n.scids.begin(), n.scids.end(), scid), n.scids.end());
      if (--n.nchans == 0) nodes.erase(nit);
    }
    chans.erase(scid);
  }
  void channel_spent(u32 blockheight, u64 scid) {  // gossmap_manage_channel_spent, :1439-1497
    auto it = chans.find(scid);
    if (it == chans.end()) return;
    if (channel_already_dying(scid)) return;
    chan &c = it->second;
    chan_dying cd;
    cd.scid = scid;
    cd.deadline = blockheight + 72;  // BOLT #7: SHOULD forget a channel after a 72-block delay
    if (on_event) ev_text(LAMD_GEV_TRACE, false, nullptr, "channel " + fmt_scid(scid) + " closing soon due to the funding outpoint being spent");
    u8 m[14];
    put_be16(m, WIRE_GS_CHAN_DYING);
    put_be64(m + 2, scid);
    put_be32(m + 10, cd.deadline);
    cd.rec = store_add(WIRE_GS_CHAN_DYING, 0, m, sizeof m);
    dying_channels.push_back(cd);
    store_set_dying(c.cann_rec);
    c.dying = true;
    for (int dir = 0; dir < 2; dir++)
      if (c.set[dir]) store_set_dying(c.cupd_rec[dir]);
    for (int dir = 0; dir < 2; dir++) {
      if (dir == 1 && c.node[1] == c.node[0]) continue;
      auto nit = nodes.find(c.node[dir]);
      if (nit == nodes.end() || !nit->second.announced) continue;
      if (all_node_channels_dying(nit->second, scid)) store_set_dying(nit->second.nann_rec);
    }
  }
  void kill_dying(u32 new_blockheight) {  // second loop of gossmap_manage_new_block, :1419-1436
    for (size_t i = 0; i < dying_channels.size(); i++) {
      if (dying_channels[i].deadline > new_blockheight) continue;
      const chan_dying cd = dying_channels[i];
      if (chans.count(cd.scid)) {  // kill_spent_channel, :1369-1387
        if (on_event) ev_text(LAMD_GEV_TRACE, false, nullptr, "Deleting channel " + fmt_scid(cd.scid) + " due to the funding outpoint being spent");
        remove_channel(cd.scid);
      }
      store_del(cd.rec);
      dying_channels.erase(dying_channels.begin() + i);
      i--;
    }
  }
  size_t prune_network() {  // :398-470; the caller owns the timer (GOSSIP_PRUNE_INTERVAL / 4)
    const int64_t highwater = (int64_t)now() - (int64_t)(cfg.prune_interval ? cfg.prune_interval : 1209600u);
    // gossmap's channel index order is the order of the announcements in the store
    std::vector<std::pair<u64, u64>> order;
    order.reserve(chans.size());
    for (const auto &kv : chans) order.emplace_back(kv.second.cann_rec, kv.first);
    std::sort(order.begin(), order.end());
    nodeid ours;
    memcpy(ours.k, cfg.our_id, 33);
    size_t pruned = 0;
    for (const auto &o : order) {
      auto it = chans.find(o.second);
      if (it == chans.end()) continue;
      const chan &c = it->second;
      const u32 ts0 = c.set[0] ? store[c.cupd_rec[0]].timestamp : 0xFFFFFFFFu, ts1 = c.set[1] ? store[c.cupd_rec[1]].timestamp : 0xFFFFFFFFu;  // get_timestamp: unknown = good
      if ((int64_t)ts0 >= highwater && (int64_t)ts1 >= highwater) continue;  // "both ends must refresh!"
      if (channel_already_dying(o.second)) continue;
      char b[96];
      if (on_event && (c.node[0] == ours || c.node[1] == ours)) {
        const int local = c.node[1] == ours;
        snprintf(b, sizeof b, ": local channel_update time %u, remote %u", local ? ts1 : ts0, local ? ts0 : ts1);
        ev_text(LAMD_GEV_TRACE, false, nullptr, "Pruning local channel " + fmt_scid(o.second) + " from gossip_store" + b);
      }
      if (on_event) {
        snprintf(b, sizeof b, " from network view (ages %u and %u)", ts0, ts1);
        ev_text(LAMD_GEV_TRACE, false, nullptr, "Pruning channel " + fmt_scid(o.second) + b);
      }
      remove_channel(o.second);
      pruned++;
    }
    return pruned;
  }

  static void sha256_single(const u8 *p, size_t len, u8 out[32]);
};

// single SHA-256 through the device header's host build (sha256.h): P2WSH scripts (71 bytes: two blocks, padded in place)
void lamd_gossipd::sha256_single(const u8 *p, size_t len, u8 out[32]) {
  u32 st[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au, 0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u};
  u32 w[16];
  size_t off = 0;
  for (; off + 64 <= len; off += 64) {
    for (int i = 0; i < 16; i++) w[i] = be32(p + off + 4 * i);
    lamd::sha256_compress(st, w);
  }
  u8 tail[128];
  const size_t rem = len - off, padded = rem < 56 ? 64 : 128;
  memset(tail, 0, sizeof tail);
  memcpy(tail, p + off, rem);
  tail[rem] = 0x80;
  put_be64(tail + padded - 8, (u64)len * 8);
  for (size_t b = 0; b < padded; b += 64) {
    for (int i = 0; i < 16; i++) w[i] = be32(tail + b + 4 * i);
    lamd::sha256_compress(st, w);
  }
  for (int i = 0; i < 8; i++) put_be32(out + 4 * i, st[i]);
}

extern "C" lamd_gossipd *lamd_gossipd_new(lamd_ctx *ctx, const lamd_gossipd_config *cfg, lamd_gossipd_event_fn on_event, void *user) {
  if (!cfg) return nullptr;
  lamd_gossipd *g = new (std::nothrow) lamd_gossipd;
  if (!g) return nullptr;
  g->ctx = ctx;
  g->cfg = *cfg;
  g->on_event = on_event;
  g->user = user;
  memset(&g->st, 0, sizeof g->st);
  if (const char *e = getenv("LAMD_INGEST_SUB")) g->sub_rows = (size_t)atoll(e) < 1 ? 1 : (size_t)atoll(e);
  if (const char *e = getenv("LAMD_INGEST_RUN_MIN")) g->run_min = (size_t)atoll(e);
  g->store_init();
  return g;
}
static void free_stages(ingest_stage *s);
extern "C" void lamd_gossipd_free(lamd_gossipd *g) {
  if (!g) return;
  delete g->pool;
  delete g->pool_bg;
  free_stages(g->w_stage);
  delete g;
}
// Both setters are DEFERRED while lamd_gossipd_process() runs (an event callback may call them): the planning thread reads cfg.now through
// timestamp_reasonable() and the verify thread the back end's pointers -- plan and apply of one batch must see one clock and one back end.  The new
// values take effect when process() returns.
extern "C" void lamd_gossipd_set_backend(lamd_gossipd *g, lamd_gossipd_sigcheck_fn sigcheck, lamd_gossipd_keyparse_fn keyparse, void *user) {
  if (!g) return;
  if (g->in_process) {
    g->deferred_be = true;
    g->d_be_sig = sigcheck; g->d_be_key = keyparse; g->d_be_user = user;
    return;
  }
  g->be_sig = sigcheck;
  g->be_key = keyparse;
  g->be_user = user;
}
extern "C" void lamd_gossipd_set_time(lamd_gossipd *g, uint64_t now) {
  if (!g) return;
  if (g->in_process) { g->deferred_now = true; g->d_now = now; return; }
  g->cfg.now = now;
}

extern "C" int lamd_gossipd_push(lamd_gossipd *g, const uint8_t *source_peer33, const uint8_t *msg, size_t len) {
  if (!g || (!msg && len) || len > 0xFFFFFFFFu) return LAMD_ERR_ARG;
  if (g->queue.size() > 500000) return LAMD_ERR_STATE;  // connectd/multiplex.c:832-833
  qent q;
  q.off = g->qarena.size();
  q.len = (u32)len;
  g->qarena.insert(g->qarena.end(), msg, msg + len);
  q.has_src = source_peer33 != nullptr;
  memset(q.src.k, 0, 33);
  if (source_peer33) memcpy(q.src.k, source_peer33, 33);
  g->queue.push_back(q);
  return LAMD_OK;
}

extern "C" int lamd_gossipd_push_batch(lamd_gossipd *g, size_t n, const uint8_t *source_peers33, size_t peer_stride, const uint8_t *msgs,
                                       const uint64_t *off) {
  if (!g || (n && (!msgs || !off))) return LAMD_ERR_ARG;
  if (g->queue.size() + n > 500000 + 1) return LAMD_ERR_STATE;
  const size_t base = g->qarena.size(), q0 = g->queue.size();
  g->qarena.resize(base + (size_t)(off[n] - off[0]));
  g->queue.resize(q0 + n);
  parallel_for(g->get_pool(), n, 8192, [&](size_t lo, size_t hi) {
    if (lo < hi) memcpy(&g->qarena[base + (size_t)(off[lo] - off[0])], msgs + off[lo], (size_t)(off[hi] - off[lo]));
    for (size_t i = lo; i < hi; i++) {
      qent &q = g->queue[q0 + i];
      q.off = base + (size_t)(off[i] - off[0]);
      q.len = (u32)(off[i + 1] - off[i]);
      q.has_src = source_peers33 != nullptr;
      if (source_peers33) memcpy(q.src.k, source_peers33 + peer_stride * i, 33);
      else memset(q.src.k, 0, 33);
    }
  });
  return LAMD_OK;
}

// messages [from, end) of a batch that could not be applied go back to the HEAD of the queue, in order (engine error paths)
static void requeue(lamd_gossipd *g, const bytes &arena, const std::vector<qent> &ents, size_t from) {
  bytes na;
  std::vector<qent> nq;
  nq.reserve(ents.size() - from + g->queue.size());
  for (size_t i = from; i < ents.size(); i++) {
    qent q = ents[i];
    const size_t o = na.size();
    na.insert(na.end(), arena.begin() + q.off, arena.begin() + q.off + q.len);
    q.off = o;
    nq.push_back(q);
  }
  for (qent q : g->queue) {  // whatever a callback pushed meanwhile stays behind them
    const size_t o = na.size();
    na.insert(na.end(), g->qarena.begin() + q.off, g->qarena.begin() + q.off + q.len);
    q.off = o;
    nq.push_back(q);
  }
  g->qarena.swap(na);
  g->queue.swap(nq);
}

// One sub-batch of a drained queue between its planning stage and its apply pass
struct ingest_stage {
  size_t lo = 0, hi = 0;
  lamd_gossipd::slotlist sl;
  std::vector<int8_t> v;       // verdict per slot
  bytes keyblob;
  std::vector<u8> keyok;
  lamd_gossipd::vbufs wb;
  lamd_gossipd::vcount cnt;
  bool has_cann = false;
  int rc = LAMD_OK;
  double t_plan = 0, t_slots = 0, t_verify = 0;
  // the waiting records the planning pass builds for its channel_announcements, one per message of the stage: planned::pre points in here (no
  // allocation per record, nothing to free; the record's message bytes are the one heap block that travels on into pending_ann)
  std::vector<pending_cannounce> pre_pool;
};
static void free_stages(ingest_stage *s) { delete[] s; }
static double ingest_now() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }

// ---- the planning stage of messages [st.lo, st.hi): which (message, signer) pairs can reach a sigcheck_*() call, judged from the maps as they
// stand, and ONE back-end call for their signatures (+ one for the keys of announcements that are dropped anyway).  Reads the maps, writes
// nothing of the ingest but `st` (and batch / plan entries of its own range): it may run on its own thread while an earlier sub-batch is applied,
// as long as that apply pass inserts nothing into the maps this stage reads (see lamd_gossipd_process).
static void ingest_stage1(lamd_gossipd *g, const bytes &arena, const std::vector<qent> &ents, std::vector<queued> &batch, std::vector<planned> &plan,
                          ingest_stage &st, thread_pool *pool) {
  const size_t lo = st.lo, n = st.hi - st.lo;
  const double t0 = ingest_now();
  // Pass 1 (parallel over the messages; reads the maps, writes nothing shared): framing, r/s range, the filters that need no
  // curve arithmetic, the expected signer, the content hash that keys the verdict, the P2WSH program of an announcement.
  st.wb.pool = pool;
  if (st.pre_pool.size() < n) st.pre_pool.resize(n);
  parallel_for(pool, n, 2048, [&](size_t l, size_t h) {
    for (size_t i = lo + l; i < lo + h; i++) {
      queued &q = batch[i];
      q.msg = mview(arena.data() + ents[i].off, ents[i].len);
      q.has_src = ents[i].has_src;
      q.src = ents[i].src;
      const mview &m = q.msg;
      planned &p = plan[i];
      const gossip_frame f = gossip_parse_frame(m.data(), m.size());
      p.type = f.type;
      p.malformed = f.bad;
      p.addrs_bad = false;
      p.slot = p.keyslot = -1;
      p.scid = 0;
      p.want_slot = p.want_keys = p.spk_set = false;
      p.signer = nullptr;
      p.pc = nullptr;
      p.h = 0;
      p.pre = nullptr;
      if (f.type == GOSSIP_CANN) {
        if (!p.malformed)
          for (int s = 0; s < 4; s++) p.malformed |= !sig_in_range(&m[2 + 64 * s]);
        if (p.malformed) continue;
        const size_t flen = be16(&m[258]);
        p.scid = be64(&m[260 + flen + 32]);
        const bool order_bad = memcmp(&m[f.keyoff], &m[f.keyoff + 33], 33) >= 0;
        const bool drop = order_bad || memcmp(&m[260 + flen], g->cfg.chain_hash, 32) != 0 || g->txout_failures.count(p.scid) || g->chans.count(p.scid);
        if (drop) {  // fromwire_pubkey still decides between "Malformed" and the silent drop / the node-order warning
          p.want_keys = true;
        } else {
          p.want_slot = true;
          // scriptpubkey_p2wsh(bitcoin_redeem_2of2(key1, key2)) (:699-702; bitcoin/script.c:149-165): keys in DER order
          const u8 *k1 = &m[f.keyoff + 66], *k2 = &m[f.keyoff + 99];
          if (memcmp(k1, k2, 33) >= 0) std::swap(k1, k2);
          u8 script[71];
          script[0] = 0x52; script[1] = 33; memcpy(script + 2, k1, 33);
          script[35] = 33; memcpy(script + 36, k2, 33);
          script[69] = 0x52; script[70] = 0xae;
          lamd_gossipd::sha256_single(script, sizeof script, p.spk);
          p.spk_set = true;
          pending_cannounce *pca = &st.pre_pool[i - st.lo];
          pca->msg.assign(m.begin(), m.end());
          pca->has_src = q.has_src;
          pca->src = q.src;
          memcpy(pca->node[0].k, &m[f.keyoff], 33);
          memcpy(pca->node[1].k, &m[f.keyoff + 33], 33);
          pca->spk.resize(34);
          pca->spk[0] = 0x00; pca->spk[1] = 0x20;
          memcpy(&pca->spk[2], p.spk, 32);
          p.pre = pca;
        }
      } else if (f.type == GOSSIP_CUPD) {
        if (!p.malformed) p.malformed = !sig_in_range(&m[2]);
        if (p.malformed) continue;
        p.scid = be64(&m[98]);
        if (memcmp(&m[66], g->cfg.chain_hash, 32) != 0 || !g->timestamp_reasonable(be32(&m[106]))) continue;
        auto it = g->chans.find(p.scid);
        if (it != g->chans.end()) { p.want_slot = true; p.pc = &it->second; p.signer = &it->second.node[m[111] & 1]; }
        else if (q.has_src) { p.want_slot = true; p.signer = &q.src; }  // the private-update probe of :1107-1109 (unused if the channel turns out to be pending)
      } else if (f.type == GOSSIP_NANN) {
        if (!p.malformed) p.malformed = !sig_in_range(&m[2]);
        if (p.malformed) continue;
        const size_t alen = be16(&m[f.keyoff + 68]);
        p.addrs_bad = !wireaddrs_ok(&m[f.keyoff + 70], alen);
        p.want_slot = !p.addrs_bad;
      }
      if (p.want_slot) p.h = g->vkey(m, p.signer).h;
    }
  });
  const double t1 = ingest_now();
  // Pass 2 (serial, in arrival order): slots -- identical (message, signer) pairs relayed by several peers share one -- and the
  // key-only list
  st.sl.reserve(n);
  st.sl.msg.reserve(n);
  st.sl.signer.reserve(n);
  st.sl.hs.reserve(n);
  for (size_t i = lo; i < st.hi; i++) {
    planned &p = plan[i];
    st.has_cann |= p.type == GOSSIP_CANN;
    if (p.want_slot) {
      p.slot = st.sl.add(&st.cnt.dups, batch[i].msg, p.signer, p.h);
    } else if (p.want_keys) {
      const mview &m = batch[i].msg;
      const gossip_frame f = gossip_parse_frame(m.data(), m.size());
      p.keyslot = (int)(st.keyblob.size() / 66);
      st.keyblob.insert(st.keyblob.end(), &m[f.keyoff + 66], &m[f.keyoff + 132]);
      st.cnt.keyparse++;
    }
  }
  const double t2 = ingest_now();
  st.t_plan = t1 - t0; st.t_slots = t2 - t1;
}
// ---- the stage's device calls: one for the signatures, one for the keys of announcements that are dropped anyway.  Touches nothing of the
// ingest but `st` and the back end (under be_mu): it runs on a thread of its own while the stage before it is applied and the one after it planned
static void ingest_stage_verify(lamd_gossipd *g, ingest_stage &st, thread_pool *pool) {
  const double t2 = ingest_now();
  st.wb.pool = pool;
  st.rc = g->verify_into(st.sl, st.v, st.wb, st.cnt);
  if (st.rc == LAMD_OK) {
    st.keyok.assign(st.keyblob.size() / 33, 0);
    if (!st.keyok.empty()) st.rc = g->backend_keyparse(st.keyok.size(), st.keyblob.data(), st.keyok.data());
  }
  st.t_verify = ingest_now() - t2;
}

extern "C" long lamd_gossipd_process(lamd_gossipd *g) {
  if (!g) return LAMD_ERR_ARG;
  if (g->in_process) return LAMD_ERR_STATE;
  bytes arena;
  std::vector<qent> ents;
  arena.swap(g->qarena);
  ents.swap(g->queue);
  const size_t n = ents.size();
  if (!n) return 0;
  // (work buffers live with the ingest and only grow: a drained queue of 400 k messages is ~100 MB of scratch, and first-touch page faults on
  // fresh memory cost more than the planning pass itself -- 2-20 us each under a hypervisor)
  static const bool prof0 = getenv("LAMD_INGEST_PROFILE") != nullptr;
  const double tp0 = prof0 ? ingest_now() : 0;
  std::vector<queued> &batch = g->w_batch;
  std::vector<planned> &plan = g->w_plan;
  reserve_prefaulted(g->get_pool(), batch, n);
  reserve_prefaulted(g->get_pool(), plan, n);
  if (batch.size() < n) batch.resize(n);
  if (plan.size() < n) plan.resize(n);
  const double tp1 = prof0 ? ingest_now() : 0;
  static const bool prof = getenv("LAMD_INGEST_PROFILE") != nullptr;
  // The drained queue is ONE batch to the caller and a pipeline inside: sub-batches of `sub` messages, the planning stage (framing, filters,
  // slots, the device call) of sub-batch k+1 on a second thread UNDER the apply pass of sub-batch k.  The planning stage reads only what no
  // apply pass changes: chans and txout_failures gain and lose entries in txout_reply / new_block / prune, never inside process().  (It does NOT
  // consult pending_ann / early_ann, which channel_announcements of the sub-batch being applied are inserted into: an announcement of a channel
  // that is already waiting for its txout is verified like a new one and dropped by the apply pass's own known_scid() check -- the same
  // outcome, four signatures wasted on a re-announcement.)  Planning from the maps as they stood a sub-batch earlier is what the single-batch
  // form does for the WHOLE queue: the apply pass re-checks every filter that state can change, and a pair the plan did not foresee is verified
  // late (stats.late_verifies).
  const size_t sub = g->sub_rows;   // (LAMD_INGEST_SUB / _RUN_MIN are read once, in lamd_gossipd_new)
  const size_t nsub = (n + sub - 1) / sub;
  if (!g->w_stage) g->w_stage = new ingest_stage[3];
  ingest_stage *stage = g->w_stage;
  auto setup = [&](ingest_stage &s, size_t k) {
    s.lo = k * sub; s.hi = std::min(n, (k + 1) * sub);
    s.sl.clear();
    s.v.clear(); s.keyblob.clear(); s.keyok.clear();
    s.cnt = lamd_gossipd::vcount();
    s.has_cann = false; s.rc = LAMD_OK;
  };
  // Three stages in flight: while sub-batch k is APPLIED (this thread, all cores for its runs), the signatures of sub-batch k+1 are on the device
  // (a thread that mostly waits) and sub-batch k+2 is PLANNED (a thread with half of the cores).  Before the loop: plan 0, then verify 0 next to plan 1.
  setup(stage[0], 0);
  ingest_stage1(g, arena, ents, batch, plan, stage[0], g->get_pool());
  g->st.sub_batches++;
  double t_grow = 0;
  {
    std::thread tb([&] { ingest_stage_verify(g, stage[0], nullptr); });
    // (every message of the batch may become a store record: grow the image and the record list once, not by doubling through the pass -- on a
    // thread of its own, next to the first device call and the second planning stage: no planning stage reads the store)
    std::thread tg([&] {
      const double t = ingest_now();
      thread_pool *pb = g->get_pool_bg();
      reserve_prefaulted(pb, g->image, g->image.size() + arena.size() + 12 * n + (g->image.size() >> 2));
      reserve_prefaulted(pb, g->store, g->store.size() + n + (g->store.size() >> 2));
      t_grow = ingest_now() - t;
    });
    if (nsub > 1) {
      setup(stage[1], 1);
      ingest_stage1(g, arena, ents, batch, plan, stage[1], g->get_pool());
      g->st.sub_batches++;
    }
    tb.join();
    tg.join();
  }
  const double tp2 = prof0 ? ingest_now() : 0;
  g->in_process = true;
  long ret = (long)n;
  if (prof0) fprintf(stderr, "[ingest] process n=%zu: work buffers %.1f ms, first two planning stages + first device call %.1f ms (store / image growth %.1f ms under them)\n", n, (tp1 - tp0) * 1e3, (tp2 - tp1) * 1e3, t_grow * 1e3);
  for (size_t k = 0; k < nsub; k++) {
    ingest_stage &cur = stage[k % 3], &nxt = stage[(k + 1) % 3], &nn = stage[(k + 2) % 3];
    if (cur.rc != LAMD_OK) {  // the device call of this sub-batch failed: it and everything behind it go back to the queue, unapplied
      g->drop_verdicts();
      requeue(g, arena, ents, cur.lo);
      ret = cur.rc;
      break;
    }
    std::thread bg, bv;
    const bool overlap = k + 1 < nsub;
    if (overlap) bv = std::thread([&] { ingest_stage_verify(g, nxt, nullptr); });   // sub-batch k+1 was planned an iteration ago
    // (the planning thread takes half of the cores while it shares the machine with the apply pass)
    if (k + 2 < nsub) {
      setup(nn, k + 2);
      thread_pool *pb = g->get_pool_bg();
      bg = std::thread([&, pb] { ingest_stage1(g, arena, ents, batch, plan, nn, pb); });
      g->st.sub_batches++;
    }
    if (overlap) g->st.overlapped_stages++;
    // ---- apply sub-batch k in arrival order.  Event callbacks fire from here: they must not re-enter lamd_gossipd_process / _txout_reply /
    // _new_block (those return LAMD_ERR_STATE while in_process is set -- answer LAMD_GEV_GET_TXOUT after process() returns; _push is fine).
    const double ta = prof ? ingest_now() : 0;
    g->drop_verdicts();
    g->cur_sl = &cur.sl;
    g->cur_v = cur.v;
    g->count(cur.cnt);
    g->fault_rc = LAMD_OK;
    if (cur.has_cann) g->pending_ann.reserve(g->pending_ann.size() + (cur.hi - cur.lo) / 2);
    const bool runs = g->run_ok();
    size_t fault_at = SIZE_MAX;
    for (size_t i = cur.lo; i < cur.hi; i++) {
      if (g->run_min != 0 && g->cann_run_member(plan[i], g->cur_v)) {  // a run of plain channel_announcements: all cores (apply_cann_run)
        size_t j = i + 1, members = 1;
        for (; j < cur.hi; j++) {
          if (g->cann_run_member(plan[j], g->cur_v)) members++;
          else if (!g->cann_run_side(plan[j], g->cur_v)) break;
        }
        while (!g->cann_run_member(plan[j - 1], g->cur_v)) j--;   // a run ends with a member
        if (members >= g->run_min) {
          g->apply_cann_run(batch, plan, i, j);
          i = j - 1;
          continue;
        }
      }
      if (runs && g->nann_run_member(plan[i], g->cur_v)) {  // a run of plain node_announcements: all cores (apply_nann_run)
        size_t j = i + 1;
        while (j < cur.hi && g->nann_run_member(plan[j], g->cur_v)) j++;
        if (j - i >= g->run_min) {
          g->apply_nann_run(batch, plan, g->cur_v, i, j);
          i = j - 1;
          continue;
        }
      }
      if (runs && g->run_member(plan[i], batch[i], g->cur_v)) {  // a run of plain updates of known channels: all cores (apply_cupd_run)
        size_t j = i + 1, members = 1;
        for (; j < cur.hi; j++) {
          if (g->run_member(plan[j], batch[j], g->cur_v)) members++;
          else if (!g->run_side(plan[j], g->cur_v)) break;
        }
        while (!g->run_member(plan[j - 1], batch[j - 1], g->cur_v)) j--;   // a run ends with a member
        if (members >= g->run_min) {
          g->apply_cupd_run(batch, plan, g->cur_v, i, j);
          i = j - 1;
          continue;
        }
      }
      // the apply pass is a chain of cache misses (channel -> its store records -> their bytes in the image): fetch ahead what the plan
      // already knows a channel_update will touch
      if (i + 16 < cur.hi && plan[i + 16].pc) __builtin_prefetch(plan[i + 16].pc);
      if (i + 8 < cur.hi && plan[i + 8].pc) {
        const chan *c = plan[i + 8].pc;
        const int dir = batch[i + 8].msg[111] & 1;
        if (c->set[dir]) { __builtin_prefetch(&g->store[c->cupd_rec[dir]]); }
        else if (!c->set[!dir]) { __builtin_prefetch(&g->store[c->cann_rec]); }
      }
      if (i + 4 < cur.hi && plan[i + 4].pc) {
        const chan *c = plan[i + 4].pc;
        const int dir = batch[i + 4].msg[111] & 1;
        if (c->set[dir]) __builtin_prefetch(g->image.data() + g->store[c->cupd_rec[dir]].off - 12);
        else if (!c->set[!dir]) { const u8 *h = g->image.data() + g->store[c->cann_rec].off - 12; __builtin_prefetch(h); __builtin_prefetch(h + 64); __builtin_prefetch(h + 448 - 64); }
      }
      // a channel_announcement probes three maps by its short_channel_id (failed txouts, channels, waiting announcements): their index slots
      // for the messages ahead
      if (i + 12 < cur.hi && plan[i + 12].type == GOSSIP_CANN) {
        const u64 sc = plan[i + 12].scid;
        g->chans.prefetch(sc);
        g->pending_ann.prefetch(sc);
        if (!g->txout_failures.empty()) g->txout_failures.prefetch(sc);
      }
      const queued &q = batch[i];
      planned &p = plan[i];
      if (p.type == GOSSIP_CANN) g->apply_cann(q, p, p.keyslot >= 0 ? (cur.keyok[2 * p.keyslot] && cur.keyok[2 * p.keyslot + 1]) : 1);
      else if (p.type == GOSSIP_CUPD) g->apply_cupd(q, p);
      else if (p.type == GOSSIP_NANN) g->apply_nann(q, p);
      // other types never reach gossipd's three handlers (gossipd.c:206-264)
      if (g->fault_rc != LAMD_OK) { fault_at = i; break; }  // an engine error in a late verify
      g->st.messages++;
    }
    const double tb = prof ? ingest_now() : 0;
    if (bg.joinable()) bg.join();
    if (bv.joinable()) bv.join();
    if (fault_at != SIZE_MAX) {  // message fault_at and everything after it go back to the queue, unapplied
      ret = g->fault_rc;
      g->fault_rc = LAMD_OK;
      g->drop_verdicts();
      requeue(g, arena, ents, fault_at);
      break;
    }
    if (prof)
      fprintf(stderr, "[ingest] sub-batch %zu/%zu n=%zu plan(parallel) %.1f ms, slots %.1f ms, verify %.1f ms | apply %.1f ms%s\n", k + 1, nsub, cur.hi - cur.lo,
              cur.t_plan * 1e3, cur.t_slots * 1e3, cur.t_verify * 1e3, (tb - ta) * 1e3, overlap ? " (the next sub-batch verified, the one after it planned under it)" : "");
  }
  g->drop_verdicts();
  g->in_process = false;
  if (g->deferred_now) { g->cfg.now = g->d_now; g->deferred_now = false; }
  if (g->deferred_be) { g->be_sig = g->d_be_sig; g->be_key = g->d_be_key; g->be_user = g->d_be_user; g->deferred_be = false; }
  if (prof0) fprintf(stderr, "[ingest] process n=%zu: %.1f ms in all\n", n, (ingest_now() - tp0) * 1e3);
  // the drained arena's memory serves the next queue (unless a callback or a requeue has already started one)
  if (g->queue.empty() && g->qarena.empty()) {
    arena.clear();
    ents.clear();
    g->qarena.swap(arena);
    g->queue.swap(ents);
  }
  return ret;
}

extern "C" int lamd_gossipd_txout_reply(lamd_gossipd *g, uint64_t scid, uint64_t sat, const uint8_t *script, size_t script_len) {
  if (!g) return LAMD_ERR_ARG;
  if (g->in_process) return LAMD_ERR_STATE;  // called from an event callback: the batch being applied assumes chans / the verdict slots stay put
  auto it = g->pending_ann.find(scid);
  if (it == g->pending_ann.end()) return LAMD_OK;  // :770-780
  pending_cannounce pca = std::move(it->second);
  g->pending_ann.erase(it);
  bool bad = false;
  if (script_len == 0) {
    bad = true;  // :789-809 (the rate-limited trace is not reproduced)
  } else if (script_len != pca.spk.size() || memcmp(script, pca.spk.data(), script_len) != 0) {
    g->peer_warning(pca.has_src, &pca.src, "channel_announcement: txout " + fmt_scid(scid) + " expected " + hexs(pca.spk.data(), pca.spk.size()) + ", got " + hexs(script, script_len));  // :811-817
    bad = true;
  }
  if (bad) {
    g->txout_failures[scid] = true;  // :868-869
    g->ev_scid(LAMD_GEV_TXOUT_FAILED, false, nullptr, scid);
    return LAMD_OK;  // (the reference does not reprocess the queues on this path either)
  }
  if (g->chans.count(scid)) return LAMD_OK;  // :825-846 "Redundant channel_announce"
  // :849-852
  chan c;
  c.node[0] = pca.node[0];
  c.node[1] = pca.node[1];
  c.set[0] = c.set[1] = false;
  c.cupd_rec[0] = c.cupd_rec[1] = 0;
  c.cann_rec = g->store_add(GOSSIP_CANN, 0, pca.msg.data(), pca.msg.size());
  u8 amt[10] = {0x10, 0x05};  // WIRE_GOSSIP_STORE_CHANNEL_AMOUNT = 4101
  for (int i = 0; i < 8; i++) amt[2 + i] = (u8)(sat >> (56 - 8 * i));
  g->store_add(4101, 0, amt, sizeof amt);
  g->chans.emplace(scid, c);
  for (int i = 0; i < 2; i++) {
    if (i == 1 && pca.node[1] == pca.node[0]) continue;  // a channel with itself is one entry of that node's list
    node &nd = g->nodes[pca.node[i]];  // value-initialised on first sight
    nd.nchans++;
    nd.scids.push_back(scid);
  }
  return g->reprocess_queued_msgs();  // :864
}

extern "C" int lamd_gossipd_txout_reply_batch(lamd_gossipd *g, size_t n, const uint64_t *scids, const uint64_t *sats, const uint8_t *scripts,
                                              const uint64_t *script_off, size_t *applied) {
  if (applied) *applied = 0;
  if (!g || (n && (!scids || !sats || !scripts || !script_off))) return LAMD_ERR_ARG;
  if (g->in_process) return LAMD_ERR_STATE;
  g->chans.reserve(g->chans.size() + n);
  reserve_prefaulted(g->get_pool(), g->store, g->store.size() + 2 * n);
  if (n >= 1024) reserve_prefaulted(g->get_pool(), g->image, g->image.size() + n * (size_t)(12 + 432 + 12 + 10));
  // A batch of replies with no listener and nothing waiting in the queues (reprocess_queued_msgs() would return at once after every reply), by
  // all cores in four passes that leave the maps, the record numbers and the store image exactly as reply-by-reply would:
  //   1 per short_channel_id SHARD (a worker owns whole shards of pending_ann / chans / txout_failures and takes its replies in arrival order):
  //     the waiting announcement leaves pending_ann, its script is compared, the channel enters chans;
  //   2 serial, light: record numbers and store offsets of the new channels in arrival order;
  //   3 per node-id SHARD: the two nodes of every new channel, in arrival order (a node's channel list comes out in the serial order);
  //   4 the two store records of every new channel -- channel_announcement + amount, 466 bytes of memcpy and crc32c -- at their offsets.
  if (!g->on_event && n >= 1024 && g->pending_cupdates.empty() && g->early_cupdates.empty() && g->pending_nannounces.empty()) {
    static const bool prof = getenv("LAMD_INGEST_PROFILE") != nullptr;
    const double t0 = prof ? ingest_now() : 0;
    struct newchan { pending_cannounce pca; chan *c; u64 scid, sat, rec, off; };
    std::unique_ptr<newchan[]> slot(new newchan[n]);
    std::vector<u8> took(n, 0);   // reply i created a channel (slot[i] is filled)
    thread_pool *pool = g->get_pool();
    const unsigned NSH = g->chans.shards();
    for (unsigned sh = 0; sh < NSH; sh++) { g->pending_ann.shard(sh).reserve(g->pending_ann.shard(sh).size() + 16); }
    parallel_for(pool, NSH, 1, [&](size_t lo, size_t hi) {
      for (size_t i = 0; i < n; i++) {
        const u64 scid = scids[i];
        const unsigned sh = g->chans.shard_of(scid);   // the three maps share traits and seed: one shard number for all of them
        if (sh < lo || sh >= hi) continue;
        auto &pend = g->pending_ann.shard(sh);
        auto it = pend.find(scid);
        if (it == pend.end()) continue;  // :770-780
        newchan &nc = slot[i];
        nc.pca = std::move(it->second);
        pend.erase(scid);
        const u8 *script = scripts + script_off[i];
        const size_t script_len = (size_t)(script_off[i + 1] - script_off[i]);
        if (script_len == 0 || script_len != nc.pca.spk.size() || memcmp(script, nc.pca.spk.data(), script_len) != 0) {  // :789-817 (the warning is an event)
          g->txout_failures.shard(sh)[scid] = true;  // :868-869
          continue;
        }
        chan c;
        c.node[0] = nc.pca.node[0];
        c.node[1] = nc.pca.node[1];
        c.set[0] = c.set[1] = false;
        c.cupd_rec[0] = c.cupd_rec[1] = 0;
        c.cann_rec = 0;
        const auto r = g->chans.shard(sh).emplace(scid, c);
        if (!r.second) continue;  // :825-846 "Redundant channel_announce"
        nc.c = &r.first->second;
        nc.scid = scid;
        nc.sat = sats[i];
        took[i] = 1;
      }
    });
    const double t1 = prof ? ingest_now() : 0;
    std::vector<u32> acc;
    acc.reserve(n);
    u64 nrec = g->store.size(), pos = g->image.size();
    for (size_t i = 0; i < n; i++) {
      if (!took[i]) continue;
      newchan &nc = slot[i];
      nc.rec = nrec;
      nc.off = pos + 12;
      nc.c->cann_rec = nrec;
      nrec += 2;
      pos += 12 + nc.pca.msg.size() + 12 + 10;
      acc.push_back((u32)i);
    }
    g->store.resize(nrec);
    g->image.resize(pos);
    const double t2 = prof ? ingest_now() : 0;
    parallel_for(pool, g->nodes.shards(), 1, [&](size_t lo, size_t hi) {
      for (const u32 i : acc) {
        const newchan &nc = slot[i];
        for (int k = 0; k < 2; k++) {
          if (k == 1 && nc.pca.node[1] == nc.pca.node[0]) continue;  // a channel with itself is one entry of that node's list
          const unsigned sh = g->nodes.shard_of(nc.pca.node[k]);
          if (sh < lo || sh >= hi) continue;
          node &nd = g->nodes.shard(sh)[nc.pca.node[k]];  // value-initialised on first sight
          nd.nchans++;
          nd.scids.push_back(nc.scid);
        }
      }
    });
    const double t3 = prof ? ingest_now() : 0;
    parallel_for(pool, acc.size(), 1024, [&](size_t lo, size_t hi) {
      for (size_t j = lo; j < hi; j++) {
        const newchan &nc = slot[acc[j]];
        const size_t mlen = nc.pca.msg.size();
        u8 *h = g->image.data() + nc.off - 12;
        put_be16(h, GS_COMPLETED); put_be16(h + 2, (u32)mlen); put_be32(h + 4, crc32c(0, nc.pca.msg.data(), mlen)); put_be32(h + 8, 0);
        memcpy(h + 12, nc.pca.msg.data(), mlen);
        g->store[nc.rec] = record{GOSSIP_CANN, 0, false, nc.off, (u32)mlen};
        u8 amt[10] = {0x10, 0x05};  // WIRE_GOSSIP_STORE_CHANNEL_AMOUNT = 4101
        for (int b = 0; b < 8; b++) amt[2 + b] = (u8)(nc.sat >> (56 - 8 * b));
        u8 *h2 = h + 12 + mlen;
        put_be16(h2, GS_COMPLETED); put_be16(h2 + 2, 10); put_be32(h2 + 4, crc32c(0, amt, 10)); put_be32(h2 + 8, 0);
        memcpy(h2 + 12, amt, 10);
        g->store[nc.rec + 1] = record{4101, 0, false, nc.off + mlen + 12, 10};
      }
    });
    if (prof) fprintf(stderr, "[ingest] txout replies n=%zu: maps by scid shard %.1f ms, numbering %.1f ms, nodes by shard %.1f ms, store records %.1f ms\n", n, (t1 - t0) * 1e3,
                      (t2 - t1) * 1e3, (t3 - t2) * 1e3, (ingest_now() - t3) * 1e3);
    if (applied) *applied = n;
    return LAMD_OK;
  }
  for (size_t i = 0; i < n; i++) {
    const int rc = lamd_gossipd_txout_reply(g, scids[i], sats[i], scripts + script_off[i], (size_t)(script_off[i + 1] - script_off[i]));
    if (rc != LAMD_OK) return rc;  // replies [0, *applied) took effect (reply i's channel is in the map as well; its waiting updates still wait)
    if (applied) *applied = i + 1;
  }
  return LAMD_OK;
}

extern "C" int lamd_gossipd_new_block(lamd_gossipd *g, uint32_t blockheight) {
  if (!g) return LAMD_ERR_ARG;
  if (g->in_process) return LAMD_ERR_STATE;
  g->cfg.blockheight = blockheight;
  for (auto it = g->early_ann.begin(); it != g->early_ann.end();) {  // :1362-1387, ascending scid
    const u64 scid = it->first;
    if (!scid_depth_announceable(scid, blockheight)) break;
    pending_cannounce pca = std::move(it->second);
    it = g->early_ann.erase(it);
    if (!g->pending_ann.emplace(scid, std::move(pca)).second) continue;
    g->ev_scid(LAMD_GEV_GET_TXOUT, false, nullptr, scid);
  }
  g->kill_dying(blockheight);  // :1419-1436 channels whose funding output was spent 72 blocks ago
  return LAMD_OK;
}

extern "C" int lamd_gossipd_channel_spent(lamd_gossipd *g, uint32_t blockheight, uint64_t scid) {
  if (!g) return LAMD_ERR_ARG;
  if (g->in_process) return LAMD_ERR_STATE;
  g->channel_spent(blockheight, scid);
  return LAMD_OK;
}

extern "C" long lamd_gossipd_prune(lamd_gossipd *g) {
  if (!g) return LAMD_ERR_ARG;
  if (g->in_process) return LAMD_ERR_STATE;
  return (long)g->prune_network();
}

extern "C" size_t lamd_gossipd_store_image(const lamd_gossipd *g, const uint8_t **data) {
  if (!g) return 0;
  if (data) *data = g->image.data();
  return g->image.size();
}

extern "C" void lamd_gossipd_get_stats(const lamd_gossipd *g, lamd_gossipd_stats *out) {
  if (!g || !out) return;
  *out = g->st;
  out->channels = g->chans.size();
  out->nodes = g->nodes.size();
  out->pending = g->pending_ann.size();
  out->early = g->early_ann.size();
  out->queued_updates = g->pending_cupdates.size() + g->early_cupdates.size();
  out->queued_nodes = g->pending_nannounces.size();
  out->store_records = g->store.size();
}

// ---- self-test of the ingest's map (stable_map / sharded_map) against std::unordered_map: `ops` random insertions, look-ups, erasures (by key and
// by iterator), operator[] and reserves over a key space small enough that keys recur, erased entries are re-used and tombstones pile up; after
// every few thousand operations the two maps are compared entry for entry in both directions (iteration included), and pointers handed out
// earlier must still point at their entries.  Returns 0, or the 1-based number of the first check that failed.  (tests/test_gossip_ingest.py)
extern "C" long lamd_gossipd_selftest_maps(uint64_t seed, long ops) {
  struct val { u64 a; std::vector<u32> v; };
  u64 st = seed * 0x9E3779B97F4A7C15ull + 1;
  auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
  long check = 0;
  {
    scid_map<val> m(seed ^ 0x5EED);
    std::unordered_map<u64, val> ref;
    std::unordered_map<u64, val *> where;   // entry addresses must never change while the entry lives
    const u64 space = 5000;
    for (long op = 0; op < ops; op++) {
      const u64 k = (rnd() % space) * 0x100000001ull + 7;
      const unsigned what = (unsigned)(rnd() % 16);
      if (what < 7) {
        val v{rnd(), std::vector<u32>((size_t)(rnd() % 5), (u32)op)};
        const auto r = m.emplace(k, v);
        const auto rr = ref.emplace(k, v);
        ++check;
        if (r.second != rr.second || r.first->first != k || r.first->second.a != rr.first->second.a) return check;
        if (r.second) where[k] = &r.first->second;
      } else if (what < 10) {
        const size_t e1 = m.erase(k), e2 = ref.erase(k);
        ++check;
        if (e1 != e2) return check;
        where.erase(k);
      } else if (what < 11) {
        auto it = m.find(k);
        ++check;
        if ((it != m.end()) != (ref.count(k) != 0)) return check;
        if (it != m.end()) { m.erase(it); ref.erase(k); where.erase(k); }
      } else if (what < 12) {
        val &x = m[k];
        val &y = ref[k];
        ++check;
        if (x.a != y.a || x.v != y.v) return check;
        x.a = y.a = rnd();
        where[k] = &x;
      } else if (what < 13 && (rnd() % 64) == 0) {
        m.reserve(m.size() + (size_t)(rnd() % 3000));
      } else {
        auto it = m.find(k);
        auto rt = ref.find(k);
        ++check;
        if ((it == m.end()) != (rt == ref.end()) || m.count(k) != ref.count(k)) return check;
        if (it != m.end() && (it->second.a != rt->second.a || it->second.v != rt->second.v || where[k] != &it->second)) return check;
        m.prefetch(k);
        m.prefetch_entry(k);
      }
      if (op % 4096 == 4095 || op + 1 == ops) {
        ++check;
        if (m.size() != ref.size() || m.empty() != ref.empty()) return check;
        size_t seen = 0;
        for (const auto &kv : m) {
          auto rt = ref.find(kv.first);
          if (rt == ref.end() || rt->second.a != kv.second.a || rt->second.v != kv.second.v) return check;
          seen++;
        }
        if (seen != ref.size()) return check;
        for (const auto &kv : ref) {
          auto it = m.find(kv.first);
          if (it == m.end() || it->second.a != kv.second.a || where[kv.first] != &it->second) return check;
        }
      }
    }
  }
  {  // node ids: the index holds a hash, equal hashes must still tell keys apart -- a seed-independent check with many keys in few slots is not
     // possible from outside, so: plain differential run over 33-byte keys
    sharded_map<nodeid, u64, nodeid_key_traits> m(seed ^ 0xABCD);
    std::map<std::string, u64> ref;
    for (long op = 0; op < ops / 4; op++) {
      nodeid id;
      const u64 r = rnd() % 3000;
      for (int i = 0; i < 33; i++) id.k[i] = (u8)((r >> ((i % 8) * 3)) + i * (r % 7));
      const std::string key((const char *)id.k, 33);
      if (rnd() % 3) { m[id] += op; ref[key] += (u64)op; }
      else { ++check; if (m.erase(id) != ref.erase(key)) return check; }
      ++check;
      auto it = m.find(id);
      auto rt = ref.find(key);
      if ((it == m.end()) != (rt == ref.end()) || (it != m.end() && it->second != rt->second)) return check;
    }
    ++check;
    if (m.size() != ref.size()) return check;
  }
  return 0;
}

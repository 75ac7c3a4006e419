// ba_pool.h -- process-wide caches of device blocks and of small pinned host blocks.
//
// MCPTAM makes one ChainBundle per BundleAdjust call (src/BundleAdjusterMulti.cc:75-76 builds it, fills it, runs it, drops it) and the
// call it makes most -- BundleAdjustRecent, src/BundleAdjusterBase.cc:188-265 -- is a few milliseconds of work: ~90 hipMalloc and as
// many hipFree (each of which waits for the device) per call were a quarter of it.  Blocks a handle drops go to per-device free lists
// by size class (four classes per octave, <= 25 % slack) and the next handle takes them from there; only what exceeds the cache's
// budget goes back to the driver.
//
// Contract of put(): nothing in flight may still touch the block.  hipFree() used to guarantee that by waiting for the device; put()
// does the same unless the caller says it has already drained whatever could touch the block (DevCache::Quiesced, set by
// mcp_ba_destroy after it has waited for the handle's streams).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <vector>

namespace mcp {

class DevCache {
 public:
  static DevCache& get() { static DevCache* c = new DevCache(); return *c; }      // (never destroyed: the HIP runtime may be gone at exit)
  // size class of a request: the smallest (4 + k) * 2^m >= bytes, k in 0..3, at least 512 bytes
  static size_t class_bytes(size_t bytes, int* cls = nullptr) {
    if (bytes < 512) bytes = 512;
    int m = 63 - __builtin_clzll((unsigned long long)bytes);          // 2^m <= bytes
    size_t base = (size_t)1 << m, step = base >> 2; int k = (int)((bytes - base + step - 1)/step);      // 0..4
    if (k == 4) { ++m; base <<= 1; step <<= 1; k = 0; }
    if (cls) *cls = 4*m + k;
    return base + (size_t)k*step;
  }
  // a block of at least `bytes` on the CURRENT device; *cap = what it really holds, *dev_out = that device (hand both back to put():
  // the thread that drops a block need not have the block's device current)
  void* take(size_t bytes, size_t* cap, int* dev_out = nullptr) {
    int cls; const size_t cb = class_bytes(bytes, &cls);
    int dev = 0; (void)hipGetDevice(&dev);
    if (dev_out) *dev_out = dev;
    if (enabled_ && cls < NCLS && dev < NDEV) {      // (a device ordinal beyond the table -- CPX mode exposes up to 64 -- bypasses the cache: plain hipMalloc / hipFree)
      std::lock_guard<std::mutex> lk(mu_);
      auto& v = free_[dev][cls];
      if (!v.empty()) { void* p = v.back(); v.pop_back(); cached_[dev] -= cb; *cap = cb; poison(p, cb, cls); return p; }
    }
    void* p = nullptr;
    if (hipMalloc(&p, cb) != hipSuccess) {
      // the cache may be what is holding the memory: give it all back and try once more
      trim(dev);
      if (hipMalloc(&p, cb) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    }
    *cap = cb;
    poison(p, cb, cls);
    return p;
  }
  void put(void* p, size_t cap, int dev = -1) {
    if (!p) return;
    int cur = 0; (void)hipGetDevice(&cur);
    if (dev < 0) dev = cur;
    if (!quiesced()) { if (dev != cur) (void)hipSetDevice(dev); (void)hipDeviceSynchronize(); if (dev != cur) (void)hipSetDevice(cur); }
    int cls; const size_t cb = class_bytes(cap, &cls);
    if (enabled_ && cb == cap && cls < NCLS && dev < NDEV) {
      std::lock_guard<std::mutex> lk(mu_);
      if (cached_[dev] + cb <= budget_) { free_[dev][cls].push_back(p); cached_[dev] += cb; return; }
    }
    (void)hipFree(p);
  }
  void trim(int dev) {
    if (dev < 0 || dev >= NDEV) return;
    std::vector<void*> all;
    { std::lock_guard<std::mutex> lk(mu_); for (auto& v : free_[dev]) { all.insert(all.end(), v.begin(), v.end()); v.clear(); } cached_[dev] = 0; }
    for (void* p : all) (void)hipFree(p);
  }
  size_t cached_bytes(int dev) { if (dev < 0 || dev >= NDEV) return 0; std::lock_guard<std::mutex> lk(mu_); return cached_[dev]; }
  // the calling thread has waited for everything that could touch the blocks it is about to put()
  struct Quiesced { Quiesced() { ++depth(); } ~Quiesced() { --depth(); } };
 private:
  // test aid (MCP_DEV_CACHE_POISON=1): every block is handed out filled with 0xFF -- NaNs, index -1 -- so that a read of something
  // never written shows instead of passing on whatever the block held; MCP_DEV_CACHE_POISON_CLASS=c: only size class c, the others
  // zero-filled (to find which buffer it is); MCP_DEV_CACHE_LOG=1 lists the classes handed out
  void poison(void* p, size_t cb, int cls) {
    const int nth = ++count_;
    if (log_) fprintf(stderr, "[dev cache] take #%d class %d (%zu bytes)\n", nth, cls, cb);
    if (!poison_) return;
    const bool hit = (poison_nth_ > 0) ? (nth == poison_nth_) : (poison_cls_ < 0 || poison_cls_ == cls);
    (void)hipMemset(p, hit ? 0xFF : 0x00, cb);
    (void)hipDeviceSynchronize();          // (the solver's streams do not wait for the null stream)
  }
  static int& depth() { static thread_local int d = 0; return d; }
  static bool quiesced() { return depth() > 0; }
  DevCache() {
    if (const char* e = getenv("MCP_DEV_CACHE_POISON")) poison_ = atoi(e) != 0;
    if (const char* e = getenv("MCP_DEV_CACHE_POISON_CLASS")) poison_cls_ = atoi(e);
    if (const char* e = getenv("MCP_DEV_CACHE_POISON_NTH")) poison_nth_ = atoi(e);      // only the n-th block handed out
    if (const char* e = getenv("MCP_DEV_CACHE_LOG")) log_ = atoi(e) != 0;
    if (const char* e = getenv("MCP_DEV_CACHE_MB")) { const long mb = atol(e); if (mb <= 0) enabled_ = false; else budget_ = (size_t)mb << 20; }
  }
  static constexpr int NDEV = 64, NCLS = 4*40;      // free lists by the REAL device ordinal (no masking: ordinal 16 is not device 0)
  std::mutex mu_;
  std::vector<void*> free_[NDEV][NCLS];
  size_t cached_[NDEV] = {0};
  size_t budget_ = (size_t)2 << 30;        // per device; MCP_DEV_CACHE_MB overrides (0 = no caching)
  bool enabled_ = true, poison_ = false, log_ = false; int poison_cls_ = -1, poison_nth_ = 0; std::atomic<int> count_{0};
};

// small pinned, device-mapped host blocks (the read-back block + trial mailboxes of a handle): hipHostMalloc/hipHostFree cost
// ~0.1 ms a pair.  Keyed by (bytes, flags); a handle that takes one clears it.
class PinnedCache {
 public:
  static PinnedCache& get() { static PinnedCache* c = new PinnedCache(); return *c; }
  // (keyed by the device that was current at allocation too: a mapped block is mapped into THAT device's address space)
  void* take(size_t bytes, unsigned flags) {
    int dev = 0; (void)hipGetDevice(&dev);
    {
      std::lock_guard<std::mutex> lk(mu_);
      for (size_t i = 0; i < free_.size(); ++i) if (free_[i].bytes == bytes && free_[i].flags == flags && free_[i].dev == dev) { void* p = free_[i].p; free_[i] = free_.back(); free_.pop_back(); return p; }
    }
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes, flags) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
  }
  void put(void* p, size_t bytes, unsigned flags, int dev) {
    if (!p) return;
    { std::lock_guard<std::mutex> lk(mu_); if (free_.size() < 64) { free_.push_back({p, bytes, flags, dev}); return; } }
    (void)hipHostFree(p);
  }
 private:
  struct B { void* p; size_t bytes; unsigned flags; int dev; };
  std::mutex mu_; std::vector<B> free_;
};

}  // namespace mcp

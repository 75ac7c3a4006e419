// ba_solver.hip -- host driver and C ABI of the MI355X ChainBundle back end.
//
// Mirrors the control flow of ChainBundle::Compute (/root/reference/src/ChainBundle.cc:1305-1451)
// with g2o's SparseOptimizer::optimize / OptimizationAlgorithmLevenberg::solve schedule
// (SURVEY.md Appendix A.5); all numeric work runs in the HIP kernels of ba_kernels.h,
// ba_select.h and ba_chol.h.  There is no CPU fallback.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <array>
#include <atomic>
#include <memory>
#include <mutex>
#include <sched.h>
#include <thread>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <list>
#include <map>
#include <string>
#include <vector>

#include "../../include/mcp_ba.h"
#include "ba_pool.h"
#include "ba_kernels.h"
#include "ba_select.h"
#include "ba_chol.h"
#include "ba_cut.h"
#include "ba_group.h"
#include "ba_schur4.h"
#include "ba_comm.h"
#include "ba_trial.h"
#include "ba_head.h"
#include "ba_small.h"
#include "ba_headl.h"

using namespace mcp;

static thread_local std::string g_err;
static void set_err(const std::string& s) { g_err = s; }
extern "C" const char* mcp_last_error(void) { return g_err.c_str(); }
void mcp_set_error(const char* s) { g_err = s; }      // shared with img_api.hip

// MCP_ERR_RUNTIME (-2): HIP / RCCL failure.  Kept apart from -1, which mcp_ba_compute also uses for the reference's
// legitimate "no iteration ran" outcome (ChainBundle.cc:1355-1366); the wrappers raise on -2.
constexpr int MCP_ERR_RUNTIME = -2;
static_assert(MCP_BA_MAX_FREE_POSES*6 == mcp::CH_SOLVE_MAX, "include/mcp_ba.h documents the solver's size limit");
#define HIPCK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { \
  set_err(std::string(#expr) + ": " + hipGetErrorString(e_)); return MCP_ERR_RUNTIME; } } while (0)
#define HIPCKV(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { \
  set_err(std::string(#expr) + ": " + hipGetErrorString(e_)); } } while (0)

static bool is_gfx950(int dev) {
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, dev) != hipSuccess) return false;
  return std::strncmp(p.gcnArchName, "gfx950", 6) == 0;
}
extern "C" int mcp_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) { set_err("hipGetDeviceCount failed: no HIP runtime/device"); return 0; }
  int c = 0;
  for (int i = 0; i < n; ++i) if (is_gfx950(i)) ++c;
  return c;
}

namespace {

template <class T> struct DevBuf {
  T* p = nullptr; size_t n = 0; size_t cap_bytes = 0; bool alias = false;      // alias: p points into another buffer (packed structure upload)
  int dev = -1;                                                                  // device the block lives on
  ~DevBuf() { release(); }
  // (blocks come from and go back to the process-wide cache, ba_pool.h: a handle lives for one BundleAdjust call)
  void release() { if (p && !alias) mcp::DevCache::get().put(p, cap_bytes, dev); p = nullptr; n = 0; cap_bytes = 0; alias = false; dev = -1; }
  int alloc(size_t count) {
    if (count == 0) count = 1;
    if (count <= n) return 0;
    release();
    p = (T*)mcp::DevCache::get().take(count*sizeof(T), &cap_bytes, &dev);
    if (!p) { set_err("hipMalloc: out of device memory"); cap_bytes = 0; return -1; }
    n = count; return 0;
  }
  template <class A> int upload(const std::vector<T, A>& v, hipStream_t st) {
    if (alloc(v.size())) return -1;
    if (!v.empty()) HIPCK(hipMemcpyAsync(p, v.data(), v.size()*sizeof(T), hipMemcpyHostToDevice, st));
    return 0;
  }
};

// cores this process may actually use: the affinity mask, capped by the cgroup CPU quota (a container often sees all of the
// host's logical CPUs in hardware_concurrency() while being allowed a fraction of them)
static int usable_cores() {
  int n = (int)std::thread::hardware_concurrency();
  cpu_set_t set; CPU_ZERO(&set);
  if (sched_getaffinity(0, sizeof set, &set) == 0) { const int c = CPU_COUNT(&set); if (c > 0) n = std::min(n > 0 ? n : c, c); }
  auto quota = [](const char* path, bool v2) -> double {
    FILE* f = std::fopen(path, "r"); if (!f) return 0.0;
    char a[64] = {0}, b[64] = {0}; double q = 0.0;
    if (v2) { if (std::fscanf(f, "%63s %63s", a, b) == 2 && std::strcmp(a, "max") != 0 && std::atof(b) > 0) q = std::atof(a)/std::atof(b); }
    else if (std::fscanf(f, "%63s", a) == 1 && std::atof(a) > 0) {
      FILE* g = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r");
      if (g) { if (std::fscanf(g, "%63s", b) == 1 && std::atof(b) > 0) q = std::atof(a)/std::atof(b); std::fclose(g); }
    }
    std::fclose(f); return q;
  };
  double q = quota("/sys/fs/cgroup/cpu.max", true);
  if (q <= 0.0) q = quota("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", false);
  if (q > 0.0) n = std::min(n, std::max(1, (int)(q + 0.5)));
  return std::max(n, 1);
}

// Persistent worker threads for the host side of Prepare() (structure build): run(fn) executes fn(tid) for tid in [0, size()),
// the caller taking tid 0, and returns when all are done.  Workers sleep on a condition variable between jobs.
// The CPUs of the NUMA node the calling thread runs on (intersected with what the process may use), or an empty set if that cannot be
// found out.  The structure build streams through arrays the CALLER allocated and first touched (the pinned staging arena, the host
// copy of the map): workers on the other socket of a two-socket host read and write all of it across the socket link -- measured on a
// 2 x 64-core host, cold Prepare() of the metric map: 6.4-8.3 ms with the workers wherever the scheduler put them (bimodal), 5.3-5.9 ms
// with everything on one node (profiles/r06/README.md).  MCP_BA_HOST_NUMA=0 leaves the workers unpinned.
static bool numa_node_cpus(cpu_set_t* out) {
  CPU_ZERO(out);
  if (const char* e = getenv("MCP_BA_HOST_NUMA")) if (atoi(e) == 0) return false;
  const int cpu = sched_getcpu();
  if (cpu < 0) return false;
  cpu_set_t allowed; CPU_ZERO(&allowed);
  if (sched_getaffinity(0, sizeof allowed, &allowed) != 0) return false;
  for (int node = 0; node < 64; ++node) {
    char path[96]; std::snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
    FILE* f = std::fopen(path, "r");
    if (!f) { if (node == 0) return false; break; }
    char buf[1024] = {0}; const bool got = std::fgets(buf, sizeof buf, f) != nullptr; std::fclose(f);
    if (!got) continue;
    cpu_set_t set; CPU_ZERO(&set); bool mine = false;
    for (char* p = buf; *p; ) {                       // "0-63,128-191"
      char* q = p; const long a = std::strtol(p, &q, 10); if (q == p) break;
      long b = a; if (*q == '-') { p = q + 1; b = std::strtol(p, &q, 10); }
      for (long c = a; c <= b && c < CPU_SETSIZE; ++c) { CPU_SET((int)c, &set); if (c == cpu) mine = true; }
      p = (*q == ',') ? q + 1 : q; if (*q != ',' ) break;
    }
    if (!mine) continue;
    int n = 0;
    for (int c = 0; c < CPU_SETSIZE; ++c) if (CPU_ISSET(c, &set) && CPU_ISSET(c, &allowed)) { CPU_SET(c, out); ++n; }
    return n >= 2;
  }
  return false;
}
class HostPool {
 public:
  explicit HostPool(int n) : n_(std::max(1, n)) {
    cpu_set_t node; const bool pin = numa_node_cpus(&node);
    for (int i = 1; i < n_; ++i) {
      th_.emplace_back([this, i] { worker(i); });
      if (pin) (void)pthread_setaffinity_np(th_.back().native_handle(), sizeof node, &node);      // (the caller's node; the caller itself stays where it is)
    }
  }
  ~HostPool() { { std::lock_guard<std::mutex> lk(m_); stop_ = true; ++gen_; } cv_.notify_all(); for (auto& t : th_) t.join(); }
  int size() const { return n_; }
  void run(const std::function<void(int)>& fn) {
    std::lock_guard<std::mutex> one_job(run_m_);
    { std::lock_guard<std::mutex> lk(m_); job_ = &fn; pending_ = n_ - 1; ++gen_; gen_a_.store(gen_, std::memory_order_release); }
    cv_.notify_all();
    fn(0);
    // the phases of one Prepare() follow each other within microseconds: poll before sleeping (workers do the same for the next job)
    for (int spin = 0; spin < 20000 && done_a_.load(std::memory_order_acquire) != gen_; ++spin) cpu_relax();
    std::unique_lock<std::mutex> lk(m_);
    done_.wait(lk, [this] { return pending_ == 0; });
    job_ = nullptr;
  }
 private:
  static void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__)
    __asm__ __volatile__("yield");
#endif
  }
  void worker(int tid) {
    unsigned long long seen = 0;
    for (;;) {
      const std::function<void(int)>* f;
      for (int spin = 0; spin < 40000 && gen_a_.load(std::memory_order_acquire) == seen; ++spin) cpu_relax();     // ~100 us, then sleep
      { std::unique_lock<std::mutex> lk(m_); cv_.wait(lk, [&] { return gen_ != seen; }); seen = gen_; if (stop_) return; f = job_; }
      (*f)(tid);
      { std::lock_guard<std::mutex> lk(m_); if (--pending_ == 0) { done_a_.store(seen, std::memory_order_release); done_.notify_one(); } }
    }
  }
  int n_; std::vector<std::thread> th_; std::mutex m_, run_m_; std::condition_variable cv_, done_;
  const std::function<void(int)>* job_ = nullptr; unsigned long long gen_ = 0; int pending_ = 0; bool stop_ = false;
  std::atomic<unsigned long long> gen_a_{0}, done_a_{0};
};

// Process-wide pinned staging arena for the structure arrays of Prepare(): the builders write them straight into page-locked
// memory (no page faults on fresh allocations, no zero fill, no pageable-to-pinned bounce inside hipMemcpyAsync), one bump
// allocation per array, everything released when the upload has completed.  One Prepare() at a time holds it.
class PinnedArena {
 public:
  std::mutex& mutex() { return m_; }
  void* alloc(size_t bytes) {
    bytes = (bytes + 255) & ~(size_t)255;
    if (blocks_.empty() || used_ + bytes > blocks_.back().size) {
      size_t want = std::max<size_t>(bytes, blocks_.empty() ? ((size_t)64 << 20) : 2*blocks_.back().size);
      Block b; b.size = want; b.pinned = true;
      if (hipHostMalloc(&b.p, want, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); b.p = std::malloc(want); b.pinned = false; if (!b.p) throw std::bad_alloc(); }
      blocks_.push_back(b); used_ = 0;
    }
    void* r = (char*)blocks_.back().p + used_; used_ += bytes; total_ += bytes;
    return r;
  }
  bool is_pinned(const void* p) const {      // (device-visible: a kernel may read it in place)
    for (auto& b : blocks_) if ((const char*)p >= (const char*)b.p && (const char*)p < (const char*)b.p + b.size) return b.pinned;
    return false;
  }
  void reset() {      // nothing of the arena is in use any more
    if (blocks_.size() > 1) {      // settle on one block that holds a whole build
      size_t sum = 0; for (auto& b : blocks_) { sum += b.size; if (b.pinned) (void)hipHostFree(b.p); else std::free(b.p); }
      blocks_.clear();
      Block b; b.size = sum; b.pinned = true;
      if (hipHostMalloc(&b.p, sum, hipHostMallocDefault) == hipSuccess) blocks_.push_back(b); else (void)hipGetLastError();
    }
    used_ = 0; total_ = 0;
  }
 private:
  struct Block { void* p = nullptr; size_t size = 0; bool pinned = false; };
  std::vector<Block> blocks_; size_t used_ = 0, total_ = 0; std::mutex m_;
};
static PinnedArena& pinned_arena() { static PinnedArena* a = new PinnedArena(); return *a; }      // (never destroyed: the HIP runtime may be gone at exit)
template <class T> struct ArenaAlloc {
  using value_type = T;
  ArenaAlloc() = default;
  template <class U> ArenaAlloc(const ArenaAlloc<U>&) {}
  T* allocate(size_t n) { return (T*)pinned_arena().alloc(n*sizeof(T)); }
  void deallocate(T*, size_t) {}
  // resize(n) default-initialises (no fill) -- the builders write every entry they need; assign(n, v) still fills
  template <class U, class... A> void construct(U* p, A&&... a) { if constexpr (sizeof...(A) == 0) ::new((void*)p) U; else ::new((void*)p) U(std::forward<A>(a)...); }
  template <class U> bool operator==(const ArenaAlloc<U>&) const { return true; }
  template <class U> bool operator!=(const ArenaAlloc<U>&) const { return false; }
};
template <class T> using avec = std::vector<T, ArenaAlloc<T>>;
// end of a build (any exit path): the copies out of the arena have to be done before it is handed to the next build
struct ArenaGuard { hipStream_t st; ~ArenaGuard() { if (st) (void)hipStreamSynchronize(st); pinned_arena().reset(); } };
// MCP_BA_HOST_THREADS: worker threads of the structure build (default: up to 16 of the cores this process may use; measured on a
// 256-thread host, cold Prepare() of the metric map: see profiles/r06/README.md)
static HostPool& host_pool() {
  static HostPool p([] { const char* e = getenv("MCP_BA_HOST_THREADS"); const int want = e ? atoi(e) : 16; return std::min(std::max(want, 1), usable_cores()); }());
  return p;
}

// Small uploads and fills of a Prepare() as ONE launch: up to eight ranges of 8-byte words, each copied from pinned host memory (read in place
// over the bus) or -- source null -- zeroed.  A copy-engine operation between kernels costs its 4-9 us plus ~9 us of queue switch on either
// side; a BundleAdjustRecent window made eleven of them per call.  Large uploads stay with the copy engine (UPSET_MAX_BYTES).
struct UploadSet { const unsigned long long* src[8]; unsigned long long* dst[8]; unsigned long long n8[8]; int n; };
constexpr size_t UPSET_MAX_BYTES = (size_t)1 << 20;
__global__ void k_upload_set(UploadSet U) {
  const int r = blockIdx.y;
  const size_t nth = (size_t)gridDim.x*blockDim.x;
  const unsigned long long* sp = U.src[r]; unsigned long long* dp = U.dst[r];
  for (size_t i = blockIdx.x*(size_t)blockDim.x + threadIdx.x; i < U.n8[r]; i += nth) dp[i] = sp ? sp[i] : 0ull;
}
static void upload_set_add(UploadSet& U, const void* src, void* dst, size_t bytes) {
  U.src[U.n] = (const unsigned long long*)src; U.dst[U.n] = (unsigned long long*)dst; U.n8[U.n] = (bytes + 7)/8; ++U.n;
}
static void upload_set_launch(const UploadSet& U, hipStream_t st) {
  if (U.n == 0) return;
  size_t mx = 0; for (int r = 0; r < U.n; ++r) mx = std::max<size_t>(mx, U.n8[r]);
  const unsigned bx = (unsigned)std::min<size_t>(64, std::max<size_t>(1, (mx + 255)/256));
  hipLaunchKernelGGL(k_upload_set, dim3(bx, U.n), dim3(256), 0, st, U);      // (callers follow with note_launch("k_upload_set"))
}

// std::allocator whose resize(n) leaves new elements uninitialised (bulk entries size the arrays once and fill every record)
template <class T> struct NoInitAlloc : std::allocator<T> {
  template <class U> struct rebind { using other = NoInitAlloc<U>; };
  NoInitAlloc() = default;
  template <class U> NoInitAlloc(const NoInitAlloc<U>&) {}
  template <class U, class... A> void construct(U* p, A&&... a) { if constexpr (sizeof...(A) == 0) ::new((void*)p) U; else ::new((void*)p) U(std::forward<A>(a)...); }
};

struct HPose { int id; int fixed; double T[12]; int unk; int active; };
struct HPoint { int id; int fixed; double x[3]; int chain; int unk; int active; };
struct HMeas { int chain, point, cam; double u, v, omega; };
struct HChain { int len; int v[MCP_MAX_CHAIN]; };

enum Stage { ST_EVAL = 0, ST_SELECT, ST_LIN, ST_SCHUR, ST_CHOL, ST_SOLVE, ST_UPDATE, ST_N };

}  // namespace

// ---- the solver's streams come from a per-device pool that is set up ONCE per process ----------------------------------------
// The HIP runtime multiplexes streams onto a few hardware queues (GPU_MAX_HW_QUEUES, 4 by default); two streams that land on the
// same queue execute strictly one after the other.  The solver lives on its main chain and its speculative chain running SIDE BY
// SIDE: measured on one box, the very same build ran 810 or 535 LM iterations/s depending only on how many streams
// the process had created before the handle's (which decides the queue each one gets).  A handle created per BundleAdjust call
// must not roll those dice every time: the first handle on a device creates candidate streams, MEASURES which ones really overlap
// with the main stream (two 150 us spin kernels: ~160 us together, ~310 us in sequence), keeps those, and every later
// handle borrows the same streams.  Handles that are alive at the same time share them -- stream order keeps each of them correct.
__global__ void k_spin_us(long long ticks) { const long long t0 = wall_clock64(); while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8); }
struct SolverStreams { hipStream_t main = nullptr, spec = nullptr, spec3 = nullptr, tr[mcp::MAX_SYS] = {nullptr, nullptr, nullptr, nullptr}; bool ok = false; int overlapping = 0; };
static SolverStreams& solver_streams(int device) {
  static std::mutex mu;
  static std::map<int, SolverStreams> pools;          // by device ordinal (a fixed array indexed by `device & 15` aliased devices 16+ onto 0-15)
  std::lock_guard<std::mutex> lk(mu);
  SolverStreams& P = pools[device];
  if (P.ok) return P;
  if (hipStreamCreateWithFlags(&P.main, hipStreamNonBlocking) != hipSuccess) return P;
  const bool calibrate = [] { const char* e = getenv("MCP_BA_STREAM_CALIBRATE"); return e ? atoi(e) != 0 : true; }();
  hipEvent_t e0 = nullptr, e1 = nullptr, ej = nullptr;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); (void)hipEventCreateWithFlags(&ej, hipEventDisableTiming);
  int rate_khz = 100000; (void)hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, device);      // wall_clock64 ticks per ms
  const long long ticks = (long long)rate_khz*150/1000;                                                    // ~150 us
  auto overlaps = [&](hipStream_t other) -> bool {
    if (!calibrate || !e0 || !e1 || !ej) return true;
    float best = 1e9f;
    for (int rep = 0; rep < 2; ++rep) {      // (the first repetition also pays the kernel's first launch; the minimum counts)
      (void)hipEventRecord(e0, P.main);
      (void)hipStreamWaitEvent(other, e0, 0);
      hipLaunchKernelGGL(k_spin_us, dim3(1), dim3(64), 0, P.main, ticks);
      hipLaunchKernelGGL(k_spin_us, dim3(1), dim3(64), 0, other, ticks);
      (void)hipEventRecord(ej, other);
      (void)hipStreamWaitEvent(P.main, ej, 0);
      (void)hipEventRecord(e1, P.main);
      if (hipEventSynchronize(e1) != hipSuccess) return true;
      float ms = 0; if (hipEventElapsedTime(&ms, e0, e1) == hipSuccess) best = std::min(best, ms);
    }
    if (getenv("MCP_BA_TRACE")) fprintf(stderr, "[mcp_ba streams]   candidate: two 150 us kernels took %.0f us\n", best*1e3);
    return best < 0.230f;                  // together ~0.16 ms, in sequence ~0.31 ms
  };
  // up to 12 candidates for the 4 side streams (speculative chain, a second one for MCP_BA_OVERLAP=2, two trial streams); the ones that
  // share the main stream's queue are kept alive unused, so that no later stream inherits their place
  hipStream_t* want[4] = { &P.spec, &P.tr[2], &P.tr[3], &P.spec3 };
  int got = 0;
  // MCP_BA_SPEC_CUS=n (experiment): the side streams may only use the first n compute units (hipExtStreamCreateWithCUMask)
  const int spec_cus = [] { const char* e = getenv("MCP_BA_SPEC_CUS"); return e ? atoi(e) : 0; }();
  for (int c = 0; c < 12 && got < 4; ++c) {
    hipStream_t s2 = nullptr;
    if (spec_cus > 0) {
      uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int b = 0; b < spec_cus && b < 256; ++b) mask[b >> 5] |= 1u << (b & 31);
      if (hipExtStreamCreateWithCUMask(&s2, 8, mask) != hipSuccess) { (void)hipGetLastError(); s2 = nullptr; }
    }
    if (!s2 && hipStreamCreateWithFlags(&s2, hipStreamNonBlocking) != hipSuccess) break;
    if (overlaps(s2)) { *want[got++] = s2; }
  }
  P.overlapping = got;
  for (int k = got; k < 4; ++k) (void)hipStreamCreateWithFlags(want[k], hipStreamNonBlocking);        // not enough queues: plain streams (still correct)
  if (e0) (void)hipEventDestroy(e0); if (e1) (void)hipEventDestroy(e1); if (ej) (void)hipEventDestroy(ej);
  (void)hipGetLastError();
  P.ok = P.main && P.spec;
  if (getenv("MCP_BA_TRACE")) fprintf(stderr, "[mcp_ba streams] device %d: %d side streams overlap with the main stream%s\n", device, got, calibrate ? "" : " (not measured)");
  return P;
}

// ---- structure cache (round 5) ---------------------------------------------------------------------------------------------------
// MCPTAM builds a fresh ChainBundle per BundleAdjust call (src/BundleAdjusterMulti.cc:75) and calls again and again while the map has
// not converged (MapMaker::run: `if (!mbBundleConverged_Full) BundleAdjustAll()`, likewise the recent window) -- on a map whose
// TOPOLOGY (which chains, which point hangs off which chain, who measures what, what is fixed) is that of the call before, only the
// numbers have moved.  Everything Prepare() builds from the topology -- point order, slots, incidences, groups, staging plan, the
// packed device block -- is kept here, keyed by a 128-bit hash of the topology: a handle that brings the same topology adopts the
// host-side results, clones the device block (a device-to-device copy) and uploads only its own numbers (cameras, measurement
// values; poses and points go up as always).  A cached Prepare() leaves bit for bit the device state of a cold one.
struct StructKey {
  unsigned long long h0 = 0, h1 = 0; int npose = 0, npoint = 0, nmeas = 0, nchain = 0, dev = 0, flags = 0;
  bool operator==(const StructKey& o) const { return h0 == o.h0 && h1 == o.h1 && npose == o.npose && npoint == o.npoint && nmeas == o.nmeas && nchain == o.nchain && dev == o.dev && flags == o.flags; }
};
struct StructEntry {
  StructKey key;
  std::vector<int> pose_unk, pt_unk, fp_pose, fl_point, perm;
  std::vector<unsigned char> pose_active, pt_active, pat;
  std::vector<int> chol_segs;
  int nfp = 0, nfl = 0, np = 0, nx = 0, nsp = 0, ninc = 0, nslot = 0, ngroup = 0, nbig = 0, grp_pts = 0, grp_blk_max = 0, grp_inc_max = 0, nrhs_rows = 0;
  size_t nstage = 0; bool asm_long = false, sch4_ok = false, sch4_order = false;
  double m_total = 0, nfl_total = 0, schur_mfma = 0, schur_flops = 0;
  std::vector<size_t> counts;            // element count of every array of the packed block, in layout order
  // for the adoption of this structure by a map that holds the same poses, points and chains and a SUBSET of these measurements (the
  // map after MCPTAM erased the outliers of the adjustment before: prepare(), "near miss"): the key without the measurements, and
  // the measurements' (point, chain, camera) in add order
  StructKey base;
  std::vector<int, NoInitAlloc<int>> add_point, add_chain, add_cam;      // (filled by the host pool in finish_prepare: no zero fill first)
  char* dblock = nullptr; size_t dbytes = 0, dcap = 0; int ddev = -1;      // the device clone (DevCache block)
  size_t host_bytes() const { return (pose_unk.size() + pt_unk.size() + fp_pose.size() + fl_point.size() + perm.size() + add_point.size() + add_chain.size() + add_cam.size())*4 + pose_active.size() + pt_active.size() + pat.size(); }
  ~StructEntry() { if (dblock) DevCache::get().put(dblock, dcap, ddev); }
};
class StructCache {
 public:
  static StructCache& get() { static StructCache* c = new StructCache(); return *c; }      // (never destroyed: the HIP runtime may be gone at exit)
  bool enabled() const { return budget_ > 0; }
  std::shared_ptr<StructEntry> find(const StructKey& k) {
    std::lock_guard<std::mutex> lk(mu_);
    for (auto it = lru_.begin(); it != lru_.end(); ++it) if ((*it)->key == k) { auto e = *it; lru_.erase(it); lru_.push_front(e); ++hits_; return e; }
    ++misses_;
    return nullptr;
  }
  // entries over the same poses / points / chains that hold MORE measurements than `nmeas` (most recently used first)
  std::vector<std::shared_ptr<StructEntry>> find_supersets(const StructKey& base, int nmeas) {
    std::vector<std::shared_ptr<StructEntry>> out;
    std::lock_guard<std::mutex> lk(mu_);
    for (auto& e : lru_) if (e->base == base && e->key.nmeas > nmeas && !e->add_point.empty()) out.push_back(e);
    return out;
  }
  void touch(const std::shared_ptr<StructEntry>& e) { std::lock_guard<std::mutex> lk(mu_); for (auto it = lru_.begin(); it != lru_.end(); ++it) if (*it == e) { lru_.erase(it); lru_.push_front(e); ++near_; break; } }
  long long near_hits() { std::lock_guard<std::mutex> lk(mu_); return near_; }
  void insert(std::shared_ptr<StructEntry> e) {
    std::vector<std::shared_ptr<StructEntry>> dropped;          // (released outside the lock: dropping a device block may wait for the device)
    {
      std::lock_guard<std::mutex> lk(mu_);
      for (auto it = lru_.begin(); it != lru_.end(); ++it) if ((*it)->key == e->key) { dropped.push_back(*it); lru_.erase(it); break; }
      lru_.push_front(e);
      size_t tot = 0; int n = 0;
      for (auto it = lru_.begin(); it != lru_.end(); ) {
        tot += (*it)->dcap + (*it)->host_bytes(); ++n;
        if (n > 1 && (tot > budget_ || n > max_entries_)) { dropped.push_back(*it); it = lru_.erase(it); } else ++it;
      }
    }
  }
  void clear() { std::vector<std::shared_ptr<StructEntry>> d; { std::lock_guard<std::mutex> lk(mu_); d.assign(lru_.begin(), lru_.end()); lru_.clear(); } }
  void stats(long long* h, long long* m) { std::lock_guard<std::mutex> lk(mu_); *h = hits_; *m = misses_; }
 private:
  StructCache() { if (const char* e = getenv("MCP_BA_STRUCT_CACHE_MB")) budget_ = (size_t)std::max(0L, atol(e)) << 20; if (const char* e = getenv("MCP_BA_STRUCT_CACHE")) if (atoi(e) == 0) budget_ = 0; }
  std::mutex mu_; std::list<std::shared_ptr<StructEntry>> lru_;
  size_t budget_ = (size_t)512 << 20; int max_entries_ = 32; long long hits_ = 0, misses_ = 0, near_ = 0;
};
// two independent 64-bit hashes of an int array, block by block (the value does not depend on how the work is split over threads)
static inline unsigned long long mix64(unsigned long long x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }
static void hash_block(const int* p, size_t n, size_t stride_ints, unsigned long long seed, unsigned long long& a, unsigned long long& b) {
  unsigned long long x = 0x9e3779b97f4a7c15ull ^ seed, y = 0xc2b2ae3d27d4eb4full + seed;
  for (size_t i = 0; i < n; ++i) {
    const unsigned long long v = (unsigned int)p[i*stride_ints];
    x = (x ^ v)*0x100000001b3ull; x ^= x >> 29;
    y = (y + v)*0x9fb21c651e98df25ull; y ^= y >> 31;
  }
  a = mix64(x); b = mix64(y);
}

struct mcp_ba {
  int device = 0;
  hipStream_t st = nullptr;
  bool pooled = false;             // the streams belong to the device's pool (solver_streams): never destroyed by a handle
  // second stream: the speculative systems of a solve are built and factored here while the main stream already factors the
  // system the trial needs (both chains are latency-bound and overlap); ev_fork / ev_spec order the two
  // (MCP_BA_OVERLAP: 0 = one stream, 1 = the speculative systems together on a second stream, 2 = system 1 on the second and
  // systems 2.. on a third stream, so that the system the SECOND trial needs is ready as early as the first one's)
  hipStream_t st2 = nullptr, st3 = nullptr; hipEvent_t ev_fork = nullptr, ev_spec = nullptr, ev_spec3 = nullptr;
  bool spec_pending = false, spec3_pending = false; int spec2_from = 1, spec3_from = MAX_SYS; int overlap_spec = 1, main_sys = 1; bool overlap_auto = true;
  std::vector<mcp_camera> cams;
  int robust = 1, tukey = 1, verbose = 0;
  mcp_ba_params prm;

  std::vector<HPose> poses;
  std::vector<HPoint> points;
  std::vector<HMeas, NoInitAlloc<HMeas>> meas;
  std::vector<int, NoInitAlloc<int>> meas_point;      // meas[i].point, compact: the bucketing by point at Prepare() reads 4 bytes per measurement instead of 40
  std::vector<int, NoInitAlloc<int>> meas_chain;      // meas[i].chain, likewise
  std::vector<HChain> chains;
  std::map<std::array<int, 1 + MCP_MAX_CHAIN>, int> chain_map;
  std::vector<int> id_kind, id_index;   // by id; kind 1 pose, 2 point
  int next_id = 1;
  bool dirty = true;

  // structure
  int nfp = 0, nfl = 0, np = 0, nx = 0, ninc = 0, nslot = 0;
  std::vector<int> perm;           // sorted position -> add-order index
  std::vector<int> fl_point, fp_pose;
  double m_total = 0;              // global measurement count (all ranks)
  double nfl_total = 0;            // global free-point count (all ranks)
  int nx_total() const { return np + 3*(int)nfl_total; }

  // device problem
  DevBuf<char> d_struct;          // the structure arrays of Prepare(), packed (finish_prepare): the d_* of this block alias into it
  DevBuf<mcp_camera> d_cams;
  DevBuf<int> d_chain_len, d_chain_pose, d_pose_unk, d_pt_chain, d_pt_unk, d_m_pt, d_m_chain;
  DevBuf<unsigned char> d_pt_fixed, d_m_cam, d_flags;
  DevBuf<unsigned short> d_m_mask;
  DevBuf<double> d_m_u, d_m_v, d_m_omega;
  DevBuf<int> d_slot_start, d_slot_unk, d_slot_inc, d_l_i0, d_l_i1, d_inc_unk, d_fl_point;
  DevBuf<int> d_sp_pt, d_sp_m, d_sp_i, d_m_sp, d_l_sp, d_g_sp0, d_g_pose, d_red_tiles;
  DevBuf<double> d_pack; int n_red_tiles = 0;
  DevBuf<unsigned char> d_sp_big, d_slot_lp, d_slot_first, d_inc_lp, d_inc_mixed;
  int nsp = 0, ngroup = 0, nbig = 0;
  // state: the current one + one candidate per system of a multi-lambda batch (trial q of an iteration writes candidate q;
  // an accepted trial's candidate becomes current by index, nothing is copied)
  static constexpr int NSTATE = 1 + MAX_SYS;
  DevBuf<double> d_pose[NSTATE], d_pt[NSTATE], d_first[NSTATE], d_second[NSTATE], d_last[NSTATE], d_chi2[NSTATE];
  int cur = 0, last_tr = 1;
  int cand(int q) const { for (int s_ = 0, k = 0;; ++s_) { if (s_ == cur) continue; if (k == q) return s_; ++k; } }
  // Speculative trial evaluation: the trials of an iteration differ only in lambda, so the step of every speculatively solved
  // system is applied and evaluated (pose/point update, chains, residuals, robust sums) on the stream that solved it, right
  // behind its back-substitution -- the host finds the result of a rejected trial's successor in its mailbox instead of
  // launching five dependent kernels and waiting.  Same kernels on the same inputs as a trial run in sequence: same numbers.
  DevBuf<double> d_sxp[MAX_SYS], d_sxl[MAX_SYS], d_sp0[MAX_SYS], d_sp1[MAX_SYS], d_sp2[MAX_SYS];
  bool pre_run[MAX_SYS] = {false, false, false, false}, ahead_enq[MAX_SYS] = {false, false, false, false}; unsigned long long pre_ticket[MAX_SYS] = {0, 0, 0, 0};
  hipEvent_t ev_tr[MAX_SYS] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t ev_wf[MAX_SYS] = {nullptr, nullptr, nullptr, nullptr};      // trial q evaluated ahead: its last reader of the linearisation (k_backsub) is through
  // MCP_BA_SPEC_TRIALS: 0 = trials strictly in sequence; 1 = one trial ahead on the stream that solved it; 2 (default since round 6)
  // = the steps of ALL speculatively solved systems are applied and evaluated as soon as their solutions exist, each on its own
  // stream (st_tr[q], behind the speculative chain's event).  Round 3 measured 819 / 835 it/s (1) vs 733 / 743 (2) -- three trial
  // evaluations at once took the compute units from the trial the host was waiting for; with a trial's step in one launch and the
  // evaluation at 17 us the balance has turned: 1518 / 1519 / 1610 (1) vs 1543 / 1538 / 1629 (2), pairs within one run.
  int spec_trials = 2;
  hipStream_t st_tr[MAX_SYS] = {nullptr, nullptr, nullptr, nullptr};
  hipStream_t tr_stream[MAX_SYS] = {nullptr, nullptr, nullptr, nullptr};      // where the trial ahead of system q was enqueued
  // system
  DevBuf<double> d_ubig;    // [U (np*np) | bp (np)] contributions of the points outside the groups (> GRP_LMAX poses); only if nbig
  DevBuf<double> d_red;     // [S (np*np) | rhs (np) | bp = J^T r (np)]   (the all-reduced block)
  // staged local tiles of the groups and the assembly plan (ba_group.h: fixed-order accumulation)
  DevBuf<double> d_stU, d_stb, d_stS, d_str, d_udiag;
  DevBuf<int> d_g_blk0, d_asm_tiles, d_pair_id, d_pr_start, d_blk_dst, d_po_start, d_rhs_dst;
  DevBuf<unsigned char> d_blk_pair;
  int grp_blk_max = 0;            // most staged blocks of any group (sizes the LDS tile of k_linearize_group)
  int head_ahead_for = -1;        // state buffer whose iteration head is already on the main stream (small bundles), or -1
  bool head_ahead_want = false; int dbg_head_ahead = 0;
  int head_ahead(int w);
  // The same for maps beyond the small-bundle limit (round 6, MCP_BA_HEAD_AHEAD; ba_head.h): EVERY trial of an iteration -- the one on
  // the main stream and the ones evaluated ahead on the second -- counts the digit histograms of its |chi2| while it evaluates them
  // and is followed by ONE kernel that finishes the median and writes the sigma block of the iteration its acceptance would start,
  // each trial into its own scratch and its own sigma / start block: slot [parity of the iteration][q].  The accepted trial's blocks
  // become the current ones (sig_idx, start_blk); the others are never looked at.
  // MEASURED (profiles/r06/README.md): with the heads on, a one-trial iteration reaches its linearisation 11 us after it starts (65 without)
  // and every iteration finds its sigma^2 ready -- but the two extra launches + events per trial on the host and the heads' kernels
  // beside the trials' cost more than that saves on the metric map: 1290-1335 it/s with them, 1335-1385 without, same box and run.
  // So the default is OFF; MCP_BA_HEAD_AHEAD=1 switches every trial's head on, =2 only the main stream's.  Bit-identical either way.
  int large_head_ahead = 0;
  struct HeadScratch { unsigned int* hist = nullptr; double* vals = nullptr; double* part = nullptr; double* out = nullptr; double* rs = nullptr; };
  HeadScratch hsc[2*MAX_SYS]; DevBuf<double> d_headbuf;
  hipStream_t st_h = nullptr; bool st_h_own = false;      // the heads of trials evaluated ahead run here: the second stream stays free for the NEXT trial ahead
  hipEvent_t head_ev[MAX_SYS] = {nullptr, nullptr, nullptr, nullptr};
  bool head_enq[MAX_SYS] = {false, false, false, false}; int head_state[MAX_SYS] = {-1, -1, -1, -1};
  unsigned long long head_ticket[MAX_SYS] = {0, 0, 0, 0}, head_ticket_ctr = 0;
  int head_par = 0, acc_head_q = -1, dbg_head_miss = 0;
  const double* start_blk = nullptr;      // the iteration-start block the first trial forwards to the host (nullptr: d_res[24..28])
  unsigned int* head_hist(int q) { return hsc[head_par*MAX_SYS + q].hist; }
  int enqueue_head(hipStream_t s, int q, int w, bool side);
  int wait_head(int q);
  bool large_heads() const { return large_head_ahead && robust && !multi() && !small_mode() && d_headbuf.p != nullptr; }
  int head_small(int w, bool sum_aside = false);
  int join_sum();
  int sum_aside(); int sum_w = -1; const double* sum_sig = nullptr;
  hipEvent_t ev_head = nullptr, ev_sum = nullptr; bool sum_pending = false;
  DevBuf<double> d_parth;         // partial sums of the robust chi2 taken on the second stream (head_small)
  int small_on = 1;               // MCP_BA_SMALL=0: a small bundle runs the same launches as a large one (ba_small.h)
  bool small_mode_for(int nmeas_, int nchain_) const { return small_on && !multi() && nmeas_ > 0 && nmeas_ <= SMALL_MEAS && nchain_ <= SMALL_CHAINS; }
  bool small_mode() const { return small_mode_for(P.nmeas, P.nchain); }
  int grp_pts = GRP_PTS;          // points per group: GRP_PTS, or LIN_QUAD_PTS for a map of few points (k_linearize_quad: four lanes per point)
  static int group_points(int nsp) {
    const char* e = getenv("MCP_BA_SMALL_POINTS"); const int small_pts = e ? atoi(e) : 16384;      // (0: the large-map layout for every map)
    return (nsp <= small_pts) ? LIN_QUAD_PTS : GRP_PTS;
  }
  bool fail_clean = false;        // d_fail was cleared by the linearisation kernel and no system has been built since
  // round 5 (ba_schur4.h): all systems of a batch in one Schur workgroup.  Groups are closed at S4_LMAX poses (a single point
  // that sees more keeps its group beyond that: then no group of the map takes the new kernel, sch4_ok = false)
  static bool env_on(const char* name, bool dflt) { const char* e = getenv(name); return e ? atoi(e) != 0 : dflt; }
  bool point_order_refine = env_on("MCP_BA_POINT_ORDER", true);     // secondary point order (refine_point_order)
  int grp_lmax_pol = env_on("MCP_BA_GROUP_LMAX13", true) ? S4_LMAX : GRP_LMAX;
  bool sch4_on = env_on("MCP_BA_SCHUR4", true), sch4_ok = false;
  double schur_mfma = 0, schur_flops = 0;       // (profile only) what one system's Schur launch executes / stands for: mcp_ba_timing
  std::vector<unsigned char> last_pat;          // tile occupancy the factorisation plan was built from (kept for the structure cache)
  std::string launch_err;          // first launch of this solve that the runtime refused: "kernel: reason" (note_launch)
  void note_launch(const char* k) { const hipError_t e = hipGetLastError(); if (e != hipSuccess && e != hipErrorNotReady && launch_err.empty()) launch_err = std::string(k) + ": " + hipGetErrorString(e); }
  std::shared_ptr<StructEntry> pending_entry;
  StructKey cache_key, cache_base; bool cache_insert = false;      // a cold Prepare() of a cacheable map leaves its results in the structure cache
  // near miss: the structure of a superset of this map's measurements was adopted; the n_masked measurements this map lacks stay in the
  // device arrays with weight 0 (chi2 = 0: they sit below every real value in the median's order, contribute nothing anywhere else)
  int n_masked = 0; int near_miss_on = 1;      // MCP_BA_NEAR_MISS=0: a map that lost measurements is built cold
  unsigned long long med_rank() const { return (unsigned long long)n_masked + (unsigned long long)(m_total/2); }      // vErrorSquared[size/2] among the real measurements
  DevBuf<int> d_m_last;                         // observer chain's last link per measurement (k_linearize_pipe)
  bool lin_pipe = env_on("MCP_BA_LIN_PIPE", true);
  DevBuf<int> d_g_order, d_sp_unk;              // launch order of the groups in k_schur4 (heaviest first; only when they outnumber the slots), free-point index per sorted point
  bool sch4_order = false;
  bool asm_long = false;          // the pose pairs' lists of staged blocks are long (a few free poses staged by every group): k_assemble_long
  int grp_inc_max = 0;            // most point-pose incidences of any group (the W area of k_linearize_quad)
  size_t nstage = 0;        // staged 6x6 blocks over all groups
  int nrhs_rows = 0;        // staged rhs rows (6 doubles each) over all groups
  AsmPlan A;
  DevBuf<double> d_V, d_g, d_W, d_Vinv, d_xl, d_xp_good, d_xl_good, d_err;
  DevBuf<double> d_selvals;     // candidates of the single-GPU selection (SEL_GATHER_CAP)
  DevBuf<double> d_seltab;      // multi-rank selection: [world x sel_cap candidates][overflow flag][world counts][gather counter]
  int sel_cap = 4096;           // candidates per rank slot (MCP_BA_SELECT_CAP)
  DevBuf<double> d_xp_cand;     // pose update of the trial in flight; swapped with d_xp_good (as d_xl with d_xl_good) when the solve succeeded
  DevBuf<double> d_part0, d_part1, d_part2, d_res, d_sigma, d_hist, d_cov;
  // the head of an iteration of a large map in one launch (ba_headl.h, MCP_BA_HEAD_LARGE=1): scratch (left zeroed by every launch).  Default off:
  // measured equal to the six launches at the metric size (1611 / 1603 vs 1606 / 1611 it/s) -- 39 us alone on the device against ~50, but its 64
  // fat workgroups wait for room beside the trials evaluated ahead on the other stream, which six small launches slip past
  DevBuf<unsigned char> d_headl; bool headl_clean = false; int head_large_on = 0; int hl_grid = 0;
  bool use_head_large() const { return head_large_on && robust && !multi() && !small_mode() && P.nmeas > 0 && hl_grid > 0 && d_headl.p != nullptr; }
  int head_large(int w, int off, double* sig_copy2);
  bool head_failed = false;
  bool hist_clean = false;          // the single-GPU median left its two histograms + counter zero (k_select_small's last act): no fill in front of the next one
  // The sigma block is double-buffered by median: a trial evaluated ahead on the speculative stream that nobody consumes may
  // still be reading its iteration's block when the next iteration's median writes the new one (the two streams only meet
  // again in linearize()'s join_spec()); the block after that is written behind that join.
  // sigma blocks: 0, 1 = the pair the iteration head alternates between (a fresh block: stragglers of the last iteration keep reading
  // theirs); 2 + parity*MAX_SYS + q = the block of a large map's per-trial head (above)
  static constexpr int N_SIG = 2 + 2*MAX_SYS;
  int sig_idx = 0;
  double* sig_block(int i) { return d_sigma.p + 8*i; }
  double* sig() { return sig_block(sig_idx); }
  void flip_sig() { sig_idx = (sig_idx == 0) ? 1 : 0; }
  DevBuf<SelState> d_selstate;
  DevBuf<int> d_fail;
  static constexpr size_t HRES = 32 + 32*MAX_SYS + 8 + 2*MAX_SYS;      // ... + the status word and ticket of every per-trial head (ba_head.h)
  static constexpr size_t HEAD_MAIL = 32 + 32*MAX_SYS + 8;
  double* h_res = nullptr;  // pinned, device-visible; [32..63] is the mailbox k_final_sums writes (ticket at 32 + MAIL_TICKET)
  unsigned char* h_exp = nullptr; size_t h_exp_cap = 0; bool exported = false;      // pinned: the state [poses | points] and the Tukey flags of a finished solve, written by kernels (final_stats)
  int ensure_export(size_t bytes) {
    if (bytes <= h_exp_cap) return 0;
    if (h_exp) mcp::PinnedCache::get().put(h_exp, h_exp_cap, hipHostMallocMapped | hipHostMallocCoherent, device);
    size_t cap = 65536; while (cap < bytes) cap <<= 1;               // (size classes: the next handle of a similar map finds it in the cache)
    h_exp = (unsigned char*)mcp::PinnedCache::get().take(cap, hipHostMallocMapped | hipHostMallocCoherent); h_exp_cap = h_exp ? cap : 0;
    if (!h_exp) { set_err("hipHostMalloc failed"); return -1; }
    return 0;
  }
  unsigned long long mail_ticket = 0; int use_mailbox = 1; double* h_mail_dev = nullptr;
  int* h_fail = nullptr;    // pinned

  // robust data
  double sigma_sq = 0, sigma_sq_lim = 0;
  // results
  int converged = 0, total_iterations = 0; double lambda = 0, max_cov = DBL_MAX, mean_chi2 = 0;
  double last_chi2_action = DBL_MAX;
  std::vector<int> outliers;
  std::vector<mcp_ba_iter_log> logs;
  mcp_ba_timing timing;

  // multi-rank
  mcp_allreduce_fn hook = nullptr; void* hook_user = nullptr; int rank = 0, world = 1;
  mcp_comm* comm = nullptr;        // native RCCL transport (takes precedence over the hook)
  // MCP_BA_FORCE_MULTI=1: a one-rank communicator / hook is driven through the whole multi-rank machine (packed tiles, per-lane
  // collectives, riding histograms) -- sums over one rank are exact, so the result must equal the plain single-rank solve bit for
  // bit; the GPU suite uses it to run the two-lane RCCL path on the one device a test box has
  int force_multi = 0;
  bool multi() const { return (world > 1 || force_multi) && (hook || comm); }
  // every host-side wait of the solve is bounded (MCP_BA_TIMEOUT_MS, default 60 s): a rank that dies inside a collective leaves the
  // others waiting on a kernel that never ends, which must surface as MCP_ERR_RUNTIME with the place, not as a hang
  double timeout_ms = 60000.0;
  unsigned long long coll_seq[2] = {0, 0}; const char* coll_what[2] = {"none", "none"};
  int watchdog_fail(const char* where);
  int wait_stream(hipStream_t s, const char* what);
  // trial buffers of the multi-rank tail (ba_trial.h) and the selection state k_trial_post leaves in them
  DevBuf<double> d_trial[MAX_SYS]; DevBuf<SelState> d_trstate[MAX_SYS];
  int spec_delay = 0;
  int test_fail_trial = 0, test_trial_no = 0;      // MCP_BA_TEST_FAIL_TRIAL=k: the k-th trial of the handle's life is treated as a failed factorisation
  int sel_ride = 1;                // MCP_BA_SELECT_RIDE=0: the median never uses the histograms that rode on the trial's all-reduce
  int pred_bin = -1;               // first digit of the last median the host has seen (the prediction the trials histogram around)
  int tr_pred_ok[MAX_SYS] = {0, 0, 0, 0}, tr_ovf[MAX_SYS] = {0, 0, 0, 0};
  int sel_src = -1;                // trial buffer whose histograms belong to the current state's chi2 (the accepted trial), or -1

  // profiling
  struct Ev { int stage; hipEvent_t a, b; };
  std::vector<Ev> evs; std::vector<hipEvent_t> ev_pool; size_t ev_used = 0;

  DevProblem P;
  CholPlan plan;                   // block-sparse structure of the reduced system

  void drain() {
    if (st) (void)hipStreamSynchronize(st);
    if (st2) (void)hipStreamSynchronize(st2);
    if (st3) (void)hipStreamSynchronize(st3);
    if (st_h) (void)hipStreamSynchronize(st_h);
    for (int q = 0; q < MAX_SYS; ++q) if (st_tr[q]) (void)hipStreamSynchronize(st_tr[q]);
    (void)hipStreamSynchronize(nullptr);          // (synchronous copies and the debug hooks use the null stream)
  }
  ~mcp_ba() {
    // nothing of this handle may still be running when its mailbox goes back to the host allocator (a trial evaluated ahead writes there)
    drain();
    for (int q = 0; q < MAX_SYS; ++q) if (st_tr[q] && !pooled) (void)hipStreamDestroy(st_tr[q]);
    if (h_res) mcp::PinnedCache::get().put(h_res, HRES*sizeof(double), hipHostMallocMapped | hipHostMallocCoherent, device);
    if (h_exp) mcp::PinnedCache::get().put(h_exp, h_exp_cap, hipHostMallocMapped | hipHostMallocCoherent, device);
    if (h_fail) mcp::PinnedCache::get().put(h_fail, 4*sizeof(int), hipHostMallocDefault, device);
    for (auto e : ev_pool) (void)hipEventDestroy(e);
    for (int q = 0; q <= MAX_SYS; ++q) if (chol_exec[q]) (void)hipGraphExecDestroy(chol_exec[q]);
    for (int q = 0; q <= MAX_SYS; ++q) for (int r = 0; r < MAX_SYS; ++r) if (chain_exec[q][r]) (void)hipGraphExecDestroy(chain_exec[q][r]);
    if (st2) { (void)hipStreamSynchronize(st2); if (!pooled) (void)hipStreamDestroy(st2); }
    if (st3) { (void)hipStreamSynchronize(st3); if (!pooled) (void)hipStreamDestroy(st3); }
    if (st_h && st_h_own) (void)hipStreamDestroy(st_h);
    for (int q = 0; q < MAX_SYS; ++q) if (head_ev[q]) (void)hipEventDestroy(head_ev[q]);
    if (ev_fork) (void)hipEventDestroy(ev_fork);
    if (ev_head) (void)hipEventDestroy(ev_head);
    if (ev_sum) (void)hipEventDestroy(ev_sum);
    if (ev_spec) (void)hipEventDestroy(ev_spec);
    if (ev_spec3) (void)hipEventDestroy(ev_spec3);
    for (int q = 0; q < MAX_SYS; ++q) if (ev_tr[q]) (void)hipEventDestroy(ev_tr[q]);
    for (int q = 0; q < MAX_SYS; ++q) if (ev_wf[q]) (void)hipEventDestroy(ev_wf[q]);
    if (st && !pooled) (void)hipStreamDestroy(st);
  }
  // everything the second stream still has in flight reads the current linearisation (W, V, g, staged blocks): the main stream
  // must not overwrite any of it, nor consume a speculative solution, before that work is done
  // q < 0: everything; otherwise only the stream that produces system q of the batch
  // The step of a trial in one launch (ba_small.h k_trial_apply): chain workgroups of that launch, 0 = the separate kernels (a map with
  // more poses than a chain workgroup keeps in LDS; MCP_BA_TRIAL_FUSE=0).  Small bundles: one workgroup for all chains, as before.
  int trial_fuse = 1;
  int dissect_on = 1;                // MCP_BA_CHOL_CHAINS=1: one chain (the poses in add order), see prepare()
  int chain_arcs = 6;                // MCP_BA_CHOL_ARCS=k: the cut has k arcs at most (2 ... 6)
  std::vector<int> chol_segs;        // first tile of every chain of the factorisation plan (empty: one)
  int trial_chain_blocks() const {
    if (P.npose > TA_MAX_POSES) return 0;
    if (small_mode()) return 1;
    return trial_fuse ? std::max(1, (P.nchain + TA_CHAINS - 1)/TA_CHAINS) : 0;
  }
  int join_spec(int q = -1) {
    if (spec_pending && (q < 0 || (q >= spec2_from && q < spec3_from))) { HIPCK(hipStreamWaitEvent(st, ev_spec, 0)); spec_pending = false; }
    if (spec3_pending && (q < 0 || q >= spec3_from)) { HIPCK(hipStreamWaitEvent(st, ev_spec3, 0)); spec3_pending = false; }
    // a full join also covers the trials evaluated ahead on those streams (enqueued after ev_spec was recorded): whatever the main
    // stream does next -- the next linearisation overwrites W, g, V; the next iteration's trials reuse the candidate states -- is
    // ordered behind them, used or not
    if (q < 0) for (int k = 0; k < MAX_SYS; ++k) if (ahead_enq[k]) { HIPCK(hipStreamWaitEvent(st, ev_tr[k], 0)); ahead_enq[k] = false; }
    return 0;
  }

  // What linearize() needs of that join: nobody may still READ the linearisation it is about to overwrite (W, V, g, staged blocks).  The
  // speculative chains are done reading it once their systems are built (ev_spec covers their whole chain); of a trial evaluated ahead
  // only the pose update and the point back-substitution read it -- its chain transforms, residuals and final sums (45 of its 70 us)
  // may run beside the linearisation.  The trial's remaining kernels stay registered (ahead_enq): the fresh solve of the iteration
  // joins them before anything reuses the candidate states (solve_trial).
  int join_spec_lin() {
    if (spec_pending) { HIPCK(hipStreamWaitEvent(st, ev_spec, 0)); spec_pending = false; }
    if (spec3_pending) { HIPCK(hipStreamWaitEvent(st, ev_spec3, 0)); spec3_pending = false; }
    for (int k = 0; k < MAX_SYS; ++k) if (ahead_enq[k]) HIPCK(hipStreamWaitEvent(st, ev_wf[k], 0));
    return 0;
  }

  double* Ubig() { return nbig ? d_ubig.p : nullptr; }
  double* bp() { return rhs() + np; }            // J^T r of the current linearisation (written by k_assemble behind every system's rhs)
  // multi-lambda batch (ba_kernels.h SysBatch): systems 1.. are speculative solves for the next lambdas of the LM
  // schedule.  S()/rhs()/Vinv() address the system the latest trial used.
  size_t red_stride = 0, vinv_stride = 0, pack_stride = 0;
  bool start_rides = false;        // the iteration-start chi2 still has to be summed over the ranks
  int sys_cur = 0; bool spec_ok = false; int batch_n = 0; double batch_lambda[MAX_SYS] = {0, 0, 0, 0};
  int speculate = 3;                 // speculative systems per solve at most; MCP_BA_SPECULATE=0 turns them off
  // Small bundles (ba_small.h) speculate only while trials are being rejected: a BundleAdjustRecent window accepts every first trial, and the
  // three systems solved beside it were 16 launches per iteration for nobody (compute 1.09 -> 1.01 ms without them).  A rejection arms the
  // speculation for the re-solve that follows and the next iterations; MCP_BA_SPECULATE_ADAPT=0: always, as large bundles do.
  int spec_adapt = 1, spec_hot = 0;
  int spec_now() const { return (spec_adapt && small_mode() && spec_hot <= 0) ? 0 : speculate; }
  int lin_join_full = 0;             // MCP_BA_LIN_JOIN=1: linearize() waits for everything the speculative stream has in flight (round 2's behaviour)
  int use_graph = 0;                 // MCP_BA_GRAPH=1: replay the factorisation chain from a captured hipGraph
  hipGraphExec_t chol_exec[MAX_SYS + 1] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  hipGraphExec_t chain_exec[MAX_SYS + 1][MAX_SYS] = {};      // [systems][first system]: factorisation + back-substitution of a sub-batch, captured once
  double* S() { return d_red.p + sys_cur*red_stride; }
  double* rhs() { return S() + (size_t)np*np; }
  double* Vinv() { return d_Vinv.p + sys_cur*vinv_stride; }

  hipEvent_t get_event() {
    if (ev_used == ev_pool.size()) { hipEvent_t e; (void)hipEventCreate(&e); ev_pool.push_back(e); }
    return ev_pool[ev_used++];
  }
  void tic(int stage) { if (!prm.profile) return; Ev e{stage, get_event(), get_event()}; (void)hipEventRecord(e.a, st); evs.push_back(e); }
  void toc() { if (!prm.profile) return; (void)hipEventRecord(evs.back().b, st); }

  int new_id(int kind, int index) {
    int id = next_id++;
    if ((int)id_kind.size() <= id) { id_kind.resize(id + 1024, 0); id_index.resize(id + 1024, 0); }
    id_kind[id] = kind; id_index[id] = index; return id;
  }
  // last chain looked up: the adapters add measurements KeyFrame by KeyFrame (BundleAdjusterMulti.cc:168-200), so consecutive
  // calls repeat the same pose chain and the ordered-map search is skipped for them
  int last_chain_n = 0, last_chain_idx = -1; int last_chain_ids[MCP_MAX_CHAIN];
  // ... and a direct-mapped cache in front of the ordered map for adapters that add point by point (every row another chain)
  struct ChainSlot { int n = 0, idx = -1; int ids[MCP_MAX_CHAIN]; };
  static constexpr int CHAIN_CACHE = 4096;
  std::vector<ChainSlot> chain_cache;
  static unsigned chain_hash(const int* ids, int n) { unsigned h = 2166136261u ^ (unsigned)n; for (int i = 0; i < n; ++i) { h ^= (unsigned)ids[i]; h *= 16777619u; } return (h ^ (h >> 13)) & (CHAIN_CACHE - 1); }
  // read-only look-up (safe from several threads while nobody adds a chain): index, -1 = not known yet, -2 = malformed
  int lookup_chain(const int* ids, int n) const {
    if (n < 1 || n > MCP_MAX_CHAIN) return -2;
    if (!chain_cache.empty()) {
      const ChainSlot& cs = chain_cache[chain_hash(ids, n)];
      if (cs.idx >= 0 && cs.n == n && std::memcmp(ids, cs.ids, sizeof(int)*n) == 0) return cs.idx;
    }
    std::array<int, 1 + MCP_MAX_CHAIN> key; key.fill(-1); key[0] = n;
    for (int i = 0; i < n; ++i) {
      if (ids[i] <= 0 || ids[i] >= next_id || id_kind[ids[i]] != 1) return -2;
      key[1 + i] = id_index[ids[i]];
    }
    auto it = chain_map.find(key);
    return it == chain_map.end() ? -1 : it->second;
  }
  int find_chain(const int* ids, int n) {
    if (n < 1 || n > MCP_MAX_CHAIN) return -1;
    if (n == last_chain_n && last_chain_idx >= 0 && std::memcmp(ids, last_chain_ids, sizeof(int)*n) == 0) return last_chain_idx;
    if (chain_cache.empty()) chain_cache.resize(CHAIN_CACHE);
    ChainSlot& cs = chain_cache[chain_hash(ids, n)];
    if (cs.idx >= 0 && cs.n == n && std::memcmp(ids, cs.ids, sizeof(int)*n) == 0) {
      last_chain_n = n; last_chain_idx = cs.idx; std::memcpy(last_chain_ids, ids, sizeof(int)*n);
      return cs.idx;
    }
    std::array<int, 1 + MCP_MAX_CHAIN> key; key.fill(-1); key[0] = n;
    for (int i = 0; i < n; ++i) {
      if (ids[i] <= 0 || ids[i] >= next_id || id_kind[ids[i]] != 1) return -1;
      key[1 + i] = id_index[ids[i]];
    }
    int idx;
    auto it = chain_map.find(key);
    if (it != chain_map.end()) idx = it->second;
    else {
      HChain c; c.len = n; for (int i = 0; i < MCP_MAX_CHAIN; ++i) c.v[i] = (i < n) ? key[1 + i] : 0;
      chains.push_back(c);
      idx = (int)chains.size() - 1; chain_map[key] = idx;
    }
    last_chain_n = n; last_chain_idx = idx; std::memcpy(last_chain_ids, ids, sizeof(int)*n);
    cs.n = n; cs.idx = idx; std::memcpy(cs.ids, ids, sizeof(int)*n);
    return idx;
  }
  // PoseChainHelper::MoveTogether, ChainBundle.cc:157-199 (structural: evaluated once per chain pair)
  bool move_together(const HChain& a, const HChain& b, int depth) const {
    int furthest = -1;
    for (;;) {
      int t = furthest + 1;
      if (a.len <= t || b.len <= t) break;
      if (a.v[t] != b.v[t]) break;
      furthest = t;
      if (furthest == depth) return true;
    }
    if (furthest == -1) return false;
    for (int i = furthest; i <= depth; ++i) if (!poses[a.v[i]].fixed) return false;
    return true;
  }

  // SUM all-reduce of `count` doubles at `buf` over the ranks, on lane 0 (main stream) or 1 (speculative stream).
  // host_sync: the caller reads the result with a blocking copy next, so the stream-ordered RCCL path has to drain the stream first.
  int allreduce(double* buf, size_t count, int lane = 0, bool host_sync = false, const char* what = "all-reduce") {
    if (!multi()) return 0;
    hipStream_t s = lane ? st2 : st;
    ++coll_seq[lane]; coll_what[lane] = what;
    if (lane) { timing.n_collectives_spec++; timing.collective_bytes_spec += 8.0*(double)count; }
    else { timing.n_collectives_main++; timing.collective_bytes_main += 8.0*(double)count; }
    if (comm) {
      if (comm->dead) { set_err("communicator was aborted by an earlier time-out"); return -1; }
      const int rc = rccl().AllReduce(buf, buf, count, RCCL_FLOAT64, RCCL_SUM, comm->lane(lane), s);
      if (rc != 0) { set_err(std::string("ncclAllReduce failed: ") + (rccl().GetErrorString ? rccl().GetErrorString(rc) : "?")); return -1; }
      if (host_sync) return wait_stream(s, what);
      return 0;
    }
    if (wait_stream(s, what)) return -1;
    if (hook(hook_user, buf, count, (void*)s) != 0) { set_err(std::string("all-reduce hook failed (") + what + ")"); return -1; }
    return 0;
  }

  struct HostStruct {      // (in the pinned staging arena)
    avec<int> m_pt, m_chain, m_sp, slot_start, slot_unk, slot_inc, l_i0, l_i1, l_sp, inc_unk, sp_pt, sp_m, sp_i, g_sp0, g_pose, g_blk0,
              pair_id, pr_start, blk_dst, po_start, rhs_dst;
    avec<unsigned char> m_cam, slot_first, sp_big, slot_lp, inc_lp, inc_mixed, blk_pair;
    avec<unsigned short> m_mask;
    avec<double> m_u, m_v, m_om;
  };
  int prepare();
  int prepare_legacy();
  int finish_prepare(HostStruct& H, std::chrono::steady_clock::time_point t0, std::chrono::steady_clock::time_point tlast, bool trace, const StructEntry* hit = nullptr);
  int upload_state();
  int download_state();
  void launch_chains(int which);
  void launch_eval(int which, bool sum, double* err_out);
  int select_kth(const double* x, int n, unsigned long long k, double* out_dev, bool huber_sigma = false);
  int select_gather_finish(const double* x, int n, const double* hist, SelState* state, double* out_dev, bool check_overflow);
  int median_sigma(int which);
  int read_results(int count);
  int wait_mail(int q, unsigned long long ticket, int count);
  int enqueue_spec_trial(hipStream_t s, int q);
  int multi_trial_tail(hipStream_t s, int lane, int q, int slot, int nbe, const double* p0, int nbb, const double* p1, const double* p2,
                       const double* pose_parts, int pose_off, bool main_block, double* mail, int mail_count, unsigned long long ticket);
  int cancel_spec_trials();
  int run_ahead(int q);
  unsigned long long mail_ticket0 = 0;
  int dbg_pre = 0; double dbg_wait_us[2] = {0, 0};
  DevBuf<double>* last_xp = nullptr; DevBuf<double>* last_xl = nullptr;     // where the last trial left the solver's x (pose part, point part)
  // MCP_BA_EVT=1: device time stamps (timing events) of the phases of every iteration on both streams, printed to stderr
  int evt_debug = 0; std::vector<std::pair<const char*, hipEvent_t>> evt_log;
  void mark(const char* what, hipStream_t s) { if (!evt_debug) return; hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return; (void)hipEventRecord(e, s); evt_log.push_back({what, e}); }
  void evt_flush() {
    if (!evt_debug || evt_log.empty()) return;
    (void)hipDeviceSynchronize();
    size_t base = 0;
    for (size_t i = 0; i < evt_log.size(); ++i) {
      if (!std::strcmp(evt_log[i].first, "iter")) {
        float gap = 0; if (i) (void)hipEventElapsedTime(&gap, evt_log[i - 1].second, evt_log[i].second);       // last mark of the previous iteration -> this one's start
        base = i; fprintf(stderr, "\n[evt] (+%.0f)", gap*1e3);
      }
      float ms = 0; (void)hipEventElapsedTime(&ms, evt_log[base].second, evt_log[i].second); fprintf(stderr, " %s %.0f", evt_log[i].first, ms*1e3);
    }
    fprintf(stderr, "\n");
    for (auto& pe : evt_log) (void)hipEventDestroy(pe.second);
    evt_log.clear();
  }
  int solve_chain(hipStream_t s, int n, int q0);
  int linearize();
  int build_system(int nsys, SysBatch& sb, int q0 = 0, hipStream_t on = nullptr);
  int solve_trial(double lam, bool& ok2, double ni = 0);
  int persist_fallback(double lam, bool& ok2, double ni);
  int compute(volatile unsigned char* abort_flag, int n_iter, double user_lambda);
  int final_stats(int nCounter);
};

// ------------------------------------------------------------------------------------------
// a wait ran into the time-out: say where, and take the communicator down so that the collective kernel this rank is stuck in
// (its peers never arrived) leaves the device
int mcp_ba::watchdog_fail(const char* where) {
  char msg[384];
  std::snprintf(msg, sizeof msg, "rank %d of %d: no progress within %.0f ms while waiting for %s (last collectives: main lane #%llu '%s', "
                "speculative lane #%llu '%s')", rank, world, timeout_ms, where, coll_seq[0], coll_what[0], coll_seq[1], coll_what[1]);
  set_err(msg);
  if (comm && !comm->dead) {
    comm->dead = true;
    if (rccl().CommAbort) { if (comm->comm2 && comm->comm2 != comm->comm) (void)rccl().CommAbort(comm->comm2); (void)rccl().CommAbort(comm->comm); comm->comm = comm->comm2 = nullptr; }
  }
  return -1;
}
// hipStreamSynchronize with a deadline
int mcp_ba::wait_stream(hipStream_t s, const char* what) {
  if (!multi()) { HIPCK(hipStreamSynchronize(s)); return 0; }
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned spins = 0;; ++spins) {
    const hipError_t e = hipStreamQuery(s);
    if (e == hipSuccess) return 0;
    if (e != hipErrorNotReady) { set_err(std::string("stream failed while waiting for ") + what + ": " + hipGetErrorString(e)); return -1; }
    if (spins > 2000) {
      std::this_thread::yield();
      if ((spins & 0xff) == 0 && std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() > timeout_ms) return watchdog_fail(what);
    }
  }
}

// Large host scratch of Prepare() (the measurements in sorted order, the per-thread counters): one grow-only block per process, kept
// between calls -- a fresh 16 MB allocation per call comes back from the allocator as untouched pages, and sixteen threads faulting
// them in cost the phase that fills them anything between 1 and 6 ms.  Held (mutex) for the duration of a Prepare().
namespace {
struct PrepScratch {
  std::mutex m; std::unique_ptr<char[]> p; size_t cap = 0, used = 0;
  void reset(size_t bytes) { if (bytes > cap) { cap = bytes + bytes/4 + 4096; p.reset(new char[cap]); } used = 0; }
  template <class T> T* take(size_t count) { used = (used + 63) & ~(size_t)63; T* r = reinterpret_cast<T*>(p.get() + used); used += count*sizeof(T); return r; }
};
PrepScratch& prep_scratch() { static PrepScratch s; return s; }
}
// Structure build of one Compute() -- the analogue of g2o's initializeOptimization + buildStructure -- on the host's cores:
// which poses / points take part, points ordered by source pose, measurements bucketed by point, per-measurement Jacobian
// slots and per-point incidences, groups of points for the LDS-tiled kernels, the fixed-order staging plan of their pose
// blocks, the tile pattern of the reduced system.  Every phase that touches the measurements runs on the worker pool over
// disjoint ranges and writes to positions that are known beforehand, so the arrays are those of the serial reference
// implementation (prepare_legacy, MCP_BA_PREPARE_LEGACY=1; tests/test_ba_gpu.py compares the two) whatever the thread count.
// Secondary point order (round 5).  Points are sorted by the pose they are expressed in; inside one such pose, points that are seen
// from the same poses now lie next to each other: key = (largest observer pose, sum of the observer poses over the point's
// measurements), ties by point index.  A group of 64 consecutive points -- and a 16-point chunk of k_schur4 -- then touches fewer
// poses: at the metric size 10 % fewer staged blocks and 23 % fewer non-empty 16 x 16 tile pairs in the Schur products.
// `order` holds the points bucketed by `pkey` (ascending); buckets [b0, b1) of it are sorted here.
static void refine_point_order(std::vector<int>& order, const std::vector<int>& pkey, const int* smax, const int* ssum, size_t i0, size_t i1) {
  size_t a = i0;
  while (a < i1) {
    size_t b = a + 1;
    while (b < i1 && pkey[order[b]] == pkey[order[a]]) ++b;
    std::sort(order.begin() + a, order.begin() + b, [&](int x, int y) {
      if (smax[x] != smax[y]) return smax[x] < smax[y];
      if (ssum[x] != ssum[y]) return ssum[x] < ssum[y];
      return x < y; });
    a = b;
  }
}
// first movable (not fixed) pose of a chain -- the pose a measurement's observer hangs off -- as an index into the pose array, or -1.
// (The pose array index, not the unknown number: it is known before the activity pass and orders the poses the same way.)
static int chain_first_movable(const HChain& c, const std::vector<HPose>& poses) {
  for (int k = 0; k < c.len; ++k) if (!poses[c.v[k]].fixed) return c.v[k];
  return -1;
}

int mcp_ba::prepare() {
  HIPCK(hipSetDevice(device));
  {
    // k_head_small keeps its candidates in 112 KB of dynamic LDS: a device (or partition) that does not grant that runs small bundles through
    // the general launches instead (the refusal used to fail every robust small-bundle solve).  Per device: asked once.
    static std::mutex hs_mu; static std::map<int, bool> hs_ok;
    std::lock_guard<std::mutex> lk(hs_mu);
    auto it = hs_ok.find(device);
    if (it == hs_ok.end()) {
      const bool ok = hipFuncSetAttribute((const void*)k_head_small, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(HS_STASH*sizeof(double))) == hipSuccess;
      (void)hipGetLastError();
      it = hs_ok.emplace(device, ok).first;
    }
    if (!it->second) small_on = 0;
  }
  if (getenv("MCP_BA_PREPARE_LEGACY")) return prepare_legacy();
  auto t0 = std::chrono::steady_clock::now();
  auto tlast = t0; const bool trace = getenv("MCP_BA_TRACE") != nullptr;
  auto lap = [&](const char* what) { if (!trace) return; auto n = std::chrono::steady_clock::now(); fprintf(stderr, "[mcp_ba prepare] %-22s %.3f ms\n", what, std::chrono::duration<double, std::milli>(n - tlast).count()); tlast = n; };
  HIPCK(hipSetDevice(device));
  const int npose = (int)poses.size(), npoint = (int)points.size(), nmeas = (int)meas.size();
  const size_t nch = chains.size();
  HostPool& pool = host_pool();
  static const int par_min = [] { const char* e = getenv("MCP_BA_PREP_PAR_MIN"); return e ? atoi(e) : 32768; }();      // measurements from which the structure is built on the worker pool
  const int T = (nmeas >= par_min) ? pool.size() : 1;
  auto par = [&](const std::function<void(int)>& fn) { if (T == 1) fn(0); else pool.run(fn); };
  auto lo_of = [&](int tid, long n) { return (long)(n*tid/T); };
  lap("  entry");
  // ---- the same topology as an earlier call's (structure cache, above)?  128-bit hash over what the structure is built from
  cache_insert = false;
  { const char* e = getenv("MCP_BA_CHOL_CHAINS"); dissect_on = !(e && atoi(e) == 1); }
  { const char* e = getenv("MCP_BA_CHOL_ARCS"); chain_arcs = e ? atoi(e) : 6; }
  if (StructCache::get().enabled() && !multi() && nmeas > 0) {
    constexpr size_t HB = 8192;
    const size_t nbm = ((size_t)nmeas + HB - 1)/HB;
    std::vector<unsigned long long> ha(3*nbm + 4), hb(3*nbm + 4);
    par([&](int tid) {
      for (size_t b = (size_t)lo_of(tid, (long)nbm), e = (size_t)lo_of(tid + 1, (long)nbm); b < e; ++b) {
        const size_t i0 = b*HB, n = std::min(HB, (size_t)nmeas - i0);
        hash_block(meas_chain.data() + i0, n, 1, 1, ha[3*b], hb[3*b]);
        hash_block(meas_point.data() + i0, n, 1, 2, ha[3*b + 1], hb[3*b + 1]);
        hash_block(&meas[i0].cam, n, sizeof(HMeas)/sizeof(int), 3, ha[3*b + 2], hb[3*b + 2]);
      }
    });
    static_assert(sizeof(HMeas) % sizeof(int) == 0 && sizeof(HPoint) % sizeof(int) == 0 && sizeof(HPose) % sizeof(int) == 0 && sizeof(HChain) == (1 + MCP_MAX_CHAIN)*sizeof(int), "strided hashing");
    if (npose) hash_block(&poses[0].fixed, npose, sizeof(HPose)/sizeof(int), 4, ha[3*nbm], hb[3*nbm]);
    if (nch) hash_block(&chains[0].len, nch*(1 + MCP_MAX_CHAIN), 1, 5, ha[3*nbm + 1], hb[3*nbm + 1]);
    if (npoint) { hash_block(&points[0].fixed, npoint, sizeof(HPoint)/sizeof(int), 6, ha[3*nbm + 2], hb[3*nbm + 2]); hash_block(&points[0].chain, npoint, sizeof(HPoint)/sizeof(int), 7, ha[3*nbm + 3], hb[3*nbm + 3]); }
    StructKey key;
    for (size_t i = 0; i < ha.size(); ++i) { key.h0 = mix64(key.h0 ^ ha[i]) + i; key.h1 = mix64(key.h1 + hb[i]) ^ (i*0x9e3779b97f4a7c15ull); }
    key.npose = npose; key.npoint = npoint; key.nmeas = nmeas; key.nchain = (int)nch; key.dev = device;
    key.h1 = mix64(key.h1 ^ (unsigned long long)cams.size());          // (the camera table is part of the packed block's layout)
    // (what else decides the structure: the grouping policy and its run-time switches)
    const char* e_al = getenv("MCP_BA_ASM_LONG");
    key.flags = grp_lmax_pol | (point_order_refine ? 32 : 0) | (sch4_on ? 64 : 0) | (env_on("MCP_BA_SCHUR4_ORDER", true) ? 128 : 0) | (group_points(npoint) << 8) |
                ((e_al ? 1 + (atoi(e_al) != 0) : 0) << 16) | (dissect_on ? (chain_arcs & 7) << 18 : 0);
    { const char* e = getenv("MCP_BA_TEST_CHOL_CUT"); if (e) key.flags ^= (unsigned)(atoi(e) & 0xff) << 21; }
    cache_key = key;
    {
      // the same key without the measurements (near miss, below)
      StructKey b = key; b.h0 = b.h1 = 0; b.nmeas = 0;
      for (size_t i = 3*nbm, r = 0; i < ha.size(); ++i, ++r) { b.h0 = mix64(b.h0 ^ ha[i]) + r; b.h1 = mix64(b.h1 + hb[i]) ^ (r*0x9e3779b97f4a7c15ull); }      // (r, not i: the number of measurement blocks must not enter)
      b.h1 = mix64(b.h1 ^ (unsigned long long)cams.size());
      cache_base = b;
    }
    n_masked = 0;
    std::shared_ptr<StructEntry> hit = StructCache::get().find(key);
    lap("  topology hash");
    std::vector<int> present;          // near miss: cached add index -> this map's add index, -1 = this map does not have it
    if (!hit && near_miss_on) {
      // NEAR MISS (round 6): MCPTAM erases the measurements an adjustment flagged as outliers and adjusts again
      // (/root/reference/src/MapMaker.cc:225-230, 283-287 -> MapMakerServerBase::HandleOutliers, src/MapMakerServerBase.cc:1198-1238:
      // kf.EraseMeasurementOfPoint) -- the next ChainBundle holds the poses, points and chains of the call before and its measurements
      // MINUS a few, in the same order.  If the cache holds such a superset and no pose or point loses its last measurement (activity,
      // hence the unknowns, unchanged), its structure is adopted as on a hit; the erased measurements keep their place in the device
      // arrays with weight 0.  Same mathematics as a cold Prepare() of the smaller map; the order of some floating-point sums differs
      // (the cold build would group the points by their NEW observer sets), so the two agree to rounding, not bit for bit.
      for (auto& c : StructCache::get().find_supersets(cache_base, nmeas)) {
        const int nc2 = c->key.nmeas;
        present.assign(nc2, -1);
        const int* ep = c->add_point.data(); const int* ec = c->add_chain.data(); const int* em = c->add_cam.data();
        const int* mp = meas_point.data(); const int* mc = meas_chain.data();
        int j = 0;
        for (int i = 0; i < nc2 && j < nmeas; ++i) if (ep[i] == mp[j] && ec[i] == mc[j] && em[i] == meas[j].cam) present[i] = j++;
        if (j != nmeas) continue;
        // activity: every point / pose that was active must still be (and none can have become active)
        std::vector<unsigned char> pa(npoint, 0), cu(nch, 0), qa(npose, 0);
        for (int i = 0; i < nmeas; ++i) { pa[mp[i]] = 1; cu[mc[i]] = 1; }
        for (int i = 0; i < npoint; ++i) if (pa[i]) cu[points[i].chain] = 1;
        for (size_t ch = 0; ch < nch; ++ch) if (cu[ch]) for (int k2 = 0; k2 < chains[ch].len; ++k2) qa[chains[ch].v[k2]] = 1;
        bool same = true;
        for (int i = 0; i < npoint && same; ++i) same = pa[i] == c->pt_active[i];
        for (int i = 0; i < npose && same; ++i) same = qa[i] == c->pose_active[i];
        if (!same) continue;
        hit = c; n_masked = nc2 - nmeas;
        StructCache::get().touch(c);
        break;
      }
      lap("  near miss search");
    }
    if (hit) {
      // adopt: the host-side results of the structure build, then the device block (finish_prepare)
      for (int i = 0; i < npose; ++i) { poses[i].unk = hit->pose_unk[i]; poses[i].active = hit->pose_active[i]; }
      for (int i = 0; i < npoint; ++i) { points[i].unk = hit->pt_unk[i]; points[i].active = hit->pt_active[i]; }
      fp_pose = hit->fp_pose; fl_point = hit->fl_point; perm = hit->perm;
      if (n_masked) for (auto& v : perm) v = present[v];          // sorted position -> THIS map's add index, -1 = erased here
      nfp = hit->nfp; nfl = hit->nfl; np = hit->np; nx = hit->nx; nsp = hit->nsp; ninc = hit->ninc; nslot = hit->nslot; ngroup = hit->ngroup; nbig = hit->nbig;
      grp_pts = hit->grp_pts; grp_blk_max = hit->grp_blk_max; grp_inc_max = hit->grp_inc_max; nrhs_rows = hit->nrhs_rows; nstage = hit->nstage;
      m_total = n_masked ? (double)nmeas : hit->m_total; nfl_total = hit->nfl_total;
      if (np > CH_SOLVE_MAX) { set_err("too many free poses for the dense reduced solve (6P > 6144)"); return -1; }
      std::unique_lock<std::mutex> arena_lock(pinned_arena().mutex());
      ArenaGuard arena_guard{st};
      HostStruct H;
      last_pat = hit->pat; chol_segs = hit->chol_segs;
      plan.persist_segs = chol_segs;
      if (np > 0) { if (plan.build(np, last_pat)) { set_err("Cholesky plan allocation failed"); return -1; } }
      else plan.all_tiles.clear();
      lap(n_masked ? "  adopted (near miss)" : "  adopted (cache hit)");
      return finish_prepare(H, t0, tlast, trace, hit.get());
    }
    cache_insert = true;
  }
  // ---- one pass over the measurements in add order, a range per thread: which chains are used, and how many measurements of
  // every point the range holds (private counters: no shared writes)
  std::vector<unsigned char> chain_used(nch, 0);
  struct SMeas { int chain, cam, mi, pad_; double u, v, omega; };
  PrepScratch& scratch = prep_scratch();
  std::unique_lock<std::mutex> scratch_lock(scratch.m);
  // (with the secondary point order: per point also the largest observer pose and the sum of the observer poses -- the key of
  //  refine_point_order -- in the same private way: a first version added them up with atomics on shared arrays, 6 ms of cache-line
  //  ping-pong between the threads)
  const bool refine = point_order_refine;
  // (for the chains of the factorisation, below: per point the poses that see it, as a 64-bit set per thread -- the adapters add the
  //  measurements KeyFrame by KeyFrame, so a thread's range of the add order holds few distinct observers: each gets a bit as it is first
  //  met, whatever its pose index; a range with more than 64 of them is reported and the sets are taken again with atomics)
  const bool want_graph = dissect_on && npose >= 96 && npose <= 1024 + 64;
  typedef unsigned long long u64;
  scratch.reset(sizeof(int)*(size_t)(refine ? 3 : 1)*T*std::max(npoint, 1) + sizeof(SMeas)*(size_t)std::max(nmeas, 1) + (want_graph ? sizeof(u64)*(size_t)T*std::max(npoint, 1) : 0) + 512);
  u64* const seen_t = want_graph ? scratch.take<u64>((size_t)T*std::max(npoint, 1)) : nullptr;
  std::vector<int> seen_ids((size_t)T*64, -1), cobs; std::vector<unsigned char> seen_over(T, 0);
  if (want_graph) { cobs.resize(nch); for (size_t c = 0; c < nch; ++c) { int one = -1, nmov = 0; for (int k = 0; k < chains[c].len; ++k) if (!poses[chains[c].v[k]].fixed) { one = chains[c].v[k]; ++nmov; } cobs[c] = nmov == 1 ? one : nmov ? -2 : -1; } }
  int* const cnt_t = scratch.take<int>((size_t)T*std::max(npoint, 1));
  int* const sig_t = refine ? scratch.take<int>((size_t)2*T*std::max(npoint, 1)) : nullptr;      // [thread][max | sum][point]
  SMeas* const sorted = scratch.take<SMeas>(std::max(nmeas, 1));
  std::vector<int> ckey;
  if (refine) { ckey.resize(nch); for (size_t c = 0; c < nch; ++c) ckey[c] = chain_first_movable(chains[c], poses) + 1; }
  {
    std::vector<unsigned char> cu_all((size_t)T*nch, 0);
    par([&](int tid) {
      unsigned char* cu = cu_all.data() + (size_t)tid*nch; int* ct = cnt_t + (size_t)tid*npoint;
      std::memset(ct, 0, sizeof(int)*(size_t)npoint);
      const int* mc = meas_chain.data(); const int* mp = meas_point.data();
      if (want_graph) {
        u64* sn = seen_t + (size_t)tid*npoint;
        std::memset(sn, 0, sizeof(u64)*(size_t)npoint);
        const int* co = cobs.data(); int* ids = &seen_ids[(size_t)tid*64]; int nids = 0, last_ob = -1, last_bit = 0; bool over = false;
        for (long i = lo_of(tid, nmeas), e = lo_of(tid + 1, nmeas); i < e; ++i) {
          const int ob = co[mc[i]];
          if (ob < 0) { over |= ob == -2; continue; }
          if (ob != last_ob) {
            int b = 0; while (b < nids && ids[b] != ob) ++b;
            if (b == nids) { if (nids == 64) { over = true; continue; } ids[nids++] = ob; }
            last_ob = ob; last_bit = b;
          }
          sn[mp[i]] |= 1ull << last_bit;
        }
        seen_over[tid] = over;
      }
      if (!refine) { for (long i = lo_of(tid, nmeas), e = lo_of(tid + 1, nmeas); i < e; ++i) { cu[mc[i]] = 1; ct[mp[i]]++; } return; }
      int* mx = sig_t + (size_t)2*tid*npoint; int* sm = mx + npoint;
      std::memset(mx, 0, sizeof(int)*(size_t)2*npoint);
      const int* ck = ckey.data();
      for (long i = lo_of(tid, nmeas), e = lo_of(tid + 1, nmeas); i < e; ++i) {
        const int c = mc[i], pt = mp[i], k = ck[c];
        cu[c] = 1; ct[pt]++;
        sm[pt] += k; if (k > mx[pt]) mx[pt] = k;
      }
    });
    for (int t = 0; t < T; ++t) for (size_t c = 0; c < nch; ++c) chain_used[c] |= cu_all[(size_t)t*nch + c];
  }
  // per point: total, and the rank at which every thread's measurements of it start (add order = thread order, then index order)
  std::vector<int> cnt(npoint + 1, 0), smax, ssum;
  if (refine) { smax.resize(std::max(npoint, 1)); ssum.resize(std::max(npoint, 1)); }
  par([&](int tid) {
    for (long p = lo_of(tid, npoint), e = lo_of(tid + 1, npoint); p < e; ++p) {
      int tot = 0;
      for (int t = 0; t < T; ++t) { int& c = cnt_t[(size_t)t*npoint + p]; const int v = c; c = tot; tot += v; }
      cnt[p + 1] = tot;
      if (refine) {
        int m = 0, sum = 0;
        for (int t = 0; t < T; ++t) { m = std::max(m, sig_t[(size_t)2*t*npoint + p]); sum += sig_t[(size_t)(2*t + 1)*npoint + p]; }
        smax[p] = m; ssum[p] = sum;
      }
    }
  });
  lap("  counts (threads)");
  for (auto& p : poses) { p.active = 0; p.unk = -1; }
  for (int i = 0; i < npoint; ++i) { points[i].active = cnt[i + 1] > 0; points[i].unk = -1; if (points[i].active) chain_used[points[i].chain] = 1; }
  for (size_t c = 0; c < nch; ++c) if (chain_used[c]) for (int k = 0; k < chains[c].len; ++k) poses[chains[c].v[k]].active = 1;
  m_total = (double)nmeas;
  if (multi()) {
    // ranks hold different measurement shards: agree on the active poses and the global count
    std::vector<double> flags(npose + 2);
    for (int i = 0; i < npose; ++i) flags[i] = poses[i].active;
    flags[npose] = (double)nmeas;
    double nfree = 0; for (const auto& p : points) if (p.active && !p.fixed) nfree += 1;
    flags[npose + 1] = nfree;
    DevBuf<double> tmp;
    if (tmp.upload(flags, st)) return -1;
    if (allreduce(tmp.p, flags.size(), 0, true, "set-up: active poses and counts")) return -1;
    HIPCK(hipMemcpy(flags.data(), tmp.p, flags.size()*sizeof(double), hipMemcpyDeviceToHost));
    for (int i = 0; i < npose; ++i) poses[i].active = flags[i] > 0;
    m_total = flags[npose]; nfl_total = flags[npose + 1];
  }
  fp_pose.clear(); fl_point.clear();
  for (int i = 0; i < npose; ++i) if (poses[i].active && !poses[i].fixed) { poses[i].unk = (int)fp_pose.size(); fp_pose.push_back(i); }
  chol_segs.clear();
  if (want_graph && (int)fp_pose.size() >= 96 && (int)fp_pose.size() <= 1024) {
    // ---- several chains instead of one (round 6; ba_cut.h has the search).  Here: the coupling graph -- per point the
    // set of free poses that see or carry it (taken in the counting pass above), every pair of a set an edge.  Fixed points are taken
    // as coupling their observers too (they do not): a superset only costs separator.  The plan checks the promise against the true
    // tile pattern (CholPersist::build: chains that couple after all are factorised as one).
    const int nf = (int)fp_pose.size(), W = (nf + 63)/64;
    bool windows = true; for (int t = 0; t < T; ++t) windows = windows && !seen_over[t];
    if (getenv("MCP_BA_TEST_CHOL_ATOMIC")) windows = false;      // (tests: the masks of an add order whose observers do not fit the windows)
    std::vector<u64> pmask, adj((size_t)nf*W, 0), adj_t((size_t)T*nf*W, 0);
    const int* mp = meas_point.data(); const int* mc = meas_chain.data();
    if (!windows) {
      pmask.assign((size_t)std::max(npoint, 1)*W, 0);
      par([&](int tid) {
        for (long i = lo_of(tid, nmeas), e = lo_of(tid + 1, nmeas); i < e; ++i) {
          const HChain& oc = chains[mc[i]];
          for (int k = 0; k < oc.len; ++k) { const int u = poses[oc.v[k]].unk; if (u >= 0) __atomic_fetch_or(&pmask[(size_t)mp[i]*W + (u >> 6)], 1ull << (u & 63), __ATOMIC_RELAXED); }
        }
      });
    }
    lap("  chains: point masks");
    std::vector<int> unk_of(npose); for (int i = 0; i < npose; ++i) unk_of[i] = poses[i].unk;
    par([&](int tid) {
      u64* at = adj_t.data() + (size_t)tid*nf*W; u64 row[16];
      // (many points are seen by the same poses: a set that was expanded a moment ago is not expanded again -- a small direct-mapped memory of rows)
      constexpr int RMEM = 512; std::vector<u64> rmem((size_t)RMEM*W, 0);
      const int* uo = unk_of.data();
      for (long p = lo_of(tid, npoint), e = lo_of(tid + 1, npoint); p < e; ++p) {
        if (!points[p].active) continue;
        if (windows) {
          for (int k = 0; k < W; ++k) row[k] = 0;
          for (int t = 0; t < T; ++t) for (u64 m = seen_t[(size_t)t*npoint + p]; m; m &= m - 1) { const int u = uo[seen_ids[(size_t)t*64 + __builtin_ctzll(m)]]; if (u >= 0) row[u >> 6] |= 1ull << (u & 63); }
        } else for (int k = 0; k < W; ++k) row[k] = pmask[(size_t)p*W + k];
        const HChain& sc = chains[points[p].chain];
        for (int k = 0; k < sc.len; ++k) { const int u = uo[sc.v[k]]; if (u >= 0) row[u >> 6] |= 1ull << (u & 63); }
        u64 h = 0; for (int k = 0; k < W; ++k) h = (h ^ row[k])*0x9e3779b97f4a7c15ull;
        u64* slot = &rmem[(size_t)(h >> 55)*W];
        bool same = true; for (int k = 0; k < W; ++k) same = same && slot[k] == row[k];
        if (same) continue;
        for (int k = 0; k < W; ++k) slot[k] = row[k];
        for (int k = 0; k < W; ++k) for (u64 m = row[k]; m; m &= m - 1) { u64* a = at + (size_t)(64*k + __builtin_ctzll(m))*W; for (int q = 0; q < W; ++q) a[q] |= row[q]; }
      }
    });
    for (int t = 0; t < T; ++t) for (size_t i = 0; i < adj.size(); ++i) adj[i] |= adj_t[(size_t)t*nf*W + i];
    if (multi()) {
      // ranks hold different shards of the measurements: the union of their graphs, so that every rank finds the same cut
      std::vector<double> bits((size_t)nf*nf);
      for (int u = 0; u < nf; ++u) for (int v = 0; v < nf; ++v) bits[(size_t)u*nf + v] = (double)((adj[(size_t)u*W + (v >> 6)] >> (v & 63)) & 1);
      DevBuf<double> tmp;
      if (tmp.upload(bits, st)) return -1;
      if (allreduce(tmp.p, bits.size(), 0, true, "set-up: pose coupling graph")) return -1;
      HIPCK(hipMemcpy(bits.data(), tmp.p, bits.size()*sizeof(double), hipMemcpyDeviceToHost));
      for (int u = 0; u < nf; ++u) for (int v = 0; v < nf; ++v) if (bits[(size_t)u*nf + v] > 0) adj[(size_t)u*W + (v >> 6)] |= 1ull << (v & 63);
    }
    for (int u = 0; u < nf; ++u) adj[(size_t)u*W + (u >> 6)] &= ~(1ull << (u & 63));
    PoseCut cut;
    pose_cut(adj, nf, chain_arcs, T, par, [&](const char* what) { lap(what); }, cut);
    if (cut.relabelled || cut.taken) {
      std::vector<int> fp2(nf);
      for (int i = 0; i < nf; ++i) fp2[i] = fp_pose[cut.order[i]];
      fp_pose.swap(fp2);
      for (int u = 0; u < nf; ++u) poses[fp_pose[u]].unk = u;
    }
    chol_segs = cut.segs;
    if (cut.taken) { const char* e = getenv("MCP_BA_TEST_CHOL_CUT"); if (e) chol_segs[1] = std::max(3, chol_segs[1] - atoi(e)); }      // (tests: the cut between the chains moved into the first one -- chains that couple: the plan must notice)
    if (trace) {
      if (cut.looked_at_order) fprintf(stderr, "[mcp_ba prepare]   chains: couplings %.1f poses apart in add order, %.1f breadth-first -> %s\n", cut.dist_add, cut.dist_cm, cut.relabelled ? "relabelled" : "add order kept");
      if (cut.found) { char arcs[192]; int o = 0; for (int i = 0; i < cut.k; ++i) o += snprintf(arcs + o, sizeof arcs - o, "%s%d", i ? " + " : "", cut.arc_len[i]);
        fprintf(stderr, "[mcp_ba prepare]   chains: %d free poses; ring opened at %d, %d arcs of %s poses, gaps of %d (%d), separator %d (%d of them for what still coupled the arcs): %d block columns on the longest path of %d -> %s\n",
                nf, cut.r, cut.k, arcs, cut.g_first, cut.g_last, cut.sep, cut.ncover, cut.steps, cut.t_all, cut.taken ? "chains + the separator's" : "one chain"); }
      else fprintf(stderr, "[mcp_ba prepare]   chains: %d free poses; no cut found -> one chain\n", nf);
    }
    lap("  pose order (chains)");
  }
  for (int i = 0; i < npoint; ++i) if (points[i].active && !points[i].fixed) { points[i].unk = (int)fl_point.size(); fl_point.push_back(i); }
  nfp = (int)fp_pose.size(); nfl = (int)fl_point.size(); np = 6*nfp; nx = np + 3*nfl;
  if (np > CH_SOLVE_MAX) { set_err("too many free poses for the dense reduced solve (6P > 6144)"); return -1; }
  if (!multi()) nfl_total = nfl;
  lap("activity");
  for (int i = 0; i < npoint; ++i) cnt[i + 1] += cnt[i];        // measurements by point (add order kept inside a point): ranges
  // ---- point order: by the first free pose of the chain the point is expressed in (counting sort: stable)
  std::vector<int> order;
  {
    std::vector<int> pkey(npoint, -1), kc(nfp + 2, 0);
    for (int i = 0; i < npoint; ++i) {
      if (!points[i].active) continue;
      const HChain& c = chains[points[i].chain];
      int key = nfp;
      for (int k = 0; k < c.len; ++k) if (poses[c.v[k]].unk >= 0) { key = poses[c.v[k]].unk; break; }
      pkey[i] = key; kc[key + 1]++;
    }
    for (int k = 0; k <= nfp; ++k) kc[k + 1] += kc[k];
    order.resize(kc[nfp + 1]);
    for (int i = 0; i < npoint; ++i) if (pkey[i] >= 0) order[kc[pkey[i]]++] = i;
    // ... and inside a pose by which poses see the point (refine_point_order; its keys were taken in the counting pass): buckets sorted
    // by ranges of the pool (kc[k] is now the end of bucket k: thread ranges cut at bucket boundaries)
    if (refine) {
      par([&](int tid) {
        const int k0 = (int)lo_of(tid, nfp + 1), k1 = (int)lo_of(tid + 1, nfp + 1);
        const size_t i0 = k0 ? (size_t)kc[k0 - 1] : 0, i1 = k1 ? (size_t)kc[k1 - 1] : 0;
        refine_point_order(order, pkey, smax.data(), ssum.data(), i0, i1);
      });
    }
  }
  nsp = (int)order.size();
  lap("  point order");
  // per (obs chain, src chain) activity mask, memoised in a dense table (chains are few: P*C for the Multi adapter)
  const bool dense_table = nch <= 4096;
  // (entries are written concurrently by the structure threads below: every writer stores the same value, relaxed atomics)
  std::unique_ptr<std::atomic<unsigned short>[]> mask_table(dense_table ? new std::atomic<unsigned short>[nch*nch] : nullptr);
  static_assert(sizeof(std::atomic<unsigned short>) == 2, "mask table is filled bytewise");
  if (dense_table) std::memset(static_cast<void*>(mask_table.get()), 0xff, nch*nch*sizeof(unsigned short));      // (no thread reads it before the pool starts)
  std::map<std::pair<int, int>, unsigned short> mask_cache;
  std::mutex mask_mutex;
  auto compute_mask = [&](int oc, int sc) -> unsigned short {
    unsigned short mk = 0;
    const HChain& o = chains[oc]; const HChain& s2 = chains[sc];
    for (int i = 0; i < o.len; ++i) if (!poses[o.v[i]].fixed && !move_together(o, s2, i)) mk |= (1 << i);
    for (int i = 0; i < s2.len; ++i) if (!poses[s2.v[i]].fixed && !move_together(s2, o, i)) mk |= (1 << (MAXC + i));
    return mk;
  };
  auto pair_mask = [&](int oc, int sc) -> unsigned short {
    if (dense_table) {
      std::atomic<unsigned short>& e = mask_table[(size_t)oc*nch + sc];
      unsigned short v = e.load(std::memory_order_relaxed);
      if (v == 0xffff) { v = compute_mask(oc, sc); e.store(v, std::memory_order_relaxed); }
      return v;
    }
    std::lock_guard<std::mutex> lk(mask_mutex);
    auto key = std::make_pair(oc, sc);
    auto it = mask_cache.find(key);
    if (it != mask_cache.end()) return it->second;
    const unsigned short mk = compute_mask(oc, sc);
    mask_cache[key] = mk; return mk;
  };
  std::unique_lock<std::mutex> arena_lock(pinned_arena().mutex());
  ArenaGuard arena_guard{st};
  HostStruct H;
  H.m_pt.resize(nmeas); H.m_chain.resize(nmeas); H.m_sp.resize(nmeas); H.slot_start.resize(nmeas + 1);
  H.m_cam.resize(nmeas); H.m_mask.resize(nmeas); H.m_u.resize(nmeas); H.m_v.resize(nmeas); H.m_om.resize(nmeas);
  H.l_i0.assign(nfl + 1, 0); H.l_i1.assign(nfl + 1, 0); H.l_sp.assign(nfl + 1, 0);
  H.sp_pt.resize(nsp); H.sp_m.assign(nsp + 1, 0); H.sp_i.assign(nsp + 1, 0); H.sp_big.assign(nsp, 0);
  perm.assign(nmeas, 0);
  lap("  host arrays");
  std::vector<int> sp_of(std::max(npoint, 1), -1);
  for (int sp = 0; sp < nsp; ++sp) { H.sp_m[sp + 1] = H.sp_m[sp] + (cnt[order[sp] + 1] - cnt[order[sp]]); sp_of[order[sp]] = sp; }
  // ---- every measurement to its final position, a range of the add order per thread (sequential reads, one scattered 40-byte
  // write each; the reference's adapters add measurements KeyFrame by KeyFrame, so a point's measurements lie far apart).
  // Measurements of a point are stored rotated by the point's position: neighbouring points (= neighbouring lanes of
  // k_linearize_group) share their observers, and walking the lists in the same order makes all lanes add to the same
  // LDS tile entries at the same time; a per-lane rotation spreads them over the observers.
  par([&](int tid) {
    int* ct = cnt_t + (size_t)tid*npoint; const int* mp = meas_point.data(); const int* smp = H.sp_m.data();
    for (long i = lo_of(tid, nmeas), e = lo_of(tid + 1, nmeas); i < e; ++i) {
      const int p = mp[i], r = ct[p]++, sp = sp_of[p], nm = cnt[p + 1] - cnt[p];
      int kk = (r - sp) % nm; if (kk < 0) kk += nm;                    // rank r sits at slot kk with (kk + sp) % nm == r
      const HMeas& m = meas[i];
      SMeas& o = sorted[smp[sp] + kk];
      o.chain = m.chain; o.cam = m.cam; o.mi = (int)i; o.u = m.u; o.v = m.v; o.omega = m.omega;
    }
  });
  lap("  by point, order");
  // ---- slots, incidences and the poses of every point: contiguous ranges of sorted points per thread (balanced by measurement
  // count), each into its own arrays with range-relative indices; concatenated below with the offsets fixed up
  std::vector<unsigned char> cov((size_t)std::max(nfp, 1)*std::max(nfp, 1), 0);        // pose-pair co-visibility, a >= b
  struct Chunk { int sp0 = 0, sp1 = 0; std::vector<int> slot_unk, slot_inc, inc_unk, slot_cnt, sp_ninc, q_n, q_data; std::vector<unsigned char> slot_first, inc_state; };
  // (eight chunks per thread (MCP_BA_PREP_CHUNKS), claimed one after the other: equal measurement counts are not equal times -- 1.0 to 1.6 ms per thread
  //  at the metric size with one chunk each -- and the concatenation below is by chunk index, so who built a chunk changes nothing)
  static const int chunks_per_thread = [] { const char* e = getenv("MCP_BA_PREP_CHUNKS"); return e ? std::max(1, atoi(e)) : 8; }();
  const int NC = T > 1 ? chunks_per_thread*T : 1;
  std::vector<Chunk> chunks(NC);
  const avec<int>& sp_m = H.sp_m;
  for (int t = 0; t < NC; ++t) {
    const long m0 = (long)nmeas*t/NC, m1 = (long)nmeas*(t + 1)/NC;
    chunks[t].sp0 = (t == 0) ? 0 : (int)(std::lower_bound(sp_m.begin(), sp_m.begin() + nsp, (int)m0) - sp_m.begin());
    chunks[t].sp1 = (t == NC - 1) ? nsp : (int)(std::lower_bound(sp_m.begin(), sp_m.begin() + nsp, (int)m1) - sp_m.begin());
  }
  std::vector<double> thr_ms(T, 0.0);
  std::atomic<int> next_chunk{0};
  par([&](int tid) {
    const auto tt0 = std::chrono::steady_clock::now();
    struct Stamp { double* d; std::chrono::steady_clock::time_point t; ~Stamp() { *d = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count(); } } stamp_{&thr_ms[tid], tt0};
    for (int ci; (ci = next_chunk.fetch_add(1, std::memory_order_relaxed)) < NC;) {
    Chunk& C = chunks[ci];
    const size_t nm_chunk = (size_t)(sp_m[C.sp1] - sp_m[C.sp0]);
    C.slot_unk.reserve(nm_chunk*2 + 16); C.slot_inc.reserve(nm_chunk*2 + 16); C.slot_first.reserve(nm_chunk*2 + 16);
    C.inc_unk.reserve(nm_chunk + 16); C.inc_state.reserve(nm_chunk + 16); C.q_data.reserve(nm_chunk + 16);
    C.sp_ninc.assign(C.sp1 - C.sp0, 0); C.q_n.assign(C.sp1 - C.sp0, 0); C.slot_cnt.assign(nm_chunk, 0);
    std::vector<int> q; q.reserve(64);
    unsigned char* covp = cov.data();
    for (int sp = C.sp0; sp < C.sp1; ++sp) {
      const int pt = order[sp];
      const int lpt = points[pt].unk, pch = points[pt].chain;
      const HChain& sc = chains[pch];
      const int ibase = (int)C.inc_unk.size();
      H.sp_pt[sp] = pt;
      q.clear();
      const int nm_pt = cnt[pt + 1] - cnt[pt];
      int j = sp_m[sp];
      for (int kk = 0; kk < nm_pt; ++kk, ++j) {
        const SMeas& m = sorted[j];
        perm[j] = m.mi;
        H.m_pt[j] = pt; H.m_chain[j] = m.chain; H.m_cam[j] = (unsigned char)m.cam; H.m_u[j] = m.u; H.m_v[j] = m.v; H.m_om[j] = m.omega; H.m_sp[j] = sp;
        const unsigned short mk = pair_mask(m.chain, pch);
        H.m_mask[j] = mk;
        int nsl = 0;
        const HChain& oc = chains[m.chain];
        for (unsigned bits = mk; bits; bits &= bits - 1) {
          const int b = __builtin_ctz(bits);
          const HChain& c = (b < MAXC) ? oc : sc;
          const int u = poses[c.v[b & (MAXC - 1)]].unk;
          C.slot_unk.push_back(u); ++nsl;
          if (std::find(q.begin(), q.end(), u) == q.end()) q.push_back(u);
          // slot_first: first contribution to its W block among the slots that write it through memory (every slot but
          // the first source link, whose block is kept in registers by k_linearize_group); inc_mixed: both kinds occur
          int inc = -1; unsigned char first = 0;
          if (lpt >= 0) {
            for (int t2 = ibase; t2 < (int)C.inc_unk.size(); ++t2) if (C.inc_unk[t2] == u) { inc = t2; break; }
            if (inc < 0) { inc = (int)C.inc_unk.size(); C.inc_unk.push_back(u); C.inc_state.push_back(0); }
            if (b == MAXC) C.inc_state[inc] |= 1;
            else { if (!(C.inc_state[inc] & 2)) first = 1; C.inc_state[inc] |= 2; }
          }
          C.slot_inc.push_back(inc); C.slot_first.push_back(first);
        }
        C.slot_cnt[j - sp_m[C.sp0]] = nsl;
      }
      C.sp_ninc[sp - C.sp0] = (int)C.inc_unk.size() - ibase;
      C.q_n[sp - C.sp0] = (int)q.size();
      C.q_data.insert(C.q_data.end(), q.begin(), q.end());
      if ((int)q.size() > GRP_LMAX) H.sp_big[sp] = 1;
      for (int a : q) for (int b2 : q) if (a >= b2 && !__atomic_load_n(covp + (size_t)a*nfp + b2, __ATOMIC_RELAXED)) __atomic_store_n(covp + (size_t)a*nfp + b2, (unsigned char)1, __ATOMIC_RELAXED);
    }
    }
  });
  lap("  slot threads");
  if (trace) { fprintf(stderr, "[mcp_ba prepare]   per-thread ms:"); for (double v : thr_ms) fprintf(stderr, " %.2f", v); fprintf(stderr, "\n"); }
  std::vector<int> sp_q0(nsp + 1, 0), sp_q;           // distinct pose unknowns touched by every sorted point
  std::vector<unsigned char> inc_state;
  {
    std::vector<size_t> soff(NC + 1, 0), ioff(NC + 1, 0), qoff(NC + 1, 0);
    for (int t = 0; t < NC; ++t) { soff[t + 1] = soff[t] + chunks[t].slot_unk.size(); ioff[t + 1] = ioff[t] + chunks[t].inc_unk.size(); qoff[t + 1] = qoff[t] + chunks[t].q_data.size(); }
    H.slot_unk.resize(soff[NC]); H.slot_inc.resize(soff[NC]); H.slot_first.resize(soff[NC]); H.inc_unk.resize(ioff[NC]); inc_state.resize(ioff[NC]); sp_q.resize(qoff[NC]);
    next_chunk.store(0);
    par([&](int) {
      for (int tid; (tid = next_chunk.fetch_add(1, std::memory_order_relaxed)) < NC;) {
      const Chunk& C = chunks[tid];
      const int io = (int)ioff[tid];
      int so = (int)soff[tid];
      for (int j = sp_m[C.sp0]; j < sp_m[C.sp1]; ++j) { H.slot_start[j] = so; so += C.slot_cnt[j - sp_m[C.sp0]]; }
      if (!C.slot_unk.empty()) {
        std::memcpy(&H.slot_unk[soff[tid]], C.slot_unk.data(), C.slot_unk.size()*sizeof(int));
        std::memcpy(&H.slot_first[soff[tid]], C.slot_first.data(), C.slot_first.size());
        int* d = &H.slot_inc[soff[tid]];
        for (size_t k = 0; k < C.slot_inc.size(); ++k) d[k] = C.slot_inc[k] < 0 ? -1 : C.slot_inc[k] + io;
      }
      if (!C.inc_unk.empty()) { std::memcpy(&H.inc_unk[ioff[tid]], C.inc_unk.data(), C.inc_unk.size()*sizeof(int)); std::memcpy(&inc_state[ioff[tid]], C.inc_state.data(), C.inc_state.size()); }
      if (!C.q_data.empty()) std::memcpy(&sp_q[qoff[tid]], C.q_data.data(), C.q_data.size()*sizeof(int));
      int ib = io, qb = (int)qoff[tid];
      for (int sp = C.sp0; sp < C.sp1; ++sp) {
        H.sp_i[sp] = ib; sp_q0[sp] = qb;
        const int lpt = points[order[sp]].unk;
        if (lpt >= 0) { H.l_i0[lpt] = ib; H.l_i1[lpt] = ib + C.sp_ninc[sp - C.sp0]; H.l_sp[lpt] = sp; }
        ib += C.sp_ninc[sp - C.sp0]; qb += C.q_n[sp - C.sp0];
      }
      }
    });
    sp_q0[nsp] = (int)qoff[NC];
    H.sp_i[nsp] = (int)ioff[NC];
    H.slot_start[nmeas] = (int)soff[NC];
  }
  ninc = (int)H.inc_unk.size(); nslot = (int)H.slot_unk.size();
  lap("sort+slots");
  // ---- groups: consecutive points, <= grp_pts points and <= GRP_LMAX distinct poses (greedy, in order).  A map of few points
  // (every BundleAdjustRecent window) gets quarter-size groups: four times the workgroups on a chip they do not fill anyway, a
  // quarter of the per-group latency in the linearisation (four lanes per point) and in the Schur complement (one chunk per group)
  grp_pts = group_points(nsp);
  nbig = 0;
  {
    std::vector<int> stamp(std::max(nfp, 1), -1), curset; curset.reserve(GRP_LMAX);
    int start = 0, gid = 0;
    auto close = [&](int end) {
      if (end <= start) return;
      std::sort(curset.begin(), curset.end());
      H.g_sp0.push_back(start);
      for (int k = 0; k < GRP_LMAX; ++k) H.g_pose.push_back(k < (int)curset.size() ? curset[k] : -1);
      curset.clear(); start = end; ++gid;
    };
    for (int sp = 0; sp < nsp; ++sp) {
      const bool big = H.sp_big[sp] != 0;                          // big points ride along with no poses
      if (big) ++nbig;
      const int* q = sp_q.data() + sp_q0[sp]; const int nq = big ? 0 : sp_q0[sp + 1] - sp_q0[sp];
      int fresh = 0;
      for (int k = 0; k < nq; ++k) if (stamp[q[k]] != gid) ++fresh;
      if ((int)curset.size() + fresh > grp_lmax_pol || sp - start >= grp_pts) close(sp);
      for (int k = 0; k < nq; ++k) if (stamp[q[k]] != gid) { stamp[q[k]] = gid; curset.push_back(q[k]); }
    }
    close(nsp);
    H.g_sp0.push_back(nsp);
  }
  ngroup = (int)H.g_sp0.size() - 1;
  lap("groups");
  // ---- tile occupancy of the reduced pose system: poses a, b interact iff some point touches both
  if (np > 0) {
    const int ntc = (np + CH_NB - 1)/CH_NB;
    std::vector<unsigned char> pat((size_t)ntc*ntc, 0);
    for (int a = 0; a < nfp; ++a) for (int b = 0; b <= a; ++b) {
      if (!cov[(size_t)a*nfp + b]) continue;
      const int ra0 = (6*a)/CH_NB, ra1 = (6*a + 5)/CH_NB, rb0 = (6*b)/CH_NB, rb1 = (6*b + 5)/CH_NB;
      for (int ra = ra0; ra <= ra1; ++ra) for (int rb = rb0; rb <= rb1; ++rb) if (ra >= rb) pat[(size_t)ra*ntc + rb] = 1;
    }
    for (int i = 0; i < ntc; ++i) pat[(size_t)i*ntc + i] = 1;
    if (multi()) {
      // the union of every rank's co-visibility pattern: agreed once, used for the factorisation plan and for the
      // packed all-reduce of the reduced system (only structurally non-zero tiles travel over xGMI)
      std::vector<double> pd(pat.begin(), pat.end());
      DevBuf<double> tmp;
      if (tmp.upload(pd, st)) return -1;
      if (allreduce(tmp.p, pd.size(), 0, true, "set-up: tile pattern")) return -1;
      HIPCK(hipMemcpy(pd.data(), tmp.p, pd.size()*sizeof(double), hipMemcpyDeviceToHost));
      for (size_t i = 0; i < pd.size(); ++i) pat[i] = pd[i] > 0;
      std::vector<int> rt;
      for (int i = 0; i < ntc; ++i) for (int j = 0; j <= i; ++j) if (pat[(size_t)i*ntc + j]) rt.push_back((i << 16) | j);
      n_red_tiles = (int)rt.size();
      if (d_red_tiles.upload(rt, st) || d_pack.alloc(MAX_SYS*((size_t)n_red_tiles*CH_NB*CH_NB + 2*(size_t)np))) return -1;
      pack_stride = (size_t)n_red_tiles*CH_NB*CH_NB + 2*(size_t)np;
    }
    if (trace) {
      std::string hs; std::vector<int> hd(ntc, 0); int tot = 0;
      for (int i = 0; i < ntc; ++i) for (int j = 0; j <= i; ++j) if (pat[(size_t)i*ntc + j]) { ++hd[i - j]; ++tot; }
      for (int d = 0; d < ntc; ++d) { char b[16]; snprintf(b, sizeof b, " %d", hd[d]); hs += b; }
      fprintf(stderr, "[mcp_ba prepare]   tile pattern before fill-in: %d of %d lower tiles; by distance from the diagonal:%s\n", tot, ntc*(ntc + 1)/2, hs.c_str());
    }
    lap("  covisibility");
    plan.persist_segs = chol_segs;
    if (plan.build(np, pat)) { set_err("Cholesky plan allocation failed"); return -1; }
    last_pat = std::move(pat);
    lap("  symbolic plan");
  } else { plan.all_tiles.clear(); last_pat.clear(); }
  // ---- per group (threads over ranges of groups): local pose indices of slots and incidences, and which local pose pairs the
  // group's points co-observe (one bit per pair of the 16 x 17 / 2)
  H.slot_lp.assign(nslot + 1, 0); H.inc_lp.assign(ninc + 1, 0); H.inc_mixed.assign(ninc + 1, 0);
  constexpr int NPAIR = GRP_LMAX*(GRP_LMAX + 1)/2;
  std::vector<unsigned char> gcov((size_t)std::max(ngroup, 1)*NPAIR, 0);
  H.g_blk0.assign(ngroup + 1, 0);
  auto ltri = [](int r, int c) { return r*(r + 1)/2 + c; };
  par([&](int tid) {
    for (long i = lo_of(tid, ninc), e = lo_of(tid + 1, ninc); i < e; ++i) H.inc_mixed[i] = (inc_state[i] == 3);
    std::vector<unsigned char> lpmap(std::max(nfp, 1), 0);
    for (int gi = (int)lo_of(tid, ngroup), ge = (int)lo_of(tid + 1, ngroup); gi < ge; ++gi) {
      const int* gp = &H.g_pose[(size_t)gi*GRP_LMAX];
      for (int k = 0; k < GRP_LMAX; ++k) if (gp[k] >= 0) lpmap[gp[k]] = (unsigned char)k;
      unsigned char* gc = &gcov[(size_t)gi*NPAIR];
      for (int sp = H.g_sp0[gi]; sp < H.g_sp0[gi + 1]; ++sp) {
        if (H.sp_big[sp]) continue;
        for (int s2 = H.slot_start[sp_m[sp]]; s2 < H.slot_start[sp_m[sp + 1]]; ++s2) H.slot_lp[s2] = lpmap[H.slot_unk[s2]];
        for (int i2 = H.sp_i[sp]; i2 < H.sp_i[sp + 1]; ++i2) H.inc_lp[i2] = lpmap[H.inc_unk[i2]];
        const int* q = sp_q.data() + sp_q0[sp]; const int nq = sp_q0[sp + 1] - sp_q0[sp];
        for (int x = 0; x < nq; ++x) for (int y = 0; y < nq; ++y) { const int lx = lpmap[q[x]], ly = lpmap[q[y]]; if (lx >= ly) gc[ltri(lx, ly)] = 1; }
      }
      int nb = 0; for (int k = 0; k < NPAIR; ++k) nb += gc[k];
      H.g_blk0[gi + 1] = nb;
    }
  });
  lap("  local indices");
  // ---- fixed-order assembly plan (ba_group.h): which local pose pairs every group stages, and for every global pose
  // pair / pose the list of staged slots in ascending group order
  grp_blk_max = 0; grp_inc_max = 0;
  for (int gi = 0; gi < ngroup; ++gi) {
    grp_blk_max = std::max(grp_blk_max, H.g_blk0[gi + 1]); H.g_blk0[gi + 1] += H.g_blk0[gi];
    grp_inc_max = std::max(grp_inc_max, H.sp_i[H.g_sp0[gi + 1]] - H.sp_i[H.g_sp0[gi]]);
  }
  nstage = (size_t)H.g_blk0[ngroup];
  H.blk_pair.resize(nstage); H.blk_dst.assign(nstage, 0);
  H.pair_id.assign((size_t)std::max(nfp, 1)*std::max(nfp, 1), -1); H.po_start.assign(nfp + 1, 0); H.rhs_dst.assign((size_t)std::max(ngroup, 1)*GRP_LMAX, -1);
  {
    std::vector<int> blk_pid(nstage);            // global pose pair (id) of every staged block
    std::vector<int> cnt_pair; int npairs = 0;
    for (int gi = 0; gi < ngroup; ++gi) {
      const int* gp = &H.g_pose[(size_t)gi*GRP_LMAX];
      const unsigned char* gc = &gcov[(size_t)gi*NPAIR];
      size_t k = (size_t)H.g_blk0[gi];
      for (int la = 0; la < GRP_LMAX; ++la) for (int lb = 0; lb <= la; ++lb) if (gc[ltri(la, lb)]) {
        H.blk_pair[k] = (unsigned char)((la << 4) | lb);
        int& id = H.pair_id[(size_t)gp[la]*nfp + gp[lb]];               // g_pose is ascending: gp[la] >= gp[lb]
        if (id < 0) { id = npairs++; cnt_pair.push_back(0); }
        cnt_pair[id]++; blk_pid[k] = id; ++k;
      }
      for (int q2 = 0; q2 < GRP_LMAX; ++q2) if (gp[q2] >= 0) H.po_start[gp[q2] + 1]++;
    }
    H.pr_start.assign(npairs + 1, 0);
    for (int i = 0; i < npairs; ++i) H.pr_start[i + 1] = H.pr_start[i] + cnt_pair[i];
    if (trace) {
      // (list lengths of the assembly's walks: diagonal pairs against the rest)
      long nd = 0, sd = 0, md = 0, no = 0, so = 0, mo = 0;
      for (int a = 0; a < nfp; ++a) for (int b = 0; b <= a; ++b) { const int pid = H.pair_id[(size_t)a*nfp + b]; if (pid < 0) continue; const long c = cnt_pair[pid];
        if (a == b) { ++nd; sd += c; md = std::max(md, c); } else { ++no; so += c; mo = std::max(mo, c); } }
      fprintf(stderr, "[mcp_ba prepare] staged blocks per pose pair: diagonal %ld pairs, mean %.1f, max %ld; off-diagonal %ld pairs, mean %.1f, max %ld\n", nd, nd ? (double)sd/nd : 0.0, md, no, no ? (double)so/no : 0.0, mo);
    }
    // destination-ordered staging: the blocks of one pose pair are consecutive, in ascending group order (blocks are
    // numbered group by group, so walking them in order fills every pair's run in that order)
    { std::vector<int> pos(H.pr_start.begin(), H.pr_start.end() - 1);
      for (size_t k = 0; k < nstage; ++k) H.blk_dst[k] = pos[blk_pid[k]]++; }
    for (int a = 0; a < nfp; ++a) H.po_start[a + 1] += H.po_start[a];
    nrhs_rows = H.po_start[nfp];
    { std::vector<int> pos(H.po_start.begin(), H.po_start.end() - 1);
      for (int gi = 0; gi < ngroup; ++gi) for (int k = 0; k < GRP_LMAX; ++k) { const int u = H.g_pose[(size_t)gi*GRP_LMAX + k]; if (u >= 0) H.rhs_dst[(size_t)gi*GRP_LMAX + k] = pos[u]++; } }
  }
  lap("pattern+plan");
  if (trace) fprintf(stderr, "[mcp_ba prepare] %d groups, %zu staged blocks (%.1f per group, at most %d), %d host threads\n", ngroup, nstage, ngroup ? (double)nstage/ngroup : 0.0, grp_blk_max, T);
  return finish_prepare(H, t0, tlast, trace);
}

int mcp_ba::prepare_legacy() {
  cache_insert = false;            // (the serial builder neither consults nor fills the structure cache)
  auto t0 = std::chrono::steady_clock::now();
  auto tlast = t0; const bool trace = getenv("MCP_BA_TRACE") != nullptr;
  auto lap = [&](const char* what) { if (!trace) return; auto n = std::chrono::steady_clock::now(); fprintf(stderr, "[mcp_ba prepare] %-22s %.3f ms\n", what, std::chrono::duration<double, std::milli>(n - tlast).count()); tlast = n; };
  HIPCK(hipSetDevice(device));
  const int npose = (int)poses.size(), npoint = (int)points.size(), nmeas = (int)meas.size();
  for (auto& p : poses) { p.active = 0; p.unk = -1; }
  for (auto& p : points) { p.active = 0; p.unk = -1; }
  for (const auto& m : meas) {
    HPoint& p = points[m.point]; p.active = 1;
    const HChain& oc = chains[m.chain]; const HChain& sc = chains[p.chain];
    for (int k = 0; k < oc.len; ++k) poses[oc.v[k]].active = 1;
    for (int k = 0; k < sc.len; ++k) poses[sc.v[k]].active = 1;
  }
  m_total = (double)nmeas;
  if (multi()) {
    // ranks hold different measurement shards: agree on the active poses and the global count
    std::vector<double> flags(npose + 2);
    for (int i = 0; i < npose; ++i) flags[i] = poses[i].active;
    flags[npose] = (double)nmeas;
    double nfree = 0; for (const auto& p : points) if (p.active && !p.fixed) nfree += 1;
    flags[npose + 1] = nfree;
    DevBuf<double> tmp;
    if (tmp.upload(flags, st)) return -1;
    if (allreduce(tmp.p, flags.size(), 0, true, "set-up: active poses and counts")) return -1;
    HIPCK(hipMemcpy(flags.data(), tmp.p, flags.size()*sizeof(double), hipMemcpyDeviceToHost));
    for (int i = 0; i < npose; ++i) poses[i].active = flags[i] > 0;
    m_total = flags[npose]; nfl_total = flags[npose + 1];
  }
  fp_pose.clear(); fl_point.clear();
  for (int i = 0; i < npose; ++i) if (poses[i].active && !poses[i].fixed) { poses[i].unk = (int)fp_pose.size(); fp_pose.push_back(i); }
  for (int i = 0; i < npoint; ++i) if (points[i].active && !points[i].fixed) { points[i].unk = (int)fl_point.size(); fl_point.push_back(i); }
  nfp = (int)fp_pose.size(); nfl = (int)fl_point.size(); np = 6*nfp; nx = np + 3*nfl;
  if (np > CH_SOLVE_MAX) { set_err("too many free poses for the dense reduced solve (6P > 6144)"); return -1; }
  if (!multi()) nfl_total = nfl;
  lap("activity");
  // ---- measurements by point (add order kept inside a point)
  std::vector<int> cnt(npoint + 1, 0);
  for (int i = 0; i < nmeas; ++i) cnt[meas_point[i] + 1]++;
  for (int i = 0; i < npoint; ++i) cnt[i + 1] += cnt[i];
  std::vector<int> by_point(nmeas);
  { std::vector<int> pos(cnt.begin(), cnt.end() - 1);
    for (int i = 0; i < nmeas; ++i) by_point[pos[meas_point[i]]++] = i; }
  // ---- point order: by the first free pose of the chain the point is expressed in
  std::vector<int> order;
  order.reserve(npoint);
  std::vector<int> pkey(npoint, INT32_MAX);
  for (int i = 0; i < npoint; ++i) {
    if (!points[i].active) continue;
    const HChain& c = chains[points[i].chain];
    for (int k = 0; k < c.len; ++k) if (poses[c.v[k]].unk >= 0) { pkey[i] = poses[c.v[k]].unk; break; }
    order.push_back(i);
  }
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return pkey[a] < pkey[b]; });
  if (point_order_refine) {      // (the same secondary order as the threaded builder's)
    std::vector<int> smax(std::max(npoint, 1), 0), ssum(std::max(npoint, 1), 0);
    for (int i = 0; i < nmeas; ++i) {
      const int k = chain_first_movable(chains[meas[i].chain], poses) + 1;
      ssum[meas_point[i]] += k; smax[meas_point[i]] = std::max(smax[meas_point[i]], k);
    }
    refine_point_order(order, pkey, smax.data(), ssum.data(), 0, order.size());
  }
  nsp = (int)order.size();
  // per (obs chain, src chain) activity mask, memoised in a dense table (chains are few: P*C for the Multi adapter)
  const size_t nch = chains.size();
  const bool dense_table = nch <= 4096;
  // (entries are written concurrently by the structure threads below: every writer stores the same value, relaxed atomics)
  std::unique_ptr<std::atomic<unsigned short>[]> mask_table(dense_table ? new std::atomic<unsigned short>[nch*nch] : nullptr);
  if (dense_table) for (size_t i = 0; i < nch*nch; ++i) mask_table[i].store(0xffff, std::memory_order_relaxed);
  std::map<std::pair<int, int>, unsigned short> mask_cache;
  std::mutex mask_mutex;
  auto compute_mask = [&](int oc, int sc) -> unsigned short {
    unsigned short mk = 0;
    const HChain& o = chains[oc]; const HChain& s = chains[sc];
    for (int i = 0; i < o.len; ++i) if (!poses[o.v[i]].fixed && !move_together(o, s, i)) mk |= (1 << i);
    for (int i = 0; i < s.len; ++i) if (!poses[s.v[i]].fixed && !move_together(s, o, i)) mk |= (1 << (MAXC + i));
    return mk;
  };
  auto pair_mask = [&](int oc, int sc) -> unsigned short {
    if (dense_table) {
      std::atomic<unsigned short>& e = mask_table[(size_t)oc*nch + sc];
      unsigned short v = e.load(std::memory_order_relaxed);
      if (v == 0xffff) { v = compute_mask(oc, sc); e.store(v, std::memory_order_relaxed); }
      return v;
    }
    std::lock_guard<std::mutex> lk(mask_mutex);
    auto key = std::make_pair(oc, sc);
    auto it = mask_cache.find(key);
    if (it != mask_cache.end()) return it->second;
    const unsigned short mk = compute_mask(oc, sc);
    mask_cache[key] = mk; return mk;
  };
  std::vector<int> m_pt(nmeas), m_chain(nmeas), m_sp(nmeas), slot_start(nmeas + 1, 0), slot_unk, slot_inc;
  std::vector<unsigned char> m_cam(nmeas), slot_first;
  std::vector<unsigned short> m_mask(nmeas);
  std::vector<double> m_u(nmeas), m_v(nmeas), m_om(nmeas);
  std::vector<int> l_i0(nfl + 1, 0), l_i1(nfl + 1, 0), l_sp(nfl + 1, 0), inc_unk;
  std::vector<int> sp_pt(nsp), sp_m(nsp + 1, 0), sp_i(nsp + 1, 0);
  std::vector<unsigned char> sp_big(nsp, 0);
  std::vector<std::vector<int>> sp_poses(nsp);       // distinct pose unknowns touched by the point
  std::vector<unsigned char> inc_state;              // bit0: fed by a first-source-link slot, bit1: fed by another slot
  perm.assign(nmeas, 0);
  // measurement ranges of the sorted points (prefix sum, serial and cheap); the per-point work below is independent per point
  for (int sp = 0; sp < nsp; ++sp) sp_m[sp + 1] = sp_m[sp] + (cnt[order[sp] + 1] - cnt[order[sp]]);
  // The slot / incidence lists are built by a few host threads over contiguous ranges of sorted points, each into its own
  // vectors (slot and incidence indices relative to the range), then concatenated with the offsets fixed up: the result is
  // identical to a serial pass.
  struct Chunk { int sp0, sp1; std::vector<int> slot_unk, slot_inc, inc_unk, slot_cnt; std::vector<unsigned char> slot_first, inc_state; std::vector<int> sp_ninc; };
  const int nthr = (nsp >= 4096) ? std::max(1, std::min(16, usable_cores())) : 1;
  std::vector<Chunk> chunks(nthr);
  auto build_chunk = [&](Chunk& C) {
    C.slot_unk.reserve((size_t)(sp_m[C.sp1] - sp_m[C.sp0])*2); C.slot_inc.reserve(C.slot_unk.capacity()); C.slot_first.reserve(C.slot_unk.capacity());
    C.sp_ninc.assign(C.sp1 - C.sp0, 0); C.slot_cnt.assign(sp_m[C.sp1] - sp_m[C.sp0], 0);
    for (int sp = C.sp0; sp < C.sp1; ++sp) {
      const int pt = order[sp];
      const int lpt = points[pt].unk;
      const int ibase = (int)C.inc_unk.size();
      sp_pt[sp] = pt;
      std::vector<int>& q = sp_poses[sp];
      // Measurements of a point are stored rotated by the point's position: neighbouring points (= neighbouring lanes of
      // k_linearize_group) share their observers, and walking the lists in the same order makes all lanes add to the same
      // LDS tile entries at the same time; a per-lane rotation spreads them over the observers.
      const int nm_pt = cnt[pt + 1] - cnt[pt];
      // (the measurements of a point are scattered over the add-order array: fetch the next point's while this one is worked on)
      if (sp + 1 < C.sp1) { const int pn = order[sp + 1]; for (int e = cnt[pn]; e < cnt[pn + 1]; ++e) __builtin_prefetch(&meas[by_point[e]], 0, 1); }
      int j = sp_m[sp];
      for (int kk = 0; kk < nm_pt; ++kk, ++j) {
        const int mi = by_point[cnt[pt] + (kk + sp) % nm_pt];
        const HMeas& m = meas[mi];
        perm[j] = mi;
        m_pt[j] = pt; m_chain[j] = m.chain; m_cam[j] = (unsigned char)m.cam; m_u[j] = m.u; m_v[j] = m.v; m_om[j] = m.omega; m_sp[j] = sp;
        const unsigned short mk = pair_mask(m.chain, points[pt].chain);
        m_mask[j] = mk;
        int nsl = 0;
        for (int b = 0; b < 2*MAXC; ++b) {
          if (!(mk & (1 << b))) continue;
          const HChain& c = (b < MAXC) ? chains[m.chain] : chains[points[pt].chain];
          const int u = poses[c.v[b & (MAXC - 1)]].unk;
          C.slot_unk.push_back(u); ++nsl;
          if (std::find(q.begin(), q.end(), u) == q.end()) q.push_back(u);
          // slot_first: first contribution to its W block among the slots that write it through memory (every slot but
          // the first source link, whose block is kept in registers by k_linearize_group); inc_mixed: both kinds occur
          int inc = -1; unsigned char first = 0;
          if (lpt >= 0) {
            for (int t = ibase; t < (int)C.inc_unk.size(); ++t) if (C.inc_unk[t] == u) { inc = t; break; }
            if (inc < 0) { inc = (int)C.inc_unk.size(); C.inc_unk.push_back(u); C.inc_state.push_back(0); }
            if (b == MAXC) C.inc_state[inc] |= 1;
            else { if (!(C.inc_state[inc] & 2)) first = 1; C.inc_state[inc] |= 2; }
          }
          C.slot_inc.push_back(inc); C.slot_first.push_back(first);
        }
        C.slot_cnt[j - sp_m[C.sp0]] = nsl;
      }
      C.sp_ninc[sp - C.sp0] = (int)C.inc_unk.size() - ibase;
      if ((int)q.size() > GRP_LMAX) sp_big[sp] = 1;
    }
  };
  {
    std::vector<std::thread> pool;
    for (int t = 0; t < nthr; ++t) {
      // ranges balanced by measurement count
      const long m0 = (long)nmeas*t/nthr, m1 = (long)nmeas*(t + 1)/nthr;
      chunks[t].sp0 = (t == 0) ? 0 : (int)(std::lower_bound(sp_m.begin(), sp_m.begin() + nsp, (int)m0) - sp_m.begin());
      chunks[t].sp1 = (t == nthr - 1) ? nsp : (int)(std::lower_bound(sp_m.begin(), sp_m.begin() + nsp, (int)m1) - sp_m.begin());
    }
    lap("  by point, order");
    for (int t = 1; t < nthr; ++t) pool.emplace_back(build_chunk, std::ref(chunks[t]));
    build_chunk(chunks[0]);
    for (auto& th : pool) th.join();
    lap("  slot threads");
  }
  {
    size_t tot_slot = 0, tot_inc = 0;
    for (const Chunk& C : chunks) { tot_slot += C.slot_unk.size(); tot_inc += C.inc_unk.size(); }
    slot_unk.reserve(tot_slot); slot_inc.reserve(tot_slot); slot_first.reserve(tot_slot); inc_unk.reserve(tot_inc); inc_state.reserve(tot_inc);
    for (const Chunk& C : chunks) {
      const int ioff = (int)inc_unk.size();
      int soff = (int)slot_unk.size();
      for (int j = sp_m[C.sp0]; j < sp_m[C.sp1]; ++j) { slot_start[j] = soff; soff += C.slot_cnt[j - sp_m[C.sp0]]; }
      slot_unk.insert(slot_unk.end(), C.slot_unk.begin(), C.slot_unk.end());
      slot_first.insert(slot_first.end(), C.slot_first.begin(), C.slot_first.end());
      for (int v : C.slot_inc) slot_inc.push_back(v < 0 ? -1 : v + ioff);
      inc_unk.insert(inc_unk.end(), C.inc_unk.begin(), C.inc_unk.end());
      inc_state.insert(inc_state.end(), C.inc_state.begin(), C.inc_state.end());
      int ib = ioff;
      for (int sp = C.sp0; sp < C.sp1; ++sp) {
        sp_i[sp] = ib;
        const int lpt = points[order[sp]].unk;
        if (lpt >= 0) { l_i0[lpt] = ib; l_i1[lpt] = ib + C.sp_ninc[sp - C.sp0]; l_sp[lpt] = sp; }
        ib += C.sp_ninc[sp - C.sp0];
      }
    }
  }
  sp_i[nsp] = (int)inc_unk.size();
  slot_start[nmeas] = (int)slot_unk.size();
  ninc = (int)inc_unk.size(); nslot = (int)slot_unk.size();
  lap("sort+slots");
  // ---- groups: consecutive points, <= grp_pts points and <= GRP_LMAX distinct poses
  std::vector<int> g_sp0, g_pose;
  grp_pts = group_points(nsp);
  nbig = 0;
  {
    std::vector<int> curset;
    int start = 0;
    auto close = [&](int end) {
      if (end <= start) return;
      std::sort(curset.begin(), curset.end());
      g_sp0.push_back(start);
      for (int k = 0; k < GRP_LMAX; ++k) g_pose.push_back(k < (int)curset.size() ? curset[k] : -1);
      curset.clear(); start = end;
    };
    for (int sp = 0; sp < nsp; ++sp) {
      const std::vector<int>& q = sp_big[sp] ? std::vector<int>() : sp_poses[sp];   // big points ride along with no poses
      if (sp_big[sp]) ++nbig;
      std::vector<int> merged = curset;
      for (int u : q) if (std::find(merged.begin(), merged.end(), u) == merged.end()) merged.push_back(u);
      if ((int)merged.size() > grp_lmax_pol || sp - start >= grp_pts) { close(sp); merged = q; }
      curset = merged;
    }
    close(nsp);
    g_sp0.push_back(nsp);
  }
  ngroup = (int)g_sp0.size() - 1;
  lap("groups");
  // ---- tile occupancy of the reduced pose system: poses a, b interact iff some point touches both
  if (np > 0) {
    const int ntc = (np + CH_NB - 1)/CH_NB;
    std::vector<unsigned char> pat((size_t)ntc*ntc, 0);
    {
      std::vector<unsigned char> cov((size_t)nfp*nfp, 0);                  // pose-pair co-visibility, a >= b
      for (int sp = 0; sp < nsp; ++sp) {
        const std::vector<int>& q = sp_poses[sp];
        for (int a : q) for (int b : q) if (a >= b) cov[(size_t)a*nfp + b] = 1;
      }
      for (int a = 0; a < nfp; ++a) for (int b = 0; b <= a; ++b) {
        if (!cov[(size_t)a*nfp + b]) continue;
        const int ra0 = (6*a)/CH_NB, ra1 = (6*a + 5)/CH_NB, rb0 = (6*b)/CH_NB, rb1 = (6*b + 5)/CH_NB;
        for (int ra = ra0; ra <= ra1; ++ra) for (int rb = rb0; rb <= rb1; ++rb) if (ra >= rb) pat[(size_t)ra*ntc + rb] = 1;
      }
    }
    for (int i = 0; i < ntc; ++i) pat[(size_t)i*ntc + i] = 1;
    if (multi()) {
      // the union of every rank's co-visibility pattern: agreed once, used for the factorisation plan and for the
      // packed all-reduce of the reduced system (only structurally non-zero tiles travel over xGMI)
      std::vector<double> pd(pat.begin(), pat.end());
      DevBuf<double> tmp;
      if (tmp.upload(pd, st)) return -1;
      if (allreduce(tmp.p, pd.size(), 0, true, "set-up: tile pattern")) return -1;
      HIPCK(hipMemcpy(pd.data(), tmp.p, pd.size()*sizeof(double), hipMemcpyDeviceToHost));
      for (size_t i = 0; i < pd.size(); ++i) pat[i] = pd[i] > 0;
      std::vector<int> rt;
      for (int i = 0; i < ntc; ++i) for (int j = 0; j <= i; ++j) if (pat[(size_t)i*ntc + j]) rt.push_back((i << 16) | j);
      n_red_tiles = (int)rt.size();
      if (d_red_tiles.upload(rt, st) || d_pack.alloc(MAX_SYS*((size_t)n_red_tiles*CH_NB*CH_NB + 2*(size_t)np))) return -1;
      pack_stride = (size_t)n_red_tiles*CH_NB*CH_NB + 2*(size_t)np;
    }
    lap("  covisibility");
    chol_segs.clear(); plan.persist_segs.clear();      // (the serial builder keeps the poses in add order: one chain)
    if (plan.build(np, pat)) { set_err("Cholesky plan allocation failed"); return -1; }
    lap("  symbolic plan");
  } else plan.all_tiles.clear();
  // local pose indices of slots and incidences
  std::vector<unsigned char> slot_lp(nslot + 1, 0), inc_lp(ninc + 1, 0);
  std::vector<unsigned char> inc_mixed(ninc + 1, 0);
  for (int i = 0; i < ninc; ++i) inc_mixed[i] = (inc_state[i] == 3);
  for (int gi = 0; gi < ngroup; ++gi) {
    const int* gp = &g_pose[(size_t)gi*GRP_LMAX];
    auto local = [&](int u) -> unsigned char { for (int k = 0; k < GRP_LMAX; ++k) if (gp[k] == u) return (unsigned char)k; return 0; };
    for (int sp = g_sp0[gi]; sp < g_sp0[gi + 1]; ++sp) {
      if (sp_big[sp]) continue;
      for (int s2 = slot_start[sp_m[sp]]; s2 < slot_start[sp_m[sp + 1]]; ++s2) slot_lp[s2] = local(slot_unk[s2]);
      for (int i2 = sp_i[sp]; i2 < sp_i[sp + 1]; ++i2) inc_lp[i2] = local(inc_unk[i2]);
    }
  }

  lap("  local indices");
  // ---- fixed-order assembly plan (ba_group.h): which local pose pairs every group stages, and for every global pose
  // pair / pose the list of staged slots in ascending group order
  std::vector<int> g_blk0(ngroup + 1, 0), pair_id((size_t)std::max(nfp, 1)*std::max(nfp, 1), -1), pr_start, blk_dst, po_start(nfp + 1, 0),
                   rhs_dst((size_t)std::max(ngroup, 1)*GRP_LMAX, -1);
  std::vector<unsigned char> blk_pair;
  {
    std::vector<std::array<int, 2>> blk_ab;                            // global pose pair (a >= b) of every staged block
    auto ltri = [](int r, int c) { return r*(r + 1)/2 + c; };
    for (int gi = 0; gi < ngroup; ++gi) {
      const int* gp = &g_pose[(size_t)gi*GRP_LMAX];
      unsigned char cov[GRP_LMAX*(GRP_LMAX + 1)/2] = {0};
      for (int sp = g_sp0[gi]; sp < g_sp0[gi + 1]; ++sp) {
        if (sp_big[sp]) continue;
        int loc[GRP_LMAX]; int nl = 0;
        for (int u : sp_poses[sp]) for (int k = 0; k < GRP_LMAX; ++k) if (gp[k] == u) { loc[nl++] = k; break; }
        for (int x = 0; x < nl; ++x) for (int y = 0; y < nl; ++y) if (loc[x] >= loc[y]) cov[ltri(loc[x], loc[y])] = 1;
      }
      g_blk0[gi] = (int)blk_pair.size();
      for (int la = 0; la < GRP_LMAX; ++la) for (int lb = 0; lb <= la; ++lb) if (cov[ltri(la, lb)]) {
        blk_pair.push_back((unsigned char)((la << 4) | lb));
        blk_ab.push_back({gp[la], gp[lb]});                            // g_pose is ascending: gp[la] >= gp[lb]
      }
      for (int k = 0; k < GRP_LMAX; ++k) if (gp[k] >= 0) po_start[gp[k] + 1]++;
    }
    g_blk0[ngroup] = (int)blk_pair.size();
    nstage = blk_pair.size();
    grp_blk_max = 0; grp_inc_max = 0;
    for (int gi = 0; gi < ngroup; ++gi) { grp_blk_max = std::max(grp_blk_max, g_blk0[gi + 1] - g_blk0[gi]); grp_inc_max = std::max(grp_inc_max, sp_i[g_sp0[gi + 1]] - sp_i[g_sp0[gi]]); }
    int npairs = 0;
    std::vector<int> cnt_pair;
    for (const auto& ab : blk_ab) {
      int& id = pair_id[(size_t)ab[0]*nfp + ab[1]];
      if (id < 0) { id = npairs++; cnt_pair.push_back(0); }
      cnt_pair[id]++;
    }
    pr_start.assign(npairs + 1, 0);
    for (int i = 0; i < npairs; ++i) pr_start[i + 1] = pr_start[i] + cnt_pair[i];
    // destination-ordered staging: the blocks of one pose pair are consecutive, in ascending group order (blocks are
    // numbered group by group, so walking them in order fills every pair's run in that order)
    blk_dst.assign(nstage, 0);
    { std::vector<int> pos(pr_start.begin(), pr_start.end() - 1);
      for (size_t k = 0; k < blk_ab.size(); ++k) blk_dst[k] = pos[pair_id[(size_t)blk_ab[k][0]*nfp + blk_ab[k][1]]]++; }
    for (int a = 0; a < nfp; ++a) po_start[a + 1] += po_start[a];
    nrhs_rows = po_start[nfp];
    { std::vector<int> pos(po_start.begin(), po_start.end() - 1);
      for (int gi = 0; gi < ngroup; ++gi) for (int k = 0; k < GRP_LMAX; ++k) { const int u = g_pose[(size_t)gi*GRP_LMAX + k]; if (u >= 0) rhs_dst[(size_t)gi*GRP_LMAX + k] = pos[u]++; } }
  }

  lap("pattern+plan");
  if (trace) fprintf(stderr, "[mcp_ba prepare] %d groups, %zu staged blocks (%.1f per group, at most %d)\n", ngroup, nstage, ngroup ? (double)nstage/ngroup : 0.0, grp_blk_max);
  std::unique_lock<std::mutex> arena_lock(pinned_arena().mutex());
  ArenaGuard arena_guard{st};
  HostStruct H;
#define CP(x) H.x.assign(x.begin(), x.end())
  CP(m_pt); CP(m_chain); CP(m_sp); CP(slot_start); CP(slot_unk); CP(slot_inc); CP(l_i0); CP(l_i1); CP(l_sp); CP(inc_unk); CP(sp_pt); CP(sp_m); CP(sp_i);
  CP(g_sp0); CP(g_pose); CP(g_blk0); CP(pair_id); CP(pr_start); CP(blk_dst); CP(po_start); CP(rhs_dst); CP(m_cam); CP(slot_first); CP(sp_big); CP(slot_lp);
  CP(inc_lp); CP(inc_mixed); CP(blk_pair); CP(m_mask); CP(m_u); CP(m_v); CP(m_om);
#undef CP
  return finish_prepare(H, t0, tlast, trace);
}

// allocation + upload of a finished host structure (common to both builders)
int mcp_ba::finish_prepare(HostStruct& H, std::chrono::steady_clock::time_point t0, std::chrono::steady_clock::time_point tlast, bool trace, const StructEntry* hit) {
  pending_entry.reset();          // (whatever an earlier, failed Prepare() of this handle left behind is not this call's to publish)
  auto lap = [&](const char* what) { if (!trace) return; auto n = std::chrono::steady_clock::now(); fprintf(stderr, "[mcp_ba prepare] %-22s %.3f ms\n", what, std::chrono::duration<double, std::milli>(n - tlast).count()); tlast = n; };
  const int npose = (int)poses.size(), npoint = (int)points.size();
  const int nmeas = hit ? hit->key.nmeas : (int)meas.size();          // (near miss: the device arrays keep the adopted structure's length)
  std::vector<int> chain_len(chains.size()), chain_pose(chains.size()*MAXC), pose_unk(npose), pt_chain(npoint), pt_unk(npoint);
  std::vector<unsigned char> pt_fixed(npoint);
  for (size_t c = 0; c < chains.size(); ++c) { chain_len[c] = chains[c].len; for (int i = 0; i < MAXC; ++i) chain_pose[c*MAXC + i] = chains[c].v[i]; }
  for (int i = 0; i < npose; ++i) pose_unk[i] = poses[i].unk;
  for (int i = 0; i < npoint; ++i) { pt_chain[i] = points[i].chain; pt_unk[i] = points[i].unk; pt_fixed[i] = (unsigned char)points[i].fixed; }
  // k_schur4 holds a group's local tile as 5 x 5 16-row tiles: every group must fit S4_LMAX poses
  sch4_ok = ngroup > 0;
  if (hit) sch4_ok = hit->sch4_ok;
  else for (int gi = 0; gi < ngroup && sch4_ok; ++gi) if (H.g_pose[(size_t)gi*GRP_LMAX + S4_LMAX] >= 0) sch4_ok = false;
  schur_mfma = schur_flops = 0;
  if (hit) { schur_mfma = hit->schur_mfma; schur_flops = hit->schur_flops; }
  else if ((prm.profile || cache_insert) && ngroup) {       // (a Prepare() that fills the structure cache computes it for whoever adopts the entry)
    // the work of one system's Schur launch, for the bench line: non-empty 16-row tiles of every 16-point chunk -> tile pairs x 12
    // instructions (k_schur4 skips empty tiles; k_schur_group multiplies every tile of the group's poses), and the structural flops
    // (ranges of groups on the worker pool, partial sums added in thread order: integers in doubles, any order gives the same value)
    HostPool& pool = host_pool();
    const int T = (nmeas >= 32768) ? pool.size() : 1;
    std::vector<double> pm(T, 0.0), pf(T, 0.0);
    auto body = [&](int tid) {
      double sm = 0, sf = 0;
      for (int gi = (int)((long)ngroup*tid/T), ge = (int)((long)ngroup*(tid + 1)/T); gi < ge; ++gi) {
        int npl = 0; for (int k = 0; k < GRP_LMAX; ++k) if (H.g_pose[(size_t)gi*GRP_LMAX + k] >= 0) npl = k + 1;
        const int ntile = (6*npl + 15)/16;
        for (int s0 = H.g_sp0[gi]; s0 < H.g_sp0[gi + 1]; s0 += 16) {
          unsigned tm = 0;
          for (int sp = s0; sp < std::min(s0 + 16, H.g_sp0[gi + 1]); ++sp) {
            if (H.sp_big[sp]) continue;
            const int k = H.sp_i[sp + 1] - H.sp_i[sp];
            sf += 324.0*k*(k + 1)/2;
            for (int i = H.sp_i[sp]; i < H.sp_i[sp + 1]; ++i) { tm |= 1u << ((6*H.inc_lp[i]) >> 4); tm |= 1u << ((6*H.inc_lp[i] + 5) >> 4); }
          }
          const int nt = (sch4_on && sch4_ok) ? __builtin_popcount(tm) : (tm ? ntile : 0);
          sm += 12.0*nt*(nt + 1)/2;
        }
      }
      pm[tid] = sm; pf[tid] = sf;
    };
    if (T == 1) body(0); else pool.run(body);
    for (int t2 = 0; t2 < T; ++t2) { schur_mfma += pm[t2]; schur_flops += pf[t2]; }
  }
  if (hit) asm_long = hit->asm_long;
  else {
    // long lists of staged blocks per pose pair (mean >= 48: a handful of free poses that every group stages) go through k_assemble_long
    const size_t npairs = H.pr_start.empty() ? 0 : H.pr_start.size() - 1;
    asm_long = npairs > 0 && nstage >= 48*npairs;
    if (const char* e = getenv("MCP_BA_ASM_LONG")) asm_long = atoi(e) != 0 && npairs > 0;
  }
  // The ~40 structure arrays go into ONE device block (d_struct; every d_* below is an alias into it): one allocation, and one
  // host-to-device copy for all the small arrays together (staged through the pinned arena) -- the large per-measurement arrays,
  // which the builders wrote into the pinned arena already, are copied from where they lie.
  {
    struct Item { const void* src; size_t bytes, off; bool direct; };
    std::vector<Item> items; items.reserve(48);
    size_t total = 0, staged = 0;
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    // (a cache hit brings no host arrays: the layout is replayed from the element counts the cold Prepare() recorded, in the same order)
    std::vector<size_t> counts; counts.reserve(48);
    size_t nadd = 0;
    auto add = [&](auto& dbuf, const auto& vec, bool pinned) {
      using E = typename std::remove_reference<decltype(vec)>::type::value_type;
      const size_t cnt = hit ? hit->counts[nadd] : vec.size();
      ++nadd;
      const size_t bytes = cnt*sizeof(E);
      dbuf.release();
      dbuf.alias = true; dbuf.n = std::max<size_t>(cnt, 1); dbuf.p = reinterpret_cast<decltype(dbuf.p)>(total);      // (offset for now)
      const bool direct = pinned && bytes >= 16384;
      counts.push_back(cnt);
      items.push_back({hit ? nullptr : (const void*)vec.data(), bytes, total, direct});
      if (!direct) staged += al(std::max<size_t>(bytes, 1));
      total += al(std::max<size_t>(bytes, 1));
    };
    add(d_cams, cams, false); add(d_chain_len, chain_len, false); add(d_chain_pose, chain_pose, false); add(d_pose_unk, pose_unk, false);
    add(d_pt_chain, pt_chain, false); add(d_pt_unk, pt_unk, false); add(d_pt_fixed, pt_fixed, false);
    add(d_m_pt, H.m_pt, true); add(d_m_chain, H.m_chain, true); add(d_m_cam, H.m_cam, true); add(d_m_mask, H.m_mask, true);
    add(d_m_u, H.m_u, true); add(d_m_v, H.m_v, true); add(d_m_omega, H.m_om, true); add(d_slot_start, H.slot_start, true);
    add(d_slot_unk, H.slot_unk, true); add(d_slot_inc, H.slot_inc, true); add(d_l_i0, H.l_i0, true); add(d_l_i1, H.l_i1, true);
    add(d_inc_unk, H.inc_unk, true); add(d_fl_point, fl_point, false); add(d_sp_pt, H.sp_pt, true); add(d_sp_m, H.sp_m, true);
    add(d_sp_i, H.sp_i, true); add(d_sp_big, H.sp_big, true); add(d_m_sp, H.m_sp, true); add(d_l_sp, H.l_sp, true);
    add(d_g_sp0, H.g_sp0, true); add(d_g_pose, H.g_pose, true); add(d_slot_lp, H.slot_lp, true); add(d_slot_first, H.slot_first, true);
    add(d_inc_lp, H.inc_lp, true); add(d_inc_mixed, H.inc_mixed, true); add(d_g_blk0, H.g_blk0, true); add(d_blk_pair, H.blk_pair, true);
    add(d_asm_tiles, plan.all_tiles, false); add(d_pair_id, H.pair_id, true); add(d_pr_start, H.pr_start, true); add(d_blk_dst, H.blk_dst, true);
    add(d_po_start, H.po_start, true); add(d_rhs_dst, H.rhs_dst, true);
    // k_schur4: free-point index per sorted point (one index hop less at the head of every workgroup), and -- when the groups
    // outnumber the workgroup slots of the device (2 per compute unit) -- the order in which they are launched: heaviest first
    // (tile pairs x chunks), so that the second round of workgroups is made of the light ones and the launch ends evenly
    std::vector<int> sp_unk(hit ? 0 : nsp), g_order;
    if (!hit) for (int sp = 0; sp < nsp; ++sp) sp_unk[sp] = H.sp_big[sp] ? -1 : pt_unk[H.sp_pt[sp]];
    sch4_order = hit ? hit->sch4_order : (sch4_on && sch4_ok && ngroup > 512 && env_on("MCP_BA_SCHUR4_ORDER", true));
    if (sch4_order && !hit) {
      std::vector<int> cost(ngroup);
      for (int gi = 0; gi < ngroup; ++gi) {
        int npl = 0; for (int k = 0; k < GRP_LMAX; ++k) if (H.g_pose[(size_t)gi*GRP_LMAX + k] >= 0) npl = k + 1;
        const int nt = (6*npl + 15)/16;
        cost[gi] = nt*(nt + 1)/2*((H.g_sp0[gi + 1] - H.g_sp0[gi] + 15)/16);
      }
      g_order.resize(ngroup);
      for (int gi = 0; gi < ngroup; ++gi) g_order[gi] = gi;
      std::stable_sort(g_order.begin(), g_order.end(), [&](int a, int b) { return cost[a] > cost[b]; });
    }
    std::vector<int, NoInitAlloc<int>> m_last(hit ? 0 : nmeas);
    if (!hit) {
      HostPool& pool = host_pool();
      const int T = (nmeas >= 32768) ? pool.size() : 1;
      auto body = [&](int tid) { for (long j = (long)nmeas*tid/T, e = (long)nmeas*(tid + 1)/T; j < e; ++j) { const int oc = H.m_chain[j]; m_last[j] = oc*MAXC + chains[oc].len - 1; } };
      if (T == 1) body(0); else pool.run(body);
    }
    add(d_sp_unk, sp_unk, false); add(d_g_order, g_order, false); add(d_m_last, m_last, false);
    if (d_struct.alloc(total)) return -1;
    char* const base = d_struct.p;
    auto fix = [&](auto& dbuf) { dbuf.p = reinterpret_cast<decltype(dbuf.p)>(base + reinterpret_cast<size_t>(dbuf.p)); };
    fix(d_cams); fix(d_chain_len); fix(d_chain_pose); fix(d_pose_unk); fix(d_pt_chain); fix(d_pt_unk); fix(d_pt_fixed);
    fix(d_m_pt); fix(d_m_chain); fix(d_m_cam); fix(d_m_mask); fix(d_m_u); fix(d_m_v); fix(d_m_omega); fix(d_slot_start);
    fix(d_slot_unk); fix(d_slot_inc); fix(d_l_i0); fix(d_l_i1); fix(d_inc_unk); fix(d_fl_point); fix(d_sp_pt); fix(d_sp_m);
    fix(d_sp_i); fix(d_sp_big); fix(d_m_sp); fix(d_l_sp); fix(d_g_sp0); fix(d_g_pose); fix(d_slot_lp); fix(d_slot_first);
    fix(d_inc_lp); fix(d_inc_mixed); fix(d_g_blk0); fix(d_blk_pair); fix(d_asm_tiles); fix(d_pair_id); fix(d_pr_start); fix(d_blk_dst);
    fix(d_po_start); fix(d_rhs_dst); fix(d_sp_unk); fix(d_g_order); fix(d_m_last);
    if (hit) {
      // the cached clone of the block, then this handle's own numbers over it: the cameras and the measurement values (u, v, the
      // weight), gathered into sorted order through the permutation the structure build left
      if (hit->dbytes != total) { set_err("structure cache: layout mismatch"); return -1; }
      HIPCK(hipMemcpyAsync(base, hit->dblock, total, hipMemcpyDeviceToDevice, st));
      const size_t nm = (size_t)nmeas;
      double* vals = (double*)pinned_arena().alloc((3*nm + 1)*sizeof(double));
      {
        HostPool& pool = host_pool();
        const int T = (nmeas >= 32768) ? pool.size() : 1;
        const int* pm = perm.data(); const HMeas* ms = meas.data();
        auto body = [&](int tid) {
          for (size_t j = nm*tid/T, e = nm*(tid + 1)/T; j < e; ++j) {
            if (pm[j] < 0) { vals[j] = 0.0; vals[nm + j] = 0.0; vals[2*nm + j] = 0.0; continue; }      // (near miss: a measurement this map does not have -- weight 0)
            const HMeas& m = ms[pm[j]]; vals[j] = m.u; vals[nm + j] = m.v; vals[2*nm + j] = m.omega;
          }
        };
        if (T == 1) body(0); else pool.run(body);
      }
      mcp_camera* cs = cams.empty() ? nullptr : (mcp_camera*)pinned_arena().alloc(cams.size()*sizeof(mcp_camera));
      if (cs) std::memcpy(cs, cams.data(), cams.size()*sizeof(mcp_camera));
      static_assert(sizeof(mcp_camera) % 8 == 0, "k_upload_set copies 8-byte words");
      if (3*nm*sizeof(double) <= UPSET_MAX_BYTES && pinned_arena().is_pinned(vals) && (!cs || pinned_arena().is_pinned(cs))) {
        UploadSet V; V.n = 0;             // a small bundle: the kernel reads the pinned values in place
        if (nm) { upload_set_add(V, vals, d_m_u.p, nm*sizeof(double)); upload_set_add(V, vals + nm, d_m_v.p, nm*sizeof(double)); upload_set_add(V, vals + 2*nm, d_m_omega.p, nm*sizeof(double)); }
        if (cs) upload_set_add(V, cs, d_cams.p, cams.size()*sizeof(mcp_camera));
        upload_set_launch(V, st); note_launch("k_upload_set");
      } else {
        if (nm) {
          HIPCK(hipMemcpyAsync(d_m_u.p, vals, nm*sizeof(double), hipMemcpyHostToDevice, st));
          HIPCK(hipMemcpyAsync(d_m_v.p, vals + nm, nm*sizeof(double), hipMemcpyHostToDevice, st));
          HIPCK(hipMemcpyAsync(d_m_omega.p, vals + 2*nm, nm*sizeof(double), hipMemcpyHostToDevice, st));
        }
        if (cs) HIPCK(hipMemcpyAsync(d_cams.p, cs, cams.size()*sizeof(mcp_camera), hipMemcpyHostToDevice, st));
      }
    }
    // the small ones: contiguous runs of staged items are contiguous in the device block too -- one copy per run
    char* stage = (staged && !hit) ? (char*)pinned_arena().alloc(staged) : nullptr;
    size_t so = 0;
    for (size_t i = 0; i < items.size() && !hit; ) {
      if (items[i].direct) { if (items[i].bytes) HIPCK(hipMemcpyAsync(base + items[i].off, items[i].src, items[i].bytes, hipMemcpyHostToDevice, st)); ++i; continue; }
      const size_t run_off = items[i].off, run_so = so;
      size_t run_bytes = 0;
      for (; i < items.size() && !items[i].direct; ++i) {
        if (items[i].bytes) std::memcpy(stage + so, items[i].src, items[i].bytes);
        const size_t a = al(std::max<size_t>(items[i].bytes, 1)); so += a; run_bytes += a;
      }
      HIPCK(hipMemcpyAsync(base + run_off, stage + run_so, run_bytes, hipMemcpyHostToDevice, st));
    }
    lap("  upload enqueued");
    if (!hit && cache_insert) {
      // leave the results in the structure cache: host-side members + a device clone of the block (behind the uploads, on this stream;
      // whoever adopts it waits for nothing: an entry becomes visible only after this Prepare()'s final synchronisation, below)
      auto e = std::make_shared<StructEntry>();
      e->key = cache_key; e->base = cache_base;
      {   // (1.6 MB each at the metric size, the cameras through the 40-byte measurement records: 0.45 ms on one thread)
        e->add_point.resize(nmeas); e->add_chain.resize(nmeas); e->add_cam.resize(nmeas);
        HostPool& pool = host_pool();
        const int T = (nmeas >= 32768) ? pool.size() : 1;
        auto body = [&](int tid) {
          const long j0 = (long)nmeas*tid/T, j1 = (long)nmeas*(tid + 1)/T;
          if (j1 > j0) { std::memcpy(e->add_point.data() + j0, meas_point.data() + j0, (size_t)(j1 - j0)*sizeof(int)); std::memcpy(e->add_chain.data() + j0, meas_chain.data() + j0, (size_t)(j1 - j0)*sizeof(int)); }
          for (long j = j0; j < j1; ++j) e->add_cam[j] = meas[j].cam;
        };
        if (T == 1) body(0); else pool.run(body);
      }
      e->pose_unk.resize(npose); e->pose_active.resize(npose); e->pt_unk.resize(npoint); e->pt_active.resize(npoint);
      for (int i = 0; i < npose; ++i) { e->pose_unk[i] = poses[i].unk; e->pose_active[i] = (unsigned char)poses[i].active; }
      for (int i = 0; i < npoint; ++i) { e->pt_unk[i] = points[i].unk; e->pt_active[i] = (unsigned char)points[i].active; }
      e->fp_pose = fp_pose; e->fl_point = fl_point; e->perm = perm; e->pat = last_pat; e->chol_segs = chol_segs;
      e->nfp = nfp; e->nfl = nfl; e->np = np; e->nx = nx; e->nsp = nsp; e->ninc = ninc; e->nslot = nslot; e->ngroup = ngroup; e->nbig = nbig;
      e->grp_pts = grp_pts; e->grp_blk_max = grp_blk_max; e->grp_inc_max = grp_inc_max; e->nrhs_rows = nrhs_rows; e->nstage = nstage;
      e->asm_long = asm_long; e->sch4_ok = sch4_ok; e->sch4_order = sch4_order; e->m_total = m_total; e->nfl_total = nfl_total;
      e->schur_mfma = schur_mfma; e->schur_flops = schur_flops; e->counts = counts;
      e->dblock = (char*)DevCache::get().take(std::max<size_t>(total, 1), &e->dcap, &e->ddev);
      if (e->dblock) {
        e->dbytes = total;
        HIPCK(hipMemcpyAsync(e->dblock, base, total, hipMemcpyDeviceToDevice, st));
        pending_entry = e;
      }
      lap("  cache entry");
    }
  }
  const size_t nc = chains.size();
  for (int b = 0; b < NSTATE; ++b)
    if (d_pose[b].alloc((size_t)npose*12) || d_pt[b].alloc((size_t)npoint*3) || d_first[b].alloc(nc*MAXC*12) ||
        d_second[b].alloc(nc*MAXC*9) || d_last[b].alloc(nc*12) || d_chi2[b].alloc(nmeas)) return -1;
  const size_t n2 = (size_t)np*np;
  const int nblk = (std::max(nmeas, nfl) + 255)/256 + 1;
  red_stride = n2 + 2*(size_t)np; vinv_stride = (size_t)nfl*6; spec_ok = false; sys_cur = 0;
  { const char* e = getenv("MCP_BA_SPECULATE"); if (e) speculate = atoi(e); }
  { const char* e = getenv("MCP_BA_SPECULATE_ADAPT"); if (e) spec_adapt = atoi(e); }
  { const char* e = getenv("MCP_BA_LIN_JOIN"); if (e) lin_join_full = atoi(e); }
  { const char* e = getenv("MCP_BA_TRIAL_FUSE"); if (e) trial_fuse = atoi(e); }
  { const char* e = getenv("MCP_BA_HEAD_LARGE"); if (e) head_large_on = atoi(e); }
  { const char* e = getenv("MCP_BA_GRAPH"); if (e) use_graph = atoi(e); }
  { const char* e = getenv("MCP_BA_SELECT_CAP"); if (e) sel_cap = std::max(1, atoi(e)); }
  if (multi()) {
    if (d_seltab.alloc((size_t)world*sel_cap + world + 2)) return -1;
    for (int q = 0; q < MAX_SYS; ++q) if (d_trial[q].alloc(TRIAL_LEN) || d_trstate[q].alloc(SEL_PASSES + 1)) return -1;
  }
  sel_src = -1; pred_bin = -1;
  for (int q = 0; q <= MAX_SYS; ++q) if (chol_exec[q]) { (void)hipGraphExecDestroy(chol_exec[q]); chol_exec[q] = nullptr; }      // plan and buffers may have changed
  for (int q = 0; q <= MAX_SYS; ++q) for (int r = 0; r < MAX_SYS; ++r) if (chain_exec[q][r]) { (void)hipGraphExecDestroy(chain_exec[q][r]); chain_exec[q][r] = nullptr; }
  hist_clean = false;                              // (the histogram block may be a fresh one)
  if ((nbig && d_ubig.alloc(n2 + np)) || d_stU.alloc(nstage*36) || d_stb.alloc((size_t)nrhs_rows*6) || d_stS.alloc(MAX_SYS*nstage*36) ||
      d_str.alloc(MAX_SYS*(size_t)nrhs_rows*6) || d_udiag.alloc(np) || d_red.alloc(MAX_SYS*(n2 + 2*(size_t)np)) || d_V.alloc((size_t)nfl*6) || d_g.alloc((size_t)nfl*3) ||
      d_W.alloc((size_t)ninc*18) || d_Vinv.alloc(MAX_SYS*(size_t)nfl*6) || d_xl.alloc((size_t)nfl*3) ||
      d_xp_good.alloc(np) || d_xp_cand.alloc(np) || d_selvals.alloc(SEL_GATHER_CAP) || d_xl_good.alloc((size_t)nfl*3) || d_part0.alloc(nblk) || d_part1.alloc(nblk) ||
      d_part2.alloc(nblk) || d_parth.alloc(nblk) || d_res.alloc(32 + 8*MAX_SYS) || d_sigma.alloc(8*N_SIG) || d_hist.alloc((size_t)SEL_PASSES*SEL_BINS) ||
      d_selstate.alloc(SEL_PASSES + 1) || d_fail.alloc(4) || d_flags.alloc(nmeas) || d_cov.alloc(nfl)) return -1;
  if (head_large_on && robust && !multi() && nmeas > 0) {
    if (d_headl.alloc(sizeof(HeadLScratch))) return -1;
    headl_clean = false;                           // (the block may be a fresh one: zeroed in front of the first head)
    if (!hl_grid) {
      int nb = 0, ncu = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)k_head_large, HL_THREADS, 0) == hipSuccess &&
          hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess) hl_grid = std::min(HL_GRID, std::max(0, nb*ncu/2));      // (co-resident with room to spare)
      (void)hipGetLastError();
    }
  } else d_headl.release();
  {
    // scratch of the per-trial iteration heads (ba_head.h): per slot [parity][q] the digit histograms + counters, the candidates,
    // the robust chi2's partial sums, the median and the iteration-start block
    for (auto& H : hsc) H = HeadScratch();
    if (large_head_ahead && robust && !multi() && !small_mode_for(nmeas, (int)nc) && nmeas > 0) {
      auto al = [](size_t x) { return (x + 31) & ~(size_t)31; };
      const size_t n_hist = al((HEAD_HIST + HEAD_CTL + 1)/2), n_vals = al(HEAD_CAND), n_part = al(std::max<size_t>(nblk, 1)), per = n_hist + n_vals + n_part + 32;
      if (d_headbuf.alloc(per*2*MAX_SYS)) return -1;
      for (int i = 0; i < 2*MAX_SYS; ++i) {
        double* b = d_headbuf.p + per*i;
        hsc[i].hist = reinterpret_cast<unsigned int*>(b); hsc[i].vals = b + n_hist; hsc[i].part = hsc[i].vals + n_vals;
        hsc[i].out = hsc[i].part + n_part; hsc[i].rs = hsc[i].out + 8;
        HIPCK(hipMemsetAsync(hsc[i].hist, 0, (HEAD_HIST + HEAD_CTL)*sizeof(unsigned int), st));       // (every head leaves them zero again)
      }
    } else d_headbuf.release();
    for (int q = 0; q < MAX_SYS; ++q) { head_enq[q] = false; head_state[q] = -1; }
    acc_head_q = -1; start_blk = nullptr;
  }
  if (const char* e = getenv("MCP_BA_DEBUG_POISON_RED")) {      // test aid: "q,lo,hi" -- the reduced-system buffer zeroed, doubles [lo, hi) of system q set to NaN
    int q = 0; long lo = 0, hi = 0;
    if (std::sscanf(e, "%d,%ld,%ld", &q, &lo, &hi) == 3 && q >= 0 && q < MAX_SYS && lo >= 0 && hi <= (long)(n2 + 2*(size_t)np) && lo < hi) {
      HIPCK(hipMemset(d_red.p, 0, MAX_SYS*(n2 + 2*(size_t)np)*sizeof(double)));
      HIPCK(hipMemset(d_red.p + q*(n2 + 2*(size_t)np) + lo, 0xFF, (size_t)(hi - lo)*sizeof(double)));
    }
  }
  // pinned, device-visible: [0..31] read-back block, [32 + 32 q ..] mailbox of trial q (ticket at + MAIL_TICKET)
  if (!h_res) {
    h_res = (double*)mcp::PinnedCache::get().take(HRES*sizeof(double), hipHostMallocMapped | hipHostMallocCoherent);
    if (!h_res) { set_err("hipHostMalloc failed"); return -1; }
    std::memset(h_res, 0, HRES*sizeof(double));
    void* dp = nullptr; HIPCK(hipHostGetDevicePointer(&dp, h_res, 0)); h_mail_dev = (double*)dp + 32;
  }
  for (int q = 1; q < MAX_SYS; ++q)
    if (d_sxp[q].alloc(np) || d_sxl[q].alloc((size_t)nfl*3) || d_sp0[q].alloc(nblk) || d_sp1[q].alloc(nblk) || d_sp2[q].alloc(nblk)) return -1;
  if (!h_fail) { h_fail = (int*)mcp::PinnedCache::get().take(4*sizeof(int), hipHostMallocDefault); if (!h_fail) { set_err("hipHostMalloc failed"); return -1; } }
  if (((size_t)np + (size_t)nfl*3)*sizeof(double) <= UPSET_MAX_BYTES) {      // x = 0 before the first solve, no sigma block yet: one launch
    UploadSet Z; Z.n = 0;
    upload_set_add(Z, nullptr, d_xp_good.p, std::max<size_t>(np, 1)*sizeof(double));
    upload_set_add(Z, nullptr, d_xl_good.p, std::max<size_t>((size_t)nfl*3, 1)*sizeof(double));
    upload_set_add(Z, nullptr, d_sigma.p, 16*sizeof(double));
    upload_set_launch(Z, st); note_launch("k_upload_set");
  } else {
    HIPCK(hipMemsetAsync(d_xp_good.p, 0, std::max<size_t>(np, 1)*sizeof(double), st));
    HIPCK(hipMemsetAsync(d_xl_good.p, 0, std::max<size_t>((size_t)nfl*3, 1)*sizeof(double), st));
    HIPCK(hipMemsetAsync(d_sigma.p, 0, 16*sizeof(double), st));
  }
  sig_idx = 0;

  P.cams = d_cams.p; P.ncam = (int)cams.size(); P.nchain = (int)nc; P.chain_len = d_chain_len.p; P.chain_pose = d_chain_pose.p;
  P.npose = npose; P.pose_unk = d_pose_unk.p; P.npoint = npoint; P.pt_chain = d_pt_chain.p; P.pt_unk = d_pt_unk.p;
  P.pt_fixed = d_pt_fixed.p; P.nmeas = nmeas; P.m_pt = d_m_pt.p; P.m_chain = d_m_chain.p; P.m_cam = d_m_cam.p;
  P.m_mask = d_m_mask.p; P.m_u = d_m_u.p; P.m_v = d_m_v.p; P.m_omega = d_m_omega.p; P.slot_start = d_slot_start.p;
  P.slot_unk = d_slot_unk.p; P.slot_inc = d_slot_inc.p; P.nfl = nfl; P.ninc = ninc; P.np = np;
  P.l_i0 = d_l_i0.p; P.l_i1 = d_l_i1.p; P.inc_unk = d_inc_unk.p; P.fl_point = d_fl_point.p; P.robust = robust;
  P.nsp = nsp; P.ngroup = ngroup; P.sp_pt = d_sp_pt.p; P.sp_m = d_sp_m.p; P.sp_i = d_sp_i.p; P.sp_big = d_sp_big.p;
  P.m_sp = d_m_sp.p; P.l_sp = d_l_sp.p; P.g_sp0 = d_g_sp0.p; P.g_pose = d_g_pose.p; P.slot_lp = d_slot_lp.p;
  P.slot_first = d_slot_first.p; P.inc_lp = d_inc_lp.p; P.inc_mixed = d_inc_mixed.p;
  P.g_blk0 = d_g_blk0.p; P.blk_pair = d_blk_pair.p; P.blk_dst = d_blk_dst.p; P.rhs_dst = d_rhs_dst.p; P.sp_unk = d_sp_unk.p; P.m_last = d_m_last.p;
  A.nfp = nfp; A.ntiles = (int)plan.all_tiles.size(); A.tiles = d_asm_tiles.p; A.pair_id = d_pair_id.p;
  A.pr_start = d_pr_start.p; A.po_start = d_po_start.p;
  // the staging arrays of a group that stages nothing for a slot are never read; slots are always fully written
  // before k_assemble runs, so they need no clearing either
  {   // (function attributes are per device: once, not once per Prepare())
    static std::atomic<unsigned long long> sch_attr{0};
    const unsigned long long bit = 1ull << (device & 63);
    if (!(sch_attr.load(std::memory_order_relaxed) & bit)) {
      HIPCK(hipFuncSetAttribute((const void*)k_schur_group, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SCH_LDS_BYTES));
      sch_attr.fetch_or(bit, std::memory_order_relaxed);
    }
  }
  lap("alloc+upload");
  if (upload_state()) { pending_entry.reset(); return -1; }
  {
    const hipError_t es = hipStreamSynchronize(st);
    //  an entry enters the process-wide structure cache only when everything that fills its device clone has COMPLETED;
    // on any failure it is dropped here, not left for the handle's next successful Prepare() to publish under the old key
    if (es != hipSuccess || !launch_err.empty()) {
      pending_entry.reset();
      if (es != hipSuccess) { set_err(std::string("Prepare: ") + hipGetErrorString(es)); return -1; }
      set_err("Prepare: a launch was refused: " + launch_err); launch_err.clear(); return -1;
    }
  }
  if (pending_entry) { StructCache::get().insert(pending_entry); pending_entry.reset(); }
  lap("state");
  dirty = false;
  timing.structure_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  return 0;
}

// poses and points of the host state into every state buffer (current + one per trial candidate): fixed poses and points are never
// written by a trial, so every buffer has to hold them.  One staged copy through the pinned arena (the caller, prepare(), holds it
// until its final stream synchronisation) and one kernel that fans it out -- ten pageable hipMemcpyAsync were 0.2 ms of a window's set-up.
struct StateFan { double* pose[MAX_SYS + 1]; double* pt[MAX_SYS + 1]; int first; };
__global__ void k_fan_state(size_t npose_d, size_t npt_d, const double* __restrict__ pose0, const double* __restrict__ pt0, StateFan f) {
  const size_t i = blockIdx.x*(size_t)blockDim.x + threadIdx.x;
  if (i < npose_d) { const double v = pose0[i];
#pragma unroll
    for (int b = 0; b <= MAX_SYS; ++b) if (b >= f.first) f.pose[b][i] = v; }
  if (i < npt_d) { const double v = pt0[i];
#pragma unroll
    for (int b = 0; b <= MAX_SYS; ++b) if (b >= f.first) f.pt[b][i] = v; }
}
int mcp_ba::upload_state() {
  const size_t nps = poses.size()*12, npt = points.size()*3;
  double* stage = (double*)pinned_arena().alloc((nps + npt + 1)*sizeof(double));
  for (size_t i = 0; i < poses.size(); ++i) std::memcpy(stage + i*12, poses[i].T, 96);
  for (size_t i = 0; i < points.size(); ++i) std::memcpy(stage + nps + i*3, points[i].x, 24);
  cur = 0;
  // a small state is read by the fan-out kernel straight from the pinned stage (no copy-engine operation); a large one is copied first
  const bool in_place = (nps + npt)*8 <= UPSET_MAX_BYTES && pinned_arena().is_pinned(stage);
  if (!in_place) {
    if (nps) HIPCK(hipMemcpyAsync(d_pose[0].p, stage, nps*8, hipMemcpyHostToDevice, st));
    if (npt) HIPCK(hipMemcpyAsync(d_pt[0].p, stage + nps, npt*8, hipMemcpyHostToDevice, st));
  }
  StateFan f; f.first = in_place ? 0 : 1;
  for (int b = 0; b <= MAX_SYS; ++b) { f.pose[b] = d_pose[b].p; f.pt[b] = d_pt[b].p; }
  const size_t nmax = std::max(nps, npt);
  if (nmax) hipLaunchKernelGGL(k_fan_state, dim3((unsigned)((nmax + 255)/256)), dim3(256), 0, st, nps, npt, in_place ? (const double*)stage : (const double*)d_pose[0].p,
                               in_place ? (const double*)(stage + nps) : (const double*)d_pt[0].p, f);
  if (nmax) note_launch("k_fan_state");                // (reads the pinned stage: a refused launch would leave the device state unset and nothing else would say so)
  return 0;
}
int mcp_ba::download_state() {
  if (exported) {                                  // final_stats() had the state written to pinned host memory, and has waited for it
    exported = false;
    const double* ps = reinterpret_cast<const double*>(h_exp); const double* pt = ps + poses.size()*12;
    for (size_t i = 0; i < poses.size(); ++i) std::memcpy(poses[i].T, ps + i*12, 96);
    for (size_t i = 0; i < points.size(); ++i) std::memcpy(points[i].x, pt + i*3, 24);
    return 0;
  }
  std::vector<double> pt((size_t)points.size()*3), ps((size_t)poses.size()*12);
  if (!ps.empty()) HIPCK(hipMemcpyAsync(ps.data(), d_pose[cur].p, ps.size()*8, hipMemcpyDeviceToHost, st));
  if (!pt.empty()) HIPCK(hipMemcpyAsync(pt.data(), d_pt[cur].p, pt.size()*8, hipMemcpyDeviceToHost, st));
  HIPCK(hipStreamSynchronize(st));
  for (size_t i = 0; i < poses.size(); ++i) std::memcpy(poses[i].T, &ps[i*12], 96);
  for (size_t i = 0; i < points.size(); ++i) std::memcpy(points[i].x, &pt[i*3], 24);
  return 0;
}

void mcp_ba::launch_chains(int w) {
  const int nc = P.nchain;
  if (nc == 0) return;
  hipLaunchKernelGGL(k_chains, dim3((nc + 63)/64), dim3(64), 0, st, P, d_pose[w].p, d_first[w].p, d_second[w].p, d_last[w].p);
}
void mcp_ba::launch_eval(int w, bool sum, double* err_out) {
  const int nb = (P.nmeas + EVAL_BLOCK - 1)/EVAL_BLOCK;
  if (nb == 0) return;
  if (sum) hipLaunchKernelGGL((k_eval<true>), dim3(nb), dim3(EVAL_BLOCK), 0, st, P, d_pt[w].p, d_last[w].p, d_chi2[w].p, err_out, sig(), d_part0.p);
  else hipLaunchKernelGGL((k_eval<false>), dim3(nb), dim3(EVAL_BLOCK), 0, st, P, d_pt[w].p, d_last[w].p, d_chi2[w].p, err_out, sig(), d_part0.p);
}

// exact k-th smallest |x| (global over ranks when a hook is installed); result left at out_dev[0]
int mcp_ba::select_kth(const double* x, int n, unsigned long long k, double* out_dev, bool huber_sigma) {
  static const int grid_max = [] { const char* e = getenv("MCP_BA_SELECT_GRID"); return e ? std::max(1, atoi(e)) : 1024; }();
  const int grid = std::max(1, std::min(grid_max, (n + SEL_BLOCK*4 - 1)/(SEL_BLOCK*4)));
  if (!multi()) {
    // single GPU: two full histogram passes, gather, one-workgroup finish (which also writes the Huber sigma block if asked)
    if (!hist_clean) HIPCK(hipMemsetAsync(d_hist.p, 0, (size_t)(2*SEL_BINS + 1)*sizeof(double), st));        // two histograms + the gather counter behind them
    unsigned int* cnt = reinterpret_cast<unsigned int*>(d_hist.p + 2*SEL_BINS);
    for (int p = 0; p < 2; ++p) hipLaunchKernelGGL(k_select_pass, dim3(grid), dim3(SEL_BLOCK), 0, st, p, n, x, d_hist.p, d_selstate.p, k);
    hipLaunchKernelGGL(k_select_gather, dim3(grid), dim3(SEL_BLOCK), 0, st, n, x, (const double*)d_hist.p, d_selstate.p, cnt, d_selvals.p);
    hipLaunchKernelGGL(k_select_small, dim3(1), dim3(1024), 0, st, n, x, (const unsigned int*)cnt, (const double*)d_selvals.p, (const SelState*)d_selstate.p,
                       m_total, prm.min_mestimator_sigma*prm.min_mestimator_sigma, out_dev, huber_sigma ? sig() : (double*)nullptr,
                       huber_sigma ? d_res.p + 25 : (double*)nullptr, (const double*)nullptr, 0, 0, d_hist.p, 2*SEL_BINS + 1);
    hist_clean = true;
    return 0;
  }
  // several ranks: two all-reduced histogram passes, then the few candidates of every rank are gathered through a
  // zero-filled (ranks x sel_cap) table summed over the ranks, and each rank finishes locally: 3 collectives
  hist_clean = false;
  HIPCK(hipMemsetAsync(d_hist.p, 0, (size_t)SEL_PASSES*SEL_BINS*sizeof(double), st));
  for (int p = 0; p < 2; ++p) {
    hipLaunchKernelGGL(k_select_pass, dim3(grid), dim3(SEL_BLOCK), 0, st, p, n, x, d_hist.p, d_selstate.p, k);
    if (allreduce(d_hist.p + (size_t)p*SEL_BINS, SEL_BINS, 0, false, "median: digit histogram")) return -1;
  }
  // the flag is the same on every rank (it is derived from the all-reduced histogram), so all ranks take the same branch
  if (select_gather_finish(x, n, d_hist.p, d_selstate.p, out_dev, true)) return -1;
  if (h_res[30] == 0.0) return 0;
  // more candidates than the table holds (tens of thousands of values equal in their top 22 bits): the remaining digits by histogram
  for (int p = 2; p < SEL_PASSES; ++p) {
    hipLaunchKernelGGL(k_select_pass, dim3(grid), dim3(SEL_BLOCK), 0, st, p, n, x, d_hist.p, d_selstate.p, k);
    if (allreduce(d_hist.p + (size_t)p*SEL_BINS, SEL_BINS, 0, false, "median: digit histogram (overflow path)")) return -1;
  }
  hipLaunchKernelGGL(k_select_final, dim3(1), dim3(SEL_BLOCK), 0, st, d_hist.p, d_selstate.p, out_dev);
  return 0;
}
// multi-rank: with the first two digits fixed (hist pass 1 + state[1] given), gather every rank's candidates through the slot table
// (one all-reduce) and resolve the remaining digits locally.  check_overflow: read the overflow flag back (h_res[30]; one host
// wait) and skip the finish if it is up; without it the caller knows from the summed histogram that no slot can overflow.
int mcp_ba::select_gather_finish(const double* x, int n, const double* hist, SelState* state, double* out_dev, bool check_overflow) {
  const int grid = std::max(1, std::min(1024, (n + SEL_BLOCK*4 - 1)/(SEL_BLOCK*4)));
  const size_t tab = (size_t)world*sel_cap;              // [tab] overflow flag, [tab+1 .. tab+world] counts, [tab+world+1] gather counter
  HIPCK(hipMemsetAsync(d_seltab.p, 0, (tab + world + 2)*sizeof(double), st));
  unsigned int* cnt = reinterpret_cast<unsigned int*>(d_seltab.p + tab + world + 1);
  hipLaunchKernelGGL(k_select_gather_slot, dim3(grid), dim3(SEL_BLOCK), 0, st, n, x, hist, state, cnt,
                     d_seltab.p + (size_t)rank*sel_cap, sel_cap, rank == 0 ? d_seltab.p + tab : (double*)nullptr);
  hipLaunchKernelGGL(k_select_publish, dim3(1), dim3(64), 0, st, (const unsigned int*)cnt, sel_cap, d_seltab.p + tab + 1 + rank);
  if (allreduce(d_seltab.p, tab + 1 + world, 0, false, "median: candidate gather")) return -1;
  if (check_overflow) {
    HIPCK(hipMemcpyAsync(h_res + 30, d_seltab.p + tab, sizeof(double), hipMemcpyDeviceToHost, st));
    if (wait_stream(st, "median: overflow flag")) return -1;
    if (h_res[30] != 0.0) return 0;
  }
  hipLaunchKernelGGL(k_select_small, dim3(1), dim3(1024), 0, st, n, x, (const unsigned int*)cnt, (const double*)d_seltab.p, (const SelState*)state,
                     m_total, 0.0, out_dev, (double*)nullptr, (double*)nullptr, (const double*)(d_seltab.p + tab + 1), world, sel_cap);
  return 0;
}
// median + sigma block + robust chi2 of buffer `w` in one launch (ba_headl.h): the sigma block to sig() and d_res[25..28] (and to
// sig_copy2), the median to d_res[8], the robust chi2 to d_res[off]
int mcp_ba::head_large(int w, int off, double* sig_copy2) {
  tic(ST_SELECT);
  flip_sig();
  if (!headl_clean) { HIPCK(hipMemsetAsync(d_headl.p, 0, offsetof(HeadLScratch, cand), st)); headl_clean = true; }
  const int n = P.nmeas;
  const int grid = std::max(1, std::min(hl_grid, (n + HL_THREADS*2 - 1)/(HL_THREADS*2)));
  hipLaunchKernelGGL(k_head_large, dim3(grid), dim3(HL_THREADS), 0, st, n, (const double*)d_chi2[w].p, med_rank(), m_total,
                     prm.min_mestimator_sigma*prm.min_mestimator_sigma, sig(), d_res.p + 25, sig_copy2, d_res.p + 8, d_part0.p, d_res.p, off,
                     reinterpret_cast<HeadLScratch*>(d_headl.p));
  note_launch("k_head_large");
  sel_src = -1;
  toc();
  return 0;
}
// RobustKernelData::RecomputeNow on the chi2 array of buffer `w`
int mcp_ba::median_sigma(int w) {
  tic(ST_SELECT);
  flip_sig();                                                        // a fresh block: stragglers of the last iteration keep reading theirs
  const unsigned long long k = med_rank();                           // vErrorSquared[size/2]
  // several ranks, and the state is the one an accepted trial left: that trial's all-reduce carried the first two digit histograms
  // of this very chi2 array (ba_trial.h) -- if its prediction held and the selected bin fits the gather table, the median costs
  // ONE collective and no host wait
  if (multi() && sel_src >= 0 && tr_pred_ok[sel_src] && !tr_ovf[sel_src]) {
    if (select_gather_finish(d_chi2[w].p, P.nmeas, d_trial[sel_src].p + TRIAL_HDR, d_trstate[sel_src].p, d_res.p + 8, false)) return -1;
    timing.n_median_fast++;
  } else if (select_kth(d_chi2[w].p, P.nmeas, k, d_res.p + 8, true)) return -1;
  sel_src = -1;
  if (multi())         // (the single-GPU path writes the sigma block in its last kernel)
    hipLaunchKernelGGL(k_sigma_from_median, dim3(1), dim3(64), 0, st, d_res.p + 8, m_total,
                       prm.min_mestimator_sigma*prm.min_mestimator_sigma, sig(), d_res.p + 25 /* compute()'s read-back block */);
  toc();
  return 0;
}
// waits for `ticket` in the mailbox of trial q (written by the trial's last kernel) and takes the forwarded block from there
int mcp_ba::wait_mail(int q, unsigned long long ticket, int count) {
  const auto w0 = std::chrono::steady_clock::now();
  struct Acc { mcp_ba* h; int q; std::chrono::steady_clock::time_point t; ~Acc() { h->dbg_wait_us[q ? 1 : 0] += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t).count(); } } acc_{this, q, w0};
  const double* box = h_res + 32 + 32*q;
  volatile unsigned long long* tk = (volatile unsigned long long*)(box + MAIL_TICKET);
  hipStream_t watched = (q == 0) ? st : (tr_stream[q] ? tr_stream[q] : st2);
  // The result is normally a few tens of microseconds away: poll hard for a while (a trial is ~70 us, a solve ~1 ms), then give
  // the core to whoever else wants it between polls -- a tracker thread beside the mapper (BASELINE c5) should not lose a core
  // to a solver that waits for a long factorisation or for another rank.  Every so often: has the stream died, drained without
  // delivering, or made no progress for timeout_ms?
  constexpr unsigned long long HARD_SPINS = 1ull << 16;
  for (unsigned long long spins = 0; __atomic_load_n(tk, __ATOMIC_ACQUIRE) != ticket; ++spins) {
    if (spins >= HARD_SPINS) {
      std::this_thread::yield();
      if ((spins & 0x3ff) == 0) {
        hipError_t e = hipStreamQuery(watched);
        if (e == hipSuccess && q > 0 && st3) e = hipStreamQuery(st3);
        if (e == hipSuccess) { if (__atomic_load_n(tk, __ATOMIC_ACQUIRE) == ticket) break; set_err("mailbox ticket never arrived"); return -1; }
        if (e != hipErrorNotReady) { set_err(std::string("stream failed while waiting for a trial: ") + hipGetErrorString(e)); return -1; }
        if (std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - w0).count() > timeout_ms)
          return watchdog_fail(q ? "the result of a trial evaluated ahead (speculative lane)" : "the result of a trial (main lane)");
      }
    } else {
#if defined(__x86_64__) || defined(__i386__)
      __builtin_ia32_pause();
#elif defined(__aarch64__)
      __asm__ __volatile__("yield");
#endif
    }
  }
  std::memcpy(h_res, box, count*sizeof(double));
  return 0;
}
// several ranks: what follows the evaluation of a trial (candidate state `slot`, system q of the batch) on stream s / lane --
// digit histograms of its chi2, final sums into the trial buffer, ONE all-reduce, scan + result block (+ mailbox)   (ba_trial.h)
int mcp_ba::multi_trial_tail(hipStream_t s, int lane, int q, int slot, int nbe, const double* p0, int nbb, const double* p1, const double* p2,
                             const double* pose_parts, int pose_off, bool main_block, double* mail, int mail_count, unsigned long long ticket) {
  double* T = d_trial[q].p;
  HIPCK(hipMemsetAsync(T, 0, TRIAL_LEN*sizeof(double), s));
  const bool ride = robust && sel_ride && m_total > 0;       // (decided on GLOBAL quantities: the payload size must agree on every rank)
  if (ride) {
    const int grid = std::max(1, std::min(256, (P.nmeas + SEL_BLOCK*8 - 1)/(SEL_BLOCK*8)));
    hipLaunchKernelGGL(k_select_hist2, dim3(grid), dim3(SEL_BLOCK), 0, s, P.nmeas, (const double*)d_chi2[slot].p, pred_bin, T + TRIAL_HDR);
  }
  const bool rides = main_block && start_rides;
  hipLaunchKernelGGL(k_final_sums, dim3(1), dim3(256), 0, s, nbe, p0, nbb, p1, nbb, p2, T, 0, (const int*)d_fail.p + q, (double*)nullptr, 0, 0ull,
                     rides ? (const double*)(d_res.p + 24) : (const double*)nullptr, 4);
  if (allreduce(T, ride ? TRIAL_LEN : TRIAL_HDR, lane, false, lane ? "trial evaluated ahead: result block + median histograms" : "trial: result block + median histograms")) return -1;
  hipLaunchKernelGGL(k_trial_post, dim3(1), dim3(SEL_BLOCK), 0, s, T, d_trstate[q].p, (unsigned long long)(m_total/2), pred_bin, (double)sel_cap, ride ? 1 : 0,
                     pose_parts, pose_off, main_block ? (const double*)(d_res.p + 24) : (const double*)nullptr, rides ? 1 : 0,
                     main_block ? d_res.p : (double*)nullptr, mail, mail_count, ticket);
  if (rides) start_rides = false;
  return 0;
}
// the step of speculative system q applied and evaluated on stream s (which has just solved it): candidate state cand(q), its own
// scratch, result block d_res[32 + 8 q ..] -> mailbox q.
int mcp_ba::enqueue_spec_trial(hipStream_t s, int q) {
  tr_stream[q] = s;
  const int slot = cand(q);
  const double lam = batch_lambda[q];
  double* Sq = d_red.p + q*red_stride; double* rhsq = Sq + (size_t)np*np;
  double* resq = d_res.p + 32 + 8*q;
  const bool small = small_mode();
  const int nbb = (nfl*BS_TPP + BS_BLOCK - 1)/BS_BLOCK;
  const int ncb = trial_chain_blocks();
  if (ncb) { hipLaunchKernelGGL(k_trial_apply, dim3(ncb + (nfl ? nbb : 0)), dim3(256), (size_t)P.npose*12*sizeof(double), s, P, ncb, lam, (const double*)rhsq, (const double*)(rhsq + np), (const double*)d_pose[cur].p, d_pose[slot].p, resq + 4, d_sxp[q].p,
                                d_first[slot].p, d_second[slot].p, d_last[slot].p, (const double*)d_g.p, (const double*)d_W.p, (const double*)(d_Vinv.p + q*vinv_stride),
                                (const double*)d_pt[cur].p, d_pt[slot].p, d_sxl[q].p, d_sp1[q].p, d_sp2[q].p); note_launch("k_trial_apply"); }
  else {
  hipLaunchKernelGGL(k_update_poses, dim3(1), dim3(256), 0, s, P, lam, (const double*)rhsq, (const double*)(rhsq + np), (const double*)d_pose[cur].p, d_pose[slot].p, resq + 4, d_sxp[q].p);
  if (nfl) hipLaunchKernelGGL(k_backsub, dim3(nbb), dim3(BS_BLOCK), 0, s, P, lam, (const double*)rhsq, (const double*)d_g.p, (const double*)d_W.p,
                              (const double*)(d_Vinv.p + q*vinv_stride), (const double*)d_pt[cur].p, d_pt[slot].p, d_sxl[q].p, d_sp1[q].p, d_sp2[q].p);
  }
  HIPCK(hipEventRecord(ev_wf[q], s));                     // the linearisation's outputs are not read below this line (join_spec_lin)
  if (P.nchain && !ncb) hipLaunchKernelGGL(k_chains, dim3((P.nchain + 63)/64), dim3(64), 0, s, P, (const double*)d_pose[slot].p, d_first[slot].p, d_second[slot].p, d_last[slot].p);
  const int nbe = (P.nmeas + EVAL_BLOCK - 1)/EVAL_BLOCK;
  const bool with_head = head_ahead_want && !small && large_heads() && large_head_ahead != 2 && nbe > 0;      // (MCP_BA_HEAD_AHEAD=2: only the main stream's trial carries a head)
  if (nbe) hipLaunchKernelGGL((k_eval<true>), dim3(nbe), dim3(EVAL_BLOCK), 0, s, P, (const double*)d_pt[slot].p, (const double*)d_last[slot].p, d_chi2[slot].p, (double*)nullptr,
                              (const double*)sig(), d_sp0[q].p);
  pre_ticket[q] = ++mail_ticket;
  if (multi()) {
    if (multi_trial_tail(s, 1, q, slot, nbe, d_sp0[q].p, nbb, nfl ? d_sp1[q].p : nullptr, nfl ? d_sp2[q].p : nullptr, resq + 4, 4, false, h_mail_dev + 32*q, MAIL_TICKET, pre_ticket[q])) return -1;
  } else
  hipLaunchKernelGGL(k_final_sums, dim3(1), dim3(256), 0, s, nbe, (const double*)d_sp0[q].p, nbb, (const double*)(nfl ? d_sp1[q].p : nullptr),
                     nbb, (const double*)(nfl ? d_sp2[q].p : nullptr), resq, 0, (const int*)d_fail.p + q, h_mail_dev + 32*q, 6, pre_ticket[q]);
  HIPCK(hipEventRecord(ev_tr[q], s));
  pre_run[q] = true; ahead_enq[q] = true;
  if (with_head) { if (enqueue_head(s, q, slot, true)) return -1; }
  return 0;
}
// the iteration has decided (or starts a new solve): a trial evaluated ahead that nobody asked for is simply never looked at
// (at most one is in flight -- see run_ahead -- and it is done before the next linearisation wants its buffers)
int mcp_ba::cancel_spec_trials() {
  for (int q = 0; q < MAX_SYS; ++q) pre_run[q] = false;
  return 0;
}
// One trial ahead: when the host turns to trial q - 1, the step of system q (if it was solved speculatively on another stream) is
// applied and evaluated there already, so that a rejection of q - 1 finds q's result waiting.
int mcp_ba::run_ahead(int q) {
  if (!spec_trials || !use_mailbox || prm.profile) return 0;
  if (multi() && overlap_spec != 1) return 0;       // (the speculative lane belongs to the second stream)
  if (!spec_pending && !spec3_pending && q >= spec2_from) { /* its stream has been joined already: still valid to use it */ }
  if (q < spec2_from || q >= batch_n || pre_run[q]) return 0;
  if (spec_trials >= 2 && !multi() && st_tr[q] && q > spec2_from) {      // (the first one stays on the chain's own stream: hardware queues are few)
    // its own stream, ordered behind the chain that produced system q: the trials ahead run side by side
    hipEvent_t done = (q >= spec3_from) ? ev_spec3 : ev_spec;
    HIPCK(hipStreamWaitEvent(st_tr[q], done, 0));
    return enqueue_spec_trial(st_tr[q], q);
  }
  return enqueue_spec_trial(q >= spec3_from ? st3 : st2, q);
}
int mcp_ba::read_results(int count) {
  // the failure flag of a trial travels inside the block (k_final_sums, d_res[3]): one copy, one wait
  HIPCK(hipMemcpyAsync(h_res, d_res.p, count*sizeof(double), hipMemcpyDeviceToHost, st));
  return wait_stream(st, "a result block");
}

// buildSystem at the current state (sigma block must be current)
int mcp_ba::linearize() {
  mark("lin_wait", st);
  if ((lin_join_full ? join_spec() : join_spec_lin())) return -1;        // the second stream may still be reading the previous linearisation
  mark("lin_go", st);
  spec_ok = false;                   // a speculative solve belongs to the linearisation it was built from
  tic(ST_LIN);
  const size_t n2 = (size_t)np*np;
  if (nbig) {      // points that touch more than GRP_LMAX poses go through the generic atomic path into their own dense block
    HIPCK(hipMemsetAsync(d_ubig.p, 0, (n2 + np)*sizeof(double), st));
    if (nfl) {
      HIPCK(hipMemsetAsync(d_V.p, 0, (size_t)nfl*6*sizeof(double), st));
      HIPCK(hipMemsetAsync(d_g.p, 0, (size_t)nfl*3*sizeof(double), st));
    }
    if (ninc) HIPCK(hipMemsetAsync(d_W.p, 0, (size_t)ninc*18*sizeof(double), st));
    if (P.nmeas) hipLaunchKernelGGL(k_linearize, dim3((P.nmeas + LIN_BLOCK - 1)/LIN_BLOCK), dim3(LIN_BLOCK), 0, st, P, 1,
                       d_pt[cur].p, d_first[cur].p, d_second[cur].p, sig(), d_ubig.p, d_ubig.p + n2, d_V.p, d_g.p, d_W.p);
  }
  // (the quad form keeps the group's W blocks in LDS beside the pose blocks: groups whose points see many poses -- 16 points x 16
  //  incidences is 36 KB on top of up to 39 KB of pose blocks -- do not fit the 64 KB a launch gets by default and take the
  //  one-lane-per-point kernel, which handles groups of any size up to 64 points)
  const size_t quad_lds = ((size_t)std::max(grp_blk_max, 1)*36 + (size_t)std::max(grp_inc_max, 1)*18)*sizeof(double);
  static const bool quad_on = [] { const char* e = getenv("MCP_BA_LIN_QUAD"); return !(e && atoi(e) == 0); }();
  if (ngroup && grp_pts <= LIN_QUAD_PTS && quad_on && quad_lds <= 58*1024)      // (+ 5 KB of static LDS: tables and cameras)
    hipLaunchKernelGGL(k_linearize_quad, dim3(ngroup), dim3(64), quad_lds, st, P,
                       d_pt[cur].p, d_first[cur].p, d_second[cur].p, sig(), d_stU.p, d_stb.p, d_V.p, d_g.p, d_W.p, std::max(grp_blk_max, 1)*36, d_fail.p);
  else if (ngroup && lin_pipe && !getenv("MCP_BA_TEST_REFUSE_LAUNCH"))
    hipLaunchKernelGGL(k_linearize_pipe, dim3(ngroup), dim3(64), (size_t)std::max(grp_blk_max, 1)*36*sizeof(double), st, P,
                       d_pt[cur].p, d_first[cur].p, d_second[cur].p, sig(), d_stU.p, d_stb.p, d_V.p, d_g.p, d_W.p, d_fail.p);
  else if (ngroup)
    hipLaunchKernelGGL(k_linearize_group, dim3(ngroup), dim3(64), getenv("MCP_BA_TEST_REFUSE_LAUNCH") ? ((size_t)1 << 20) /* test hook: more LDS than a compute unit has */ : (size_t)std::max(grp_blk_max, 1)*36*sizeof(double), st, P,
                       d_pt[cur].p, d_first[cur].p, d_second[cur].p, sig(), d_stU.p, d_stb.p, d_V.p, d_g.p, d_W.p, d_fail.p);
  if (ngroup) note_launch("k_linearize_group / k_linearize_quad");
  fail_clean = ngroup > 0;            // (the group kernel has cleared the failure flags: the first solve of this linearisation needs no fill)
#ifdef MCP_LIN_PROF
  {
    HIPCK(hipStreamSynchronize(st));
    unsigned long long pr[64]; hipMemcpyFromSymbol(pr, HIP_SYMBOL(g_lin_prof), sizeof pr);
    double a[6] = {0}; int cnt = 0;
    for (int b2 = 0; b2 < 6; ++b2) { const unsigned long long* q = pr + 8*b2; if (!q[6]) continue; for (int i2 = 0; i2 < 6; ++i2) a[i2] += (double)(q[i2 + 1] - q[i2]); ++cnt; }
    if (cnt) fprintf(stderr, "[lin prof] zero %.0f  setup %.0f  measurements %.0f  V/g/W out %.0f  seg-reduce %.0f  flush %.0f  (cycles, %d groups)\n", a[0]/cnt, a[1]/cnt, a[2]/cnt, a[3]/cnt, a[4]/cnt, a[5]/cnt, cnt);
  }
#endif

  toc();
  timing.n_linearize++;
  return 0;
}

// The reduced system(s) of the current linearisation for the lambdas of `sb`: every rank contributes U_r - Schur_r
// (+ lambda I once, on rank 0: sb.lambda_init), bp_r - W V^-1 g and bp_r; summed over the ranks when there are several.
int mcp_ba::build_system(int nsys, SysBatch& sbfull, int q0, hipStream_t on) {
  hipStream_t s = on ? on : st;
  const bool main_stream = (s == st);
  // the systems [q0, q0 + nsys) of the batch: a shifted copy of the lambda table, buffers offset by q0 system strides
  SysBatch sb = sbfull;
  for (int i = 0; i < nsys; ++i) { sb.lambda[i] = sbfull.lambda[q0 + i]; sb.lambda_init[i] = sbfull.lambda_init[q0 + i]; }
  sb.sstride = red_stride; sb.vstride = vinv_stride; sb.ststride = nstage*36; sb.strstride = (size_t)nrhs_rows*6;
  sbfull.sstride = sb.sstride; sbfull.vstride = sb.vstride; sbfull.ststride = sb.ststride; sbfull.strstride = sb.strstride;
  double* Sq = d_red.p + q0*red_stride; double* Vq = d_Vinv.p + q0*vinv_stride;
  double* stSq = d_stS.p + q0*sb.ststride; double* strq = d_str.p + q0*sb.strstride; int* failq = d_fail.p + q0;
  if (main_stream) tic(ST_SCHUR);
  if (nfl && ngroup && sch4_on && sch4_ok)      // every system of the batch in one workgroup per group (ba_schur4.h)
    hipLaunchKernelGGL(k_schur4, dim3(ngroup), dim3(256), S4_LDS_BYTES, s, P, nsys, (const int*)(sch4_order ? d_g_order.p : nullptr), (const double*)d_V.p, (const double*)d_g.p, (const double*)d_W.p, Vq, stSq, strq, failq, sb);
  else if (nfl && ngroup) hipLaunchKernelGGL(k_schur_group, dim3(ngroup, nsys), dim3(256), SCH_LDS_BYTES, s, P, sb.lambda[0], d_V.p, d_g.p, d_W.p, Vq, stSq, strq, failq, sb);
  if (nfl && ngroup) note_launch("k_schur4 / k_schur_group");
  else if (ngroup && nstage) {      // no free point: nothing is eliminated, the staged Schur blocks are zero
    HIPCK(hipMemsetAsync(stSq, 0, (size_t)nsys*nstage*36*sizeof(double), s));
    HIPCK(hipMemsetAsync(strq, 0, (size_t)nsys*nrhs_rows*6*sizeof(double), s));
  }
  if (np && asm_long) hipLaunchKernelGGL(k_assemble_long, dim3(A.ntiles*32, nsys), dim3(32*ASML_LPE), 0, s, A, np, (const double*)d_stU.p, (const double*)d_stb.p,
                             (const double*)stSq, (const double*)strq, (const double*)Ubig(), Sq, sb);
  else if (np) hipLaunchKernelGGL(k_assemble, dim3(A.ntiles, nsys), dim3(ASM_NT), 0, s, A, np, (const double*)d_stU.p, (const double*)d_stb.p,
                             (const double*)stSq, (const double*)strq, (const double*)Ubig(), Sq, sb);
  if (nfl && nbig) hipLaunchKernelGGL(k_schur, dim3((nfl + 3)/4, nsys), dim3(256), 0, s, P, 1, sb.lambda[0], d_V.p, d_g.p, d_W.p, Vq, Sq, Sq + (size_t)np*np, failq, sb);
  if (main_stream) toc();
#ifdef MCP_SCH_PROF
  {
    HIPCK(hipStreamSynchronize(s));
    unsigned long long pr[64]; hipMemcpyFromSymbol(pr, HIP_SYMBOL(g_sch_prof), sizeof pr);
    double a[8] = {0}; int cnt = 0;
    for (int b2 = 0; b2 < 8; ++b2) { const unsigned long long* q = pr + 8*b2; if (!q[0]) continue; for (int i2 = 0; i2 < 8; ++i2) a[i2] += (double)q[i2]; ++cnt; }
    if (cnt && sch4_on && sch4_ok)
      fprintf(stderr, "[sch4 prof] prologue %.0f  clear+scatter %.0f  barrier %.0f  prefetch+mfma %.0f  barrier %.0f  flush %.0f  (cycles of wavefront 0, %d systems, %d groups)\n",
              a[0]/cnt, a[1]/cnt, a[2]/cnt, a[3]/cnt, a[4]/cnt, a[5]/cnt, nsys, cnt);
    else if (cnt) fprintf(stderr, "[sch prof] prologue %.0f  inverse+barrier %.0f  scatter %.0f  prefetch+barrier %.0f  mfma %.0f  rhs %.0f  barrier+clear %.0f  flush %.0f  (cycles, %d systems, %d groups)\n",
                     a[0]/cnt, a[1]/cnt, a[2]/cnt, a[3]/cnt, a[4]/cnt, a[5]/cnt, a[6]/cnt, a[7]/cnt, nsys, cnt);
  }
#endif
  if (np && multi()) {
    // pack the structurally non-zero tiles of S with rhs and bp, sum over the ranks, unpack -- on the stream that built the systems,
    // through that stream's lane: the trial's own system never queues behind the speculative ones
    const size_t npack = pack_stride*nsys;
    double* pk = d_pack.p + q0*pack_stride;
    hipLaunchKernelGGL(k_pack_tiles, dim3(n_red_tiles + 1, nsys), dim3(256), 0, s, (const double*)Sq, np, (const int*)d_red_tiles.p, n_red_tiles, pk, 1, red_stride, pack_stride);
    if (allreduce(pk, npack, main_stream ? 0 : 1, false, main_stream ? "reduced system of the trial (packed tiles)" : "speculative reduced systems (packed tiles)")) return -1;
    hipLaunchKernelGGL(k_pack_tiles, dim3(n_red_tiles + 1, nsys), dim3(256), 0, s, (const double*)pk, np, (const int*)d_red_tiles.p, n_red_tiles, Sq, 0, red_stride, pack_stride);
  }
  return 0;
}

// small bundles: the whole head of an iteration for state `w` in one launch (ba_small.h) -- the sigma block of the other parity
// becomes current (as median_sigma() does), d_res[24] = robust chi2 of the state, d_res[25..28] = copy of the sigma block
int mcp_ba::head_small(int w, bool sum_aside) {
  tic(ST_SELECT);
  const double* prev = sig();
  if (robust) { flip_sig(); sel_src = -1; }
  // sum_aside: linearize() needs the sigma block, nobody on the device needs the robust chi2 (the host reads it with the next
  // trial's results): that sum -- a third of this kernel's time -- goes to the second stream, next to the linearisation
  const bool aside = sum_aside && st2 && ev_head && ev_sum && d_parth.p;
  // a sum still on its way on the second stream (the head of a trial that was then rejected) writes the same d_res[24]: it must not
  // land after the one this launch takes itself
  if (!aside && join_sum()) return -1;
  hipLaunchKernelGGL(k_head_small, dim3(1), dim3(1024), HS_STASH*sizeof(double), st, P.nmeas, robust ? 1 : 0, (const double*)d_chi2[w].p, med_rank(), m_total,
                     prm.min_mestimator_sigma*prm.min_mestimator_sigma, prev, d_res.p + 8, sig(), d_res.p + 25, d_res.p, 24, aside ? 0 : 1);
  note_launch("k_head_small");
  if (aside) { HIPCK(hipEventRecord(ev_head, st)); sum_w = w; sum_sig = sig(); }      // (the second stream's part: sum_aside(), once the caller has queued what else it has for that stream)
  toc();
  return 0;
}
// the robust chi2 of state sum_w on the second stream, behind the head kernel (ev_head) -- enqueued AFTER the speculative work of
// the iteration, which must not wait for the main stream's trial
int mcp_ba::sum_aside() {
  if (sum_w < 0) return 0;
  const int nbe = (P.nmeas + EVAL_BLOCK - 1)/EVAL_BLOCK;
  HIPCK(hipStreamWaitEvent(st2, ev_head, 0));
  hipLaunchKernelGGL(k_robust_sum, dim3(nbe), dim3(EVAL_BLOCK), 0, st2, P.nmeas, robust, (const double*)d_chi2[sum_w].p, (const double*)sum_sig, d_parth.p);
  hipLaunchKernelGGL(k_final_sums, dim3(1), dim3(256), 0, st2, nbe, (const double*)d_parth.p, 0, (const double*)nullptr, 0, (const double*)nullptr, d_res.p, 24, (const int*)nullptr);
  HIPCK(hipEventRecord(ev_sum, st2));
  sum_pending = true; sum_w = -1;
  return 0;
}
// the sum head_small() put on the second stream has to be in d_res[24] before anything on the main stream reads that block
int mcp_ba::join_sum() {
  if (sum_pending) { HIPCK(hipStreamWaitEvent(st, ev_sum, 0)); sum_pending = false; }
  return 0;
}
// Small bundles: the head of the NEXT iteration (median of |chi2|, sigma block, robust chi2 of the state -> d_res[24..28]) for the
// trial state `w`, enqueued right behind the trial's own kernels, before the host has seen its result.  A trial is accepted nine
// times out of ten; then the device has spent the host's turn-around (mailbox -> accept/reject -> first launch of the next
// iteration: ~30 us of an ~170 us iteration) on work the next iteration needs first.  If the trial is rejected the block is never
// looked at: it went to the sigma block of the other parity (which nobody reads: the trials of the last iteration that may still
// have been evaluating with it are waited for below) and to d_res[24..28], which the host has already taken for this iteration.
int mcp_ba::head_ahead(int w) {
  for (int q = 1; q < MAX_SYS; ++q) if (ev_tr[q]) HIPCK(hipStreamWaitEvent(st, ev_tr[q], 0));
  if (head_small(w, true)) return -1;                // (flips to the fresh sigma block)
  if (robust) flip_sig();                            // ... which becomes the current one only if the trial is accepted (compute())
  head_ahead_for = w;
  return 0;
}

// The head of the iteration that accepting trial q would start, for a map beyond the small-bundle limit (round 6; SURVEY 8(a) a10 / a11,
// src/ChainBundle.cc:810-833, 913-917, MEstimator.h:194-204; kernels in ba_head.h): TWO launches behind the trial's result block count
// the first digits of its |chi2|, finish the median and write slot [head_par][q]'s sigma block, and tell the host through a pinned
// status word whether they could.  Round 5 started the selection (four launches) when
// the host had seen an accepted trial's result: 65 us between that trial and the next linearisation; now the next iteration starts at
// its linearisation.  The robust chi2 at the new sigma -- which only the HOST needs, with the first trial's result -- is summed beside
// the linearisation on the second stream (compute()).
int mcp_ba::enqueue_head(hipStream_t s_trial, int q, int w, bool side) {
  HeadScratch& H = hsc[head_par*MAX_SYS + q];
  // a trial evaluated ahead: its head forks off its last kernel (ev_tr[q]) onto the head stream -- on the second stream it would sit in
  // front of the next trial ahead (measured: +38 us per rejected trial, more than the heads save)
  hipStream_t s = (side && st_h) ? st_h : s_trial;
  if (s != s_trial) HIPCK(hipStreamWaitEvent(s, ev_tr[q], 0));
  head_ticket[q] = ++head_ticket_ctr;
  hipLaunchKernelGGL(k_head_hist, dim3(HEAD_GRID), dim3(HEAD_THREADS), 0, s, P.nmeas, (const double*)d_chi2[w].p, (const double*)sig(), H.hist);
  hipLaunchKernelGGL(k_head_finish, dim3(HEAD_GRID), dim3(HEAD_THREADS), 0, s, P.nmeas, (const double*)d_chi2[w].p, H.hist, H.vals, (const double*)sig(),
                     med_rank(), m_total, prm.min_mestimator_sigma*prm.min_mestimator_sigma,
                     sig_block(2 + head_par*MAX_SYS + q), H.rs + 1, H.out, h_mail_dev - 32 + HEAD_MAIL + 2*q, head_ticket[q]);
  note_launch("k_head_finish");
  HIPCK(hipEventRecord(head_ev[q], s));
  head_enq[q] = true; head_state[q] = w;
  return 0;
}
// the status word of trial q's head: 1 = its sigma block is written, 2 = it could not (prediction missed: plain selection), -1 = error
int mcp_ba::wait_head(int q) {
  const double* box = h_res + HEAD_MAIL + 2*q;
  volatile unsigned long long* tk = (volatile unsigned long long*)(box + 1);
  const auto w0 = std::chrono::steady_clock::now();
  for (unsigned long long spins = 0; __atomic_load_n(tk, __ATOMIC_ACQUIRE) != head_ticket[q]; ++spins) {
    if ((spins & 0xfff) == 0xfff) {
      if (std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - w0).count() > timeout_ms) return watchdog_fail("the status word of an iteration head");
      std::this_thread::yield();
    }
  }
  return box[0] == 1.0 ? 1 : 2;
}

// one LM trial up to and including the evaluation of the trial state.
// factorisation + back-substitution of systems [q0, q0 + n) of the batch on stream s: ~40 dependent launches, or -- with
// MCP_BA_GRAPH=1 -- one launch of a graph captured the first time this sub-batch shape occurs (same buffers, same plan every time).
// With two chains in flight the host's enqueue rate decides when the second one can start; a graph hands it over at once.
int mcp_ba::solve_chain(hipStream_t s, int n, int q0) {
  if (use_graph && !prm.profile) {
    hipGraphExec_t& ex = chain_exec[n][q0];
    if (!ex) {
      hipGraph_t g = nullptr;
      if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) == hipSuccess) {
        chol_factor(s, plan, d_red.p, d_fail.p, n, red_stride, q0);
        chol_back(s, plan, d_red.p, n, red_stride, q0);
        if (hipStreamEndCapture(s, &g) != hipSuccess || !g || hipGraphInstantiate(&ex, g, nullptr, nullptr, 0) != hipSuccess) { ex = nullptr; use_graph = 0; }
        if (g) (void)hipGraphDestroy(g);
      } else use_graph = 0;
      (void)hipGetLastError();
    }
    if (ex) { HIPCK(hipGraphLaunch(ex, s)); return 0; }
  }
  chol_factor(s, plan, d_red.p, d_fail.p, n, red_stride, q0);
  chol_back(s, plan, d_red.p, n, red_stride, q0);
  return 0;
}

// on return h_res: [0] robust chi2 of the trial, [1] sum x(lambda x + b), [2] sum x^2
int mcp_ba::solve_trial(double lam, bool& ok2, double ni) {
  int defer_n1 = 0, defer_n2 = 0, defer_nsys = 0;
  head_ahead_for = -1;
  bool ahead_single = false;      // whole batch solved on the main stream: the speculative systems' trials can still be evaluated ahead on the second
  if (spec_ok && sys_cur + 1 < batch_n && batch_lambda[sys_cur + 1] == lam) {
    // an earlier trial of this iteration already built and solved this system speculatively (possibly on the second stream)
    ++sys_cur;
    timing.n_spec_hits++;
    if (pre_run[sys_cur]) {
      // ... and its step has been applied and evaluated there as well: the result is in (or on its way to) the trial's mailbox
      const int q = sys_cur;
      last_tr = cand(q); last_xp = &d_sxp[q]; last_xl = &d_sxl[q];
      if (run_ahead(q + 1)) return -1;
      if (wait_mail(q, pre_ticket[q], multi() ? MAIL_TICKET : 6)) return -1;
      if (multi()) { tr_pred_ok[q] = h_res[MAIL_PRED_OK] != 0.0; tr_ovf[q] = h_res[MAIL_OVERFLOW] != 0.0; }
      pre_run[q] = false;
      HIPCK(hipStreamWaitEvent(st, ev_tr[q], 0));         // whatever the main stream does next with the trial's state comes after the kernels that made it
      mark("pre_used", st);
      ++dbg_pre;
      if (!nfl_total) h_res[1] = h_res[2] = 0.0;
      h_res[1] += h_res[4]; h_res[2] += h_res[5];
      if (h_res[3] >= 1e9) return persist_fallback(lam, ok2, ni);
      ok2 = (h_res[3] == 0.0);
      timing.n_trials++;
      return 0;
    }
    if (join_spec(sys_cur)) return -1;
  } else {
    sys_cur = 0;
    // the lambdas of the rejection branch of the LM schedule: lambda *= ni; ni *= 2 (same operations as compute())
    const int nsys = (ni > 0) ? 1 + std::max(0, std::min(spec_now(), MAX_SYS - 1)) : 1;
    SysBatch sb; std::memset(&sb, 0, sizeof sb);
    { double l = lam, f = ni; for (int q = 0; q < nsys; ++q) { batch_lambda[q] = sb.lambda[q] = l; sb.lambda_init[q] = (rank == 0) ? l : 0.0; l *= f; f *= 2; } }
    batch_n = nsys;
    if (cancel_spec_trials()) return -1;   // (a re-solve inside an iteration: nothing of the previous batch is wanted any more)
    if (join_spec()) return -1;            // (... and its stragglers first)
    if (!fail_clean) HIPCK(hipMemsetAsync(d_fail.p, 0, 4*sizeof(int), st));
    fail_clean = false;
    // One rank, one-launch factorisation: every system of the batch has its own critical workgroup, four systems take the time of
    // one, so the batch stays on the main stream (measured: 990 vs 940 LM it/s; MCP_BA_OVERLAP=1 splits it as the per-step kernels
    // want it).  Several ranks: one speculative stream, the one lane 1 belongs to.
    const int ov_single = (overlap_auto && plan.use_persist && plan.persist.ok) ? 0 : overlap_spec;
    const int ov = multi() ? std::min(overlap_spec, 1) : ov_single;
    const bool split = ov && nsys > 1 && np > 0 && st2;
    if (split) {
      // system 0 -- the one this trial needs -- alone on the main stream; the speculative systems behind the fork on the second
      // stream.  Same kernels on the same data as the batched path: the numbers do not depend on which stream produced them.
      // main_sys systems stay on the main stream: 1 = only the trial's own; 2 = also the first speculative one, so that a second trial
      // never waits for the other stream (its system comes out of the same launches as the first trial's)
      const int n1 = std::min(std::max(main_sys, 1), nsys - 1);
      HIPCK(hipEventRecord(ev_fork, st));
      plan.persist.batch_sys = nsys;
      mark("fork", st);
      if (build_system(n1, sb, 0, st)) return -1;
      mark("main_built", st);
      // MCP_BA_SPEC_DELAY=1: the speculative systems' Schur complements start only when the trial's own is through (they share the
      // compute units otherwise: 85 us alone, 170 us side by side) -- the first trial sooner, a rejected trial's successor later
      if (spec_delay) HIPCK(hipEventRecord(ev_fork, st));
      // host enqueue order: the other streams' Schur complements right away (they start at the fork on the device and run beside
      // the main stream's), their factorisation chains only after this trial's own chain and tail (below) -- the main stream never
      // runs dry behind the ~45 launches of another chain
      defer_n2 = (ov >= 2 && nsys - n1 > 1 && st3) ? 1 : nsys - n1;       // systems on the second stream
      HIPCK(hipStreamWaitEvent(st2, ev_fork, 0));
      mark("spec_start", st2);
      if (build_system(defer_n2, sb, n1, st2)) return -1;
      mark("spec_built", st2);
      if (n1 + defer_n2 < nsys) { HIPCK(hipStreamWaitEvent(st3, ev_fork, 0)); if (build_system(nsys - n1 - defer_n2, sb, n1 + defer_n2, st3)) return -1; }
      if (use_graph && !prm.profile) { if (solve_chain(st, n1, 0)) return -1; }
      else {
        tic(ST_CHOL); chol_factor(st, plan, d_red.p, d_fail.p, n1, red_stride, 0); toc();
        tic(ST_SOLVE); chol_back(st, plan, d_red.p, n1, red_stride, 0); toc();
      }
      defer_n1 = n1; defer_nsys = nsys;
    } else {
    plan.persist.batch_sys = nsys;
    if (build_system(nsys, sb)) return -1;
    if (np) {
      if (use_graph && !prm.profile) {
        // the ~40 dependent launches of one factorisation + back-substitution replayed from a captured graph
        // (arguments never change between solves: same buffers, same plan); one graph per batch width
        if (!chol_exec[nsys]) {
          hipGraph_t g = nullptr;
          if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) == hipSuccess) {
            chol_factor(st, plan, d_red.p, d_fail.p, nsys, red_stride);
            chol_back(st, plan, d_red.p, nsys, red_stride);
            if (hipStreamEndCapture(st, &g) != hipSuccess || !g || hipGraphInstantiate(&chol_exec[nsys], g, nullptr, nullptr, 0) != hipSuccess) { chol_exec[nsys] = nullptr; use_graph = 0; }
            if (g) (void)hipGraphDestroy(g);
          } else use_graph = 0;
          (void)hipGetLastError();
        }
        if (chol_exec[nsys]) HIPCK(hipGraphLaunch(chol_exec[nsys], st));
        else { chol_factor(st, plan, S(), d_fail.p, nsys, red_stride); chol_back(st, plan, S(), nsys, red_stride); }
      } else {
        tic(ST_CHOL); chol_factor(st, plan, S(), d_fail.p, nsys, red_stride); toc();
        tic(ST_SOLVE); chol_back(st, plan, S(), nsys, red_stride); toc();
      }
      if (st2 && nsys > 1 && !multi() && spec_trials && use_mailbox && !prm.profile) {
        // every system of the batch is solved: while this trial runs on the main stream, the next one's step is applied and evaluated
        // on the second stream (enqueue_spec_trial), so that a rejection finds its result waiting
        HIPCK(hipEventRecord(ev_fork, st)); HIPCK(hipStreamWaitEvent(st2, ev_fork, 0));
        spec2_from = 1; spec3_from = MAX_SYS; ahead_single = true;
      }
    }
    }
    spec_ok = (nsys > 1);
    timing.n_solves++;
  }
  mark("solved", st);
  const int tr = cand(sys_cur); last_tr = tr; last_xp = &d_xp_cand; last_xl = &d_xl;
  const double* bp_glob = bp();
  tic(ST_UPDATE);
  const bool small = small_mode();
  const int nbb = (nfl*BS_TPP + BS_BLOCK - 1)/BS_BLOCK;
  const int ncb = trial_chain_blocks();
  if (ncb) { hipLaunchKernelGGL(k_trial_apply, dim3(ncb + (nfl ? nbb : 0)), dim3(256), (size_t)P.npose*12*sizeof(double), st, P, ncb, lam, (const double*)rhs(), bp_glob, (const double*)d_pose[cur].p, d_pose[tr].p, d_res.p + 6, d_xp_cand.p,
                                d_first[tr].p, d_second[tr].p, d_last[tr].p, (const double*)d_g.p, (const double*)d_W.p, (const double*)Vinv(),
                                (const double*)d_pt[cur].p, d_pt[tr].p, d_xl.p, d_part1.p, d_part2.p); note_launch("k_trial_apply"); }
  else {
  hipLaunchKernelGGL(k_update_poses, dim3(1), dim3(256), 0, st, P, lam, rhs(), bp_glob, d_pose[cur].p, d_pose[tr].p, d_res.p + 6, d_xp_cand.p);
  if (nfl) hipLaunchKernelGGL(k_backsub, dim3(nbb), dim3(BS_BLOCK), 0, st, P, lam, rhs(), d_g.p, d_W.p, Vinv(),
                              d_pt[cur].p, d_pt[tr].p, d_xl.p, d_part1.p, d_part2.p);
  }
  toc();
  tic(ST_EVAL);
  if (!ncb) launch_chains(tr);
  const int nbe = (P.nmeas + EVAL_BLOCK - 1)/EVAL_BLOCK;
  const bool mailbox = use_mailbox && !prm.profile;
  mail_ticket0 = ++mail_ticket;
  const bool with_head = head_ahead_want && mailbox && !small && large_heads();
  launch_eval(tr, true, nullptr);
  if (multi()) {
    // (several ranks: the sums are rank-local until the trial's all-reduce; k_trial_post writes the block and the mailbox)
    if (multi_trial_tail(st, 0, sys_cur, tr, nbe, d_part0.p, nbb, nfl ? d_part1.p : nullptr, nfl ? d_part2.p : nullptr, d_res.p + 6, 6, true,
                         mailbox ? h_mail_dev : (double*)nullptr, MAIL_TICKET, mail_ticket0)) return -1;
  } else {
    if (join_sum()) return -1;                       // (this launch forwards d_res[24..28], the head of the iteration, to the host)
    hipLaunchKernelGGL(k_final_sums, dim3(1), dim3(256), 0, st, nbe, (const double*)d_part0.p, nbb, (const double*)(nfl ? d_part1.p : nullptr),
                       nbb, (const double*)(nfl ? d_part2.p : nullptr), d_res.p, 0, (const int*)d_fail.p + sys_cur,
                       mailbox ? h_mail_dev : (double*)nullptr, 29, mail_ticket0, (const double*)nullptr, 0, start_blk);
    if (!nfl) HIPCK(hipMemsetAsync(d_res.p + 1, 0, 2*sizeof(double), st));
  }
  toc();
  mark("trial_end", st);
  if (head_ahead_want && mailbox && !multi() && small) { if (head_ahead(tr)) return -1; mark("head_ahead", st); }
  if (defer_nsys) {
    const int n2 = defer_n2;
    if (solve_chain(st2, n2, defer_n1)) return -1;
    mark("spec_done", st2);
    HIPCK(hipEventRecord(ev_spec, st2));
    spec_pending = true; spec2_from = defer_n1; spec3_from = defer_n1 + n2;
    if (defer_n1 + n2 < defer_nsys) {
      const int n3 = defer_nsys - defer_n1 - n2;
      if (solve_chain(st3, n3, defer_n1 + n2)) return -1;
      HIPCK(hipEventRecord(ev_spec3, st3));
      spec3_pending = true;
    }
  }
  if (defer_nsys) {
    if (run_ahead(1)) return -1;
    if (spec_trials >= 2 && !multi()) for (int q = 2; q < defer_nsys; ++q) if (run_ahead(q)) return -1;
  } else if (ahead_single) { if (run_ahead(1)) return -1; }
  // (the head of this trial's state goes out AFTER the next trial ahead has been handed to the second stream: the device needs ~70 us
  //  for the trial's own kernels, the host ~4 us per launch -- the trial ahead must not wait for the host to get through the head's)
  if (with_head && mailbox && !multi()) { if (enqueue_head(st, sys_cur, tr, false)) return -1; mark("head_ahead", st); }
  if (sum_aside()) return -1;
  if (mailbox) {
    // the block (trial results [0..7], iteration-start block [24..28]) is already on its way to the host
    if (wait_mail(0, mail_ticket0, multi() ? MAIL_TICKET : 29)) return -1;
  } else if (read_results(MAIL_TICKET)) return -1;          // trial results [0..7] and, for compute(), the iteration-start block [24..28]
  if (multi()) { tr_pred_ok[sys_cur] = h_res[MAIL_PRED_OK] != 0.0; tr_ovf[sys_cur] = h_res[MAIL_OVERFLOW] != 0.0; }
  if (!nfl_total) h_res[1] = h_res[2] = 0.0;
  h_res[1] += h_res[6]; h_res[2] += h_res[7];
  if (h_res[3] >= 1e9) return persist_fallback(lam, ok2, ni);
  ok2 = (h_res[3] == 0.0);
  timing.n_trials++;
  return 0;
}
// A hand-off of the one-launch factorisation (ba_chol2.h) timed out somewhere in the batch this trial belongs to: its numbers
// are void.  The reduced systems themselves are intact (that factorisation never writes S), but simplest and safest is to redo
// the whole solve of this lambda with the per-step kernels, which this handle then keeps for the rest of its life.
int mcp_ba::persist_fallback(double lam, bool& ok2, double ni) {
  if (!plan.use_persist) { set_err("the factorisation reported a hand-off time-out although the one-launch path is off"); return -1; }
  static std::atomic<int> warned{0};
  if (!warned.exchange(1)) fprintf(stderr, "mcptam_hip: a hand-off of the one-launch Cholesky factorisation timed out; falling back to the per-step kernels\n");
  plan.use_persist = false;
  // A real time-out (not the test hook's) is remembered per device: after the second one the plans of later handles do not take the
  // one-launch path on this device any more -- a handle lives for one BundleAdjust call, and paying the deadline again in every call
  // would be worse than the per-step kernels.  (With claimed work a time-out needs a device that stays saturated for 20 ms.)
  if (plan.persist.test_fail_launch < 0) {
    CpDevice& dv = cp_device(device);
    if (++dv.real_fallbacks >= 2 && !dv.disabled) { dv.disabled = true; fprintf(stderr, "mcptam_hip: device %d: one-launch Cholesky factorisation switched off for this process after %d hand-off time-outs\n", device, dv.real_fallbacks); }
  }
  spec_ok = false;
  if (cancel_spec_trials()) return -1;
  if (join_spec()) return -1;
  HIPCK(hipStreamSynchronize(st));
  if (st2) HIPCK(hipStreamSynchronize(st2));
  HIPCK(hipMemsetAsync(plan.persist.d_err, 0, 2*sizeof(int)*CholPersist::max_sys, st));      // (error words and claim counters)
  timing.n_persist_fallbacks++;
  return solve_trial(lam, ok2, ni);
}

int mcp_ba::compute(volatile unsigned char* abort_flag, int n_iter, double user_lambda) {
  auto t_begin = std::chrono::steady_clock::now();
  HIPCK(hipSetDevice(device));
  (void)hipGetLastError(); launch_err.clear(); plan.launch_failed = nullptr;      // (launch checks below are about THIS solve)
  std::memset(&timing, 0, sizeof timing);
  evs.clear(); ev_used = 0;
  if (n_iter < 0) n_iter = prm.max_iterations;      // 0 runs nothing, as g2o optimize(0) (-> -1 unless externally aborted)
  auto terminate = [&]() { return abort_flag && *abort_flag; };
  // Initialize(), ChainBundle.cc:1284-1298
  outliers.clear(); logs.clear();
  int conv_mag = 0, conv_res = 0;
  auto t_first_end = t_begin;
  static const bool trace_compute = [] { const char* e = getenv("MCP_BA_TRACE"); return e && atoi(e) >= 2; }();
  if (dirty) { if (prepare()) return MCP_ERR_RUNTIME; }
  timing.schur_mfma_per_system = schur_mfma; timing.schur_flops_structural = schur_flops;
  timing.chol_flops_plan = (np > 0 && plan.persist.ok) ? plan.persist.flops : 0.0;
  timing.chol_chains = (np > 0 && plan.persist.ok) ? plan.persist.nseg : 0;
  converged = 0; total_iterations = 0; spec_hot = 0;
  int nCounter = 0;
  // emptiness is decided on the GLOBAL totals: a rank whose shard holds no measurement (or no free point) still runs every
  // kernel with zero-size inputs and joins every collective, or the other ranks would wait in them for ever
  if (nx_total() == 0 || m_total == 0) { nCounter = -1; if (P.nmeas) { launch_chains(cur); launch_eval(cur, false, nullptr); } }
  else {
    tic(ST_EVAL); launch_chains(cur); launch_eval(cur, false, nullptr); toc();
    double ni = 2; bool ok = true; int cj = 0;
    for (int it = 0; it < n_iter && !terminate() && ok; ++it) {
      mcp_ba_iter_log lg; std::memset(&lg, 0, sizeof lg);
      mark("iter", st);
      // preIteration + first robustify: sigma^2 from |chi2| at the iteration-start state
      constexpr int RS = 24;
      const int nbe = (P.nmeas + EVAL_BLOCK - 1)/EVAL_BLOCK;
      // (small bundle: the head of this iteration was enqueued behind the trial that produced this state, before the host knew it
      //  would be accepted -- head_ahead(); all that is left is to make its sigma block the current one)
      bool head_done = (it > 0 && head_ahead_for == cur);
      head_ahead_for = -1;
      if (head_done) { if (robust) flip_sig(); ++dbg_head_ahead; }
      start_blk = nullptr;
      bool lin_done = false;
      if (it > 0 && large_heads() && acc_head_q >= 0 && head_enq[acc_head_q] && head_state[acc_head_q] == cur) {
        // The accepted trial brought its head along (enqueue_head).  The linearisation goes out behind it AT ONCE, on the assumption that
        // the head could finish the median (it nearly always can); only then does the host look at the head's status word -- if the
        // prediction missed (nothing was written to the block), the plain selection runs and the linearisation is simply redone.
        const int q = acc_head_q, keep = sig_idx;
        HeadScratch& H = hsc[head_par*MAX_SYS + q];
        HIPCK(hipStreamWaitEvent(st, head_ev[q], 0));          // (a trial evaluated ahead: its head ran on the head stream)
        sig_idx = 2 + head_par*MAX_SYS + q;
        start_rides = false;
        if (linearize()) return MCP_ERR_RUNTIME;
        const int hs = wait_head(q);
        if (hs < 0) return MCP_ERR_RUNTIME;
        if (hs == 1) {
          start_blk = H.rs;
          // the robust chi2 of the new current state at the new sigma: the host's business (it arrives with the first trial's result), so
          // it is summed BESIDE the linearisation, on the second stream if there is one
          hipStream_t ss = (st2 && ev_sum) ? st2 : st;
          if (ss != st) HIPCK(hipStreamWaitEvent(ss, head_ev[q], 0));
          if (nbe) hipLaunchKernelGGL(k_robust_sum, dim3(nbe), dim3(EVAL_BLOCK), 0, ss, P.nmeas, robust, (const double*)d_chi2[cur].p, (const double*)sig(), H.part);
          hipLaunchKernelGGL(k_final_sums, dim3(1), dim3(256), 0, ss, nbe, (const double*)H.part, 0, (const double*)nullptr, 0, (const double*)nullptr, H.rs, 0, (const int*)nullptr);
          if (ss != st) { HIPCK(hipEventRecord(ev_sum, ss)); sum_pending = true; }
          head_done = true; lin_done = true; ++dbg_head_ahead;
        } else { sig_idx = keep; ++dbg_head_miss; }
      }
      acc_head_q = -1;
      for (int q = 0; q < MAX_SYS; ++q) head_enq[q] = false;      // (what this iteration's trials enqueue goes to the other parity's slots)
      head_par = it & 1;
      head_ahead_want = (small_mode() || large_heads()) && use_mailbox && !prm.profile && it + 1 < n_iter;
      if (head_done) { }
      else if (small_mode()) { if (head_small(cur)) return MCP_ERR_RUNTIME; }
      else if (use_head_large()) { if (head_large(cur, RS, nullptr)) return MCP_ERR_RUNTIME; }
      else {
      if (robust) { if (median_sigma(cur)) return MCP_ERR_RUNTIME; }
      tic(ST_EVAL);
      if (nbe) hipLaunchKernelGGL(k_robust_sum, dim3(nbe), dim3(EVAL_BLOCK), 0, st, P.nmeas, robust, (const double*)d_chi2[cur].p, (const double*)sig(), d_part0.p);
      // iteration-start robust chi2 and the sigma block go to d_res[24..28]; they are read back together with the first
      // trial's results (one host synchronisation less per iteration) -- except in the first iteration, whose lambda comes
      // from the diagonal of the freshly built system
      hipLaunchKernelGGL(k_final_sums, dim3(1), dim3(256), 0, st, nbe, (const double*)d_part0.p, 0, (const double*)nullptr, 0, (const double*)nullptr, d_res.p, RS, (const int*)nullptr);
      toc();
      }
      // several ranks: the sum over the ranks rides on the first trial's all-reduce (d_res[0..3] + d_res[4], see solve_trial);
      // the first iteration needs it before its first trial
      if (it == 0) { if (allreduce(d_res.p + RS, 1, 0, false, "iteration-start chi2")) return MCP_ERR_RUNTIME; }
      start_rides = (it > 0 && multi());
      if (!lin_done) { if (linearize()) return MCP_ERR_RUNTIME; }
      if (it == 0 && !(user_lambda > 0)) {
        // computeLambdaInit [g2o]: 1e-5 * max |H_jj| over the pose and point diagonals; the pose diagonal is summed from
        // the staged blocks (and over the ranks), the point diagonals are rank-local (maximum over ranks taken below)
        if (np && asm_long) hipLaunchKernelGGL(k_udiag_long, dim3((np*ASML_LPE + 255)/256), dim3(256), 0, st, A, np, (const double*)d_stU.p, (const double*)Ubig(), d_udiag.p);
        else if (np) hipLaunchKernelGGL(k_udiag, dim3((np + 255)/256), dim3(256), 0, st, A, np, (const double*)d_stU.p, (const double*)Ubig(), d_udiag.p);
        if (multi() && np) { if (allreduce(d_udiag.p, np, 0, false, "pose diagonal for the initial lambda")) return MCP_ERR_RUNTIME; }
        const int nbm = std::min(256, (std::max(np, nfl) + 1023)/1024);
        if (nbm > 1) {          // (d_part1: the trials' partial sums live there later; nothing of this iteration has used it yet)
          hipLaunchKernelGGL(k_max_diag_part, dim3(nbm), dim3(256), 0, st, np, (const double*)d_udiag.p, 1, nfl, (const double*)d_V.p, d_part1.p);
          hipLaunchKernelGGL(k_max_of, dim3(1), dim3(256), 0, st, nbm, (const double*)d_part1.p, d_res.p + 5);
        } else
        hipLaunchKernelGGL(k_max_diag, dim3(1), dim3(256), 0, st, np, (const double*)d_udiag.p, 1, nfl, (const double*)d_V.p, d_res.p + 5);
      }
      static_assert(RS + 1 == 25, "median_sigma() writes the sigma block to d_res + 25");
      double currentChi = 0, tempChi = 0;
      bool start_pending = true;
      auto take_start = [&]() {
        currentChi = tempChi = h_res[RS];
        { unsigned long long b; std::memcpy(&b, &h_res[RS], 8); if (b == HL_FAILED_BITS) head_failed = true; }
        if (robust) { sigma_sq = h_res[RS + 1]; sigma_sq_lim = h_res[RS + 2]; pred_bin = sel_coarse_bin(h_res[RS + 4]); }
        lg.chi2_start = currentChi; lg.sigma_sq = sigma_sq;
        start_pending = false;
      };
      if (it == 0) { if (read_results(RS + 5)) return MCP_ERR_RUNTIME; take_start(); }
      if (head_failed) { head_failed = false; set_err("the head of an iteration (k_head_large) gave up at a barrier"); return MCP_ERR_RUNTIME; }
      if (it == 0) {
        if (user_lambda > 0) lambda = user_lambda;
        else {
          double md = h_res[5];
          if (multi()) {   // max over ranks of the V diagonals (U is already global)
            if (world > 8) { set_err("more than 8 ranks"); return MCP_ERR_RUNTIME; }
            std::vector<double> slots(world, 0.0); slots[rank] = md;
            HIPCK(hipMemcpyAsync(d_res.p + 16, slots.data(), world*sizeof(double), hipMemcpyHostToDevice, st));
            if (allreduce(d_res.p + 16, world, 0, true, "initial lambda")) return MCP_ERR_RUNTIME;
            HIPCK(hipMemcpy(slots.data(), d_res.p + 16, world*sizeof(double), hipMemcpyDeviceToHost));
            for (double v : slots) md = std::max(md, v);
          }
          lambda = 1e-5*md;
        }
        ni = 2;
      }
      double rho = 0; int qmax = 0; int accepted = 0; double ss_last = 0; double trial_chi_raw = currentChi;
      static const bool trace_trials = [] { const char* e = getenv("MCP_BA_TRACE"); return e && atoi(e) >= 3; }();
      const auto it_t0 = std::chrono::steady_clock::now();
      std::string it_line;
      do {
        bool ok2 = true;
        const int pre0 = dbg_pre;
        if (solve_trial(lambda, ok2, ni)) return MCP_ERR_RUNTIME;
        if (trace_trials) { char b[64]; snprintf(b, sizeof b, " %s%.0f", dbg_pre != pre0 ? "a" : (sys_cur ? "s" : "m"), std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - it_t0).count()); it_line += b; }
        if (start_pending) take_start();
        if (head_failed) { head_failed = false; set_err("the head of an iteration (k_head_large) gave up at a barrier"); return MCP_ERR_RUNTIME; }
        if (test_fail_trial > 0 && ++test_trial_no == test_fail_trial) ok2 = false;      // (test hook: this trial's factorisation "failed")
        double scale, ss;
        trial_chi_raw = h_res[0];
        if (ok2) {
          tempChi = h_res[0]; scale = h_res[1]; ss = h_res[2];
          // the solver's x of the last successful solve (what a later failed solve falls back on): swap, no copy
          std::swap(d_xp_good.p, last_xp->p); std::swap(d_xp_good.n, last_xp->n);
          std::swap(d_xl_good.p, last_xl->p); std::swap(d_xl_good.n, last_xl->n);
        } else {
          // CHOLMOD-failure analogue: the solver's x keeps its previous content [g2o]; recompute the
          // scale terms from it with the current lambda and b
          tempChi = DBL_MAX;
          scale = 0; ss = 0;
          head_ahead_for = -1;                       // (the state is about to be rewritten with the stale step)
          for (int q = 0; q < MAX_SYS; ++q) head_enq[q] = false;
          std::vector<double> xp(np), xl((size_t)nfl*3), bpv(np), gv((size_t)nfl*3);
          if (np) { HIPCK(hipMemcpy(xp.data(), d_xp_good.p, (size_t)np*8, hipMemcpyDeviceToHost)); HIPCK(hipMemcpy(bpv.data(), bp(), (size_t)np*8, hipMemcpyDeviceToHost)); }
          if (nfl) { HIPCK(hipMemcpy(xl.data(), d_xl_good.p, (size_t)nfl*24, hipMemcpyDeviceToHost)); HIPCK(hipMemcpy(gv.data(), d_g.p, (size_t)nfl*24, hipMemcpyDeviceToHost)); }
          for (int j = 0; j < np; ++j) { scale += xp[j]*(lambda*xp[j] + bpv[j]); ss += xp[j]*xp[j]; }
          double sl = 0, sq = 0;
          for (size_t j = 0; j < xl.size(); ++j) { sl += xl[j]*(lambda*xl[j] + gv[j]); sq += xl[j]*xl[j]; }
          if (multi()) {   // point parts are rank-local
            double v2[2] = { sl, sq };
            HIPCK(hipMemcpy(d_res.p + 16, v2, 16, hipMemcpyHostToDevice));
            if (allreduce(d_res.p + 16, 2, 0, true, "step statistics after a failed factorisation")) return MCP_ERR_RUNTIME;
            HIPCK(hipMemcpy(v2, d_res.p + 16, 16, hipMemcpyDeviceToHost)); sl = v2[0]; sq = v2[1];
          }
          scale += sl; ss += sq;
          // g2o goes on with update(_solver->x()) -- the solver's x is what the last successful solve left -- and computeActiveErrors:
          // the edges then hold the errors of (current state (+) stale x), which is what the residual action reads if this was the
          // iteration's last trial (OptimizationAlgorithmLevenberg::solve [3P-memory]; ChainBundle.cc:1096-1116).  Same here.
          {
            const int tr = last_tr;
            hipLaunchKernelGGL(k_update_poses, dim3(1), dim3(256), 0, st, P, lambda, (const double*)d_xp_good.p, (const double*)bp(), (const double*)d_pose[cur].p, d_pose[tr].p, d_res.p + 6, d_xp_cand.p);
            if (nfl) hipLaunchKernelGGL(k_apply_point_step, dim3((nfl + 255)/256), dim3(256), 0, st, P, (const double*)d_xl_good.p, (const double*)d_pt[cur].p, d_pt[tr].p);
            launch_chains(tr);
            launch_eval(tr, true, nullptr);
            const int nbe2 = (P.nmeas + EVAL_BLOCK - 1)/EVAL_BLOCK;
            hipLaunchKernelGGL(k_final_sums, dim3(1), dim3(256), 0, st, nbe2, (const double*)d_part0.p, 0, (const double*)nullptr, 0, (const double*)nullptr, d_res.p, 0, (const int*)nullptr);
            if (allreduce(d_res.p, 1, 0, false, "chi2 of the stale step after a failed factorisation")) return MCP_ERR_RUNTIME;
            if (read_results(1)) return MCP_ERR_RUNTIME;
            trial_chi_raw = h_res[0];
            sel_src = -1;
          }
        }
        ss_last = ss;
        rho = currentChi - tempChi;
        scale += 1e-3;
        rho /= scale;
        if (rho > 0 && std::isfinite(tempChi)) {
          double alpha = 1. - std::pow((2*rho - 1), 3);
          alpha = std::min(alpha, 2./3.);
          const double sf = std::max(1./3., alpha);
          lambda *= sf; ni = 2; currentChi = tempChi; accepted = 1;
          cur = last_tr;                             // discardTop: the trial state becomes current
          acc_head_q = ok2 ? sys_cur : -1;           // (its head, if one was enqueued behind it, is for exactly this state)
          // ... and the digit histograms that rode on its all-reduce describe the new chi2 array -- unless the factorisation had failed:
          // then the accepted state is the STALE step's (negative scale), whose chi2 never rode on anything
          sel_src = (multi() && ok2) ? sys_cur : -1;
        } else {
          lambda *= ni; ni *= 2; accepted = 0;       // pop: the current buffers were never touched
          spec_hot = 4;                              // (small bundles: from here on the rejection branch's systems are solved ahead again)
        }
        ++qmax;
      } while (rho < 0 && qmax < prm.max_trials_after_failure && !terminate());
      if (trace_trials) fprintf(stderr, "[mcp_ba trials] it %d: results at (us; m = solved for this trial, a = evaluated ahead, s = system of the batch, trial run now)%s\n", it, it_line.c_str());
      if (cancel_spec_trials()) return MCP_ERR_RUNTIME;
      ok = !(qmax == prm.max_trials_after_failure || rho == 0);
      ++cj;
      // post-iteration actions
      const double rms = std::sqrt(ss_last/(double)nx_total());
      if (rms < prm.update_rms_limit && !prm.disable_convergence) { conv_mag = 1; if (abort_flag) *abort_flag = 1; }
      {
        // activeRobustChi2 of whatever errors the edges hold: the last trial's (or, when verbose, the
        // recomputed errors of the current state)
        // (an accepted trial's errors ARE the edges' errors; only after a failed factorisation whose stale step was accepted --
        // negative scale -- does that differ from the schedule's currentChi, which is DBL_MAX then)
        double curchi = accepted ? trial_chi_raw : (verbose ? currentChi : trial_chi_raw);
        const double pct = (last_chi2_action - curchi)/last_chi2_action;
        if (!prm.disable_convergence) {
          if (pct >= 0 && pct <= prm.update_percent_limit) { conv_res = 1; if (abort_flag) *abort_flag = 1; }
          else if (curchi == 0) { conv_res = 1; if (abort_flag) *abort_flag = 1; }
        }
        last_chi2_action = curchi;
        lg.chi2_end = curchi;
      }
      if (qmax == 1 && spec_hot > 0) --spec_hot;     // (an iteration that accepted its first trial)
      total_iterations += qmax;
      lg.lambda_end = lambda; lg.trials = qmax; lg.accepted = accepted; lg.rms_update = rms;
      logs.push_back(lg);
      if (it == 0) t_first_end = std::chrono::steady_clock::now();
    }
    nCounter = cj;
  }
  const auto t_loop_end = std::chrono::steady_clock::now();
  if (join_spec()) return MCP_ERR_RUNTIME;
  if (join_sum()) return MCP_ERR_RUNTIME;
  if (evt_debug) { fprintf(stderr, "[evt] iteration heads enqueued ahead and used: %d (median prediction missed: %d)\n", dbg_head_ahead, dbg_head_miss); dbg_head_ahead = 0; dbg_head_miss = 0; }
#ifdef MCP_HL_PROF
  if (evt_debug) {
    (void)hipDeviceSynchronize();
    long long pr[16]; (void)hipMemcpyFromSymbol(pr, HIP_SYMBOL(g_hl_prof), sizeof pr);
    fprintf(stderr, "[hl prof] workgroup 0 (10 ns ticks): A %lld | barrier %lld | B %lld | barrier %lld | C %lld | barrier %lld | D %lld | barrier %lld | E %lld | last workgroup done +%lld\n",
            pr[1] - pr[0], pr[2] - pr[1], pr[3] - pr[2], pr[4] - pr[3], pr[5] - pr[4], pr[6] - pr[5], pr[7] - pr[6], pr[8] - pr[7], pr[9] - pr[8], pr[10] - pr[9]);
    static long long wgs[256][4]; (void)hipMemcpyFromSymbol(wgs, HIP_SYMBOL(g_hl_wg), sizeof wgs);
    long long lo[3] = {1ll << 62, 1ll << 62, 1ll << 62}, hi[3] = {0, 0, 0};
    for (int w = 0; w < std::min(256, hl_grid); ++w) for (int j = 0; j < 3; ++j) { lo[j] = std::min(lo[j], wgs[w][j] - pr[0]); hi[j] = std::max(hi[j], wgs[w][j] - pr[0]); }
    fprintf(stderr, "[hl prof] over the workgroups, ticks from workgroup 0's start: local histogram done %lld..%lld, flushed %lld..%lld, barrier passed %lld..%lld\n", lo[0], hi[0], lo[1], hi[1], lo[2], hi[2]);
  }
#endif
#ifdef MCP_HS_PROF
  if (evt_debug) {
    (void)hipDeviceSynchronize();
    unsigned long long pr[16]; (void)hipMemcpyFromSymbol(pr, HIP_SYMBOL(g_hs_prof), sizeof pr);
    fprintf(stderr, "[hs prof] sweep1 %llu  find0 %llu  (sweep1b) %llu  find1 %llu  sweep2 %llu  lds-select %llu  sigma %llu  sum-sweep %llu  final %llu  (clock64 ticks; %llu candidates)\n",
            pr[1] - pr[0], pr[2] - pr[1], pr[3] - pr[2], pr[4] - pr[3], pr[5] - pr[4], pr[6] - pr[5], pr[7] - pr[6], pr[8] - pr[7], pr[9] - pr[8], pr[10]);
  }
#endif
  if (evt_debug) { fprintf(stderr, "[evt] trials evaluated ahead and used: %d; host wait on mailbox: own trials %.0f us, ahead trials %.0f us\n", dbg_pre, dbg_wait_us[0], dbg_wait_us[1]); dbg_pre = 0; dbg_wait_us[0] = dbg_wait_us[1] = 0; }
  evt_flush();
  int rc = final_stats(nCounter);
  const auto t_stats_end = std::chrono::steady_clock::now();
  if (rc == MCP_ERR_RUNTIME) return rc;
  converged = (conv_mag || conv_res);
  bool external_abort = terminate() && !converged;
  if (nCounter == 0 && !external_abort) rc = -1;
  else if (nCounter == 0 && terminate()) rc = 0;
  else rc = nCounter;
  if (download_state()) return MCP_ERR_RUNTIME;
  // A launch the runtime refused (a configuration a kernel cannot run with) leaves no trace in the stream: the state simply stays
  // what it was.  Kernel launches are not checked one by one; whatever one of them reported is still the thread's last error here.
  // The launches whose configuration a device can refuse (dynamic LDS, persistent grids) are checked where they are made
  // (note_launch / chol_persist_*): the first refusal is reported here by kernel name.  (Round 4 looked at the thread's sticky last
  // error at this point, which also caught benign leftovers of unrelated runtime calls and blamed "a launch".)
  if (plan.launch_failed && launch_err.empty()) launch_err = std::string(plan.launch_failed) + ": launch refused";
  if (!launch_err.empty()) { set_err("a launch of this solve failed: " + launch_err); launch_err.clear(); plan.launch_failed = nullptr; return MCP_ERR_RUNTIME; }
  // stage timings
  if (prm.profile) {
    (void)hipStreamSynchronize(st);
    double acc[ST_N] = {0};
    for (auto& e : evs) { float ms = 0; if (hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) acc[e.stage] += ms; }
    timing.eval_ms = acc[ST_EVAL]; timing.select_ms = acc[ST_SELECT]; timing.linearize_ms = acc[ST_LIN];
    timing.schur_ms = acc[ST_SCHUR]; timing.cholesky_ms = acc[ST_CHOL]; timing.solve_ms = acc[ST_SOLVE]; timing.update_ms = acc[ST_UPDATE];
  }
  timing.total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
  if (trace_compute) {
    auto ms = [&](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    fprintf(stderr, "[compute] %d iterations: to the end of the first %.3f ms, the others %.3f ms, final statistics + export %.3f ms, state and outliers on the host %.3f ms; total %.3f ms\n",
            nCounter, ms(t_begin, t_first_end), ms(t_first_end, t_loop_end), ms(t_loop_end, t_stats_end), ms(t_stats_end, std::chrono::steady_clock::now()), timing.total_ms);
  }
  return rc;
}

// ChainBundle.cc:1339-1345 (final sigma^2), :1368-1399 (Tukey outliers), :1401-1448 (depth covariance)
int mcp_ba::final_stats(int nCounter) {
  exported = false;
  if (m_total == 0 || dirty) { max_cov = 0; return 0; }
  // Behind the kernels that leave the median in the result block, and BEFORE the one wait for that block: the Tukey flags (threshold from the
  // median on the device) and the state, both written to pinned host memory by kernels -- one wait at the end of a solve instead of three,
  // no copy into pageable memory.
  const size_t exp_state = (poses.size()*12 + points.size()*3)*sizeof(double);
  const bool flags_dev = tukey && nCounter != 0 && P.nmeas > 0;
  auto enqueue_export = [&](int med_idx) -> int {
    if (ensure_export(exp_state + ((size_t)P.nmeas + 63)/64*8)) return -1;
    if (flags_dev) hipLaunchKernelGGL(k_tukey_flags_dev, dim3((P.nmeas + 255)/256), dim3(256), 0, st, P.nmeas, (const double*)d_chi2[cur].p, (const double*)d_res.p, med_idx, (double)m_total,
                                      prm.min_mestimator_sigma, reinterpret_cast<unsigned long long*>(h_exp + exp_state));
    const size_t nps = poses.size()*12, npt = points.size()*3;
    if (flags_dev) note_launch("k_tukey_flags_dev");
    if (nps + npt) hipLaunchKernelGGL(k_export_state, dim3((unsigned)((nps + npt + 255)/256)), dim3(256), 0, st, nps, npt, (const double*)d_pose[cur].p, (const double*)d_pt[cur].p, reinterpret_cast<double*>(h_exp));
    if (nps + npt) note_launch("k_export_state");
    // (these two write the pinned block download_state() and the outlier list read; a launch the runtime refused must not
    //  pass for an export -- compute() then reports launch_err, and `exported` stays false so nothing is read from the stale block)
    if (!launch_err.empty()) { set_err("a launch of this solve failed: " + launch_err); return -1; }      // (the first refusal of the solve, by kernel name)
    return 0;
  };
  // a trial evaluated ahead that nobody consumed may still be running on the second stream and reads the sigma block of the parity
  // median_sigma() is about to rewrite: order everything behind it first (it had only ever been joined by the next solve)
  if (join_spec()) return -2;
  if (small_mode() && robust) {
    // small bundle: median, sigma block and robust chi2 of the final state in one launch (ba_small.h), one read-back
    if (head_small(cur)) return -2;
    if (enqueue_export(28)) return -2;
    if (read_results(29)) return -2;
    exported = true;
    h_res[0] = h_res[24];
    for (int i = 0; i < 4; ++i) h_res[9 + i] = h_res[25 + i];
  } else {
  const auto tf0 = std::chrono::steady_clock::now();
  if (use_head_large()) { if (head_large(cur, 0, d_res.p + 9)) return -2; }      // (the sigma block lands in d_res[9..12] too: no copy behind it)
  else {
  if (median_sigma(cur)) return -2;
  const int nbe = (P.nmeas + EVAL_BLOCK - 1)/EVAL_BLOCK;
  if (nbe) hipLaunchKernelGGL(k_robust_sum, dim3(nbe), dim3(EVAL_BLOCK), 0, st, P.nmeas, robust, (const double*)d_chi2[cur].p, (const double*)sig(), d_part0.p);
  hipLaunchKernelGGL(k_final_sums, dim3(1), dim3(256), 0, st, nbe, (const double*)d_part0.p, 0, (const double*)nullptr, 0, (const double*)nullptr, d_res.p, 0, (const int*)nullptr);
  if (allreduce(d_res.p, 1, 0, false, "final robust chi2")) return -2;
  HIPCK(hipMemcpyAsync(d_res.p + 9, sig(), 4*sizeof(double), hipMemcpyDeviceToDevice, st));
  }
  if (enqueue_export(12)) return -2;
  const auto tf1 = std::chrono::steady_clock::now();
  if (read_results(13)) return -2;
  { unsigned long long b; std::memcpy(&b, &h_res[0], 8); if (b == HL_FAILED_BITS) { set_err("the final statistics (k_head_large) gave up at a barrier"); return -2; } }
  exported = true;
  { static const bool tr = [] { const char* e = getenv("MCP_BA_TRACE"); return e && atoi(e) >= 2; }();
    if (tr) fprintf(stderr, "[final] median + sums + export enqueued in %.3f ms, waited for %.3f ms\n", std::chrono::duration<double, std::milli>(tf1 - tf0).count(), std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tf1).count()); }
  }
  if (robust) { sigma_sq = h_res[9]; sigma_sq_lim = h_res[10]; }
  mean_chi2 = h_res[0]/m_total;
  const double median = h_res[12];
  if (nCounter == 0) return 0;
  if (tukey) {
    double s = 1.4826*(1 + 5.0/mest_denom(m_total))*std::sqrt(median);     // Tukey::FindSigmaSquared, MEstimator.h:109-124
    s = 4.6851*s;
    double s2 = s*s;
    const double mins = prm.min_mestimator_sigma*prm.min_mestimator_sigma;
    if (s2 < mins) s2 = mins;
    (void)s2;                                      // (the flags were taken on the device with this very threshold: k_tukey_flags_dev, above)
    // The flags arrive as one bit per measurement in the structure's order.  Set bits only are visited: structure order -> add order
    // through a bit map of the add indices (the list is in add order, as the reference's walk over its edges), then the three
    // numbers of every outlier with the look-ups requested a few entries ahead -- each is a miss in a 16 MB array on a cold cache.
    // (Round 6: byte flags walked one by one + the list built on demand misses took 2.3 ms of a 15 ms call at the metric size.)
    const auto tq0 = std::chrono::steady_clock::now();
    const unsigned long long* mk = reinterpret_cast<const unsigned long long*>(h_exp + exp_state);
    const int nw = (P.nmeas + 63)/64;
    std::vector<unsigned long long> addmask((meas.size() + 63)/64, 0ull);
    size_t nout = 0;
    for (int w = 0; w < nw; ++w) {
      unsigned long long bts = mk[w];
      while (bts) {
        const int j = w*64 + __builtin_ctzll(bts); bts &= bts - 1;
        const int a = perm[j];
        if (a >= 0) { addmask[(size_t)a >> 6] |= 1ull << (a & 63); ++nout; }      // (perm < 0: a measurement this map does not have -- near miss; its chi2 is 0)
      }
    }
    const auto tq1 = std::chrono::steady_clock::now();
    std::vector<int> idx; idx.reserve(nout);
    for (size_t w = 0; w < addmask.size(); ++w) {
      unsigned long long bts = addmask[w];
      while (bts) { idx.push_back((int)(w*64 + __builtin_ctzll(bts))); bts &= bts - 1; }
    }
    const auto tq2 = std::chrono::steady_clock::now();
    outliers.resize(3*idx.size());
    const size_t no = idx.size();
    constexpr size_t AHEAD = 16;
    for (size_t k = 0; k < std::min(no, AHEAD); ++k) __builtin_prefetch(&meas[idx[k]]);
    for (size_t k = 0; k < std::min(no, AHEAD/2); ++k) __builtin_prefetch(&points[meas[idx[k]].point]);
    for (size_t k = 0; k < no; ++k) {
      if (k + AHEAD < no) __builtin_prefetch(&meas[idx[k + AHEAD]]);
      if (k + AHEAD/2 < no) __builtin_prefetch(&points[meas[idx[k + AHEAD/2]].point]);
      const HMeas& m = meas[idx[k]];
      outliers[3*k] = points[m.point].id;
      outliers[3*k + 1] = poses[chains[m.chain].v[0]].id;      // vertices().front(), :1394
      outliers[3*k + 2] = m.cam;
    }
    { static const bool tr = [] { const char* e = getenv("MCP_BA_TRACE"); return e && atoi(e) >= 2; }();
      auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
      if (tr) fprintf(stderr, "[final] outliers: set bits to add order %.3f ms, index list %.3f ms, %zu triples %.3f ms\n", ms(tq0, tq1), ms(tq1, tq2), no, ms(tq2, std::chrono::steady_clock::now())); }
  }
  // depth covariance only when fewer than 3 free poses (:1419); Hessian of the last buildSystem, no lambda
  if (nfp < 3 && nCounter > 0) {
    // (several ranks: S is the all-reduced system, every rank inverts it, the point covariances are rank-local and their
    // median is the global order statistic -- same collectives on every rank, nfp and nfl_total are global)
    bool okm = true;
    sys_cur = 0; spec_ok = false;
    SysBatch sb; std::memset(&sb, 0, sizeof sb);
    if (join_spec()) return -2;
    HIPCK(hipMemsetAsync(d_fail.p, 0, 4*sizeof(int), st));
    if (build_system(1, sb)) return -2;
    const size_t n2 = (size_t)np*np;
    hipLaunchKernelGGL(k_final_sums, dim3(1), dim3(256), 0, st, 0, (const double*)nullptr, 0, (const double*)nullptr, 0, (const double*)nullptr, d_res.p, 0, (const int*)d_fail.p);
    if (allreduce(d_res.p + 3, 1, 0, false, "covariance: failure flag")) return -2;
    std::vector<double> Sh(n2 + 1), Sinv(n2 + 1, 0.0);
    if (np) HIPCK(hipMemcpyAsync(Sh.data(), S(), n2*8, hipMemcpyDeviceToHost, st));
    HIPCK(hipMemcpyAsync(h_res + 3, d_res.p + 3, sizeof(double), hipMemcpyDeviceToHost, st));
    HIPCK(hipStreamSynchronize(st));
    if (h_res[3] != 0.0) okm = false;
    if (okm && np) {
      // <= 12x12 control-plane inverse on the host (lower triangle of S is valid)
      std::vector<double> L(n2, 0.0);
      for (int i = 0; i < np && okm; ++i) for (int j = 0; j <= i; ++j) {
        double sacc = Sh[(size_t)i*np + j];
        for (int k = 0; k < j; ++k) sacc -= L[(size_t)i*np + k]*L[(size_t)j*np + k];
        if (i == j) { if (!(sacc > 0)) { okm = false; break; } L[(size_t)i*np + j] = std::sqrt(sacc); }
        else L[(size_t)i*np + j] = sacc/L[(size_t)j*np + j];
      }
      for (int c = 0; c < np && okm; ++c) {
        std::vector<double> e(np, 0.0); e[c] = 1;
        for (int i = 0; i < np; ++i) { double sacc = e[i]; for (int k = 0; k < i; ++k) sacc -= L[(size_t)i*np + k]*e[k]; e[i] = sacc/L[(size_t)i*np + i]; }
        for (int i = np - 1; i >= 0; --i) { double sacc = e[i]; for (int k = i + 1; k < np; ++k) sacc -= L[(size_t)k*np + i]*e[k]; e[i] = sacc/L[(size_t)i*np + i]; }
        for (int i = 0; i < np; ++i) Sinv[(size_t)i*np + c] = e[i];
      }
    }
    if (okm) {
      if (nfl_total > 0) {
        DevBuf<double> dSinv;
        std::vector<double> tmp(Sinv.begin(), Sinv.begin() + std::max<size_t>(n2, 1));
        if (dSinv.upload(tmp, st)) return -2;
        if (nfl) hipLaunchKernelGGL(k_point_cov22, dim3((nfl + 63)/64), dim3(64), 0, st, P, (const double*)d_Vinv.p, (const double*)d_W.p, (const double*)dSinv.p, d_cov.p);
        if (select_kth(d_cov.p, nfl, (unsigned long long)(nfl_total/2), d_res.p + 8)) return -2;
        HIPCK(hipMemcpyAsync(h_res, d_res.p + 8, sizeof(double), hipMemcpyDeviceToHost, st));
        HIPCK(hipStreamSynchronize(st));
        max_cov = h_res[0];
      } else max_cov = DBL_MAX;                                   // :1441
    } else max_cov = 0;                                           // :1447
  } else max_cov = 0;                                             // :1444-1448
  return 0;
}

// ------------------------------------------------------------------------------------------ C ABI
extern "C" {

mcp_ba* mcp_ba_create(const mcp_camera* cams, int ncam, int use_robust, int use_tukey, int verbose,
                      const mcp_ba_params* params) {
  if (!cams || ncam <= 0 || ncam > 255) { set_err("mcp_ba_create: need 1..255 cameras"); return nullptr; }
  for (int i = 0; i < ncam; ++i)
    if (cams[i].n_inv < 0 || cams[i].n_inv > MCP_MAX_INV) { set_err("mcp_ba_create: bad inverse polynomial length"); return nullptr; }      // 0 = Newton mode
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { set_err("mcp_ba_create: no HIP device available (the HIP path has no CPU fallback)"); return nullptr; }
  mcp_ba_params p;
  p.max_iterations = 100; p.max_trials_after_failure = 100; p.update_percent_limit = 1e-10; p.update_rms_limit = 1e-10;
  p.min_mestimator_sigma = 0.5; p.disable_convergence = 0; p.device = -1; p.profile = 0;
  if (params) p = *params;
  int dev = p.device;
  if (dev < 0) { if (hipGetDevice(&dev) != hipSuccess) dev = 0; }
  if (dev >= ndev) { set_err("mcp_ba_create: device ordinal out of range"); return nullptr; }
  if (!is_gfx950(dev)) { set_err("mcp_ba_create: device is not gfx950 (MI355X); this library carries gfx950 code only"); return nullptr; }
  if (hipSetDevice(dev) != hipSuccess) { set_err("hipSetDevice failed"); return nullptr; }
  mcp_ba* h = new mcp_ba();
  h->device = dev; h->prm = p; h->robust = use_robust ? 1 : 0; h->tukey = use_tukey ? 1 : 0; h->verbose = verbose;
  h->cams.assign(cams, cams + ncam);
  std::memset(&h->timing, 0, sizeof h->timing);
  std::memset(&h->P, 0, sizeof h->P);
  const bool use_pool = [] { const char* e = getenv("MCP_BA_STREAM_POOL"); return e ? atoi(e) != 0 : true; }();
  SolverStreams* pool = use_pool ? &solver_streams(dev) : nullptr;
  if (pool && pool->ok) { h->pooled = true; h->st = pool->main; }
  else if (hipStreamCreateWithFlags(&h->st, hipStreamNonBlocking) != hipSuccess) { set_err("hipStreamCreate failed"); delete h; return nullptr; }
  { const char* e = getenv("MCP_BA_OVERLAP"); if (e) { h->overlap_spec = atoi(e); h->overlap_auto = false; } }
  { const char* e = getenv("MCP_BA_MAILBOX"); if (e) h->use_mailbox = atoi(e); }
  { const char* e = getenv("MCP_BA_MAIN_SYS"); if (e) h->main_sys = atoi(e); }
  { const char* e = getenv("MCP_BA_EVT"); if (e) h->evt_debug = atoi(e); }
  { const char* e = getenv("MCP_BA_SPEC_TRIALS"); if (e) h->spec_trials = atoi(e); }
  { const char* e = getenv("MCP_BA_HEAD_AHEAD"); if (e) h->large_head_ahead = atoi(e); }
  { const char* e = getenv("MCP_BA_NEAR_MISS"); if (e) h->near_miss_on = atoi(e); }
  { const char* e = getenv("MCP_BA_FORCE_MULTI"); if (e) h->force_multi = atoi(e); }
  { const char* e = getenv("MCP_BA_TEST_FAIL_TRIAL"); if (e) h->test_fail_trial = atoi(e); }
  { const char* e = getenv("MCP_BA_SPEC_DELAY"); if (e) h->spec_delay = atoi(e); }
  { const char* e = getenv("MCP_BA_SMALL"); if (e) h->small_on = atoi(e); }
  { const char* e = getenv("MCP_BA_SELECT_RIDE"); if (e) h->sel_ride = atoi(e); }
  { const char* e = getenv("MCP_BA_TIMEOUT_MS"); if (e && atof(e) > 0) h->timeout_ms = atof(e); }
  for (int q = 0; q < mcp::MAX_SYS; ++q) if (hipEventCreateWithFlags(&h->ev_tr[q], hipEventDisableTiming) != hipSuccess ||
                                              hipEventCreateWithFlags(&h->ev_wf[q], hipEventDisableTiming) != hipSuccess) { set_err("hipEventCreate failed"); delete h; return nullptr; }
  // (the runtime multiplexes streams onto a handful of hardware queues -- GPU_MAX_HW_QUEUES, 4 by default: main, speculative and two
  // trial streams use them up; a third stream for MCP_BA_OVERLAP=2 is only created when asked for)
  if (h->pooled) {
    if (h->spec_trials >= 2) for (int q = 2; q < mcp::MAX_SYS; ++q) h->st_tr[q] = pool->tr[q];
    if (h->overlap_spec) { h->st2 = pool->spec; if (h->overlap_spec >= 2) h->st3 = pool->spec3; }
  } else {
    if (h->spec_trials >= 2) for (int q = 2; q < mcp::MAX_SYS; ++q) if (hipStreamCreateWithFlags(&h->st_tr[q], hipStreamNonBlocking) != hipSuccess) { set_err("hipStreamCreate failed"); delete h; return nullptr; }
    if (h->overlap_spec && (hipStreamCreateWithFlags(&h->st2, hipStreamNonBlocking) != hipSuccess || (h->overlap_spec >= 2 && hipStreamCreateWithFlags(&h->st3, hipStreamNonBlocking) != hipSuccess))) { set_err("second stream could not be created"); delete h; return nullptr; }
  }
  if (h->large_head_ahead) {
    // the head stream: the pool's fourth side stream (measured to overlap with the main one if a hardware queue was left), else an own one
    if (h->pooled && pool->spec3 && pool->spec3 != h->st3) h->st_h = pool->spec3;
    else if (hipStreamCreateWithFlags(&h->st_h, hipStreamNonBlocking) == hipSuccess) h->st_h_own = true;
    else { (void)hipGetLastError(); h->st_h = nullptr; }
    for (int q = 0; q < mcp::MAX_SYS; ++q) if (hipEventCreateWithFlags(&h->head_ev[q], hipEventDisableTiming) != hipSuccess) { set_err("hipEventCreate failed"); delete h; return nullptr; }
  }
  if (h->overlap_spec && (
                          hipEventCreateWithFlags(&h->ev_spec3, hipEventDisableTiming) != hipSuccess ||
                          hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) != hipSuccess ||
                          hipEventCreateWithFlags(&h->ev_spec, hipEventDisableTiming) != hipSuccess ||
                          hipEventCreateWithFlags(&h->ev_head, hipEventDisableTiming) != hipSuccess ||
                          hipEventCreateWithFlags(&h->ev_sum, hipEventDisableTiming) != hipSuccess)) { set_err("second stream / events could not be created"); delete h; return nullptr; }
  return h;
}
void mcp_ba_destroy(mcp_ba* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  h->drain();                          // nothing of this handle is in flight any more ...
  mcp::DevCache::Quiesced q;           // ... so its blocks can go straight back to the cache (ba_pool.h)
  delete h;
}

int mcp_ba_add_pose(mcp_ba* h, const double R[9], const double t[3], int fixed) {
  HPose p; std::memset(&p, 0, sizeof p);
  std::memcpy(p.T, R, 72); std::memcpy(p.T + 9, t, 24); p.fixed = fixed ? 1 : 0; p.unk = -1;
  p.id = h->new_id(1, (int)h->poses.size());
  h->poses.push_back(p); h->dirty = true;
  return p.id;
}
int mcp_ba_add_point(mcp_ba* h, const double x[3], const int* chain, int n, int fixed) {
  int c = h->find_chain(chain, n);
  if (c < 0) { set_err("mcp_ba_add_point: bad chain"); return -1; }
  HPoint p; std::memset(&p, 0, sizeof p);
  std::memcpy(p.x, x, 24); p.chain = c; p.fixed = fixed ? 1 : 0; p.unk = -1;
  p.id = h->new_id(2, (int)h->points.size());
  h->points.push_back(p); h->dirty = true;
  return p.id;
}
int mcp_ba_add_meas(mcp_ba* h, const int* chain, int n, int point_id, const double uv[2], double sigma_sq, int cam_index) {
  if (point_id <= 0 || point_id >= h->next_id || h->id_kind[point_id] != 2) { set_err("mcp_ba_add_meas: unknown point id"); return -1; }
  if (cam_index < 0 || cam_index >= (int)h->cams.size()) { set_err("mcp_ba_add_meas: bad camera index"); return -1; }
  int c = h->find_chain(chain, n);
  if (c < 0) { set_err("mcp_ba_add_meas: bad chain"); return -1; }
  HMeas m; m.chain = c; m.point = h->id_index[point_id]; m.cam = cam_index; m.u = uv[0]; m.v = uv[1];
  m.omega = 1/std::sqrt(sigma_sq);                    // information = I / sqrt(sigma^2), ChainBundle.cc:1244-1245
  h->meas.push_back(m); h->meas_point.push_back(m.point); h->meas_chain.push_back(m.chain); h->dirty = true;
  return 0;
}
int mcp_ba_add_points(mcp_ba* h, int count, const double* x, const int* chains, int stride, const int* chain_len,
                      const unsigned char* fixed, int* ids_out) {
  static const bool trace = getenv("MCP_BA_TRACE") != nullptr;
  const auto t0 = std::chrono::steady_clock::now();
  // (reserve(size + count) allocates exactly that much with libstdc++ -- an adapter adding points in many small batches would
  //  reallocate and copy the whole arrays every call; grow geometrically instead, and only when the batch does not fit)
  auto grow = [&](auto& v) { const size_t need = v.size() + (size_t)count; if (need > v.capacity()) v.reserve(std::max(need, 2*v.capacity())); };
  grow(h->points); grow(h->id_kind); grow(h->id_index);
  for (int i = 0; i < count; ++i) {
    int id = mcp_ba_add_point(h, x + 3*(size_t)i, chains + (size_t)stride*i, chain_len[i], fixed ? fixed[i] : 0);
    if (id < 0) return -1;
    if (ids_out) ids_out[i] = id;
  }
  if (trace) fprintf(stderr, "[mcp_ba add] %d points %.3f ms\n", count, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
  return 0;
}
int mcp_ba_add_measurements(mcp_ba* h, int count, const int* chains, int stride, const int* chain_len, const int* point_ids,
                            const double* uv, const double* sigma_sq, const int* cam_index) {
  // one crossing for a whole map: the arrays are sized once and filled in place by the worker pool (same checks and the same
  // records as `count` calls of mcp_ba_add_meas; on a bad row the rows before it stay added, as they would).  Pose chains that
  // are known already -- all of them, normally: an observer chain (MKF, camera) was met as some point's source chain -- are
  // resolved read-only by the threads; rows with a new chain are completed afterwards in row order, so chains are numbered as
  // a serial replay numbers them.
  if (count <= 0) return 0;
  static const bool trace = getenv("MCP_BA_TRACE") != nullptr;
  const auto t0 = std::chrono::steady_clock::now();
  struct Tr { bool on; int n; std::chrono::steady_clock::time_point t0; double sized = 0; ~Tr() { if (on) fprintf(stderr, "[mcp_ba add] %d measurements %.3f ms (arrays sized after %.3f)\n", n, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), sized); } } tr{trace, count, t0};
  const size_t n0 = h->meas.size();
  h->meas.resize(n0 + count); h->meas_point.resize(n0 + count); h->meas_chain.resize(n0 + count);
  tr.sized = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  HMeas* mo = h->meas.data() + n0; int* po = h->meas_point.data() + n0; int* co = h->meas_chain.data() + n0;
  const int ncam = (int)h->cams.size(), next_id = h->next_id;
  const int* kind = h->id_kind.data(); const int* index = h->id_index.data();
  HostPool& pool = host_pool();
  const int T = (count >= 32768) ? pool.size() : 1;
  std::vector<int> bad_row(T, count), bad_why(T, 0);
  auto body = [&](int tid) {
    const int lo = (int)((long)count*tid/T), hi = (int)((long)count*(tid + 1)/T);
    int last_n = 0, last_idx = -3; const int* last_ids = nullptr;
    for (int i = lo; i < hi; ++i) {
      const int pid = point_ids[i], cam = cam_index[i];
      if (pid <= 0 || pid >= next_id || kind[pid] != 2) { bad_row[tid] = i; bad_why[tid] = 1; return; }
      if (cam < 0 || cam >= ncam) { bad_row[tid] = i; bad_why[tid] = 2; return; }
      const int* ids = chains + (size_t)stride*i; const int n = chain_len[i];
      int c;
      if (last_idx >= -1 && n == last_n && std::memcmp(ids, last_ids, sizeof(int)*n) == 0) c = last_idx;
      else { c = h->lookup_chain(ids, n); if (c == -2) { bad_row[tid] = i; bad_why[tid] = 3; return; } last_n = n; last_idx = c; last_ids = ids; }
      HMeas& m = mo[i];
      m.chain = c; m.point = index[pid]; m.cam = cam; m.u = uv[2*(size_t)i]; m.v = uv[2*(size_t)i + 1];
      m.omega = 1/std::sqrt(sigma_sq[i]);                   // information = I / sqrt(sigma^2), ChainBundle.cc:1244-1245
      po[i] = m.point; co[i] = c;
    }
  };
  if (T == 1) body(0); else pool.run(body);
  int ibad = count, why = 0;
  for (int t = 0; t < T; ++t) if (bad_row[t] < ibad) { ibad = bad_row[t]; why = bad_why[t]; }
  for (int i = 0; i < ibad; ++i) if (co[i] == -1) {          // chains met for the first time, in row order
    const int c = h->find_chain(chains + (size_t)stride*i, chain_len[i]);
    if (c < 0) { ibad = i; why = 3; break; }
    co[i] = c; mo[i].chain = c;
  }
  h->dirty = true;
  if (ibad < count) {
    h->meas.resize(n0 + ibad); h->meas_point.resize(n0 + ibad); h->meas_chain.resize(n0 + ibad);
    set_err(why == 1 ? "mcp_ba_add_measurements: unknown point id" : why == 2 ? "mcp_ba_add_measurements: bad camera index" : "mcp_ba_add_measurements: bad chain");
    return -1;
  }
  return 0;
}

int mcp_ba_compute(mcp_ba* h, volatile unsigned char* abort_flag, int n_iter, double user_lambda) {
  return h->compute(abort_flag, n_iter, user_lambda);
}
int mcp_ba_converged(mcp_ba* h) { return h->converged; }
int mcp_ba_total_iterations(mcp_ba* h) { return h->total_iterations; }
int mcp_ba_get_point(mcp_ba* h, int id, double x[3]) {
  if (id <= 0 || id >= h->next_id || h->id_kind[id] != 2) { set_err("mcp_ba_get_point: unknown id"); return -1; }
  std::memcpy(x, h->points[h->id_index[id]].x, 24); return 0;
}
int mcp_ba_get_pose(mcp_ba* h, int id, double R[9], double t[3]) {
  if (id <= 0 || id >= h->next_id || h->id_kind[id] != 1) { set_err("mcp_ba_get_pose: unknown id"); return -1; }
  const HPose& p = h->poses[h->id_index[id]];
  std::memcpy(R, p.T, 72); std::memcpy(t, p.T + 9, 24); return 0;
}
int mcp_ba_get_points(mcp_ba* h, int count, const int* ids, double* x) {
  for (int i = 0; i < count; ++i) if (mcp_ba_get_point(h, ids[i], x + 3*(size_t)i)) return -1;
  return 0;
}
int mcp_ba_get_poses(mcp_ba* h, int count, const int* ids, double* R, double* t) {
  for (int i = 0; i < count; ++i) if (mcp_ba_get_pose(h, ids[i], R + 9*(size_t)i, t + 3*(size_t)i)) return -1;
  return 0;
}
int mcp_ba_num_outliers(mcp_ba* h) { return (int)h->outliers.size()/3; }
int mcp_ba_get_outliers(mcp_ba* h, int* out, int cap) {
  int n = std::min((int)h->outliers.size()/3, cap);
  if (n > 0) std::memcpy(out, h->outliers.data(), sizeof(int)*3*(size_t)n);
  return n;
}
double mcp_ba_sigma_squared(mcp_ba* h) { return h->sigma_sq; }
double mcp_ba_mean_chi_squared(mcp_ba* h) { return h->mean_chi2; }
double mcp_ba_max_cov(mcp_ba* h) { return h->max_cov; }
double mcp_ba_lambda(mcp_ba* h) { return h->lambda; }
int mcp_ba_num_iter_logs(mcp_ba* h) { return (int)h->logs.size(); }
int mcp_ba_get_iter_logs(mcp_ba* h, mcp_ba_iter_log* out, int cap) {
  int n = std::min((int)h->logs.size(), cap);
  if (n > 0) std::memcpy(out, h->logs.data(), sizeof(mcp_ba_iter_log)*(size_t)n);
  return n;
}
int mcp_ba_get_timing(mcp_ba* h, mcp_ba_timing* out) { *out = h->timing; return 0; }

int mcp_ba_set_allreduce(mcp_ba* h, mcp_allreduce_fn hook, void* user, int rank, int world_size) {
  if (world_size < 1 || rank < 0 || rank >= world_size) { set_err("mcp_ba_set_allreduce: bad rank/world"); return -1; }
  h->hook = hook; h->hook_user = user; h->comm = nullptr; h->rank = hook ? rank : 0; h->world = hook ? world_size : 1; h->dirty = true;
  return 0;
}

int mcp_comm_unique_id(void* id_out) {
  if (!rccl().load()) { set_err("mcp_comm_unique_id: librccl not found"); return -1; }
  RcclUniqueId id; std::memset(&id, 0, sizeof id);
  const int rc = rccl().GetUniqueId(&id);
  if (rc != 0) { set_err("ncclGetUniqueId failed"); return -1; }
  std::memcpy(id_out, &id, sizeof id);
  return 0;
}
mcp_comm* mcp_comm_init(const void* id, int rank, int world_size, int device) {
  if (!rccl().load()) { set_err("mcp_comm_init: librccl not found"); return nullptr; }
  if (world_size < 1 || rank < 0 || rank >= world_size) { set_err("mcp_comm_init: bad rank/world"); return nullptr; }
  if (device < 0 && hipGetDevice(&device) != hipSuccess) device = 0;
  if (hipSetDevice(device) != hipSuccess) { set_err("mcp_comm_init: hipSetDevice failed"); return nullptr; }
  RcclUniqueId uid; std::memcpy(&uid, id, sizeof uid);
  mcp_comm* c = new mcp_comm(); c->rank = rank; c->world = world_size; c->device = device;
  const int rc = rccl().CommInitRank(&c->comm, world_size, uid, rank);
  if (rc != 0) { set_err(std::string("ncclCommInitRank failed: ") + (rccl().GetErrorString ? rccl().GetErrorString(rc) : "?")); delete c; return nullptr; }
  // lane 1 (speculative stream): the same ranks in the same order, split off lane 0 -- collective over all ranks like the init itself
  if (!rccl().CommSplit) { set_err("mcp_comm_init: this librccl has no ncclCommSplit (needed for the second lane)"); (void)rccl().CommDestroy(c->comm); delete c; return nullptr; }
  const int rc2 = rccl().CommSplit(c->comm, 0, rank, &c->comm2, nullptr);
  if (rc2 != 0 || !c->comm2) { set_err(std::string("ncclCommSplit failed: ") + (rccl().GetErrorString ? rccl().GetErrorString(rc2) : "?")); (void)rccl().CommDestroy(c->comm); delete c; return nullptr; }
  return c;
}
void mcp_comm_destroy(mcp_comm* c) {
  if (!c) return;
  if (c->comm2 && c->comm2 != c->comm) (void)rccl().CommDestroy(c->comm2);
  if (c->comm) (void)rccl().CommDestroy(c->comm);
  delete c;
}
int mcp_ba_set_comm(mcp_ba* h, mcp_comm* c) {
  if (c && c->device != h->device) { set_err("mcp_ba_set_comm: communicator lives on another device"); return -1; }
  h->comm = c; h->rank = c ? c->rank : 0; h->world = c ? c->world : 1; h->dirty = true;
  return 0;
}
int mcp_comm_allreduce(mcp_comm* c, void* buf, size_t count) { return mcp_comm_allreduce_lane(c, buf, count, 0); }
int mcp_comm_allreduce_lane(mcp_comm* c, void* buf, size_t count, int lane) {
  if (!c || !c->comm || c->dead) { set_err("mcp_comm_allreduce: no communicator"); return -1; }
  HIPCK(hipSetDevice(c->device));
  const int rc = rccl().AllReduce(buf, buf, count, RCCL_FLOAT64, RCCL_SUM, c->lane(lane), nullptr);
  if (rc != 0) { set_err("ncclAllReduce failed"); return -1; }
  HIPCK(hipStreamSynchronize(nullptr));
  return 0;
}

int mcp_ba_prepare(mcp_ba* h) {
  if (h->prepare()) return -1;
  return h->nx_total();
}
int mcp_ba_eval(mcp_ba* h, double* chi2_out, double* err_out) {
  if (h->dirty && h->prepare()) return -1;
  const int n = h->P.nmeas;
  if (n == 0) return 0;
  if (err_out && h->d_err.alloc((size_t)n*2)) return -1;
  h->launch_chains(h->cur);
  h->launch_eval(h->cur, false, err_out ? h->d_err.p : nullptr);
  std::vector<double> c(n), e(err_out ? (size_t)n*2 : 0);
  HIPCK(hipMemcpyAsync(c.data(), h->d_chi2[h->cur].p, (size_t)n*8, hipMemcpyDeviceToHost, h->st));
  if (err_out) HIPCK(hipMemcpyAsync(e.data(), h->d_err.p, (size_t)n*16, hipMemcpyDeviceToHost, h->st));
  HIPCK(hipStreamSynchronize(h->st));
  for (int j = 0; j < n; ++j) {
    const int i = h->perm[j];
    if (i < 0) continue;               // (near miss: not a measurement of this map)
    if (chi2_out) chi2_out[i] = c[j];
    if (err_out) { err_out[2*(size_t)i] = e[2*(size_t)j]; err_out[2*(size_t)i + 1] = e[2*(size_t)j + 1]; }
  }
  return 0;
}
int mcp_ba_robust_chi2(mcp_ba* h, double* sigma_sq_raw, double* chi2_sum) {
  if (h->dirty && h->prepare()) return -1;
  const int n = h->P.nmeas;
  if (n == 0) return -1;
  h->launch_chains(h->cur);
  h->launch_eval(h->cur, false, nullptr);
  if (h->robust && h->median_sigma(h->cur)) return -1;
  const int nbe = (n + EVAL_BLOCK - 1)/EVAL_BLOCK;
  hipLaunchKernelGGL(k_robust_sum, dim3(nbe), dim3(EVAL_BLOCK), 0, h->st, n, h->robust, (const double*)h->d_chi2[h->cur].p, (const double*)h->sig(), h->d_part0.p);
  hipLaunchKernelGGL(k_final_sums, dim3(1), dim3(256), 0, h->st, nbe, (const double*)h->d_part0.p, 0, (const double*)nullptr, 0, (const double*)nullptr, h->d_res.p, 0, (const int*)nullptr);
  if (h->allreduce(h->d_res.p, 1, 0, false, "robust chi2")) return -1;
  HIPCK(hipMemcpyAsync(h->d_res.p + 9, h->sig(), 4*sizeof(double), hipMemcpyDeviceToDevice, h->st));
  if (h->read_results(13)) return -1;
  if (sigma_sq_raw) *sigma_sq_raw = h->h_res[9];
  if (chi2_sum) *chi2_sum = h->h_res[0];
  return 0;
}
int mcp_ba_debug_solve(mcp_ba* h, double lambda, double* x_out) {
  if (h->dirty && h->prepare()) return -1;
  if (h->P.nmeas == 0 || h->nx == 0) { set_err("mcp_ba_debug_solve: empty problem"); return -1; }
  h->launch_chains(h->cur);
  h->launch_eval(h->cur, false, nullptr);
  if (h->robust && h->median_sigma(h->cur)) return -1;
  if (h->linearize()) return -1;
  bool ok2 = true;
  if (h->solve_trial(lambda, ok2)) return -1;
  if (!ok2) { set_err("mcp_ba_debug_solve: system not positive definite"); return -1; }
  if (h->np) HIPCK(hipMemcpy(x_out, h->rhs(), (size_t)h->np*8, hipMemcpyDeviceToHost));
  if (h->nfl) HIPCK(hipMemcpy(x_out + h->np, h->d_xl.p, (size_t)h->nfl*24, hipMemcpyDeviceToHost));
  return 0;
}

int mcp_ba_debug_system(mcp_ba* h, double lambda, double* out) {
  if (h->dirty && h->prepare()) return -1;
  if (!out) return h->np;
  if (h->m_total == 0 || h->nx_total() == 0) { set_err("mcp_ba_debug_system: empty problem"); return -1; }
  h->launch_chains(h->cur);
  h->launch_eval(h->cur, false, nullptr);
  if (h->robust && h->median_sigma(h->cur)) return -1;
  if (h->linearize()) return -1;
  h->sys_cur = 0; h->spec_ok = false; h->fail_clean = false;
  SysBatch sb; std::memset(&sb, 0, sizeof sb);
  sb.lambda[0] = lambda; sb.lambda_init[0] = (h->rank == 0) ? lambda : 0.0;
  HIPCK(hipMemsetAsync(h->d_fail.p, 0, 4*sizeof(int), h->st));
  if (h->build_system(1, sb)) return -1;
  const size_t n2 = (size_t)h->np*h->np;
  // the entries outside the plan's tiles are never written by the assembly: report them as zeros
  std::vector<double> full(n2 + 2*(size_t)h->np, 0.0), dev(n2 + 2*(size_t)h->np);
  HIPCK(hipMemcpyAsync(dev.data(), h->d_red.p, dev.size()*8, hipMemcpyDeviceToHost, h->st));
  HIPCK(hipStreamSynchronize(h->st));
  for (int tpk : h->plan.all_tiles) {
    const int ti = tpk >> 16, tj = tpk & 0xffff;
    for (int r = 32*ti; r < std::min(32*ti + 32, h->np); ++r) for (int c = 32*tj; c < std::min(32*tj + 32, h->np); ++c) full[(size_t)r*h->np + c] = dev[(size_t)r*h->np + c];
  }
  for (int i = 0; i < 2*h->np; ++i) full[n2 + i] = dev[n2 + i];
  std::memcpy(out, full.data(), full.size()*8);
  return h->np;
}

// reproducibility of the factorisation + back-substitution chain (test hook): see mcp_ba.h
int mcp_dense_spd_stress(const double* A, int n, const double* b, int nsys, int reps, double* x, int* n_mismatch) {
  if (n <= 0 || n > CH_SOLVE_MAX || nsys < 1 || nsys > MAX_SYS || reps < 1) { set_err("mcp_dense_spd_stress: bad arguments"); return -1; }
  const size_t stride = (size_t)n*n + n;
  DevBuf<double> pristine, work; DevBuf<int> f;
  if (pristine.alloc(stride*nsys) || work.alloc(stride*nsys) || f.alloc(4)) return -1;
  {
    std::vector<double> host(stride);
    for (int q = 0; q < nsys; ++q) {
      std::memcpy(host.data(), A, (size_t)n*n*8);
      for (int i = 0; i < n; ++i) host[(size_t)i*n + i] += (double)q;
      std::memcpy(host.data() + (size_t)n*n, b, (size_t)n*8);
      HIPCK(hipMemcpy(pristine.p + q*stride, host.data(), stride*8, hipMemcpyHostToDevice));
    }
  }
  CholPlan plan;
  if (plan.build(n, std::vector<unsigned char>())) { set_err("mcp_dense_spd_stress: plan allocation failed"); return -1; }
  std::vector<double> first((size_t)nsys*n), cur((size_t)nsys*n);
  int bad = 0, fl[4] = {0, 0, 0, 0};
  for (int rep = 0; rep < reps; ++rep) {
    HIPCK(hipMemcpyAsync(work.p, pristine.p, stride*nsys*8, hipMemcpyDeviceToDevice, nullptr));
    HIPCK(hipMemsetAsync(f.p, 0, 16, nullptr));
    chol_factor(nullptr, plan, work.p, f.p, nsys, stride);
    chol_back(nullptr, plan, work.p, nsys, stride);
    std::vector<double>& dst = rep ? cur : first;
    for (int q = 0; q < nsys; ++q) HIPCK(hipMemcpyAsync(dst.data() + (size_t)q*n, work.p + q*stride + (size_t)n*n, (size_t)n*8, hipMemcpyDeviceToHost, nullptr));
    HIPCK(hipMemcpyAsync(fl, f.p, 16, hipMemcpyDeviceToHost, nullptr));
    HIPCK(hipStreamSynchronize(nullptr));
    for (int q = 0; q < nsys; ++q) if (fl[q]) { set_err("mcp_dense_spd_stress: matrix not positive definite"); return -1; }
    if (plan.use_persist && plan.persist.ok) {
      int ew[4] = {0, 0, 0, 0}; HIPCK(hipMemcpy(ew, plan.persist.d_err, 16, hipMemcpyDeviceToHost));
      for (int q = 0; q < nsys; ++q) if (ew[q]) { char m[112]; std::snprintf(m, sizeof m, "mcp_dense_spd_stress: a hand-off of the one-launch factorisation timed out (system %d, code 0x%x, repetition %d)", q, ew[q], rep); set_err(m); return -3; }
    }
    if (rep && std::memcmp(first.data(), cur.data(), first.size()*8) != 0) ++bad;
  }
  std::memcpy(x, first.data(), first.size()*8);
  if (n_mismatch) *n_mismatch = bad;
  return 0;
}

// structure cache (test / diagnostic hooks): hits and misses so far in this process; drop every entry
void mcp_ba_struct_cache_stats(long long* hits, long long* misses) { long long h = 0, m = 0; StructCache::get().stats(&h, &m); if (hits) *hits = h; if (misses) *misses = m; }
long long mcp_ba_struct_cache_near_hits(void) { return StructCache::get().near_hits(); }
void mcp_ba_struct_cache_clear(void) { StructCache::get().clear(); }

// the one-launch factorisation of ba_chol2.h looked at from outside (test hook): L (n x n, row-major, lower triangle; the
// diagonal BLOCKS hold L_kk^-1, which is what the kernels keep) and y = L^-1 b; info[0] = error word, info[1] = failure flag
int mcp_chol_debug_factor(const double* A, int n, const double* b, double* L_out, double* y_out, int* info) {
  if (n <= 0 || n > CH_SOLVE_MAX) { set_err("mcp_chol_debug_factor: bad n"); return -1; }
  DevBuf<double> d; DevBuf<int> f;
  if (d.alloc((size_t)n*n + n) || f.alloc(4)) return -1;
  HIPCK(hipMemcpy(d.p, A, (size_t)n*n*8, hipMemcpyHostToDevice));
  HIPCK(hipMemcpy(d.p + (size_t)n*n, b, (size_t)n*8, hipMemcpyHostToDevice));
  HIPCK(hipMemset(f.p, 0, 16));
  CholPlan plan; plan.persist_min_ntc = 1;
  if (plan.build(n, std::vector<unsigned char>())) { set_err("mcp_chol_debug_factor: plan allocation failed"); return -1; }
  CholPersist& P = plan.persist;
  if (!plan.use_persist || !P.ok) { set_err("mcp_chol_debug_factor: the persistent factorisation is switched off"); return -1; }
  if (chol_persist_factor(nullptr, P, d.p, f.p, 1, 0, 0)) { set_err("mcp_chol_debug_factor: launch failed"); return -1; }
  hipLaunchKernelGGL(k_cp_bump, dim3(1), dim3(64), 0, nullptr, P.d_epoch, P.d_err + CholPersist::max_sys, 1);
  HIPCK(hipDeviceSynchronize());
  std::vector<double> lt(P.lt_stride);
  HIPCK(hipMemcpy(lt.data(), P.d_Lt, P.lt_stride*8, hipMemcpyDeviceToHost));
  HIPCK(hipMemcpy(info, P.d_err, 4, hipMemcpyDeviceToHost));
  HIPCK(hipMemcpy(info + 1, f.p, 4, hipMemcpyDeviceToHost));
  HIPCK(hipMemset(P.d_err, 0, 4));
  const int ntc = P.ntc;
  auto at = [&](int slot, int r, int c) {
    const int qd = (r >> 4)*2 + (c >> 4), ri = r & 15;
    return lt[(size_t)slot*CP_TQ + (qd*64 + ((ri & 3) << 4 | (c & 15)))*4 + (ri >> 2)];
  };
  std::memset(L_out, 0, (size_t)n*n*8);
  for (int i = 0; i < ntc; ++i) for (int j = 0; j <= i; ++j) {
    const int sl = P.slot_of[(size_t)i*ntc + j];
    if (sl < 0) continue;
    for (int r = 0; r < 32 && 32*i + r < n; ++r) for (int c = 0; c < 32 && 32*j + c < n; ++c) L_out[(size_t)(32*i + r)*n + 32*j + c] = at(sl, r, c);
  }
  for (int j = 0; j < ntc; ++j) { const int sl = P.slot_of[(size_t)ntc*ntc + j]; for (int c = 0; c < 32 && 32*j + c < n; ++c) y_out[32*j + c] = at(sl, 0, c); }
  return 0;
}

// the cut of a coupling graph handed in from outside (ba_cut.h; tests/test_pose_cut.py) -- host code, runs without a device
int mcp_debug_pose_cut(const unsigned char* adjacency, int nf, int max_arcs, int threads, int* order_out, int* segs_out, int* info_out) {
  using namespace mcp;
  if (!adjacency || nf < 1 || nf > 1024 || !order_out || !segs_out || !info_out) { set_err("mcp_debug_pose_cut: bad arguments"); return -1; }
  const int W = (nf + 63)/64, T = std::max(1, std::min(threads, 64));
  std::vector<cut_u64> adj((size_t)nf*W, 0);
  for (int u = 0; u < nf; ++u) for (int v = 0; v < nf; ++v) if (u != v && (adjacency[(size_t)u*nf + v] || adjacency[(size_t)v*nf + u])) adj[(size_t)u*W + (v >> 6)] |= 1ull << (v & 63);
  PoseCut cut;
  // (the thread ranges one after the other: the search must give the same cut however it is split)
  pose_cut(adj, nf, max_arcs, T, [&](const std::function<void(int)>& fn) { for (int t = 0; t < T; ++t) fn(t); }, [](const char*) {}, cut);
  for (int i = 0; i < nf; ++i) order_out[i] = cut.order[i];
  for (int i = 0; i < 8; ++i) segs_out[i] = i < (int)cut.segs.size() ? cut.segs[i] : -1;
  const int info[14] = { cut.found, cut.taken, cut.relabelled, cut.k, cut.r, cut.steps, cut.t_all, cut.sep, cut.arc_len[0], cut.arc_len[1], cut.arc_len[2], cut.arc_len[3], cut.arc_len[4], cut.arc_len[5] };
  for (int i = 0; i < 14; ++i) info_out[i] = info[i];
  return cut.taken ? (int)cut.segs.size() : 1;
}
// device time of the factorisation and of the back-substitution launches (test / tuning hook): `reps` solves of (A + q I) x = b,
// q < nsys, from a device-resident copy; band > 0 restricts the plan to a banded + bordered tile pattern like a loop trajectory's
// (tiles with i - j <= band, and the last `band` block rows dense; the matrix should have that structure then)
int mcp_chol_time(const double* A, int n, const double* b, int nsys, int reps, int band, double* ms_factor, double* ms_back, double* x) {
  if (n <= 0 || n > CH_SOLVE_MAX || nsys < 1 || nsys > MAX_SYS || reps < 1) { set_err("mcp_chol_time: bad arguments"); return -1; }
  const size_t stride = (size_t)n*n + n;
  DevBuf<double> pristine, work; DevBuf<int> f;
  if (pristine.alloc(stride*nsys) || work.alloc(stride*nsys) || f.alloc(4)) return -1;
  {
    std::vector<double> host(stride);
    for (int q = 0; q < nsys; ++q) {
      std::memcpy(host.data(), A, (size_t)n*n*8);
      for (int i = 0; i < n; ++i) host[(size_t)i*n + i] += (double)q;
      std::memcpy(host.data() + (size_t)n*n, b, (size_t)n*8);
      HIPCK(hipMemcpy(pristine.p + q*stride, host.data(), stride*8, hipMemcpyHostToDevice));
    }
  }
  CholPlan plan;
  std::vector<unsigned char> pattern;
  if (band > 0) {
    const int ntc = (n + CH_NB - 1)/CH_NB;
    pattern.assign((size_t)ntc*ntc, 0);
    for (int i = 0; i < ntc; ++i) for (int j = 0; j <= i; ++j) if (i - j <= band || i >= ntc - band) pattern[(size_t)i*ntc + j] = 1;
  } else if (band < 0) {
    // a band of -band tiles DISSECTED: the caller's matrix is a banded one with its block rows / columns ordered
    // [left half ascending | right half descending | the -band blocks between them] (scripts/gpu_chol2.py dissect()): two chains + a border
    const int b = -band, ntc = (n + CH_NB - 1)/CH_NB, h = (ntc - b)/2;
    if (n % CH_NB || h < 3 || ntc - b - h < 3) { set_err("mcp_chol_time: a dissected band needs whole tiles and halves of at least three"); return -1; }
    auto orig = [&](int i) { return i < h ? i : (i < ntc - b ? (ntc - 1) - (i - h) : h + (i - (ntc - b))); };
    pattern.assign((size_t)ntc*ntc, 0);
    for (int i = 0; i < ntc; ++i) for (int j = 0; j <= i; ++j) if (std::abs(orig(i) - orig(j)) <= b) pattern[(size_t)i*ntc + j] = 1;
    plan.persist_segs = {0, h, ntc - b};
  }
  if (plan.build(n, pattern)) { set_err("mcp_chol_time: plan allocation failed"); return -1; }
  std::vector<hipEvent_t> ev((size_t)3*reps);
  for (auto& e : ev) HIPCK(hipEventCreate(&e));
  for (int rep = 0; rep < reps; ++rep) {
    HIPCK(hipMemcpyAsync(work.p, pristine.p, stride*nsys*8, hipMemcpyDeviceToDevice, nullptr));
    HIPCK(hipMemsetAsync(f.p, 0, 16, nullptr));
    HIPCK(hipEventRecord(ev[3*rep], nullptr));
    chol_factor(nullptr, plan, work.p, f.p, nsys, stride);
    HIPCK(hipEventRecord(ev[3*rep + 1], nullptr));
    chol_back(nullptr, plan, work.p, nsys, stride);
    HIPCK(hipEventRecord(ev[3*rep + 2], nullptr));
  }
  HIPCK(hipStreamSynchronize(nullptr));
  double tf = 0, tb = 0;
  for (int rep = 0; rep < reps; ++rep) {
    float a = 0, c = 0;
    HIPCK(hipEventElapsedTime(&a, ev[3*rep], ev[3*rep + 1])); HIPCK(hipEventElapsedTime(&c, ev[3*rep + 1], ev[3*rep + 2]));
    if (rep > 0 || reps == 1) { tf += a; tb += c; }
  }
  const int cnt = std::max(1, reps - 1);
  *ms_factor = tf/cnt; *ms_back = tb/cnt;
#ifdef MCP_CP_PROF
  if (plan.use_persist && plan.persist.ok) {
    // stamps of the LAST repetition, system 0: 10 ns ticks
    std::vector<unsigned long long> pr(256*16), hp(8192*4);
    (void)hipMemcpyFromSymbol(pr.data(), HIP_SYMBOL(g_cp_prof), pr.size()*8); (void)hipMemcpyFromSymbol(hp.data(), HIP_SYMBOL(g_cp_hprof), hp.size()*8);
    const CholPersist& P = plan.persist;
    { unsigned int ms[8] = {0}; (void)hipMemcpyFromSymbol(ms, HIP_SYMBOL(g_cp_miss), sizeof ms); fprintf(stderr, "[cp prof] over %d launches: band rows asked for again %u times, late products %u times (per wavefront)\n", reps, ms[0], ms[1]); }
    auto us = [&](unsigned long long a, unsigned long long b0) { return a && b0 ? ((double)a - (double)b0)*0.01 : -1.0; };
    fprintf(stderr, "[cp prof] n=%d ntc=%d helpers=%d   (us from the start of P3: wave 0 reads D | pivots | writes L^-1;  team: flags of L^-1, L(s+1,s) | [next row polled, measured from P3 as well] | staged row in LDS | trsm | last flag;  step)\n", n, P.ntc, P.nhelpers);
#if MCP_CP_PROF == 5
    { double a[4] = {0, 0, 0, 0}, st = 0, w2[6] = {0, 0, 0, 0, 0, 0}; int c5 = 0;
      for (int s = 2; s + 3 < P.ntc; ++s, ++c5) { const unsigned long long* q0 = &pr[(s + 1)*16]; a[0] += us(q0[2], q0[0]); for (int w = 1; w < 4; ++w) a[w] += us(q0[4 + w], q0[0]); st += us(q0[8], q0[0]);
        w2[0] += us(q0[3], q0[0]); w2[1] += us(q0[11], q0[0]); w2[2] += us(q0[12], q0[0]); w2[3] += us(q0[13], q0[0]); w2[4] += us(q0[14], q0[0]); w2[5] += us(q0[15], q0[0]); }
      if (c5) fprintf(stderr, "  solve done and L(s+2,s) on its way: wave 2 %.2f  wave 3 %.2f\n", w2[3]/c5, w2[4]/c5);
      if (c5) fprintf(stderr, "  wave 2, us from the step's start: flags of L^-1 / L(s+1,s) %.2f  solve done, X2 stores issued %.2f  products done %.2f\n", w2[0]/c5, w2[1]/c5, w2[2]/c5);
      if (c5) fprintf(stderr, "  arrival at the step's last barrier, us from the step's start (mean over %d steps): wave 0 %.2f  wave 1 %.2f  wave 2 %.2f  wave 3 %.2f | step %.2f\n", c5, a[0]/c5, a[1]/c5, a[2]/c5, a[3]/c5, st/c5); }
#endif
    double sum[10] = {0}; int c2 = 0;
    for (int s = 2; s + 3 < P.ntc; ++s, ++c2) {
      const unsigned long long* q0 = &pr[(s + 1)*16];
      const double v[10] = {us(q0[1], q0[0]), us(q0[9], q0[1]), us(q0[10], q0[9]), us(q0[2], q0[10]), us(q0[3], q0[1]), us(q0[4], q0[1]), us(q0[5], q0[1]), us(q0[6], q0[1]), us(q0[7], q0[1]), us(q0[8], q0[0])};
      for (int i = 0; i < 10; ++i) sum[i] += v[i];
      if (s == 10) fprintf(stderr, "  step 10, panel: first half %.2f  hand-over %.2f  second half %.2f\n", us(q0[14], q0[9]), us(q0[15], q0[14]), us(q0[10], q0[15]));
      if (s == 10) fprintf(stderr, "  step 10, wave 1 after the trsm: X2 stores issued %+.2f  products done %+.2f  drained %+.2f  flag %+.2f\n", us(q0[11], q0[6]), us(q0[12], q0[6]), us(q0[13], q0[6]), us(q0[7], q0[6]));
      if (s >= 10 && s < 13) {
        fprintf(stderr, "  step %2d: P1+P2 %5.2f | D in %5.2f pivots %5.2f L^-1 out %5.2f | published %5.2f polled %5.2f in LDS %5.2f trsm %5.2f flag %5.2f | step %5.2f\n", s, v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], v[8], v[9]);
        const unsigned long long pub = pr[(s - 1 + 1)*16 + 3], pub2 = pr[(s - 1 + 1)*16 + 7];
        for (int hi = 0; hi < P.nhelpers && hi < 8192; ++hi) {
          const CpHelper& h = P.helpers[hi];
          if (h.ti != s + 2 || !((h.kind == 0 && h.tj == s - 1) || h.kind == 1)) continue;
          fprintf(stderr, "      helper (%d,%d) %s: last poll ok %+6.2f  diag poll ok %+6.2f  published %+6.2f   (vs publish of L^-1(%d) / L(%d,%d); L(%d,%d) came %+5.2f)\n", h.ti, h.tj, h.kind ? "band" : "far ",
                  us(hp[hi*4 + 1], pub), us(hp[hi*4 + 2], pub), us(hp[hi*4 + 3], pub), s - 1, s, s - 1, s + 1, s - 1, us(pub2, pub));
        }
      }
    }
    if (c2) fprintf(stderr, "  mean over %d steps: P1+P2 %.2f | D in %.2f pivots %.2f L^-1 out %.2f | published %.2f polled %.2f in LDS %.2f trsm %.2f flag %.2f | step %.2f\n", c2,
                    sum[0]/c2, sum[1]/c2, sum[2]/c2, sum[3]/c2, sum[4]/c2, sum[5]/c2, sum[6]/c2, sum[7]/c2, sum[8]/c2, sum[9]/c2);
  }
#endif
  for (auto& e : ev) (void)hipEventDestroy(e);
  for (int q = 0; q < nsys; ++q) HIPCK(hipMemcpy(x + (size_t)q*n, work.p + q*stride + (size_t)n*n, (size_t)n*8, hipMemcpyDeviceToHost));
  if (plan.use_persist && plan.persist.ok) {
    int ew[4] = {0, 0, 0, 0}; HIPCK(hipMemcpy(ew, plan.persist.d_err, 16, hipMemcpyDeviceToHost));
    for (int q = 0; q < nsys; ++q) if (ew[q]) { char m[112]; std::snprintf(m, sizeof m, "mcp_chol_time: a hand-off of the one-launch factorisation timed out (system %d, code 0x%x)", q, ew[q]); set_err(m); return -3; }
  }
  return 0;
}

// Dense SPD solve A x = b on the device with the reduced-system kernels (test hook for ba_chol.h)
int mcp_dense_spd_solve(const double* A, int n, const double* b, double* x) {
  if (n <= 0 || n > CH_SOLVE_MAX) { set_err("mcp_dense_spd_solve: bad n"); return -1; }
  DevBuf<double> d; DevBuf<int> f;
  if (d.alloc((size_t)n*n + n) || f.alloc(4)) return -1;
  HIPCK(hipMemcpy(d.p, A, (size_t)n*n*8, hipMemcpyHostToDevice));
  HIPCK(hipMemcpy(d.p + (size_t)n*n, b, (size_t)n*8, hipMemcpyHostToDevice));
  HIPCK(hipMemset(f.p, 0, 16));
  CholPlan plan;
  std::vector<unsigned char> pattern;
#ifdef MCP_CHOL_PROF
  if (const char* e = getenv("MCP_CHOL_TEST_BAND")) {      // profiling only: a banded + bordered tile pattern like a loop trajectory's (results are not checked)
    const int bw = atoi(e), ntc = (n + CH_NB - 1)/CH_NB;
    pattern.assign((size_t)ntc*ntc, 0);
    for (int i = 0; i < ntc; ++i) for (int j = 0; j <= i; ++j) if (i - j <= bw || i >= ntc - bw) pattern[(size_t)i*ntc + j] = 1;
  }
#endif
  if (plan.build(n, pattern)) { set_err("mcp_dense_spd_solve: plan allocation failed"); return -1; }
#ifdef MCP_CHOL_PROF
  {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    chol_factor(nullptr, plan, d.p, f.p);      // warm
    HIPCK(hipMemcpy(d.p, A, (size_t)n*n*8, hipMemcpyHostToDevice));
    HIPCK(hipMemcpy(d.p + (size_t)n*n, b, (size_t)n*8, hipMemcpyHostToDevice));
    HIPCK(hipMemset(f.p, 0, 16));
    hipEventRecord(e0, nullptr);
  }
#endif
  chol_factor(nullptr, plan, d.p, f.p);
#ifdef MCP_CHOL_PROF
  {
    hipEvent_t e1; hipEventCreate(&e1); hipEventRecord(e1, nullptr); hipEventSynchronize(e1);
    std::vector<unsigned long long> pr(256*2*8);
    hipMemcpyFromSymbol(pr.data(), HIP_SYMBOL(g_chol_prof), pr.size()*8);
    const int ntc = plan.ntc;
    double acc[2][4] = {{0}}; double gap = 0; int cnt = 0;
    for (int k = 2; k + 2 < ntc; ++k) for (int b2 = 0; b2 < 2; ++b2) {
      const unsigned long long* q = &pr[(k*2 + b2)*8];
      for (int i = 0; i < 4; ++i) acc[b2][i] += (double)(q[i + 1] - q[i]);
      if (b2 == 1) { gap += (double)(pr[((k + 1)*2 + 1)*8] - q[4]); ++cnt; }
    }
    fprintf(stderr, "[chol prof] n=%d steps=%d clock64 ticks per phase (avg over %d steps)\n", n, ntc, cnt);
    for (int b2 = 0; b2 < 2; ++b2) fprintf(stderr, "  block %d: load %.0f  mfma %.0f  panel %.0f  store %.0f\n", b2, acc[b2][0]/cnt, acc[b2][1]/cnt, acc[b2][2]/cnt, acc[b2][3]/cnt);
    fprintf(stderr, "  end(k) -> start(k+1) gap %.0f ticks; total first->last %.0f ticks\n", gap/cnt, (double)(pr[((ntc - 3)*2 + 1)*8] - pr[(2*2 + 1)*8]));
  }
#endif
  chol_back(nullptr, plan, d.p);
#ifdef MCP_CHOL_PROF
  {
    HIPCK(hipDeviceSynchronize());
    std::vector<unsigned long long> pr(256*2*8);
    hipMemcpyFromSymbol(pr.data(), HIP_SYMBOL(g_back_prof), pr.size()*8);
    const int ntc = plan.ntc; double sv = 0, rl = 0, sb_ = 0, up = 0, ub_ = 0; int cnt = 0;
    for (int k = 2; k + 2 < ntc; ++k) {
      const unsigned long long* a = &pr[(k*2 + 0)*8]; const unsigned long long* u = &pr[(k*2 + 1)*8];
      sv += (double)(a[1] - a[0]); rl += (double)(a[3] - a[1]); sb_ += (double)(a[4] - a[3]);
      up += (double)(u[3] - u[0]); ub_ += (double)(u[4] - u[3]); ++cnt;
    }
    fprintf(stderr, "[back prof] solver:  fast tile + solve %.0f  reload issue %.0f  barrier %.0f\n", sv/cnt, rl/cnt, sb_/cnt);
    fprintf(stderr, "[back prof] updater: update %.0f  barrier %.0f\n", up/cnt, ub_/cnt);
  }
#endif
  int fl = 0;
  HIPCK(hipMemcpy(&fl, f.p, 4, hipMemcpyDeviceToHost));
  if (plan.use_persist && plan.persist.ok) {
    int ew = 0; HIPCK(hipMemcpy(&ew, plan.persist.d_err, 4, hipMemcpyDeviceToHost));
    if (ew) { char m[96]; std::snprintf(m, sizeof m, "mcp_dense_spd_solve: a hand-off of the one-launch factorisation timed out (code 0x%x)", ew); set_err(m); return -3; }
  }
  HIPCK(hipMemcpy(x, d.p + (size_t)n*n, (size_t)n*8, hipMemcpyDeviceToHost));
  if (fl) { set_err("mcp_dense_spd_solve: matrix not positive definite"); return -1; }
  return 0;
}

}  // extern "C"

// context.cpp -- the C ABI (include/gadfit_hip.h): data residency, kernel launches, the
// cross-rank sum.  Everything N-sized stays in HBM; per call only the parameter block goes
// down (<= n_datasets*n_pars doubles) and the packed [JTJ | JTres | chi2] comes back.
#include "context.h"
#include <dlfcn.h>
#include "group.h"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <ctime>
#include <exception>
#include <memory>
#include <mutex>

using namespace gfh;

namespace gfh {
static std::string g_err;
static std::mutex g_err_mutex;     // the members of a device group fail on their own threads
void set_global_error(const std::string& m) { std::lock_guard<std::mutex> lk(g_err_mutex); g_err = m; }
int fail(gfh_ctx* c, const std::string& msg) { if (c) c->err = msg; set_global_error(msg); return 1; }
}  // namespace gfh

#define HIPCHK(c, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) \
  return fail(c, std::string(#call) + ": " + hipGetErrorString(e_)); } while (0)
// (a pass may end with kUnseen instead of 0 / 1: passed up unchanged to the loop that recovers and repeats it)
#define PASS(expr) do { const int rc_ = (expr); if (rc_) return rc_; } while (0)
#define NCCLCHK(c, call) do { ncclResult_t r_ = (call); if (r_ != ncclSuccess) \
  return fail(c, std::string(#call) + ": " + ncclGetErrorString(r_)); } while (0)
#define NEED_GPU(c) do { if (!(c)) return 1; if ((c)->device < 0) \
  return fail(c, "no GPU bound to this context (libgadfit_hip has no CPU fallback)"); \
  if (gfh::join_pending(c)) return 1; \
  if ((c)->create_failed) return fail(c, (c)->create_err); \
  hipError_t e_ = hipSetDevice((c)->device); if (e_ != hipSuccess) return fail(c, "hipSetDevice failed"); } while (0)
// a device-group handle: the same call on every member, each on its own thread (k = member, r = its rank)
#define GROUP(c, expr) do { if ((c) && (c)->grp) return gfh::group_run((c), [&](gfh_ctx* k, int r) -> int { (void)k; (void)r; return (expr); }); } while (0)
#define NOT_FOR_GROUP(c, what) do { if ((c) && (c)->grp) return fail(c, what " is not available on a device-group handle"); } while (0)

// A batch of small fits -- gadf_init ... gadf_close per spectrum -- creates and destroys a context per fit, and what that costs is
// the runtime's own calls: ~25 hipFree (each waits for the device) and as many hipMalloc, a stream, six events, three pinned
// allocations: 3.6 ms around a fit of 0.9 ms (tools/probes/context_cycle.py).  So what a destroyed context held is kept for the
// next one of the same device: its small device blocks (up to 4 MB each, 64 MB per device in all, in power-of-two classes) and
// its stream, events, status word and pinned buffers (BaseRes, one parked set per device).  GADFIT_HIP_POOL=0: everything is
// returned to the runtime as before.  Blocks enter the pool only from gfh_destroy, after the context's stream has drained.
namespace {
constexpr size_t kPoolBlockMax = (size_t)4 << 20, kPoolCap = (size_t)64 << 20;
struct BaseRes {
  hipStream_t stream = nullptr; hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  void* status = nullptr; int* h_status = nullptr;
  double* h_pinned = nullptr; size_t h_pinned_bytes = 0;
  double* h_pars = nullptr; size_t h_pars_bytes = 0;
  double* h_dpars = nullptr; size_t h_dpars_bytes = 0;
};
struct DevicePool { std::vector<void*> blocks[32]; size_t cached = 0; bool has_base = false; BaseRes base; };
std::mutex g_pool_mutex;
std::map<int, DevicePool> g_pool;
bool pool_on() { static const bool on = [] { const char* e = getenv("GADFIT_HIP_POOL"); return !e || atoi(e) != 0; }(); return on; }
int pool_class(size_t bytes) { int c = 8; while (((size_t)1 << c) < bytes) c++; return c; }       // 256 B ... 4 MB
}  // namespace

static int dev_alloc(gfh_ctx* c, DevBuf& b, size_t bytes) {
  if (b.bytes >= bytes && b.p) return 0;
  if (b.p) { hipFree(b.p); b.p = nullptr; b.bytes = 0; }
  if (bytes == 0) bytes = 8;
  if (bytes <= kPoolBlockMax && pool_on()) {
    const int cls = pool_class(bytes);
    {
      std::lock_guard<std::mutex> lk(g_pool_mutex);
      auto it = g_pool.find(c->device);
      if (it != g_pool.end() && !it->second.blocks[cls].empty()) {
        b.p = it->second.blocks[cls].back(); it->second.blocks[cls].pop_back();
        it->second.cached -= (size_t)1 << cls;
      }
    }
    if (!b.p) HIPCHK(c, hipMalloc(&b.p, (size_t)1 << cls));
    b.bytes = bytes;
    return 0;
  }
  HIPCHK(c, hipMalloc(&b.p, bytes));
  b.bytes = bytes;
  return 0;
}
static void dev_free(DevBuf& b) { if (b.p) hipFree(b.p); b.p = nullptr; b.bytes = 0; }
// gfh_destroy's form (the stream has drained): a small block goes to the pool of its device
static void dev_release(int device, DevBuf& b) {
  if (b.p && b.bytes <= kPoolBlockMax && pool_on()) {
    const int cls = pool_class(b.bytes);
    std::lock_guard<std::mutex> lk(g_pool_mutex);
    DevicePool& dp = g_pool[device];
    if (dp.cached + ((size_t)1 << cls) <= kPoolCap) {
      dp.blocks[cls].push_back(b.p); dp.cached += (size_t)1 << cls;
      b.p = nullptr; b.bytes = 0;
      return;
    }
  }
  dev_free(b);
}

static int pinned_reserve(gfh_ctx* c, size_t bytes) {
  if (c->h_pinned_bytes >= bytes) return 0;
  if (c->h_pinned) hipHostFree(c->h_pinned);
  c->h_pinned = nullptr; c->h_pinned_bytes = 0;
  // host-coherent and mapped: k_publish writes results into it from the device (result mailbox)
  HIPCHK(c, hipHostMalloc((void**)&c->h_pinned, bytes, hipHostMallocCoherent | hipHostMallocMapped));
  c->h_pinned_bytes = bytes;
  return 0;
}

namespace gfh {
namespace {
struct Roctx {
  int (*push)(const char*) = nullptr;
  int (*pop)() = nullptr;
  Roctx() {
    const char* e = getenv("GADFIT_HIP_ROCTX");
    if (!e || atoi(e) == 0) return;
    void* h = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return;
    push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
    pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
    if (!push || !pop) { push = nullptr; pop = nullptr; }
  }
};
const Roctx& roctx() { static Roctx r; return r; }
}  // namespace
Range::Range(const char* name) : on_(roctx().push != nullptr) { if (on_) roctx().push(name); }
Range::~Range() { if (on_) roctx().pop(); }

int join_pending(gfh_ctx* c) {
  if (!c->pending.joinable()) return 0;
  c->stop_warm.store(true);
  c->pending.join();
  c->creating = false;
  const int rc = c->pending_rc;
  c->pending_rc = 0;
  return rc;
}
// choose whether the next sweeps write the Jacobian (only the fused kernel can do without it)
// does STEP 3 (J^T omega) read the Jacobian back from HBM for the current model and options?
bool omega_needs_jacobian(const gfh_ctx* c, int n_active) {
  return !(c->gen.omega_jt && !c->gen.finite_diff && c->has_model && !c->model.has_integrals() && c->gen.loss == 0 && n_active <= kOmegaJtMaxActive);
}

void set_store_j(gfh_ctx* c, bool on) {
  if (!c->fused || (c->has_model && c->model.has_integrals())) on = true;   // the two-kernel path re-reads J
  if (on != c->gen.store_j) { c->gen.store_j = on; c->cur = nullptr; c->have_sweep = false; c->j_valid = false; }
}
// chi2() overwrites the residual vector in the reference (gadfit.F90:1024-1026); only the grad_chi2 / cos_phi tests
// and read-backs ever look at it, so gfh_fit under keep_jacobian mode 2 lets the chi2 kernel skip the 8 B/point store
void set_store_res(gfh_ctx* c, bool on) {
  if (on != c->gen.store_res) { c->gen.store_res = on; c->cur = nullptr; c->prepared = false; c->have_sweep = false; }
}
}  // namespace gfh

// The first host-to-device copy of more than a few KB in a process costs ~10 ms on top of its transfer (the runtime sets its copy
// path up: tools/probes/upload_warm.py -- a first upload of 3 x 80 MB takes 15 ms, after ANY earlier copy of 0.8 MB it takes 5.8).
// The first context of a process makes that copy on a thread of its own, beside whatever the caller does between creating the
// context and handing its data over; an upload waits for it (copy_path_ready), since two first copies at once pay twice.
namespace {
std::once_flag g_copy_warm_once;
std::thread g_copy_warm;
std::mutex g_copy_warm_mutex;
void copy_path_ready();
void warm_copy_path(int device) {
  if (const char* e = getenv("GADFIT_HIP_WARM_COPY")) if (atoi(e) == 0) return;
  std::call_once(g_copy_warm_once, [device]() {
    try {
      std::lock_guard<std::mutex> lk(g_copy_warm_mutex);
      g_copy_warm = std::thread([device]() {
        if (hipSetDevice(device) != hipSuccess) { (void)hipGetLastError(); return; }
        const size_t bytes = (size_t)1 << 20;
        std::vector<char> host(bytes, 1);
        void* dev = nullptr;
        if (hipMalloc(&dev, bytes) != hipSuccess) { (void)hipGetLastError(); return; }
        if (hipMemcpy(dev, host.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) (void)hipGetLastError();
        (void)hipFree(dev);
      });
      std::atexit(copy_path_ready);       // a process that ends without gfh_destroy: the thread is joined before the statics go
    } catch (const std::exception&) {}
  });
}
void copy_path_ready() {
  std::lock_guard<std::mutex> lk(g_copy_warm_mutex);
  if (g_copy_warm.joinable()) g_copy_warm.join();
}
}  // namespace

// the device part of gfh_create: runtime initialisation (hipGetDeviceCount is where a process pays for it: 80 ms, 240 ms for the first
// process on a box), stream, events, status word, result mailbox -- on the caller's thread (gfh_create) or on the context's own
// (gfh_create_begin); failure leaves the message in the context and in the global slot
static int init_device(gfh_ctx* c) {
  const int device = c->device;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) return fail(c, "no HIP device available (libgadfit_hip has no CPU fallback)");
  if (device >= n) return fail(c, "device index out of range");
  if (hipSetDevice(device) != hipSuccess) return fail(c, "cannot initialise HIP device");
  // what the last context destroyed on this device left behind (BaseRes), if anything
  bool adopted = false;
  if (pool_on()) {
    std::lock_guard<std::mutex> lk(g_pool_mutex);
    auto it = g_pool.find(device);
    if (it != g_pool.end() && it->second.has_base) {
      BaseRes& r = it->second.base;
      c->stream = r.stream; for (int k = 0; k < 6; k++) c->ev[k] = r.ev[k];
      c->status.p = r.status; c->h_status = r.h_status;
      c->h_pinned = r.h_pinned; c->h_pinned_bytes = r.h_pinned_bytes;
      c->h_pars = r.h_pars; c->h_pars_bytes = r.h_pars_bytes; c->h_dpars = r.h_dpars; c->h_dpars_bytes = r.h_dpars_bytes;
      it->second.has_base = false; r = BaseRes();
      adopted = true;
    }
  }
  if (!adopted && hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
    return fail(c, "cannot initialise HIP device");
  }
  // the status word (+ the report area of unseen branches, kStatusBytes), the result mailbox's flag and the timing events: every
  // later call dereferences them, so a context without them is not handed out
  bool ok = true;
  if (!adopted) {
    for (auto& ev : c->ev) ok = ok && hipEventCreate(&ev) == hipSuccess;
    ok = ok && hipMalloc(&c->status.p, gfh::kStatusBytes) == hipSuccess;
  }
  if (ok) { c->status.bytes = gfh::kStatusBytes; ok = hipMemset(c->status.p, 0, gfh::kStatusBytes) == hipSuccess; }
  if (!adopted) ok = ok && hipHostMalloc((void**)&c->h_status, 64, hipHostMallocCoherent | hipHostMallocMapped) == hipSuccess && c->h_status;
  if (!ok) {
    (void)hipGetLastError();
    for (auto& ev : c->ev) { if (ev) hipEventDestroy(ev); ev = nullptr; }
    if (c->status.p) hipFree(c->status.p);
    if (c->h_status) hipHostFree(c->h_status);
    hipStreamDestroy(c->stream);
    c->status.p = nullptr; c->status.bytes = 0; c->h_status = nullptr; c->stream = nullptr;
    return fail(c, "cannot allocate the status word, the result mailbox or the timing events of a context");
  }
  memset(c->h_status, 0, 64); c->h_flag = reinterpret_cast<unsigned long long*>(c->h_status + 2);
  warm_copy_path(device);
  return 0;
}

extern "C" {

int gfh_version(void) { return 100; }

// (the process-wide message is copied under its mutex into a per-thread snapshot: device-group members fail on their own threads)
const char* gfh_last_error(const gfh_ctx* ctx) {
  if (ctx) return ctx->err.c_str();
  static thread_local std::string snap;
  { std::lock_guard<std::mutex> lk(g_err_mutex); snap = g_err; }
  return snap.c_str();
}

int gfh_create(int device, gfh_ctx** out) {
  if (!out) return 1;
  *out = nullptr;
  gfh_ctx* c = new gfh_ctx();
  c->device = device;
  if (const char* e = getenv("GADFIT_HIP_OMEGA_JT")) c->gen.omega_jt = atoi(e) != 0;
  if (const char* e = getenv("GADFIT_HIP_KERNARG")) c->kernarg = atoi(e) != 0;
  if (const char* e = getenv("GADFIT_HIP_TAIL")) c->tail = atoi(e) != 0;
  if (const char* e = getenv("GADFIT_HIP_SPARSE")) c->sparse_ok = atoi(e) != 0;
  if (const char* e = getenv("GADFIT_HIP_MERGE_SMALL")) c->merge_small = atoi(e) != 0;
  if (const char* e = getenv("GADFIT_HIP_FAST_DIV")) c->gen.fast_div = atoi(e) != 0;
  if (const char* e = getenv("GADFIT_HIP_FUSED")) c->fused = atoi(e) != 0;
  if (const char* e = getenv("GADFIT_HIP_LOOKAHEAD")) c->lookahead = atoi(e) != 0;
  if (const char* e = getenv("GADFIT_HIP_KEEP_J")) { int v = atoi(e); if (v >= 0 && v <= 2) { c->keep_jacobian = v; c->gen.store_j = v != 0; } }
  if (const char* e = getenv("GADFIT_HIP_MESH")) c->mesh_on = atoi(e) != 0;
  if (const char* e = getenv("GADFIT_HIP_ORDER")) c->order_on = atoi(e) != 0;
  if (const char* e = getenv("GADFIT_HIP_KEEP_WARM")) c->keep_warm = atoi(e) != 0;
  if (const char* e = getenv("GADFIT_HIP_PLACEMENT_AFTER")) { int v = atoi(e); if (v >= 0) c->placement_after = v; }
  if (const char* e = getenv("GADFIT_HIP_WS_FAST")) { int v = atoi(e); if (v >= 0) c->ws_fast = v; }
  if (const char* e = getenv("GADFIT_HIP_HALF_STAGE")) { int v = atoi(e); if (v >= -1 && v <= 1) c->gen.half_stage = v; }
  if (const char* e = getenv("GADFIT_HIP_FUSED_WAVES")) { int v = atoi(e); if (v == 1 || v == 2 || v == 4 || v == 8) c->gen.fused_waves = v; }
  if (const char* e = getenv("GADFIT_HIP_SINGLE_IMAGE")) c->gen.single_image = atoi(e) != 0;
  if (const char* e = getenv("GADFIT_HIP_FRAG_LATE")) c->gen.frag_late = atoi(e) != 0;
  if (const char* e = getenv("GADFIT_HIP_FUSED_WPE")) { int v = atoi(e); if (v >= 0 && v <= 8) c->gen.fused_wpe = v; }
  if (const char* e = getenv("GADFIT_HIP_COOP")) { int v = atoi(e); if (v >= 0 && v <= 2) c->gen.coop = v; }
  if (const char* e = getenv("GADFIT_HIP_VALU_AHEAD")) { int v = atoi(e); if (v >= 0 && v <= 2) c->gen.valu_ahead = v; }
  if (const char* e = getenv("GADFIT_HIP_TIMERS")) { int v = atoi(e); if (v >= 0 && v <= 2) c->timer_detail = v; }
  if (device >= 0 && init_device(c)) { delete c; return 1; }
  *out = c;
  return 0;
}

// gfh_create that returns at once: the device part runs on a thread of the context, beside whatever the caller does next on the
// host; the first call that needs the device waits for it (NEED_GPU), and gfh_set_data_begin queues its upload behind it.
int gfh_create_begin(int device, gfh_ctx** out) {
  if (device < 0) return gfh_create(device, out);
  if (const char* e = getenv("GADFIT_HIP_ASYNC_INIT")) if (atoi(e) == 0) return gfh_create(device, out);
  if (!out) return 1;
  *out = nullptr;
  gfh_ctx* c = nullptr;
  if (gfh_create(-1, &c)) return 1;          // (the host part: configuration from the environment)
  c->device = device;
  c->pending_rc = 0; c->creating = true;
  try {
    c->pending = std::thread([c]() {
      const int rc = init_device(c);
      if (rc) { c->create_failed = true; c->create_err = c->err; }
      c->pending_rc = rc;
    });
  } catch (const std::exception&) {
    c->creating = false;
    if (init_device(c)) { c->device = -1; gfh_destroy(c); return 1; }
  }
  *out = c;
  return 0;
}

int gfh_create_group(int n_devices, const int* devices, gfh_ctx** out) { return gfh::group_create(n_devices, devices, out); }
int gfh_group_size(const gfh_ctx* c) { return c ? (c->grp ? gfh::group_size(c) : 1) : 0; }

// Test hook: member r sums bufs[r][0..n) over the group in place through the same barrier + ordered host sum the
// passes use (status[r] in, max over the members out); member `fail_member` (>= 0) fails before it reaches the
// barrier, which must release the others with an error instead of leaving them waiting.
int gfh_debug_group_allreduce(gfh_ctx* c, double* bufs, int n, int* status, int fail_member) {
  if (!c || !c->grp) return fail(c, "gfh_debug_group_allreduce needs a device-group handle");
  return gfh::group_run(c, [&](gfh_ctx* k, int r) -> int {
    if (r == fail_member) return fail(k, "member " + std::to_string(r) + " failed on purpose");
    return gfh::group_allreduce(k, bufs + (size_t)r * n, (size_t)n, status + r);
  });
}

int gfh_debug_group_latency(gfh_ctx* c, int n, int rounds, double* out2) {
  if (!c || !c->grp || n < 1 || rounds < 1 || !out2) return fail(c, "gfh_debug_group_latency needs a device-group handle, n >= 1, rounds >= 1");
  const int N = gfh_group_size(c);
  std::vector<std::vector<double>> bufs((size_t)N, std::vector<double>((size_t)n, 1.0));
  auto sums = [&](int count) {
    return gfh::group_run(c, [&](gfh_ctx* k, int r) -> int {
      int st = 0;
      for (int i = 0; i < count; i++) {
        for (int j = 0; j < n; j++) bufs[(size_t)r][(size_t)j] = 1.0 + r;         // (a member's pass leaves fresh numbers in its mailbox)
        if (gfh::group_allreduce(k, bufs[(size_t)r].data(), (size_t)n, &st)) return 1;
      }
      return 0;
    });
  };
  if (sums(std::min(rounds, 200))) return 1;                                       // (threads awake, pages touched)
  auto t0 = std::chrono::steady_clock::now();
  if (sums(rounds)) return 1;
  out2[0] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / rounds;
  const double want = 0.5 * N * (N + 1);
  for (int r = 0; r < N; r++) if (bufs[(size_t)r][0] != want || bufs[(size_t)r][(size_t)n - 1] != want) return fail(c, "gfh_debug_group_latency: wrong sum");
  t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < rounds; i++) if (gfh::group_run(c, [](gfh_ctx*, int) -> int { return 0; })) return 1;
  out2[1] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / rounds;
  return 0;
}

void gfh_destroy(gfh_ctx* c) {
  if (!c) return;
  (void)gfh::join_pending(c);          // (stops the upload thread's keep-warm loop instead of waiting it out)
  if (c->host_copy.joinable()) c->host_copy.join();
  copy_path_ready();
  if (c->grp) gfh::group_destroy(c);
  if (c->device >= 0) {
    hipSetDevice(c->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    if (c->comm) ncclCommDestroy(c->comm);
    for (auto& kv : c->kernel_cache) release_loaded(c->device, &kv.second);
    DevBuf* bufs[] = {&c->x, &c->y, &c->w, &c->res, &c->omega, &c->is_pad, &c->J, &c->tile_ds, &c->gb_start, &c->gb_slots,
                      &c->gb_ds, &c->ds_first_gb, &c->partial, &c->G, &c->chi2_partial, &c->packed, &c->pars, &c->dpars,
                      &c->inv, &c->dl, &c->vec, &c->slice, &c->counters, &c->tail_dev, &c->aux, &c->mesh, &c->tile_cost, &c->tile_order, &c->gb_order, &c->owner, &c->nz_row, &c->nz_col, &c->gs_meta, &c->gs_list, &c->wsg};
    for (DevBuf* b : bufs) dev_release(c->device, *b);
    // stream, events, status word and pinned buffers: parked for the next context of this device (one set), else given back
    bool parked = false;
    if (pool_on() && c->stream && c->status.p && c->h_status) {
      std::lock_guard<std::mutex> lk(g_pool_mutex);
      DevicePool& dp = g_pool[c->device];
      if (!dp.has_base) {
        BaseRes& r = dp.base;
        r.stream = c->stream; for (int k = 0; k < 6; k++) r.ev[k] = c->ev[k];
        r.status = c->status.p; r.h_status = c->h_status;
        r.h_pinned = c->h_pinned; r.h_pinned_bytes = c->h_pinned_bytes;
        r.h_pars = c->h_pars; r.h_pars_bytes = c->h_pars_bytes; r.h_dpars = c->h_dpars; r.h_dpars_bytes = c->h_dpars_bytes;
        dp.has_base = true; parked = true;
      }
    }
    if (!parked) {
      if (c->status.p) hipFree(c->status.p);
      if (c->h_pinned) hipHostFree(c->h_pinned);
      if (c->h_pars) hipHostFree(c->h_pars);
      if (c->h_dpars) hipHostFree(c->h_dpars);
      if (c->h_status) hipHostFree(c->h_status);
      for (auto& ev : c->ev) if (ev) hipEventDestroy(ev);
      if (c->stream) hipStreamDestroy(c->stream);
    }
  }
  delete c;
}

// ------------------------------------------------------------------------- communicator
int gfh_comm_unique_id(void* id) {
  static_assert(sizeof(ncclUniqueId) == GFH_UNIQUE_ID_BYTES, "unique id size");
  ncclUniqueId u;
  ncclResult_t r = ncclGetUniqueId(&u);
  if (r != ncclSuccess) { set_global_error(std::string("ncclGetUniqueId: ") + ncclGetErrorString(r)); return 1; }
  memcpy(id, &u, sizeof u);
  return 0;
}

int gfh_comm_init(gfh_ctx* c, int nranks, int rank, const void* id) {
  NOT_FOR_GROUP(c, "gfh_comm_init (a device group is its own communicator)");
  if (c && c->member_of) return fail(c, "context belongs to a device group");
  NEED_GPU(c);
  if (nranks < 1 || rank < 0 || rank >= nranks) return fail(c, "bad communicator geometry");
  if (c->count) return fail(c, "gfh_comm_init must precede gfh_set_data");
  ncclUniqueId u; memcpy(&u, id, sizeof u);
  NCCLCHK(c, ncclCommInitRank(&c->comm, nranks, u, rank));
  c->nranks = nranks; c->rank = rank;
  return 0;
}

// How the cross-rank sums of this context travel: ranks of its RCCL communicator as RCCL itself reports them
// (ncclCommCount; 0 = no communicator: a single image, or a device group that sums on the host), and the number of
// all-reduces issued since gfh_reset_timers.
int gfh_comm_info(gfh_ctx* c, int* rccl_nranks, int64_t* n_allreduce) {
  if (!c) return 1;
  gfh_ctx* k = c->grp ? gfh::group_member(c, 0) : c;
  int n = 0;
  if (k->comm) NCCLCHK(c, ncclCommCount(k->comm, &n));
  if (rccl_nranks) *rccl_nranks = n;
  if (n_allreduce) *n_allreduce = k->n_allreduce;
  return 0;
}

int gfh_set_loss(gfh_ctx* c, int loss) {
  if (!c) return 1;
  GROUP(c, gfh_set_loss(k, loss));
  if (loss < GFH_LOSS_LINEAR || loss > GFH_LOSS_HUBER) return fail(c, "gfh_set_loss: unknown loss function");
  if (loss != c->gen.loss) { c->gen.loss = loss; c->cur = nullptr; c->have_sweep = false; }
  return 0;
}

int gfh_set_use_ad(gfh_ctx* c, int on) {
  if (!c) return 1;
  GROUP(c, gfh_set_use_ad(k, on));
  const bool fd = on == 0;
  if (fd != c->gen.finite_diff) { c->gen.finite_diff = fd; c->cur = nullptr; c->have_sweep = false; c->prepared = false; c->mesh_valid = false; }
  return 0;
}

int gfh_set_fd_column_sets(gfh_ctx* c, int on) {
  if (!c) return 1;
  GROUP(c, gfh_set_fd_column_sets(k, on));
  const bool v = on != 0;
  if (v != c->gen.fd_col_sets) { c->gen.fd_col_sets = v; c->cur = nullptr; c->have_sweep = false; c->prepared = false; }
  return 0;
}

int gfh_set_keep_jacobian(gfh_ctx* c, int mode) {
  if (!c) return 1;
  GROUP(c, gfh_set_keep_jacobian(k, mode));
  if (mode < 0 || mode > 2) return fail(c, "gfh_set_keep_jacobian: mode must be 0, 1 or 2");
  c->keep_jacobian = mode;
  gfh::set_store_j(c, mode != 0);
  if (mode != 2) gfh::set_store_res(c, true);
  return 0;
}

int gfh_set_placement_tries(gfh_ctx* c, int tries) {
  if (!c) return 1;
  GROUP(c, gfh_set_placement_tries(k, tries));
  if (tries < 1 || tries > 16) return fail(c, "gfh_set_placement_tries: between 1 and 16");
  c->placement_tries = tries;
  return 0;
}
int gfh_set_placement_after(gfh_ctx* c, int sweeps) {
  if (!c) return 1;
  GROUP(c, gfh_set_placement_after(k, sweeps));
  if (sweeps < 0) return fail(c, "gfh_set_placement_after: a number of sweeps >= 0");
  c->placement_after = sweeps;
  return 0;
}
int gfh_get_placement(gfh_ctx* c, double* out8) {
  if (!c) return 1;
  if (c->grp) return gfh_get_placement(gfh::group_member(c, 0), out8);
  for (int k = 0; k < 7; k++) out8[k] = k < c->placement_n ? c->placement_ms[k] : 0.0;
  out8[7] = c->placement_n ? 1e-9 * c->placement_copy_rate : 0.0;      // GB/s of the copy the thresholds were scaled with
  return 0;
}
int gfh_set_timer_detail(gfh_ctx* c, int level) {
  if (!c) return 1;
  GROUP(c, gfh_set_timer_detail(k, level));
  if (level < 0 || level > 2) return fail(c, "gfh_set_timer_detail: level must be 0, 1 or 2");
  c->timer_detail = level;
  return 0;
}

int gfh_set_lookahead(gfh_ctx* c, int on) {
  if (!c) return 1;
  GROUP(c, gfh_set_lookahead(k, on));
  c->lookahead = on != 0;
  return 0;
}

int gfh_debug_set_rank(gfh_ctx* c, int nranks, int rank) {
  if (!c) return 1;
  NOT_FOR_GROUP(c, "gfh_debug_set_rank");
  if (c->member_of) return fail(c, "context belongs to a device group");
  if (nranks < 1 || rank < 0 || rank >= nranks) return fail(c, "bad communicator geometry");
  if (c->comm) return fail(c, "context already has a communicator");
  c->nranks = nranks; c->rank = rank;
  return 0;
}

int gfh_comm_init_from_env(gfh_ctx* c) {
  const char* nr = getenv("GADFIT_HIP_NRANKS");
  if (c && c->grp) return nr ? fail(c, "GADFIT_HIP_NRANKS (one process per GPU) and a device group exclude each other") : 0;
  if (!nr) return c ? 0 : 1;      // (before the device is needed: a context from gfh_create_begin may still be setting it up)
  NEED_GPU(c);
  if (atoi(nr) < 1) return fail(c, "GADFIT_HIP_NRANKS must be >= 1");
  const char* rk = getenv("GADFIT_HIP_RANK");
  const char* path = getenv("GADFIT_HIP_IDFILE");
  if (!rk || !path) return fail(c, "GADFIT_HIP_NRANKS needs GADFIT_HIP_RANK and GADFIT_HIP_IDFILE");
  const int nranks = atoi(nr), rank = atoi(rk);
  unsigned char id[GFH_UNIQUE_ID_BYTES];
  if (rank == 0) {
    if (gfh_comm_unique_id(id)) return fail(c, std::string(gfh_last_error(nullptr)));
    std::string tmp = std::string(path) + ".tmp";
    FILE* f = fopen(tmp.c_str(), "wb");
    if (!f || fwrite(id, 1, sizeof id, f) != sizeof id) { if (f) fclose(f); return fail(c, "cannot write GADFIT_HIP_IDFILE"); }
    fclose(f);
    if (rename(tmp.c_str(), path) != 0) return fail(c, "cannot publish GADFIT_HIP_IDFILE");
  } else {
    bool ok = false;
    for (int tries = 0; tries < 6000 && !ok; tries++) {      // up to ~60 s
      FILE* f = fopen(path, "rb");
      if (f) { ok = fread(id, 1, sizeof id, f) == sizeof id; fclose(f); }
      if (!ok) { struct timespec ts = {0, 10 * 1000 * 1000}; nanosleep(&ts, nullptr); }
    }
    if (!ok) return fail(c, "timed out waiting for GADFIT_HIP_IDFILE");
  }
  const int rc = gfh_comm_init(c, nranks, rank, id);
  // (ncclCommInitRank is a rendezvous: when it has returned here every rank holds the id, and a file left behind would be read as
  // the id of the NEXT run that names the same path)
  if (rank == 0) (void)remove(path);
  return rc;
}

void gfh_partition(int64_t n_total, int nranks, int rank, int64_t* begin, int64_t* count) {
  // gadfit.F90:978-983 with img_weights = 1/num_images: sizes = int(w*N), remainder +1 to
  // the first images.
  std::vector<int64_t> sizes(nranks);
  int64_t tmp = 0;
  for (int i = 0; i < nranks; i++) { sizes[i] = (int64_t)((1.0 / nranks) * (double)n_total); tmp += sizes[i]; }
  for (int i = 0; i < nranks; i++) if (i + 1 <= n_total - tmp) sizes[i]++;
  int64_t b = 0;
  for (int i = 0; i < rank; i++) b += sizes[i];
  *begin = b; *count = sizes[rank];
}

// gadfit.F90:977-983 with arbitrary image weights (re_initialize STEP 2): sizes = int(w*N), remainder +1 to the first images
static void partition_weighted(int64_t n_total, const std::vector<double>& w, int rank, int64_t* begin, int64_t* count) {
  const int n = (int)w.size();
  std::vector<int64_t> sizes(n);
  int64_t tmp = 0;
  for (int i = 0; i < n; i++) { sizes[i] = (int64_t)(w[i] * (double)n_total); if (sizes[i] < 0) sizes[i] = 0; tmp += sizes[i]; }
  for (int i = 0; i < n; i++) if (i + 1 <= n_total - tmp) sizes[i]++;
  int64_t b = 0;
  for (int i = 0; i < rank; i++) b += sizes[i];
  *begin = b; *count = sizes[rank];
}

// ------------------------------------------------------------------------- data
constexpr int kGramTarget = 512;       // aimed number of gram workgroups (about two per CU)
constexpr int kGramTargetFine = 8192;  // models with integrate(): the cost of a point varies along x (number of bisections), so
                                       // the contiguous blocks are kept small and the hardware deals them out as workgroups retire
constexpr int kPassGranule = 512;      // slots one pass of an 8-wave workgroup covers; divides kPadGranule
// Number of gram workgroups to aim for: about two 8-wave workgroups per CU; many small ones for quadrature models.
// (4-wave workgroups on 768 blocks for the VALU form of the fused kernel were measured: cfg 2 0.174 against 0.166 ms, and
// gfh_k_chi2 on the same partition 0.077 against 0.064 ms.)
static int gb_target_for(const gfh_ctx* c) {
  if (const char* e = getenv("GADFIT_HIP_GB_TARGET")) { const int v = atoi(e); if (v > 0) return v; }      // (experiments)
  return c->has_model && c->model.has_integrals() ? kGramTargetFine : kGramTarget;
}

static int build_layout(gfh_ctx* c) {
  // local per-dataset ranges = intersection of [begin, begin+count) with each dataset
  // (equivalent to img_bounds, gadfit.F90:984-1002)
  const int nd = c->nd;
  c->lb.assign(nd + 1, 0);
  c->ds_slot.assign(nd + 1, 0);
  const int64_t lo = c->begin, hi = c->begin + c->count;
  for (int d = 0; d < nd; d++) {
    int64_t a = std::max(lo, c->dp[d]), b = std::min(hi, c->dp[d + 1]);
    int64_t len = b > a ? b - a : 0;
    c->lb[d + 1] = c->lb[d] + len;
    int64_t padded = (len + kPadGranule - 1) / kPadGranule * kPadGranule;
    c->ds_slot[d + 1] = c->ds_slot[d] + padded;
  }
  c->n_slots = c->ds_slot[nd];
  c->ldj = c->n_slots;
  // gram workgroups: whole 256-slot tiles of one dataset each
  c->gb_target = gb_target_for(c);
  const int target = c->gb_target;
  int64_t per = (c->n_slots + target - 1) / target;
  per = std::max<int64_t>(kPassGranule, (per + kPassGranule - 1) / kPassGranule * kPassGranule);   // whole passes of the widest workgroup (8 waves)
  // a few passes in all (the fits of a few hundred points most of gadfit's use consists of): one workgroup per dataset -- a pass
  // costs ~2 us, a hand-off between workgroups ~5, and a single workgroup takes the fused kernel's short tail
  if (c->n_slots <= 4 * kPassGranule) per = std::max<int64_t>(per, c->n_slots);
  c->h_gb_start.clear(); c->h_gb_slots.clear(); c->h_gb_ds.clear(); c->h_ds_first_gb.assign(nd + 1, 0);
  for (int d = 0; d < nd; d++) {
    c->h_ds_first_gb[d] = (int)c->h_gb_start.size();
    for (int64_t s = c->ds_slot[d]; s < c->ds_slot[d + 1]; s += per) {
      c->h_gb_start.push_back(s);
      c->h_gb_slots.push_back((int)std::min<int64_t>(per, c->ds_slot[d + 1] - s));
      c->h_gb_ds.push_back(d);
    }
  }
  c->h_ds_first_gb[nd] = (int)c->h_gb_start.size();
  c->n_gb = (int)c->h_gb_start.size();
  return 0;
}

// The Jacobian buffer: `na` column streams ldj * 8 bytes apart, written concurrently by every workgroup -- the traffic that
// bounds the sweep.  How fast the part absorbs them is a matter of the physical pages behind the allocation, and that is the
// luck of the draw: over a row of fresh allocations of the 2.6 GB buffer of the headline size the store stream alone takes
// 0.41 ... 0.47 ms (the same virtual address, different pages, reads either) and the fused kernel 0.46 ... 0.52 ms -- what
// rounds 1 and 2 first read as a power state of the box.  So a large buffer is PLACED: allocated here, and once `placement_after`
// sweeps have written it (a job that has run that long is taken to run on: the search costs as much as 50-110 sweeps)
// up to `placement_tries` allocations are held at once (place_jacobian_now), each timed with four launches of
// the kernel that is about to run, the fastest kept, the others freed.
static int place_jacobian(gfh_ctx* c, int na) {
  const size_t bytes = sizeof(double) * (size_t)na * (size_t)std::max<int64_t>(1, c->ldj);
  if (c->J.bytes >= bytes && c->J.p) return 0;
  if (dev_alloc(c, c->J, bytes)) return 1;
  c->placement_n = 0; c->sweeps_on_J = 0;
  // (only where the kernel that writes the buffer is bound by its store stream: the sweeps of models with integrate() are bound by
  // the quadrature arithmetic, no placement could show in their time)
  c->placement_pending = c->placement_tries >= 2 && bytes >= ((size_t)256 << 20) && c->n_gb > 0 && !(c->has_model && c->model.has_integrals());
  return 0;
}

static int upload_tables(gfh_ctx* c) {
  const int ngb = std::max(1, c->n_gb);
  if (dev_alloc(c, c->gb_start, sizeof(int64_t) * ngb) || dev_alloc(c, c->gb_slots, sizeof(int) * ngb) ||
      dev_alloc(c, c->gb_ds, sizeof(int) * ngb) || dev_alloc(c, c->ds_first_gb, sizeof(int) * (c->nd + 1))) return 1;
  if (c->n_gb) {
    HIPCHK(c, hipMemcpy(c->gb_start.p, c->h_gb_start.data(), sizeof(int64_t) * c->n_gb, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->gb_slots.p, c->h_gb_slots.data(), sizeof(int) * c->n_gb, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->gb_ds.p, c->h_gb_ds.data(), sizeof(int) * c->n_gb, hipMemcpyHostToDevice));
  }
  HIPCHK(c, hipMemcpy(c->ds_first_gb.p, c->h_ds_first_gb.data(), sizeof(int) * (c->nd + 1), hipMemcpyHostToDevice));
  c->tile = 0;   // tile_ds is rebuilt lazily for the kernel's tile size
  return 0;
}

// the gram-block partition follows the model kind (build_layout): rebuilt when a model set AFTER the data changes it
static int ensure_gb_partition(gfh_ctx* c) {
  if (!c->nd || c->gb_target == gb_target_for(c)) return 0;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (build_layout(c)) return 1;
  c->prepared = false; c->have_sweep = false; c->tail_host.clear();
  return upload_tables(c);
}

static int ensure_tile_table(gfh_ctx* c) {
  const int tile = c->gen.block;
  if (c->tile == tile) return 0;
  if (kPadGranule % tile) return fail(c, "tile size must divide the pad granule");
  c->n_tiles = (int)(c->n_slots / tile);
  std::vector<int> t(std::max(1, c->n_tiles));
  for (int d = 0; d < c->nd; d++)
    for (int64_t s = c->ds_slot[d] / tile; s < c->ds_slot[d + 1] / tile; s++) t[s] = d;
  if (dev_alloc(c, c->tile_ds, sizeof(int) * t.size())) return 1;
  HIPCHK(c, hipMemcpy(c->tile_ds.p, t.data(), sizeof(int) * t.size(), hipMemcpyHostToDevice));
  c->tile = tile;
  return 0;
}

// xs/ys/ws point at the element with global index `begin` (local slice)
static int upload_points_impl(gfh_ctx* c, const double* xs, const double* ys, const double* ws);
static int upload_points(gfh_ctx* c, const double* xs, const double* ys, const double* ws) {
  // no C++ exception may cross the C ABI: host staging of N-sized arrays can run out of memory
  try { return upload_points_impl(c, xs, ys, ws); }
  catch (const std::exception& e) { return fail(c, std::string("gfh_set_data: ") + e.what()); }
}
static int upload_points_impl(gfh_ctx* c, const double* xs, const double* ys, const double* ws) {
  gfh::Range range("gadfit upload of the data points");
  copy_path_ready();
  const size_t nb = sizeof(double) * (size_t)std::max<int64_t>(1, c->n_slots);
  if (dev_alloc(c, c->x, nb) || dev_alloc(c, c->y, nb) || dev_alloc(c, c->w, nb) || dev_alloc(c, c->res, nb) ||
      dev_alloc(c, c->omega, nb) || dev_alloc(c, c->is_pad, (size_t)std::max<int64_t>(1, c->n_slots))) return 1;
  const double* src[3] = {xs, ys, ws};
  DevBuf* dst[3] = {&c->x, &c->y, &c->w};
  if (c->nd <= 256 && c->n_slots) {
    // Few, long datasets (the large-N case): every dataset's points go down straight from the caller's arrays, one copy per
    // array and dataset, and a small kernel writes the pad slots -- no host-side staging pass over N-sized arrays (that pass and the
    // staged copies were 35 ms of a 50 ms hand-over at N = 1e7; a ten-iteration fit is 5-6 ms).
    std::vector<int64_t> seg((size_t)3 * c->nd);
    for (int d = 0; d < c->nd; d++) { seg[3 * d] = c->ds_slot[d]; seg[3 * d + 1] = c->lb[d + 1] - c->lb[d]; seg[3 * d + 2] = c->ds_slot[d + 1]; }
    DevBuf dseg;
    if (dev_alloc(c, dseg, sizeof(int64_t) * seg.size())) return 1;
    hipError_t e = hipMemcpy(dseg.p, seg.data(), sizeof(int64_t) * seg.size(), hipMemcpyHostToDevice);
    // (the FIRST upload of a process takes ~16 ms for 3 x 80 MB, every later one ~5 ms -- fresh arrays, a second context alike,
    // tools/probes/upload_cost.py: a one-time cost of the runtime's copy path, not of these arrays; three threads, one per array,
    // change nothing.  warm_copy_path pays it beside the caller's own work after gfh_create, where there is any.)
    for (int k = 0; k < 3 && e == hipSuccess; k++)
      for (int d = 0; d < c->nd && e == hipSuccess; d++) {
        const int64_t len = c->lb[d + 1] - c->lb[d];
        if (len) e = hipMemcpy(dst[k]->as<double>() + c->ds_slot[d], src[k] + c->lb[d], sizeof(double) * (size_t)len, hipMemcpyHostToDevice);
      }
    if (e == hipSuccess) e = hipMemsetAsync(c->is_pad.p, 0, (size_t)c->n_slots, c->stream);
    if (e == hipSuccess) e = launch_fill_pads(c->stream, c->nd, dseg.as<i64>(), c->x.as<double>(), c->y.as<double>(), c->w.as<double>(), c->is_pad.as<unsigned char>());
    if (e == hipSuccess) e = hipMemsetAsync(c->res.p, 0, sizeof(double) * (size_t)c->n_slots, c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(c->omega.p, 0, sizeof(double) * (size_t)c->n_slots, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    dev_free(dseg);
    if (e != hipSuccess) return fail(c, std::string("gfh_set_data: ") + hipGetErrorString(e));
    c->have_sweep = false;
    return 0;
  }
  std::vector<double> stage((size_t)c->n_slots);
  std::vector<unsigned char> pad((size_t)c->n_slots, 1);
  for (int k = 0; k < 3; k++) {
    for (int d = 0; d < c->nd; d++) {
      const int64_t len = c->lb[d + 1] - c->lb[d];
      const int64_t s0 = c->ds_slot[d], s1 = c->ds_slot[d + 1];
      if (len) memcpy(&stage[(size_t)s0], src[k] + c->lb[d], sizeof(double) * (size_t)len);
      // pad slots: a real abscissa of the same dataset (so f stays finite), y = 0, w = 0
      const double fill = (k == 0 && len) ? src[0][c->lb[d] + len - 1] : 0.0;
      for (int64_t s = s0 + len; s < s1; s++) stage[(size_t)s] = fill;
      if (k == 0) for (int64_t s = s0; s < s0 + len; s++) pad[(size_t)s] = 0;
    }
    if (c->n_slots) HIPCHK(c, hipMemcpy(dst[k]->p, stage.data(), sizeof(double) * (size_t)c->n_slots, hipMemcpyHostToDevice));
  }
  if (c->n_slots) {
    HIPCHK(c, hipMemcpy(c->is_pad.p, pad.data(), (size_t)c->n_slots, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemset(c->res.p, 0, sizeof(double) * (size_t)c->n_slots));
    HIPCHK(c, hipMemset(c->omega.p, 0, sizeof(double) * (size_t)c->n_slots));
  }
  c->have_sweep = false;
  return 0;
}

static int set_geometry(gfh_ctx* c, int64_t n_total, int nd, const int64_t* dp) {
  if (nd < 1 || !dp || dp[0] != 0 || dp[nd] != n_total) return fail(c, "data_positions must start at 0 and end at n_total");
  for (int d = 0; d < nd; d++) if (dp[d + 1] < dp[d]) return fail(c, "data_positions must be non-decreasing");
  c->n_total = n_total; c->nd = nd; c->dp.assign(dp, dp + nd + 1);
  // new data: the Jacobian/residuals on the device are stale, and the kernel form follows n_datasets
  c->cur = nullptr; c->cur_active.clear(); c->have_sweep = false; c->j_valid = false; c->prepared = false;
  c->n_aux = 0;                     // auxiliary columns belong to the data they were tabulated for
  c->mesh_valid = false;
  c->order_ready = false; c->order_want = true;
  if ((int)c->part_w.size() == c->nranks) partition_weighted(n_total, c->part_w, c->rank, &c->begin, &c->count);
  else gfh_partition(n_total, c->nranks, c->rank, &c->begin, &c->count);
  return build_layout(c);
}

int gfh_set_data(gfh_ctx* c, int64_t n_total, const double* x, const double* y, const double* w, int nd, const int64_t* dp) {
  GROUP(c, gfh_set_data(k, n_total, x, y, w, nd, dp));      // every member uploads its own contiguous range (gadfit.F90:977-983)
  NEED_GPU(c);
  if (!x || !y || !w) return fail(c, "null data array");
  c->part_w.clear(); c->lb_t_prev = 0.0; c->weights_type = -1; c->haux.clear(); c->h_n_aux = 0;
  if (set_geometry(c, n_total, nd, dp)) return 1;
  if (c->load_balancing) {
    try { c->hx.assign(x, x + n_total); c->hy.assign(y, y + n_total); c->hw.assign(w, w + n_total); }
    catch (const std::exception& e) { return fail(c, std::string("gfh_set_data (host copy for load balancing): ") + e.what()); }
  } else { c->hx.clear(); c->hy.clear(); c->hw.clear(); }
  if (upload_tables(c)) return 1;
  return upload_points(c, x + c->begin, y + c->begin, w + c->begin);
}

// gfh_set_data that returns at once: geometry and tables are set here, the N-sized copies run on a thread of the library and are
// waited for by the next call on this context (whose return code then carries a failure of the upload).  For callers that have
// host work of their own to do meanwhile -- the Fortran layer records eval() over the data (gadfit.F90, discover).
// The copy queued by gfh_queue_host_copy is made whatever becomes of the call it was queued for (an early return through
// gfh_set_data under load balancing, an error): on a thread of its own, or at once if none can be started; nothing stays queued.
static void start_host_copy(gfh_ctx* c) {
  if (c->host_copy.joinable()) c->host_copy.join();
  void* dst = c->hc_dst; const void* src = c->hc_src; const size_t bytes = c->hc_bytes;
  c->hc_dst = nullptr; c->hc_src = nullptr; c->hc_bytes = 0;
  if (!dst || !src || !bytes) return;
  try { c->host_copy = std::thread([dst, src, bytes]() { memcpy(dst, src, bytes); }); }
  catch (const std::exception&) { memcpy(dst, src, bytes); }
}

int gfh_set_data_begin(gfh_ctx* c, int64_t n_total, const double* x, const double* y, const double* w, int nd, const int64_t* dp) {
  GROUP(c, gfh_set_data_begin(k, n_total, x, y, w, nd, dp));
  // the caller's own copy of its abscissas (gfh_queue_host_copy): beside the upload, on a thread of its own -- 80 MB into fresh
  // pages take longer than the upload of 240 MB, and nothing on the device waits for them (gfh_wait_host_copy)
  if (c) start_host_copy(c);
  // (a context whose device part is still being set up, gfh_create_begin: the upload is queued behind it instead of waiting here)
  std::thread creation;
  if (c && c->device >= 0 && c->creating && c->pending.joinable() && !c->load_balancing) { creation = std::move(c->pending); c->creating = false; }
  else NEED_GPU(c);
  auto bail = [&](int rc) { if (creation.joinable()) { creation.join(); if (c->pending_rc) rc = 1; c->pending_rc = 0; } return rc; };
  if (!x || !y || !w) return bail(fail(c, "null data array"));
  if (c->load_balancing) return gfh_set_data(c, n_total, x, y, w, nd, dp);      // (keeps a host copy: nothing to overlap)
  c->part_w.clear(); c->lb_t_prev = 0.0; c->weights_type = -1; c->haux.clear(); c->h_n_aux = 0;
  if (set_geometry(c, n_total, nd, dp)) return bail(1);
  c->hx.clear(); c->hy.clear(); c->hw.clear();
  const int64_t b = c->begin;
  if (!creation.joinable()) c->pending_rc = 0;
  c->stop_warm.store(false);
  // (the creation thread, still running, travels into the upload thread inside `prev`.  Should that thread not start -- std::thread
  // throws on EAGAIN -- `prev` must be joined HERE: unwinding would destroy a joinable std::thread, which is std::terminate
  // before any handler runs, and bail() only knows `creation`, moved from by then: round-5 advisor)
  std::shared_ptr<std::thread> prev;
  try {
    prev = std::make_shared<std::thread>(std::move(creation));
    c->pending = std::thread([c, x, y, w, b, prev]() {
      int rc = 0;
      if (prev->joinable()) { prev->join(); rc = c->pending_rc; }      // (its failure is this upload's: create_failed holds the message)
      if (!rc) rc = hipSetDevice(c->device) == hipSuccess ? 0 : fail(c, "hipSetDevice failed");
      if (!rc) rc = upload_tables(c);
      if (!rc) rc = upload_points(c, x + b, y + b, w + b);
      // The caller is still busy on the host (that is why it asked for an upload in the background), and its first passes are
      // about to come: the part is kept busy until the caller is back (join_pending), so those passes do not start in the clock
      // ramp that follows an idle gap (20-35 % slower launches, tools/probes/transient.py).  At most 200 ms.
      c->warm_ms = 0;
      if (!rc && c->keep_warm && c->n_slots > 0) {
        const auto t0 = std::chrono::steady_clock::now();
        while (!c->stop_warm.load()) {
          if (launch_keep_warm(c->stream, c->x.as<double>(), c->n_slots, 8, c->res.as<double>()) != hipSuccess) { (void)hipGetLastError(); break; }
          if (hipStreamSynchronize(c->stream) != hipSuccess) { (void)hipGetLastError(); break; }
          c->warm_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
          if (c->warm_ms > 200.0) break;
        }
      }
      c->pending_rc = rc;
    });
  } catch (const std::exception& e) {
    int rc = fail(c, std::string("gfh_set_data_begin: ") + e.what());
    if (prev && prev->joinable()) { prev->join(); if (c->pending_rc) rc = 1; c->pending_rc = 0; }
    return bail(rc);
  }
  return 0;
}

// A host-to-host copy for the thread of the next gfh_set_data_begin to make once its upload is done (handle of a device group: member 0's thread).
int gfh_queue_host_copy(gfh_ctx* c, void* dst, const void* src, int64_t bytes) {
  if (!c) return 1;
  gfh_ctx* k = c->grp ? gfh::group_member(c, 0) : c;
  if (k->pending.joinable() && !k->creating) return fail(c, "gfh_queue_host_copy: an upload is in flight already");
  if (k->host_copy.joinable()) k->host_copy.join();
  k->hc_dst = dst; k->hc_src = src; k->hc_bytes = bytes > 0 ? (size_t)bytes : 0;
  return 0;
}
int gfh_wait_host_copy(gfh_ctx* c) {
  if (!c) return 1;
  gfh_ctx* k = c->grp ? gfh::group_member(c, 0) : c;
  if (k->host_copy.joinable()) k->host_copy.join();
  return 0;
}

int gfh_set_data_local(gfh_ctx* c, int64_t n_total, int nd, const int64_t* dp, int64_t begin, int64_t count,
                       const double* x, const double* y, const double* w) {
  NOT_FOR_GROUP(c, "gfh_set_data_local");
  NEED_GPU(c);
  if (c->load_balancing) return fail(c, "load balancing needs the whole arrays: use gfh_set_data");
  c->part_w.clear();
  if (set_geometry(c, n_total, nd, dp)) return 1;
  if (begin != c->begin || count != c->count) return fail(c, "local slice does not match gfh_partition for this rank");
  if (upload_tables(c)) return 1;
  return upload_points(c, x, y, w);
}

// Auxiliary per-point columns (GFH_AUX nodes): column k of the caller's [n_aux][ld] array, laid out on
// the device like x (per-dataset padding; pad slots repeat the dataset's last real point, their w is 0).
static int upload_aux(gfh_ctx* c, int n_aux, const double* aux_local, int64_t ld) try {
  if (!c->nd) return fail(c, "gfh_set_aux: set the data first (gfh_set_data)");
  if (n_aux < 0 || (n_aux > 0 && !aux_local)) return fail(c, "gfh_set_aux: bad arguments");
  c->n_aux = n_aux; c->aux_serial++; c->mesh_valid = false;
  if (!n_aux) return 0;
  if (dev_alloc(c, c->aux, sizeof(double) * (size_t)n_aux * (size_t)std::max<int64_t>(1, c->n_slots))) return 1;
  // From inside the parameter hook (columns that follow the parameters, refreshed before a pass) the copies below overwrite what the
  // kernels of the PREVIOUS pass read, and they are synchronous copies on the null stream while c->stream is non-blocking: nothing
  // but this wait orders them behind those kernels (the host has seen the previous pass's mailbox, but a result can arrive before
  // its kernel has retired: round-5 advisor).  A few microseconds before a tabulation of milliseconds.
  if (c->in_pars_hook) HIPCHK(c, hipStreamSynchronize(c->stream));
  // each dataset's segment straight from the caller's column (no staging copy of the whole column: at 1e7 points that copy and
  // its fresh pages cost more than the transfer), then its pad slots (fewer than 512 per dataset)
  std::vector<double> pads;
  for (int k = 0; k < n_aux; k++) {
    const double* src = aux_local + (size_t)k * (size_t)ld;
    double* dst = c->aux.as<double>() + (size_t)k * (size_t)c->n_slots;
    for (int d = 0; d < c->nd; d++) {
      const int64_t len = c->lb[d + 1] - c->lb[d];
      const int64_t s0 = c->ds_slot[d], s1 = c->ds_slot[d + 1];
      if (len) HIPCHK(c, hipMemcpy(dst + s0, src + c->lb[d], sizeof(double) * (size_t)len, hipMemcpyHostToDevice));
      if (s1 > s0 + len) {
        pads.assign((size_t)(s1 - s0 - len), len ? src[c->lb[d] + len - 1] : 0.0);
        HIPCHK(c, hipMemcpy(dst + s0 + len, pads.data(), sizeof(double) * pads.size(), hipMemcpyHostToDevice));
      }
    }
  }
  if (!c->in_pars_hook) c->have_sweep = false;
  return 0;
} catch (const std::exception& e) { return fail(c, std::string("gfh_set_aux: ") + e.what()); }

int gfh_set_aux(gfh_ctx* c, int n_aux, const double* aux) {
  GROUP(c, gfh_set_aux(k, n_aux, aux));
  NEED_GPU(c);
  if (c->load_balancing && n_aux > 0 && aux) {
    try { c->haux.assign(aux, aux + (size_t)n_aux * (size_t)c->n_total); c->h_n_aux = n_aux; }
    catch (const std::exception& e) { return fail(c, std::string("gfh_set_aux (host copy for load balancing): ") + e.what()); }
  } else { c->haux.clear(); c->h_n_aux = 0; }
  return upload_aux(c, n_aux, aux ? aux + c->begin : nullptr, c->n_total);
}
int gfh_set_aux_local(gfh_ctx* c, int n_aux, const double* aux_local) {
  NOT_FOR_GROUP(c, "gfh_set_aux_local");
  NEED_GPU(c);
  return upload_aux(c, n_aux, aux_local, c->count);
}

int gfh_set_load_balancing(gfh_ctx* c, int on) {
  if (!c) return 1;
  GROUP(c, gfh_set_load_balancing(k, on));
  c->load_balancing = on != 0;      // takes effect for data set from now on (the host copy is made by gfh_set_data)
  if (!on) { c->hx.clear(); c->hy.clear(); c->hw.clear(); c->haux.clear(); c->hx.shrink_to_fit(); c->hy.shrink_to_fit(); c->hw.shrink_to_fit(); c->haux.shrink_to_fit(); }
  return 0;
}

// New ranges for every rank from image weights (all ranks pass the same): layout, tables and this rank's points are
// rebuilt from the host copy; weights (gfh_init_weights) and auxiliary columns are re-applied.
int gfh_repartition(gfh_ctx* c, const double* weights) {
  GROUP(c, gfh_repartition(k, weights));
  NEED_GPU(c);
  if (!c->load_balancing || c->hx.empty()) return fail(c, "gfh_repartition needs gfh_set_load_balancing(1) before gfh_set_data");
  double sum = 0.0;
  for (int i = 0; i < c->nranks; i++) { if (!(weights[i] >= 0.0)) return fail(c, "gfh_repartition: negative weight"); sum += weights[i]; }
  if (!(sum > 0.0)) return fail(c, "gfh_repartition: weights sum to zero");
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->part_w.assign(weights, weights + c->nranks);
  for (double& v : c->part_w) v /= sum;                  // the sizes int(w*N) + remainder only add up to N for weights that sum to one
  const std::vector<int64_t> dp = c->dp;                 // set_geometry assigns c->dp from its argument
  const int n_aux = c->h_n_aux;
  if (set_geometry(c, c->n_total, c->nd, dp.data())) return 1;
  if (upload_tables(c)) return 1;
  if (upload_points(c, c->hx.data() + c->begin, c->hy.data() + c->begin, c->hw.data() + c->begin)) return 1;
  if (c->weights_type >= 0 && gfh_init_weights(c, c->weights_type)) return 1;
  if (n_aux && upload_aux(c, n_aux, c->haux.data() + c->begin, c->n_total)) return 1;
  c->lb_moves++;
  return 0;
}

int gfh_init_weights(gfh_ctx* c, int type) {
  GROUP(c, gfh_init_weights(k, type));
  NEED_GPU(c);
  if (type < 0 || type > 4) return fail(c, "Unknown weight specifier. Allowed values are NONE, SQRT_Y, PROPTO_Y, INVERSE_Y, and USER.");
  c->weights_type = type;
  if (!c->n_slots) return 0;
  HIPCHK(c, launch_init_weights(c->stream, type, c->n_slots, c->y.as<double>(), c->w.as<double>(), c->is_pad.as<unsigned char>()));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}

int64_t gfh_local_count(gfh_ctx* c) {
  if (c && c->grp) { int64_t n = 0; for (int r = 0; r < gfh::group_size(c); r++) n += gfh::group_member(c, r)->count; return n; }   // the whole array
  return c ? c->count : 0;
}
int64_t gfh_local_begin(gfh_ctx* c) { return c && !c->grp ? c->begin : 0; }
int gfh_group_ranges(gfh_ctx* c, int64_t* begins, int64_t* counts) {
  if (!c) return 1;
  if (!c->grp) { begins[0] = c->begin; counts[0] = c->count; return 0; }
  for (int r = 0; r < gfh::group_size(c); r++) { begins[r] = gfh::group_member(c, r)->begin; counts[r] = gfh::group_member(c, r)->count; }
  return 0;
}

// ------------------------------------------------------------------------- model
// The quadrature workspaces the next kernels carry (model.h, plan_workspaces): the fast form in scratch, or the user's sizes -- in
// scratch while they fit the budget, else in the context's global pool.
static void apply_ws_plan(gfh_ctx* c) {
  const gfh::WsPlan p = gfh::plan_workspaces(c->model, c->ws_fast, c->ws_grown);
  c->gen.ws_size = p.ws_size; c->gen.ws_size_inner = p.ws_size_inner; c->gen.ws_global = p.global;
  const int64_t wave = p.global ? gfh::wsg_wave_doubles(c->model, p.ws_size, p.ws_size_inner) : 0;
  if (wave != c->wsg_wave_doubles) {          // (another slot size: the pool is cut anew at the next launch that needs it)
    if (c->wsg.p && c->device >= 0) { hipSetDevice(c->device); if (c->stream) hipStreamSynchronize(c->stream); dev_free(c->wsg); }
    c->wsg_waves = 0; c->wsg_tried = 0; c->wsg_wave_doubles = wave;
  }
}

int gfh_set_model_variants(gfh_ctx* c, int n, const gfh_tape* const* t, int hint_aux) try {
  if (!c) return 1;
  GROUP(c, gfh_set_model_variants(k, n, t, hint_aux));
  std::string err;
  Model m;
  // (per-tape hint columns left by gfh_set_variant_hint_columns for exactly this hand-over)
  const std::vector<int32_t> cols = std::move(c->pending_hint_cols);
  c->pending_hint_cols.clear();
  if (!m.load_variants(n, t, hint_aux, &err, (int)cols.size() == n ? &cols : nullptr)) return fail(c, "gfh_set_model: " + err);
  if (gfh::join_pending(c)) return 1;
  if (c->device >= 0) { hipSetDevice(c->device); if (c->stream) hipStreamSynchronize(c->stream); for (auto& kv : c->kernel_cache) release_loaded(c->device, &kv.second); }
  c->kernel_cache.clear(); c->cur = nullptr; c->cur_active.clear(); c->have_sweep = false; c->prepared = false;
  c->model = std::move(m); c->has_model = true; c->model_serial++; c->mesh_valid = false;
  c->order_ready = false; c->order_want = true;
  // the kernels first carry small quadrature workspaces (fast: 3.2 KB of scratch per lane and level); a pass that exhausts them is
  // repeated with the user's sizes (grow_workspace)
  // (a model handed over by a recovery's handler keeps the grown state: the pass that is about to be repeated has needed it)
  c->ws_grown = c->ws_grown && c->in_recovery;
  apply_ws_plan(c);
  {
    // Models with integrate(): the plain kernels are bound by VALU issue and their bodies take 135-150 VGPRs as the compiler
    // allocates them (3 waves per SIMD; gfh_k_chi2's 8-wave workgroups then fit once per CU = 2 waves per SIMD).  Capped at 128
    // registers (4 waves) a handful of values spill and chi2 runs 20 % faster, the sweep 4 %; at 96 (5 waves) the spills cost
    // more than the waves bring (profiles/r03_cfg4.md).  GADFIT_HIP_WAVES_PER_EU overrides (0: the compiler's choice).
    const char* e = getenv("GADFIT_HIP_WAVES_PER_EU");
    c->gen.waves_per_eu = e ? atoi(e) : (c->model.has_integrals() ? 4 : 0);
    if (const char* pr = getenv("GADFIT_HIP_MATRIX_PRIO")) c->gen.matrix_prio = std::max(-3, std::min(3, atoi(pr)));
    if (const char* ab = getenv("GADFIT_HIP_ABLATE")) c->gen.ablate = atoi(ab);
    if (const char* fa = getenv("GADFIT_HIP_FRAG_AHEAD")) c->gen.frag_ahead = std::max(1, std::min(2, atoi(fa)));
  }
  return 0;
} catch (const std::exception& e) { return fail(c, std::string("gfh_set_model: ") + e.what()); }

int gfh_set_model(gfh_ctx* c, const gfh_tape* t) { return gfh_set_model_variants(c, 1, &t, -1); }

int gfh_set_variant_hint_columns(gfh_ctx* c, int n_tapes, const int32_t* cols) {
  if (!c) return 1;
  GROUP(c, gfh_set_variant_hint_columns(k, n_tapes, cols));
  if (n_tapes < 0 || (n_tapes > 0 && !cols)) return fail(c, "gfh_set_variant_hint_columns: bad arguments");
  c->pending_hint_cols.assign(cols, cols + n_tapes);
  return 0;
}

int gfh_model_needs_hint(gfh_ctx* c) {
  if (!c) return -1;
  if (c->grp) return gfh_model_needs_hint(gfh::group_member(c, 0));
  if (!c->has_model) return -1;
  try { return c->model.needs_hint() ? 1 : 0; } catch (const std::exception&) { return -1; }
}
int gfh_model_n_variants(gfh_ctx* c) {
  if (!c) return 0;
  if (c->grp) return gfh_model_n_variants(gfh::group_member(c, 0));
  return c->has_model ? c->model.n_variants() : 0;
}
int gfh_model_n_tapes(gfh_ctx* c) {
  if (!c) return 0;
  if (c->grp) return gfh_model_n_tapes(gfh::group_member(c, 0));
  return c->has_model ? c->model.n_tapes : 0;
}
int gfh_get_counters(gfh_ctx* c, int64_t* out4) {
  if (!c || !out4) return 1;
  gfh_ctx* k = c->grp ? gfh::group_member(c, 0) : c;
  out4[0] = k->n_unseen_rounds; out4[1] = k->n_mesh_replays; out4[2] = k->has_model ? k->model.n_variants() : 0;
  out4[3] = k->has_model ? ((int64_t)k->gen.ws_size << 32) + k->gen.ws_size_inner : 0;
  return 0;
}
// What the last recording pass of a quadrature model did, from the device's own mesh records (one per slot and outermost integrate()
// call site: byte 0 = bisections of that adaptive integral, 255 = none recorded): the work count behind the algorithmic roofline of
// BASELINE config 4 (numerical_integration.F90:236-284: n intervals = (2n - 1) panels of the bisection + n of the final pass).
int gfh_debug_mesh_stats(gfh_ctx* c, int64_t* out4) {
  if (!c || !out4) return 1;
  NOT_FOR_GROUP(c, "gfh_debug_mesh_stats");
  gfh_ctx* k = c;
  NEED_GPU(k);
  out4[0] = out4[1] = out4[2] = out4[3] = 0;
  if (!k->mesh.p || !k->mesh_stride || !k->mesh_valid) return fail(c, "gfh_debug_mesh_stats: no recorded quadrature mesh (a pass of a model with integrate() must have run)");
  HIPCHK(k, hipStreamSynchronize(k->stream));
  const size_t bytes = (size_t)k->mesh_stride * (size_t)k->n_slots;
  std::vector<unsigned char> h(bytes);
  HIPCHK(k, hipMemcpy(h.data(), k->mesh.p, bytes, hipMemcpyDeviceToHost));
  const int sites = k->mesh_stride / kMeshRecord;
  // (data slots only: the pads between datasets carry w = 0 and are evaluated like any other slot, but are not data)
  for (int d = 0; d < k->nd; d++)
    for (int64_t sl = k->ds_slot[d], e = k->ds_slot[d] + (k->lb[d + 1] - k->lb[d]); sl < e; sl++)
      for (int q = 0; q < sites; q++) {
        const unsigned char v = h[(size_t)sl * k->mesh_stride + (size_t)q * kMeshRecord];
        if (v == 255) out4[2]++; else { out4[0]++; out4[1] += v; }
      }
  out4[3] = (int64_t)sites;
  return 0;
}
int gfh_set_pars_hook(gfh_ctx* c, gfh_pars_hook fn, void* user) {
  if (!c) return 1;
  GROUP(c, gfh_set_pars_hook(k, fn, user));
  c->pars_fn = fn; c->pars_user = user;
  return 0;
}
int gfh_device_memory(gfh_ctx* c, int64_t* out3) {
  if (!c || !out3) return 1;
  gfh_ctx* k = c->grp ? gfh::group_member(c, 0) : c;
  if (k->device < 0) return fail(c, "no GPU bound to this context");
  if (hipSetDevice(k->device) != hipSuccess) return fail(c, "hipSetDevice failed");
  size_t free_b = 0, total_b = 0;
  HIPCHK(c, hipMemGetInfo(&free_b, &total_b));
  out3[0] = (int64_t)free_b; out3[1] = (int64_t)total_b; out3[2] = 0;
  const int n = c->grp ? gfh_group_size(c) : 1;
  for (int r = 0; r < n; r++) out3[2] += (int64_t)(c->grp ? gfh::group_member(c, r) : c)->wsg.bytes;
  return 0;
}
int gfh_set_unseen_handler(gfh_ctx* c, gfh_unseen_handler fn, void* user) {
  if (!c) return 1;
  GROUP(c, gfh_set_unseen_handler(k, fn, user));
  c->unseen_fn = fn; c->unseen_user = user;
  return 0;
}

constexpr int kMaxKernargPars = 480;   // doubles; the kernel-argument segment holds 4 KiB

int64_t gfh_model_source(gfh_ctx* c, int n_act, const int32_t* active, char* buf, int64_t cap) {
  if (c && c->grp) {
    gfh_ctx* k0 = gfh::group_member(c, 0);
    const int64_t n = gfh_model_source(k0, n_act, active, buf, cap);
    if (n < 0) fail(c, k0->err);
    return n;
  }
  if (!c || !c->has_model) { fail(c, "no model set"); return -1; }
  std::string src, err;
  std::vector<int32_t> a(active, active + n_act);
  GenConfig cfg = c->gen;
  const int np = c->model.n_pars;
  if (c->kernarg && np >= 1 && (int64_t)std::max(1, c->nd) * np <= kMaxKernargPars) cfg.kernarg_pars = std::max(1, c->nd) * np;
  if (!generate_source(c->model, a, cfg, &src, &err)) { fail(c, err); return -1; }
  if (buf && cap > 0) { size_t n = std::min<size_t>((size_t)cap - 1, src.size()); memcpy(buf, src.data(), n); buf[n] = 0; }
  return (int64_t)src.size() + 1;
}

static int get_kernels_variant(gfh_ctx* c, const std::vector<int32_t>& active, bool load, int kernarg_pars) {
  // loaded kernels are keyed by the active set and the generator options that can change per context
  std::vector<int32_t> key = active;
  key.push_back(-1 - c->gen.loss - 4 * (c->gen.finite_diff ? 1 : 0) - 8 * (c->gen.store_j ? 0 : 1) - 16 * (c->gen.store_res ? 0 : 1) - 32 * kernarg_pars);
  key.push_back(-1 - c->gen.ws_size); key.push_back(-1 - c->gen.ws_size_inner); key.push_back(c->gen.ws_global ? -2 : -1);
  key.push_back(c->gen.finite_diff && c->gen.fd_col_sets ? -2 : -1);
  auto it = c->kernel_cache.find(key);
  if (it != c->kernel_cache.end()) { c->cur = &it->second; return 0; }
  std::string src, err;
  GenConfig cfg = c->gen; cfg.kernarg_pars = kernarg_pars;
  if (!generate_source(c->model, active, cfg, &src, &err)) return fail(c, err);
  ModelKernels mk;
  const uint64_t skey = load ? source_key(src) : 0;
  if (!(load && acquire_loaded(c->device, skey, &mk))) {       // (a code object this process already has loaded on this card: rtc.h)
    std::vector<char> code; bool cached = false;
    if (!compile_to_code_object(src, &code, &err, &cached)) return fail(c, err);
    if (!load) return 0;
    if (!load_kernels(code, &mk, &err)) return fail(c, err);
    publish_loaded(c->device, skey, mk);
  }
  mk.kernarg_pars = kernarg_pars; mk.n_active = (int)active.size();
  c->cur = &c->kernel_cache.emplace(key, mk).first->second;
  return 0;
}

static int get_kernels(gfh_ctx* c, const std::vector<int32_t>& active, bool load) {
  if (!c->has_model) return fail(c, "no model set (gfh_set_model)");
  const int np = c->model.n_pars;
  const bool can = c->kernarg && np >= 1 && np <= kMaxKernargPars;
  if (c->device < 0 && !c->nd) {          // compile-only context without data: the one-dataset and the pointer form go to the cache
    if (can && get_kernels_variant(c, active, load, np)) return 1;
    return get_kernels_variant(c, active, load, 0);
  }
  // the whole [n_datasets][n_pars] block by value while it fits the kernel-argument segment
  const bool fits = can && c->nd >= 1 && (int64_t)c->nd * np <= kMaxKernargPars;
  return get_kernels_variant(c, active, load, fits ? c->nd * np : 0);
}

int gfh_model_prepare(gfh_ctx* c, int n_act, const int32_t* active) {
  if (!c) return 1;
  GROUP(c, gfh_model_prepare(k, n_act, active));      // compiled once: rtc.cpp serialises, the other members load the cached code object
  std::vector<int32_t> a(active, active + n_act);
  if (c->device >= 0) return get_kernels(c, a, false);
  // compile-only context (build time): also the forms gfh_fit switches to under keep_jacobian mode 2 -- without the Jacobian
  // store (plain fits) and without the residual store -- so that a GPU box finds them in the cache
  const bool sj = c->gen.store_j, sr = c->gen.store_res;
  int rc = get_kernels(c, a, false);
  const bool combos[2][2] = {{false, false}, {true, false}};
  for (int k = 0; k < 2 && !rc; k++) {
    c->gen.store_j = combos[k][0] || !c->fused || c->model.has_integrals() || n_act > fused_max_active(c->gen); c->gen.store_res = combos[k][1];
    rc = get_kernels(c, a, false);
  }
  c->gen.store_j = sj; c->gen.store_res = sr;
  return rc;
}

// ------------------------------------------------------------------------- launches
// recorders (Fortran module state, the Python tracer) are not re-entrant: ONE callback into the host layer at a time, whichever it is --
// the parameter hook of one member of a device group must not run beside the unseen-branch handler of another
static std::recursive_mutex g_handler_mutex;      // (recursive: a callback that makes a call which calls back stays on its own thread)
static int upload_pars(gfh_ctx* c, const double* pars) {
  const size_t n = (size_t)c->nd * c->model.n_pars;
  if (pinned_reserve(c, 4096)) return 1;
  if (c->h_pars_bytes < sizeof(double) * n) {
    if (c->h_pars) hipHostFree(c->h_pars);
    c->h_pars = nullptr; c->h_pars_bytes = 0;
    HIPCHK(c, hipHostMalloc((void**)&c->h_pars, sizeof(double) * n, hipHostMallocDefault));
    c->h_pars_bytes = sizeof(double) * n;
  }
  if (dev_alloc(c, c->pars, sizeof(double) * n)) return 1;
  // every public call ends with a stream synchronise, so the staging buffer is free here
  memcpy(c->h_pars, pars, sizeof(double) * n);
  if (c->pars_fn) {            // (gfh_set_pars_hook: the host's reals that follow the parameters, refreshed in the staging copy)
    int rc;
    // (in_pars_hook: columns the hook uploads belong to the parameters of THIS pass -- a real that follows the parameters and the
    // abscissa, tabulated anew; what the device holds of the last sweep -- active set, Jacobian, residuals -- stays what it was)
    { std::lock_guard<std::recursive_mutex> lk(g_handler_mutex); c->in_pars_hook = true; rc = c->pars_fn(c->pars_user, c, c->h_pars); c->in_pars_hook = false; }
    if (rc) return fail(c, "the parameter hook failed (gfh_set_pars_hook)" + (c->err.empty() ? std::string() : ": " + c->err));
  }
  // kernels that take the block by value read it from c->h_pars at launch (the runtime copies kernel
  // arguments during the launch call); nothing is queued on the stream
  if (c->cur && c->cur->kernarg_pars) return 0;
  HIPCHK(c, hipMemcpyAsync(c->pars.p, c->h_pars, sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
  return 0;
}

// STEP 1 + STEP 2 in one kernel?  Up to 64 active parameters (4 tiles of 16); beyond that the plain sweep
// writes J and k_gram_block forms the Gram image from it.
// Models with integrate() also take the two-kernel path: the adaptive quadrature makes the per-point work
// long and uneven, and the fused kernel's one 8-wave workgroup per CU with its LDS stage loses to the plain
// sweep's small workgroups (cfg 4: 2.39 ms fused against 1.77 + 0.02 ms).
static bool fusable_model(const gfh_ctx* c) { return !(c->has_model && c->model.has_integrals()); }
static bool use_fused(const gfh_ctx* c) {
  return c->fused && fusable_model(c) && (int)c->cur_active.size() <= fused_max_active(c->gen) && c->cur && c->cur->sweep_gram;
}

extern "C++" { namespace gfh {
bool uses_fused_kernel(const gfh_ctx* c) { return use_fused(c); }
// Is the sum r^2 a sweep returns bitwise what chi2() returns at the same parameters (what the look-ahead schedule needs)?  The fused
// kernel: by construction (same partition and order of additions as gfh_k_chi2).  The two-kernel path with up to 8 active parameters
// (quadrature models; GADFIT_HIP_FUSED=0): k_gram_small sums r^2 per lane over the lane's points in ascending order, wave tree, waves
// in order -- gfh_k_chi2's map and order at its 8 waves per workgroup -- and k_reduce_partials / k_gather_sum are the order gfh_k_chi2's
// tail restates; the residuals themselves agree bit for bit (same value expressions; the quadrature's final pass rounds its panel
// sums like the value-only pass).  Pinned by test_chi2_is_bitwise_the_sweeps_sum_of_squares*.
bool sweep_chi2_is_bitwise(const gfh_ctx* c) {
  if (use_fused(c)) return true;
  return c->cur && c->cur_active.size() <= 8 && fused_waves_for((int)c->cur_active.size(), c->gen) == 8 && !c->gen.finite_diff;
}
} }

// The mode a pass at `pars` runs its quadrature in (generated kernels, mesh_build): 2 = replay the recorded bisections (they were made
// at exactly these parameters), 1 = bisect and record (recording pass: from now on the record belongs to these parameters), 0 = bisect.
static int mesh_mode_for(gfh_ctx* c, const double* pars, bool recording_pass) {
  if (!c->mesh.p || !c->mesh_stride || !pars) return 0;
  const size_t n = (size_t)c->nd * c->model.n_pars;
  if (c->mesh_valid && c->mesh_pars.size() == n && !memcmp(c->mesh_pars.data(), pars, sizeof(double) * n)) { c->n_mesh_replays++; return 2; }
  if (!recording_pass) return 0;
  c->mesh_pars.assign(pars, pars + n); c->mesh_valid = true;
  return 1;
}

static int resident_grid(gfh_ctx* c, hipFunction_t f, int threads);

// Kernels whose quadrature workspaces are the global pool (GenConfig::ws_global): the pool holds one slot per wave of a launch, so the
// grid is capped at the slots there are -- as many workgroups as are resident at once where the memory allows (more would only wait
// for a second round) -- and the kernels stride over their tiles / gram blocks.  The pool is an ordinary allocation of the context:
// cut at the first launch that needs it, halved until the card can provide it (never more than half of what is free), an error code
// if not even one workgroup's slots fit, freed by gfh_destroy.  *grid: the workgroups to launch for `blocks` units of work.
static int wsg_grid(gfh_ctx* c, hipFunction_t f, int threads, int64_t blocks, int* grid) {
  *grid = (int)blocks;
  if (!c->gen.ws_global || !c->wsg_wave_doubles) return 0;
  const int wpb = threads / 64;
  const int64_t want = std::min<int64_t>(blocks, resident_grid(c, f, threads)) * wpb;
  // (a pool the card cut short stays as it is until a launch wants MORE slots than the cut was made for -- kernels of different
  // workgroup sizes then alternate on the same pool instead of each freeing and cutting it again at every pass)
  // (... and once more, whatever was asked before, when the pool at hand cannot serve even ONE workgroup of this kernel: memory may
  // have come free since the card cut it short -- the Jacobian dropped, another context destroyed: round-5 advisor)
  for (int attempt = 0; attempt < 2 && (attempt == 0 || c->wsg_waves < wpb); attempt++)
  if (c->wsg_waves < want && (want > c->wsg_tried || attempt == 1)) {
    HIPCHK(c, hipStreamSynchronize(c->stream));
    dev_free(c->wsg); c->wsg_waves = 0; c->wsg_tried = want;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); free_b = (size_t)1 << 40; }
    int64_t n = want;
    for (;;) {
      const size_t bytes = (size_t)n * (size_t)c->wsg_wave_doubles * sizeof(double);
      if (bytes <= free_b / 2) {
        if (hipMalloc(&c->wsg.p, bytes) == hipSuccess) { c->wsg.bytes = bytes; break; }
        (void)hipGetLastError(); c->wsg.p = nullptr;
      }
      if (n <= wpb) return fail(c, "the device cannot provide the quadrature workspaces of one workgroup (" + std::to_string(bytes >> 20) +
                                   " MB at ws_size " + std::to_string(c->gen.ws_size) + " / " + std::to_string(c->gen.ws_size_inner) + "): lower ws_size");
      n = std::max<int64_t>(wpb, (n / 2 + wpb - 1) / wpb * wpb);
    }
    c->wsg_waves = n;
  }
  if (c->wsg_waves < wpb) return fail(c, "the pool of quadrature workspaces holds fewer slots than one workgroup of this kernel needs: lower ws_size");
  *grid = (int)std::min<int64_t>(blocks, c->wsg_waves / wpb);
  return 0;
}
// (the generated kernels take the mesh / order arguments and the pool's address only where the model has them: codegen.cpp,
// GFH_MESH_KPARAMS, GFH_ORDER_KPARAMS, GFH_WSG_KPARAMS)
static bool takes_mesh_args(const gfh_ctx* c) { return !c->gen.finite_diff && mesh_sites(c->model) > 0; }

static int launch_model_sweep(gfh_ctx* c, int mesh_mode = 0) {
  if (!c->n_tiles) return 0;
  void* x = c->x.p; void* y = c->y.p; void* w = c->w.p; void* pars = c->pars.p; void* parg = c->cur->kernarg_pars ? (void*)c->h_pars : (void*)&pars; void* tds = c->tile_ds.p;
  void* res = c->res.p; void* J = c->J.p; long long ldj = c->ldj; int nt = c->n_tiles; void* stp = c->status.p;
  void* ax = c->aux.p; long long lda = c->n_slots; void* mesh = c->mesh.p;
  // (mesh ... cost: kernels of models with integrate() only.  The sweep that bisects measures the cost of its tiles when an order
  // of dispatch is wanted: build_orders)
  const bool ordered = c->order_on && !c->gen.finite_diff && mesh_sites(c->model) > 0;      // (kernels that take the arguments: codegen.cpp, GFH_ORDER_KPARAMS)
  void* ord = ordered && c->order_ready ? c->tile_order.p : nullptr;
  void* cst = nullptr;
  if (ordered && c->order_ready && ++c->order_age >= 64) c->order_want = true;      // (the profile moves with the parameters: measured again now and then)
  if (ordered && c->order_want) {       // (a sweep that replays meshes ranks its tiles like one that bisects: by the number of intervals)
    if (c->tile_cost.bytes < sizeof(int) * (size_t)c->n_tiles && dev_alloc(c, c->tile_cost, sizeof(int) * (size_t)c->n_tiles)) return 1;
    cst = c->tile_cost.p; c->order_measured = true;
  }
  int grid; if (wsg_grid(c, c->cur->sweep, c->gen.block, c->n_tiles, &grid)) return 1;
  void* pool = c->wsg.p;
  std::vector<void*> args{&x, &y, &w, parg, &tds, &nt, &res, &J, &ldj, &stp, &ax, &lda};
  if (takes_mesh_args(c)) { args.push_back(&mesh); args.push_back(&mesh_mode); args.push_back(&ord); args.push_back(&cst); }
  if (c->gen.ws_global) args.push_back(&pool);
  HIPCHK(c, hipModuleLaunchKernel(c->cur->sweep, grid, 1, 1, c->gen.block, 1, 1, 0, c->stream, args.data(), nullptr));
  return 0;
}

// Tiles and gram blocks in the order of their measured cost, expensive first (codegen.cpp, GFH_ORD): called once the sweep that
// measured has completed.  16 KB down, two sorts of a few thousand keys, 24 KB up: a few tenths of a millisecond, once per data set /
// model and again after every 64 sweeps (the profile moves with the parameters).
static int build_orders(gfh_ctx* c) {
  gfh::Range range("gadfit order of dispatch");
  c->order_measured = false; c->order_want = false; c->order_age = 0;
  const size_t nt = (size_t)c->n_tiles, ngb = (size_t)c->n_gb;
  if (!nt || !ngb) return 0;
  std::vector<int> cost(nt);
  HIPCHK(c, hipMemcpyAsync(cost.data(), c->tile_cost.p, sizeof(int) * nt, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  std::vector<int> to(nt), go(ngb);
  for (size_t t = 0; t < nt; t++) to[t] = (int)t;
  std::stable_sort(to.begin(), to.end(), [&](int a, int b) { return cost[(size_t)a] > cost[(size_t)b]; });
  std::vector<long long> gc(ngb, 0);
  const int64_t tile = c->gen.block;
  for (size_t b = 0; b < ngb; b++) {
    const int64_t t0 = c->h_gb_start[b] / tile, t1 = (c->h_gb_start[b] + c->h_gb_slots[b] + tile - 1) / tile;
    for (int64_t t = t0; t < t1 && t < (int64_t)nt; t++) gc[b] += cost[(size_t)t];
    go[b] = (int)b;
  }
  std::stable_sort(go.begin(), go.end(), [&](int a, int b) { return gc[(size_t)a] > gc[(size_t)b]; });
  if (dev_alloc(c, c->tile_order, sizeof(int) * nt) || dev_alloc(c, c->gb_order, sizeof(int) * ngb)) return 1;
  HIPCHK(c, hipMemcpyAsync(c->tile_order.p, to.data(), sizeof(int) * nt, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->gb_order.p, go.data(), sizeof(int) * ngb, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));          // (the vectors go out of scope)
  c->order_ready = true;
  return 0;
}

// tail_mode 0: workgroup partials only; 1: + in-kernel reduction and assembly into c->packed;
// 2: + the result mailbox (sequence number seq).  Modes 1/2 need update_tail().
static int launch_model_sweep_gram(gfh_ctx* c, int tail_mode = 0, unsigned long long seq = 0, unsigned lds_pad = 0) {
  if (!c->n_gb) return 0;
  void* x = c->x.p; void* y = c->y.p; void* w = c->w.p; void* pars = c->pars.p; void* parg = c->cur->kernarg_pars ? (void*)c->h_pars : (void*)&pars;
  void* gs = c->gb_start.p; void* gn = c->gb_slots.p; void* gd = c->gb_ds.p;
  void* res = c->res.p; void* J = c->J.p; long long ldj = c->ldj; void* part = c->partial.p;
  int ps = gram_partial_stride(c->cur_T); void* stp = c->status.p; void* tl = c->tail_dev.p;
  void* ax = c->aux.p; long long lda = c->n_slots;
  void* args[] = {&x, &y, &w, parg, &gs, &gn, &gd, &res, &J, &ldj, &part, &ps, &stp, &ax, &lda, &tl, &seq, &tail_mode};
  const int fw = fused_waves_for((int)c->cur_active.size(), c->gen);
  HIPCHK(c, hipModuleLaunchKernel(c->cur->sweep_gram, c->n_gb, 1, 1, 64 * fw, 1, 1, lds_pad, c->stream, args, nullptr));
  return 0;
}

// The tail's fence-free hand-off is the form measured with ONE workgroup per CU (MI355X_MICROARCH.md, inter-workgroup
// visibility, table).  Up to 16 active parameters two workgroups of the fused kernel fit a CU's LDS (and the kernel wants
// them: padding it down to one costs 15 % at cfg 2); those models keep the three-launch chain.
static long fused_lds_bytes(const gfh_ctx* c) {
  const int na = (int)c->cur_active.size(), fw = fused_waves_for(na, c->gen);
  if (na <= kValuGramMax) return (fw + 1) * (na * (na + 1) / 2 + na + 1) * 8 + 273 * 8 + 64;      // the VALU path: the cross-wave reduction and the image
  return fused_lds_bytes_for(na, fw, c->gen);
}
static bool tail_one_workgroup_per_cu(const gfh_ctx* c) { return fused_lds_bytes(c) > 80 * 1024; }
// Grids of at most 256 workgroups (one per CU at most) may take the tail with <= 16 parameters too: a dynamic LDS pad makes
// a second workgroup on a CU impossible, and with so few workgroups the occupancy it costs is not there to lose.
static unsigned tail_lds_pad(const gfh_ctx* c) {
  return (!tail_one_workgroup_per_cu(c) && c->n_gb > 1 && c->n_gb <= 256) ? (unsigned)(81 * 1024 - fused_lds_bytes(c)) : 0u;
}

// Device-side descriptor of the fused kernel's tail (layout = struct gfh_tail of the generated source).
struct TailDesc {
  const int* ds_first_gb; const int* inv; double* slice; double* G; double* packed; double* host_out;
  unsigned long long* host_flag; unsigned* counters; int nd, dim, n_slices, pad;
};

static int update_tail(gfh_ctx* c) {
  const int ps = gram_partial_stride(c->cur_T);
  if (dev_alloc(c, c->slice, sizeof(double) * (size_t)c->nd * 32 * ps)) return 1;
  const size_t cb = sizeof(unsigned) * (size_t)(1 + c->nd * 32);
  if (c->counters.bytes < cb) {
    if (dev_alloc(c, c->counters, cb)) return 1;
    HIPCHK(c, hipMemsetAsync(c->counters.p, 0, c->counters.bytes, c->stream));
  }
  if (dev_alloc(c, c->tail_dev, sizeof(TailDesc))) return 1;
  TailDesc t;
  memset(&t, 0, sizeof t);
  t.ds_first_gb = c->ds_first_gb.as<int>(); t.inv = c->inv.as<int>(); t.slice = c->slice.as<double>(); t.G = c->G.as<double>();
  t.packed = c->packed.as<double>(); t.host_out = c->h_pinned; t.host_flag = c->h_flag; t.counters = c->counters.as<unsigned>();
  t.nd = c->nd; t.dim = c->cur_dim; t.n_slices = 0;
  for (int d = 0; d < c->nd; d++) t.n_slices += std::min(32, c->h_ds_first_gb[d + 1] - c->h_ds_first_gb[d]);
  if (c->tail_host.size() == sizeof t && !memcmp(c->tail_host.data(), &t, sizeof t)) return 0;
  c->tail_host.assign(reinterpret_cast<const char*>(&t), reinterpret_cast<const char*>(&t) + sizeof t);
  HIPCHK(c, hipMemcpyAsync(c->tail_dev.p, c->tail_host.data(), sizeof t, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}

// Workgroups of `threads` threads that are resident on the chip at once for this kernel: the occupancy the runtime reports,
// capped at 6 per CU for 256 threads -- with more than 96 SGPRs (a by-value parameter block) the hardware admits
// fewer than the API says (MI355X_MICROARCH.md, residency).  Kernels whose workgroups each own a fixed share of the
// points are launched with at most this many, so no workgroup waits for a second round behind the first.
static int resident_grid(gfh_ctx* c, hipFunction_t f, int threads) {
  int per_cu = 0, cus = 0;
  if (hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, f, threads, 0) != hipSuccess || per_cu < 1) { (void)hipGetLastError(); per_cu = 1; }
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device) != hipSuccess || cus < 1) { (void)hipGetLastError(); cus = 256; }
  const int cap = std::max(1, 6 * 256 / threads);
  return cus * std::min(per_cu, cap);
}

// tail_mode 0: workgroup sums only; 1: total in c->vec[0]; 2: and in the host mailbox under sequence number seq
static int launch_model_chi2(gfh_ctx* c, int tail_mode, unsigned long long seq, int mesh_mode = 0) {
  if (!c->n_gb) return 0;
  void* x = c->x.p; void* y = c->y.p; void* w = c->w.p; void* pars = c->pars.p; void* parg = c->cur->kernarg_pars ? (void*)c->h_pars : (void*)&pars;
  void* gs = c->gb_start.p; void* gn = c->gb_slots.p; void* gd = c->gb_ds.p;
  void* res = c->res.p; void* part = c->chi2_partial.p; void* stp = c->status.p;
  void* ax = c->aux.p; long long lda = c->n_slots; void* dfg = c->ds_first_gb.p; int nd = c->nd;
  void* out = c->vec.p; void* hout = c->h_pinned; void* hflag = c->h_flag; void* cnt = c->status.as<char>() + 24;
  void* mesh = c->mesh.p;
  void* ord = c->order_on && c->order_ready && !c->gen.finite_diff && mesh_sites(c->model) > 0 ? c->gb_order.p : nullptr; void* cst = nullptr;
  const int cw = c->cur->n_active <= fused_max_active(c->gen) ? fused_waves_for(c->cur->n_active, c->gen) : 8;     // GFH_CW of the generated source
  int grid; if (wsg_grid(c, c->cur->chi2, 64 * cw, c->n_gb, &grid)) return 1;
  void* pool = c->wsg.p;
  std::vector<void*> args{&x, &y, &w, parg, &gs, &gn, &gd, &res, &part, &stp, &ax, &lda, &dfg, &nd, &out, &hout, &hflag, &cnt, &seq, &tail_mode};
  if (takes_mesh_args(c)) { args.push_back(&mesh); args.push_back(&mesh_mode); args.push_back(&ord); args.push_back(&cst); }
  if (c->gen.ws_global) args.push_back(&pool);
  HIPCHK(c, hipModuleLaunchKernel(c->cur->chi2, grid, 1, 1, 64 * cw, 1, 1, 0, c->stream, args.data(), nullptr));
  return 0;
}

static int launch_model_omega(gfh_ctx* c, int mesh_mode = 0) {
  if (!c->n_tiles) return 0;
  void* x = c->x.p; void* w = c->w.p; void* pars = c->pars.p; void* parg = c->cur->kernarg_pars ? (void*)c->h_pars : (void*)&pars; void* dpp = c->dpars.p; void* dp = c->cur->kernarg_pars ? (void*)c->h_dpars : (void*)&dpp; void* tds = c->tile_ds.p; void* om = c->omega.p;
  int nt = c->n_tiles; void* stp = c->status.p;
  void* ax = c->aux.p; long long lda = c->n_slots;
  void* mesh = c->mesh.p;
  void* ord = c->order_on && c->order_ready && !c->gen.finite_diff && mesh_sites(c->model) > 0 ? c->tile_order.p : nullptr; void* cst = nullptr;
  // (quadrature models: uneven cost per point -- one tile per workgroup, dealt out as workgroups retire)
  if (!c->cur->omega_grid) c->cur->omega_grid = c->model.has_integrals() ? (1 << 30) : resident_grid(c, c->cur->omega, c->gen.block);
  int grid; if (wsg_grid(c, c->cur->omega, c->gen.block, std::min(c->n_tiles, c->cur->omega_grid), &grid)) return 1;
  void* pool = c->wsg.p;
  std::vector<void*> args{&x, &w, parg, dp, &tds, &nt, &om, &stp, &ax, &lda};
  if (takes_mesh_args(c)) { args.push_back(&mesh); args.push_back(&mesh_mode); args.push_back(&ord); args.push_back(&cst); }
  if (c->gen.ws_global) args.push_back(&pool);
  HIPCHK(c, hipModuleLaunchKernel(c->cur->omega, grid, 1, 1, c->gen.block, 1, 1, 0, c->stream, args.data(), nullptr));
  return 0;
}

// publish_seq != 0 (single rank, pattern-only image through k_gather_sum): the assembling kernel writes the result mailbox itself
static int launch_gram_chain(gfh_ctx* c, bool time_it, bool with_gram = true, bool sparse = false, unsigned long long publish_seq = 0) {
  const int na = (int)c->cur_active.size(), T = c->cur_T, ps = gram_partial_stride(T);
  const int gw = ps;
  if (c->n_gb && with_gram) HIPCHK(c, launch_gram(c->stream, T, c->J.as<double>(), c->ldj, na, c->res.as<double>(), c->gb_start.as<i64>(),
                                      c->gb_slots.as<int>(), c->n_gb, c->partial.as<double>()));
  if (time_it) HIPCHK(c, hipEventRecord(c->ev[2], c->stream));
  HIPCHK(c, launch_reduce_partials(c->stream, c->partial.as<double>(), ps, gw, c->ds_first_gb.as<int>(), c->nd, c->G.as<double>()));
  if (c->gs_meta.p && c->gs_n && c->gs_sparse == sparse)
    HIPCHK(c, launch_gather_sum(c->stream, c->G.as<double>(), c->gs_meta.as<int>(), c->gs_list.as<int>(), c->gs_n, c->packed.as<double>(),
                                c->status.as<int>(), publish_seq ? c->h_pinned : nullptr, reinterpret_cast<unsigned*>(c->status.as<char>() + 16),
                                c->h_flag, publish_seq));
  else if (sparse)
    HIPCHK(c, launch_assemble_sparse(c->stream, c->G.as<double>(), gw, T, c->nd, c->cur_dim, c->inv.as<int>(), c->owner.as<int>(),
                                     c->nz_row.as<int>(), c->nz_col.as<int>(), c->nnz, c->packed.as<double>()));
  else
    HIPCHK(c, launch_assemble(c->stream, c->G.as<double>(), gw, T, c->nd, c->cur_dim, c->inv.as<int>(), c->owner.as<int>(), c->packed.as<double>()));
  return 0;
}

static int check_aux(gfh_ctx* c) {
  if (c->has_model && c->model.n_aux > c->n_aux)
    return fail(c, "the model reads " + std::to_string(c->model.n_aux) + " auxiliary per-point column(s); call gfh_set_aux after gfh_set_data");
  return 0;
}

// What the ranks all-reduce after a sweep is the image `packed`: [JTJ (dim*dim, column-major) | JTres | chi2], or for global
// fits beyond the in-kernel tail's reach the pattern-only [nnz values | JTres | chi2].  ncclAllReduce needs the same length
// and the same meaning of every element on every rank, so the layout may depend only on what all ranks share -- the column
// map, dim, the number of datasets -- never on which points (or whether any) THIS rank holds.  Host-only: also what
// gfh_debug_packed_layout reports for compile-only contexts (CPU tests of the multi-rank bookkeeping).
struct PackedLayout {
  std::vector<int> inv, owner, nz_row, nz_col;
  bool sparse = false;         // the pattern is a quarter of the dense image or less
  bool small = false;          // dim*dim*n_datasets <= 65536: dense image, assembled by the fused kernel's tail where it applies
  int nnz = 0;
  bool transfer_sparse() const { return sparse && !small; }
  size_t packed_n(int dim) const { return transfer_sparse() ? (size_t)nnz + dim + 1 : (size_t)dim * dim + dim + 1; }
};

static int compute_layout(gfh_ctx* c, int nd, int na, const int32_t* jac, int dim, bool sparse_ok, PackedLayout* L) {
  L->inv.assign((size_t)nd * dim, -1);
  for (int d = 0; d < nd; d++)
    for (int k = 0; k < na; k++) {
      const int col = jac[d * na + k];
      if (col < 0 || col >= dim) return fail(c, "Jacobian index out of range");
      L->inv[(size_t)d * dim + col] = k;
    }
  // owner[col]: the single dataset that uses column col (local parameter) or -1 (several: global parameter)
  L->owner.assign(dim, -1);
  std::vector<int> users(dim, 0);
  for (int d = 0; d < nd; d++) for (int k = 0; k < na; k++) { const int col = jac[d * na + k]; if (users[col]++ == 0) L->owner[col] = d; }
  for (int col = 0; col < dim; col++) if (users[col] != 1) L->owner[col] = -1;
  if (nd == 1) std::fill(L->owner.begin(), L->owner.end(), 0);
  // pattern of the normal equations: (row <= col) pairs of columns that share a dataset, column-major order
  L->sparse = false; L->nnz = 0; L->nz_row.clear(); L->nz_col.clear();
  L->small = (int64_t)dim * dim * nd <= 65536;
  if (sparse_ok && nd > 1) {
    // the (row <= col) pairs some dataset couples, in column-major order (sorted keys: a dim x dim map costs 16 MB and 8e6 tests
    // per call at the 4003 columns of a 1000-curve fit)
    std::vector<int64_t> keys;
    keys.reserve((size_t)nd * na * (na + 1) / 2);
    for (int d = 0; d < nd; d++)
      for (int k = 0; k < na; k++) for (int m = 0; m < na; m++) {
        const int r_ = jac[d * na + k], c_ = jac[d * na + m];
        if (r_ <= c_) keys.push_back((int64_t)c_ * dim + r_);
      }
    std::sort(keys.begin(), keys.end());
    keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
    for (const int64_t key : keys) { L->nz_row.push_back((int)(key % dim)); L->nz_col.push_back((int)(key / dim)); }
    L->nnz = (int)L->nz_row.size();
    L->sparse = 4 * ((int64_t)L->nnz + dim + 1) < (int64_t)dim * dim + dim + 1;      // worth it when the pattern is a quarter or less
  }
  return 0;
}

// the buffer of the quadrature meshes (context.h): one record per slot and outermost integrate() call site of the model
static int ensure_mesh(gfh_ctx* c) {
  const int sites = (c->has_model && c->mesh_on && c->gen.fast_div && !c->gen.finite_diff) ? mesh_sites(c->model) : 0;
  const int stride = sites * kMeshRecord;
  if (stride != c->mesh_stride) { c->mesh_stride = stride; c->mesh_valid = false; }
  if (stride) {
    const size_t need = (size_t)stride * (size_t)std::max<int64_t>(1, c->n_slots);
    if (c->mesh.bytes < need) { c->mesh_valid = false; if (dev_alloc(c, c->mesh, need)) return 1; }
  } else dev_free(c->mesh);
  return 0;
}

static int prepare_active(gfh_ctx* c, const int32_t* active, int na, const int32_t* jac, int dim) {
  if (na < 1) return fail(c, "There are no active parameters.");
  if (check_aux(c) || ensure_gb_partition(c)) return 1;
  if ((na > fused_max_active(c->gen) || (c->has_model && c->model.has_integrals())) && !c->gen.store_j)
    set_store_j(c, true);   // beyond 4 tiles, and for quadrature models, STEP 2 is a separate pass over the stored Jacobian
  // fast path of the LM loop: the same active set, column map and kernels as in the previous call
  if (c->cur && c->prepared && c->cur == c->prepared_cur && dim == c->cur_dim && (int)c->cur_active.size() == na && c->prepared_store_j == c->gen.store_j &&
      std::equal(active, active + na, c->cur_active.begin()) && c->cur_jac.size() == (size_t)c->nd * na &&
      std::equal(jac, jac + (size_t)c->nd * na, c->cur_jac.begin()))
    return 0;
  c->prepared = false;
  std::vector<int32_t> a(active, active + na);
  if (get_kernels(c, a, true)) return 1;
  if (ensure_tile_table(c)) return 1;
  std::vector<int32_t> j(jac, jac + (size_t)c->nd * na);
  const bool same = (a == c->cur_active) && (j == c->cur_jac) && dim == c->cur_dim;
  c->cur_T = (na + 15) / 16;
  if (!same) {
    PackedLayout L;
    if (compute_layout(c, c->nd, na, jac, dim, c->sparse_ok, &L)) return 1;
    const std::vector<int>& inv = L.inv;
    if (dev_alloc(c, c->inv, sizeof(int) * inv.size())) return 1;
    HIPCHK(c, hipMemcpy(c->inv.p, inv.data(), sizeof(int) * inv.size(), hipMemcpyHostToDevice));
    if (dev_alloc(c, c->owner, sizeof(int) * (size_t)dim)) return 1;
    HIPCHK(c, hipMemcpy(c->owner.p, L.owner.data(), sizeof(int) * (size_t)dim, hipMemcpyHostToDevice));
    c->sparse = L.sparse; c->nnz = L.nnz; c->h_nz_row = L.nz_row; c->h_nz_col = L.nz_col;
    if (c->sparse) {
      if (dev_alloc(c, c->nz_row, sizeof(int) * (size_t)c->nnz) || dev_alloc(c, c->nz_col, sizeof(int) * (size_t)c->nnz)) return 1;
      HIPCHK(c, hipMemcpy(c->nz_row.p, c->h_nz_row.data(), sizeof(int) * (size_t)c->nnz, hipMemcpyHostToDevice));
      HIPCHK(c, hipMemcpy(c->nz_col.p, c->h_nz_col.data(), sizeof(int) * (size_t)c->nnz, hipMemcpyHostToDevice));
    }
    // source lists for k_gather_sum: where in G (the per-dataset Gram images, [nd][gw]) the terms of every element of the packed
    // image sit, in dataset order -- what k_assemble / k_assemble_sparse find through owner/inv at run time.  Built for the layout
    // the launch chain will use: pattern-only [nnz values | JTres | chi2] or dense [JTJ column-major | JTres | chi2].
    {
      const int T = c->cur_T, gw = gram_partial_stride(T), npair = T * (T + 1) / 2;
      const bool lay_sparse = L.transfer_sparse();
      const int64_t n_img = (int64_t)L.packed_n(dim);
      dev_free(c->gs_meta); c->gs_n = 0; c->gs_sparse = lay_sparse;
      if ((int64_t)c->nd * gw < (int64_t(1) << 31) && n_img <= (int64_t(1) << 18)) {
        std::vector<int> meta((size_t)n_img), list, terms;
        auto put = [&](size_t idx) {
          if (terms.empty()) meta[idx] = (int)0x80000000;
          else if (terms.size() == 1) meta[idx] = terms[0];
          else { meta[idx] = -((int)list.size() + 1); list.push_back((int)terms.size()); list.insert(list.end(), terms.begin(), terms.end()); }
        };
        auto entry = [&](int row, int col) {
          terms.clear();
          for (int d = 0; d < c->nd; d++) {
            int a_ = inv[(size_t)d * dim + row], b_ = inv[(size_t)d * dim + col];
            if (a_ < 0 || b_ < 0) continue;
            if (a_ > b_) std::swap(a_, b_);                 // upper triangle of tile pairs is stored
            const int ti = a_ >> 4, tj = b_ >> 4, p = ti * T - ti * (ti - 1) / 2 + (tj - ti);
            terms.push_back(d * gw + p * 256 + (a_ & 15) * 16 + (b_ & 15));
          }
        };
        const size_t nn = lay_sparse ? (size_t)c->nnz : (size_t)dim * dim;
        if (lay_sparse) for (int k = 0; k < c->nnz; k++) { entry(c->h_nz_row[k], c->h_nz_col[k]); put((size_t)k); }
        else for (int col = 0; col < dim; col++) for (int row = 0; row < dim; row++) { entry(row, col); put((size_t)col * dim + row); }
        for (int row = 0; row < dim; row++) {
          terms.clear();
          for (int d = 0; d < c->nd; d++) { const int a_ = inv[(size_t)d * dim + row]; if (a_ >= 0) terms.push_back(d * gw + npair * 256 + a_); }
          put(nn + row);
        }
        terms.clear();
        for (int d = 0; d < c->nd; d++) terms.push_back(d * gw + npair * 256 + 16 * T);
        put(nn + dim);
        if (list.empty()) list.push_back(0);
        if (dev_alloc(c, c->gs_meta, sizeof(int) * meta.size()) || dev_alloc(c, c->gs_list, sizeof(int) * list.size())) return 1;
        HIPCHK(c, hipMemcpy(c->gs_meta.p, meta.data(), sizeof(int) * meta.size(), hipMemcpyHostToDevice));
        HIPCHK(c, hipMemcpy(c->gs_list.p, list.data(), sizeof(int) * list.size(), hipMemcpyHostToDevice));
        c->gs_n = (int)n_img;
      }
    }
    c->cur_active = a; c->cur_jac = j; c->cur_dim = dim; c->have_sweep = false;
  }
  if (ensure_mesh(c)) return 1;
  const int ps = gram_partial_stride(c->cur_T);
  const size_t packed_n = (size_t)dim * dim + dim + 2;       // (+ the status slot that travels with a cross-rank sum)
  if ((c->gen.store_j && place_jacobian(c, na)) ||
      dev_alloc(c, c->partial, sizeof(double) * (size_t)std::max(1, c->n_gb) * ps) ||
      dev_alloc(c, c->G, sizeof(double) * (size_t)c->nd * ps) ||
      dev_alloc(c, c->packed, sizeof(double) * packed_n) ||
      dev_alloc(c, c->chi2_partial, sizeof(double) * (size_t)std::max(1, c->n_gb)) ||
      dev_alloc(c, c->vec, sizeof(double) * (size_t)(dim + 8)) ||
      pinned_reserve(c, sizeof(double) * std::max<size_t>(packed_n + 1, 4096))) return 1;
  c->prepared = true; c->prepared_store_j = c->gen.store_j; c->prepared_cur = c->cur;
  return 0;
}

// Test hook (no GPU needed): the geometry and the layout of the all-reduced image as rank `rank` of `nranks` derives them.
// out[0] = length of the packed image, out[1] = pattern-only transfer (0/1), out[2] = nnz, out[3] = FNV-1a hash of the
// pattern lists and of inv/owner, out[4] = first global point of this rank, out[5] = its point count, out[6] = number
// of datasets it holds points of, out[7] = its number of gram workgroups.  Every rank must report the same out[0..3].
int gfh_debug_packed_layout(int nranks, int rank, int64_t n_total, int nd, const int64_t* dp, int na, const int32_t* jac, int dim,
                            int sparse_ok, int64_t* out, int32_t* nz_row, int32_t* nz_col, int nz_cap) {
  if (nranks < 1 || rank < 0 || rank >= nranks || !dp || !jac || !out || na < 1 || nd < 1) { set_global_error("gfh_debug_packed_layout: bad arguments"); return 1; }
  gfh_ctx c;
  c.nranks = nranks; c.rank = rank;
  if (set_geometry(&c, n_total, nd, dp)) { set_global_error(c.err); return 1; }
  PackedLayout L;
  if (compute_layout(&c, nd, na, jac, dim, sparse_ok != 0, &L)) { set_global_error(c.err); return 1; }
  uint64_t h = 1469598103934665603ull;
  auto mix = [&](const std::vector<int>& v) { for (int x : v) { h ^= (uint32_t)x; h *= 1099511628211ull; } h ^= 0xffu; h *= 1099511628211ull; };
  mix(L.nz_row); mix(L.nz_col); mix(L.inv); mix(L.owner);
  int held = 0;
  for (int d = 0; d < nd; d++) if (c.lb[d + 1] > c.lb[d]) held++;
  out[0] = (int64_t)L.packed_n(dim); out[1] = L.transfer_sparse() ? 1 : 0; out[2] = L.nnz; out[3] = (int64_t)(h >> 1);
  out[4] = c.begin; out[5] = c.count; out[6] = held; out[7] = c.n_gb;
  for (int k = 0; k < L.nnz && k < nz_cap; k++) { if (nz_row) nz_row[k] = L.nz_row[k]; if (nz_col) nz_col[k] = L.nz_col[k]; }
  return 0;
}

int gfh_set_active(gfh_ctx* c, const int32_t* active, int na, const int32_t* jac, int dim) {
  GROUP(c, gfh_set_active(k, active, na, jac, dim));
  NEED_GPU(c);
  if (!c->nd) return fail(c, "no data set (gfh_set_data)");
  return prepare_active(c, active, na, jac, dim);
}

// kernels raise the status word (1: quadrature workspace exhausted, 2: an integrand met a path through its comparisons
// that no recording of it has, 3: a data point took a branch of eval() no recorded variant covers).  Queue its
// read-back; check after the stream synchronise.
constexpr int kUnseen = 77;      // internal return code: a point left the recorded decision tree (status 3); the caller recovers and repeats the pass
constexpr int kGrowWs = 78;      // internal return code: the compiled-in quadrature workspace was exhausted but the user's is larger
constexpr int kIntegrandPath = 79;   // internal return code: an integrand met a path through its comparisons that no recording has (status 2)
static bool workspace_can_grow(const gfh_ctx* c) {
  return c->has_model && c->model.has_integrals() && (c->gen.ws_size < c->model.ws_size || c->gen.ws_size_inner < c->model.ws_size_inner);
}
static int status_check(gfh_ctx* c, int st) {
  if (!st) { c->n_integrand_rounds = 0; return 0; }
  if (st == 3 && c->has_model && c->model.branching()) return kUnseen;       // (the status word and the report are read and cleared by recover_unseen)
  if (st == 1 && workspace_can_grow(c)) return kGrowWs;
  if (st == 2 && c->unseen_fn && c->n_integrand_rounds < 3) return kIntegrandPath;
  hipMemsetAsync(c->status.p, 0, sizeof(int), c->stream);
  hipStreamSynchronize(c->stream);
  if (st == 1) return fail(c, "Number of iterations was insufficient. Increase either workspace size or the error bound(s).");
  if (st == 2) return fail(c, "an integrand took a path through its comparisons of AD variables that no recording of it has (the recordings place the "
                               "integration variable at a few points of its range: record eval() at more abscissas or parameter values)");
  return fail(c, "device kernel reported status " + std::to_string(st));
}

// End of every result-returning call: k_publish (kernels.hip) moves n doubles at `src` and the
// kernels' status word into the pinned mailbox c->h_pinned and stores this call's sequence number
// into the host flag; the host spins on the flag.  Everything queued on the stream before it has
// finished when the flag flips (it is the last operation of the call).  hipStreamQuery is polled
// now and then so that a failed launch or a device fault ends the wait with an error.
static int await_result(gfh_ctx* c, unsigned long long seq, size_t n, bool summed = false) {
  for (unsigned spin = 1;; spin++) {
    if (__atomic_load_n(c->h_flag, __ATOMIC_ACQUIRE) == seq) break;
    __builtin_ia32_pause();
    if ((spin & 0x3FF) == 0) {
      const hipError_t e = hipStreamQuery(c->stream);
      if (e == hipSuccess) {
        if (__atomic_load_n(c->h_flag, __ATOMIC_ACQUIRE) == seq) break;
        return fail(c, "result mailbox was not written");
      }
      if (e != hipErrorNotReady) return fail(c, std::string("HIP error while waiting for a result: ") + hipGetErrorString(e));
    }
  }
  // summed: the n doubles are a cross-rank sum whose element n is the sum of the ranks' encoded status words (allreduce_sum),
  // so a quadrature failure on one rank raises the reference's error on every rank (and none waits in a later collective)
  int st = (int)c->h_pinned[n + (summed ? 1 : 0)];
  if (summed && !st) { const double g = c->h_pinned[n]; st = g >= 16777216.0 ? 3 : g >= 4096.0 ? 2 : g >= 1.0 ? 1 : 0; }
  // member of a single-process device group: the sum over the members (co_sum, misc.F90:133-170) is taken here,
  // on the host, in rank order; the status word travels with it so every member raises the same error
  if (c->member_of && !c->comm && gfh::group_allreduce(c, c->h_pinned, n, &st)) return 1;
  return status_check(c, st);
}

static int fetch_result(gfh_ctx* c, const double* src, size_t n, bool summed = false) {
  if (pinned_reserve(c, sizeof(double) * std::max<size_t>(n + 2, 4096))) return 1;
  const unsigned long long seq = ++c->mail_seq;
  unsigned* counter = reinterpret_cast<unsigned*>(c->status.as<char>() + 16);
  HIPCHK(c, launch_publish(c->stream, src, (int)(n + (summed ? 1 : 0)), c->status.as<int>(), c->h_pinned, counter, c->h_flag, seq));
  return await_result(c, seq, n, summed);
}

// co_sum (misc.F90:133-170) of n doubles at buf over the ranks: ONE ncclAllReduce per call site of the reference; the kernels'
// status word rides along as element n (buf has room for it), encoded so that the sum still tells the codes apart
static int allreduce_sum(gfh_ctx* c, double* buf, size_t n, bool slot_written = false) {
  // (slot_written: the kernel that produced buf -- the fused kernel's or gfh_k_chi2's tail in mode 1 -- has put the slot there itself)
  if (!slot_written) HIPCHK(c, launch_status_slot(c->stream, c->status.as<int>(), buf + n));
  NCCLCHK(c, ncclAllReduce(buf, buf, n + 1, ncclDouble, ncclSum, c->comm, c->stream));
  c->n_allreduce++;
  return 0;
}

// (a result can reach the host mailbox a moment before its kernel has formally retired: wait for the closing event)
static double ev_ms(hipEvent_t a, hipEvent_t b) { float ms = 0; hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b); return ms; }

// gfh_debug_allreduce_latency on one context (a rank with a communicator, or a member of a device group on its own thread)
static int allreduce_latency_one(gfh_ctx* c, int n, int rounds, double* out6) {
  std::vector<double> dev_us, host_us;
  dev_us.reserve((size_t)rounds); host_us.reserve((size_t)rounds);
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
  int nranks = 1;
  if (c->comm) {
    NEED_GPU(c);
    NCCLCHK(c, ncclCommCount(c->comm, &nranks));
    DevBuf buf;
    if (dev_alloc(c, buf, sizeof(double) * ((size_t)n + 2))) return 1;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { if (e0) hipEventDestroy(e0); dev_free(buf); return fail(c, "hipEventCreate failed"); }
    int rc = 0;
    auto body = [&]() -> int {
      HIPCHK(c, hipMemsetAsync(buf.p, 0, sizeof(double) * ((size_t)n + 2), c->stream));
      // (a) the collective alone, between two events on an otherwise idle stream (the first rounds wake the ranks up and are dropped)
      const int warm = std::min(rounds, 20);
      for (int i = 0; i < warm + rounds; i++) {
        HIPCHK(c, hipEventRecord(e0, c->stream));
        NCCLCHK(c, ncclAllReduce(buf.p, buf.p, (size_t)n + 1, ncclDouble, ncclSum, c->comm, c->stream));
        HIPCHK(c, hipEventRecord(e1, c->stream));
        const double ms = ev_ms(e0, e1);
        if (i >= warm) dev_us.push_back(1e3 * ms);
      }
      // (b) as a pass pays it: enqueue the all-reduce, publish the sums into the host mailbox, spin on its flag
      for (int i = 0; i < warm + rounds; i++) {
        const auto t0 = now();
        NCCLCHK(c, ncclAllReduce(buf.p, buf.p, (size_t)n + 1, ncclDouble, ncclSum, c->comm, c->stream));
        if (fetch_result(c, static_cast<double*>(buf.p), (size_t)n, true)) return 1;
        if (i >= warm) host_us.push_back(us(t0, now()));
      }
      return 0;
    };
    rc = body();
    (void)hipStreamSynchronize(c->stream);
    hipEventDestroy(e0); hipEventDestroy(e1);
    dev_free(buf);
    if (rc) return 1;
  } else if (c->member_of) {
    nranks = c->nranks;
    std::vector<double> v((size_t)n + 1, 0.0);
    int st = 0;
    const int warm = std::min(rounds, 200);
    for (int i = 0; i < warm + rounds; i++) {
      for (int j = 0; j < n; j++) v[(size_t)j] = 1.0 + c->rank;
      const auto t0 = now();
      if (gfh::group_allreduce(c, v.data(), (size_t)n, &st)) return 1;
      if (i >= warm) { const double t = us(t0, now()); dev_us.push_back(t); host_us.push_back(t); }
    }
    if (v[0] != 0.5 * nranks * (nranks + 1)) return fail(c, "gfh_debug_allreduce_latency: wrong sum");
  } else {
    return fail(c, "gfh_debug_allreduce_latency needs a communicator (gfh_comm_init) or a device-group handle");
  }
  if (c->rank == 0 || !c->member_of) {
    std::sort(dev_us.begin(), dev_us.end()); std::sort(host_us.begin(), host_us.end());
    auto q = [](const std::vector<double>& s, double f) { return s.empty() ? 0.0 : s[std::min(s.size() - 1, (size_t)(f * (double)s.size()))]; };
    if (out6) {
      out6[0] = q(dev_us, 0.5); out6[1] = q(dev_us, 0.95); out6[2] = dev_us.empty() ? 0.0 : dev_us.front(); out6[3] = dev_us.empty() ? 0.0 : dev_us.back();
      out6[4] = q(host_us, 0.5); out6[5] = (double)nranks;
    }
  }
  return 0;
}

// How long ONE cross-rank sum of n doubles (+ the status slot) takes on this context's path, measured by the library itself:
// through ncclAllReduce (processes with a communicator; members of a device group with RCCL) or through the group's ordered
// host sum.  Collective: every rank (or the group handle) calls it with the same n and rounds.
int gfh_debug_allreduce_latency(gfh_ctx* c, int n, int rounds, double* out6) {
  if (!c || n < 1 || rounds < 1 || !out6) return fail(c, "gfh_debug_allreduce_latency: n >= 1, rounds >= 1");
  GROUP(c, allreduce_latency_one(k, n, rounds, r ? nullptr : out6));
  return allreduce_latency_one(c, n, rounds, out6);
}

// timer level 1 brackets every 8th launch (every launch under adaptive load balancing, whose shares follow these times): the
// sum over the timed launches, scaled to all launches since gfh_reset_timers
static bool timed_launch(const gfh_ctx* c, long n_so_far) {
  return c->timer_detail >= 2 || (c->timer_detail == 1 && (!(n_so_far & 7) || (c->load_balancing && c->nranks > 1)));
}
static double scaled_time(double t_timed, long n_all, long n_timed) { return n_timed > 0 ? t_timed * (double)n_all / (double)n_timed : 0.0; }

// sweep timers from the events of the last gfh_sweep (deferred while the kernel may still be finishing)
static void harvest_events(gfh_ctx* c) {
  const int td = c->ev_pending;
  c->ev_pending = 0;
  if (td < 1) return;
  hipEventSynchronize(c->ev[td >= 2 ? 4 : 1]);
  const double ts = 1e-3 * ev_ms(c->ev[0], c->ev[1]);
  c->t_sweep += ts; c->t_sweep_last = ts;
  if (!c->n_sweep_timed || ts < c->t_sweep_min) c->t_sweep_min = ts;
  if (!c->n_sweep_timed || ts > c->t_sweep_max) c->t_sweep_max = ts;
  c->n_sweep_timed++;
  if (td >= 2) {
    c->t_gram += 1e-3 * ev_ms(c->ev[1], c->ev[2]);
    c->t_reduce += 1e-3 * ev_ms(c->ev[2], c->ev[3]); c->t_allreduce += 1e-3 * ev_ms(c->ev[3], c->ev[4]);
    c->n_chain_timed++;
  }
}

// see place_jacobian.  The parameters of the call are uploaded already: the candidates are timed on the kernel and the numbers
// that are about to run (no tail, nothing read back).  One-time cost per (re)allocation: ~3 ms per candidate at the headline size.
static int place_jacobian_now(gfh_ctx* c, bool fused) {
  c->placement_pending = false;
  const size_t bytes = c->J.bytes;
  size_t free_b = 0, total_b = 0;
  const int tries = std::min(c->placement_tries, 16);
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess) return 0;
  if (hipEventCreate(&e1) != hipSuccess) { hipEventDestroy(e0); return 0; }
  int rc = 0;
  auto probe = [&](void* p, int launches) -> double {
    c->J.p = p;
    hipEventRecord(e0, c->stream);
    for (int k = 0; k < launches && !rc; k++) rc = fused ? launch_model_sweep_gram(c) : launch_model_sweep(c);
    hipEventRecord(e1, c->stream);
    if (rc || hipEventSynchronize(e1) != hipSuccess) return 1e30;
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    return (double)ms / launches;
  };
  void* const first = c->J.p;
  probe(first, 8);                                     // common warm-up (the first launches after an idle gap run slow)
  // Stop at the first candidate on the fast side.  Where that side lies is measured, not assumed: a device-to-device copy inside
  // the first candidate (read + write bytes over its duration) gives this card's copy rate; in fast pages the fused kernel moves
  // its algorithmic bytes at 1.22-1.24 x that rate and the plain sweep at 1.33-1.39 x, in slow pages at 1.09-1.14 x and
  // 1.19-1.25 x (round 2's kernel rates, profiles/r02_placement_probe.txt, over that round's copy rate of 5.05 TB/s): the
  // thresholds sit between.  (Without a usable measurement: round 2's absolute rates.)
  const double algo = (double)(32 + 8 * c->cur_active.size()) * (double)c->n_slots;
  double copy_rate = 0.0;
  {
    const size_t half = (bytes / 2) & ~(size_t)255;
    for (int rep = 0; rep < 3 && half; rep++) {
      hipEventRecord(e0, c->stream);
      if (hipMemcpyAsync(static_cast<char*>(first) + half, first, half, hipMemcpyDeviceToDevice, c->stream) != hipSuccess) { (void)hipGetLastError(); break; }
      hipEventRecord(e1, c->stream);
      if (hipEventSynchronize(e1) != hipSuccess) break;
      float ms = 0; hipEventElapsedTime(&ms, e0, e1);
      if (ms > 0) copy_rate = std::max(copy_rate, 2.0 * (double)half / (1e-3 * ms));
    }
  }
  c->placement_copy_rate = copy_rate;
  // (round 6: the <= 8-parameter form of the fused kernel -- no LDS stage, no matrix phase, and since this round a short epilogue --
  // moves its bytes at 1.27-1.30 x the copy rate in fast pages and 1.12-1.15 x in slow ones: 0.149 against 0.167-0.174 ms at BASELINE
  // config 2, profiles/r06_valu_form_ab.txt; with the matrix form's 1.19 a candidate at 0.160 ms counted as fast and ended the search)
  const bool valu_form = fused && (int)c->cur_active.size() <= kValuGramMax;
  // (... and never below an absolute rate: the copy is made INSIDE the first candidate, so slow pages under it lower the bar for
  // themselves -- a bench line of this round kept 0.485 ms after two candidates because its copy ran at 4.7 TB/s, in a process whose
  // other kernels all ran fast; 6.3 TB/s is what fast pages give the fused kernel on every box met: 0.426-0.448 ms at the headline size)
  const double floor_rate = valu_form ? 6.3e12 : fused ? 6.3e12 : 6.6e12;
  const double good_rate = std::max(floor_rate, copy_rate > 1e12 ? (valu_form ? 1.25 : fused ? 1.19 : 1.30) * copy_rate : 0.0);
  const double good_ms = algo / good_rate * 1e3;
  std::vector<void*> cand{first};
  std::vector<double> t{probe(first, 4)};
  for (int k = 1; k < tries && !rc && t.back() > good_ms; k++) {
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < total_b / 2 || free_b < 2 * bytes + ((size_t)1 << 30)) break;   // (never crowd the card)
    void* p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) { (void)hipGetLastError(); break; }
    cand.push_back(p); t.push_back(probe(p, 4));
    // (a kernel the pages do not matter to -- bound by its arithmetic, never near the rate above -- shows it after four candidates:
    // all within 1.5 % of each other.  The search ends there instead of trying every allocation for nothing.)
    if (t.size() == 4) {
      const double lo = *std::min_element(t.begin(), t.end()), hi = *std::max_element(t.begin(), t.end());
      if (hi - lo < 0.015 * lo) break;
    }
  }
  // the part's clocks are still ramping while the first candidates are timed (launches 3-40 after an idle gap): those are
  // timed again now that it has settled
  for (size_t k = 0; k < cand.size() && k < 8 && 8 + 4 * k < 40 && cand.size() > 1 && !rc; k++) t[k] = std::min(t[k], probe(cand[k], 4));
  size_t best = 0;
  for (size_t k = 1; k < t.size(); k++) if (t[k] < t[best]) best = k;
  for (size_t k = 0; k < cand.size(); k++) if (k != best) hipFree(cand[k]);
  c->J.p = cand[best];
  c->placement_n = (int)t.size();
  c->placement_ms[0] = t[best];
  for (size_t k = 0, o = 1; k < t.size() && o < 7; k++) if (k != best) c->placement_ms[o++] = t[k];
  // Round 6: the kernel's OTHER streams -- x, y, w read, res written: 32 of the 32 + 8 p bytes per point, a third of the traffic at 8
  // parameters -- sit in allocations of their own, and the pages behind THEM decide as much: BASELINE config 2 ran at 0.150-0.152 ms
  // or at 0.169-0.173 ms from process to process with every candidate of the Jacobian buffer alike within the process
  // (profiles/r06_data_placement.txt).  So while the kernel is still on the slow side the four arrays are re-placed together: a
  // new set allocated, the contents copied device to device, the kernel timed, the faster set kept.  GADFIT_HIP_PLACE_DATA=0: not.
  static const bool place_data = [] { const char* e = getenv("GADFIT_HIP_PLACE_DATA"); return !e || atoi(e) != 0; }();
  c->placement_data_n = 0;
  if (place_data && !rc && c->n_slots > 0 && c->x.p && c->y.p && c->w.p && c->res.p) {
    const size_t nb = sizeof(double) * (size_t)c->n_slots;
    double best_t = c->placement_ms[0];
    int stale = 0;
    for (int k = 0; k < tries && !rc && best_t > good_ms; k++) {
      if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < total_b / 2 || free_b < 8 * nb + ((size_t)1 << 30)) break;
      void* nw[4] = {nullptr, nullptr, nullptr, nullptr};
      DevBuf* cur[4] = {&c->x, &c->y, &c->w, &c->res};
      bool ok = true;
      for (int a = 0; a < 4 && ok; a++) ok = hipMalloc(&nw[a], std::max(nb, cur[a]->bytes)) == hipSuccess;
      for (int a = 0; a < 4 && ok; a++) ok = hipMemcpyAsync(nw[a], cur[a]->p, cur[a]->bytes, hipMemcpyDeviceToDevice, c->stream) == hipSuccess;
      if (!ok) { (void)hipGetLastError(); hipStreamSynchronize(c->stream); for (int a = 0; a < 4; a++) if (nw[a]) hipFree(nw[a]); break; }
      void* old[4];
      for (int a = 0; a < 4; a++) { old[a] = cur[a]->p; cur[a]->p = nw[a]; }
      const double tk = std::min(probe(c->J.p, 4), probe(c->J.p, 4));
      c->placement_data_n++;
      if (!rc && tk < best_t) { if (tk < 0.99 * best_t) stale = 0; best_t = tk; for (int a = 0; a < 4; a++) hipFree(old[a]); }
      else { hipStreamSynchronize(c->stream); for (int a = 0; a < 4; a++) { cur[a]->p = old[a]; hipFree(nw[a]); } }
      if (++stale >= 4) break;                         // (four sets in a row that gained nothing: these arrays are not what holds the kernel)
    }
    c->placement_data_ms = best_t;
    c->placement_ms[0] = best_t;
  }
  hipEventDestroy(e0); hipEventDestroy(e1);
  return rc;
}

// A point has left the recorded decision tree of a branching eval() (status 3; codegen.cpp, gfh_select): read the report, hand
// the points to the handler -- which records eval() there and extends the model -- and let the caller repeat the pass.  In a
// multi-rank run every rank comes here (the status word is part of the cross-rank sum); a rank whose own points were all covered
// has an empty report and simply repeats its pass, so the collectives stay in step.
static int recover_unseen(gfh_ctx* c, const double* pars) {
  gfh::Range range("gadfit unseen branch: record and extend the model");
  HIPCHK(c, hipStreamSynchronize(c->stream));
  std::vector<unsigned char> raw(kStatusBytes);
  HIPCHK(c, hipMemcpy(raw.data(), c->status.p, kStatusBytes, hipMemcpyDeviceToHost));
  HIPCHK(c, hipMemset(c->status.p, 0, sizeof(int)));
  HIPCHK(c, hipMemset(c->status.as<char>() + 64, 0, sizeof(unsigned)));
  if (++c->n_unseen_rounds > 4096) return fail(c, "a branching eval() keeps producing paths that were not recorded (4096 passes repeated)");
  unsigned cnt = 0; memcpy(&cnt, raw.data() + 64, sizeof cnt);
  const int n = (int)std::min<unsigned>(cnt, (unsigned)kUnseenCap);
  if (!n) return 0;
  const UnseenEntry* e = reinterpret_cast<const UnseenEntry*>(raw.data() + 128);
  std::vector<int64_t> index((size_t)n); std::vector<int32_t> ds((size_t)n), ng((size_t)n);
  std::vector<double> xs((size_t)n); std::vector<uint64_t> path((size_t)n);
  for (int k = 0; k < n; k++) {
    int64_t slot = e[k].slot;
    if (slot < 0 || slot >= c->n_slots) return fail(c, "corrupt report of an unseen branch");
    int d = 0;
    while (d + 1 < c->nd && slot >= c->ds_slot[(size_t)d + 1]) d++;
    const int64_t len = c->lb[(size_t)d + 1] - c->lb[(size_t)d];
    int64_t off = slot - c->ds_slot[(size_t)d];
    if (off >= len) off = len - 1;                          // a pad slot repeats its dataset's last point
    if (off < 0) off = 0;
    index[(size_t)k] = c->begin + c->lb[(size_t)d] + off; ds[(size_t)k] = d; ng[(size_t)k] = e[k].n_guards; path[(size_t)k] = e[k].path;
    HIPCHK(c, hipMemcpy(&xs[(size_t)k], c->x.as<double>() + slot, sizeof(double), hipMemcpyDeviceToHost));
  }
  char where[160];
  snprintf(where, sizeof where, " (first such point: x = %.17g, dataset %d, %u point(s) in this pass)", xs[0], ds[0] + 1, cnt);
  if (!c->unseen_fn)
    return fail(c, std::string("eval() takes a branch at a data point that none of the recorded variants covers, and no handler is "
                               "registered to record it (gfh_set_unseen_handler)") + where);
  const long ms = c->model_serial, as = c->aux_serial;
  int rc;
  { std::lock_guard<std::recursive_mutex> lk(g_handler_mutex);
    c->in_recovery = true;
    rc = c->unseen_fn(c->unseen_user, c, n, index.data(), ds.data(), xs.data(), path.data(), ng.data(), pars);
    c->in_recovery = false; }
  if (rc) return fail(c, std::string("the handler for unrecorded branches of eval() failed") + where + (c->err.empty() ? "" : ": " + c->err));
  if (ms == c->model_serial && as == c->aux_serial)
    return fail(c, std::string("eval() takes a branch that the recorder cannot reproduce on the host") + where);
  return 0;
}

// An adaptive integral ran out of the compiled-in workspace (status 1) while the user's workspace (the reference's default:
// 1000 intervals, NI:40) is larger: from now on this context's kernels carry the user's sizes; the caller repeats the pass.
// Only a pass that exhausts THOSE raises "Number of iterations was insufficient" (NI:282-283).
static int grow_workspace(gfh_ctx* c) {
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMemset(c->status.p, 0, sizeof(int)));
  c->ws_grown = true;
  apply_ws_plan(c);
  c->cur = nullptr; c->prepared = false; c->mesh_valid = false;
  return 0;
}
// An integrand met a path through its comparisons of AD variables that no recording of it has (status 2): the parameters have
// moved since the integrands were recorded (a kink has entered or left some point's range of integration).  The handler is
// called with NO points (n = 0): it records eval() over its sample of the data again, at the parameters of this pass, with the
// integration variable at its several places, and hands the extended model over; the pass is repeated.  Three such rounds in a
// row without a clean pass in between, or a handler that adds nothing, end in the error.
static int recover_integrand_path(gfh_ctx* c, const double* pars) {
  gfh::Range range("gadfit integrand path: record again and extend the model");
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMemset(c->status.p, 0, sizeof(int)));
  c->n_integrand_rounds++;
  const long ms = c->model_serial;
  int rc;
  { std::lock_guard<std::recursive_mutex> lk(g_handler_mutex);
    c->in_recovery = true;
    rc = c->unseen_fn(c->unseen_user, c, 0, nullptr, nullptr, nullptr, nullptr, nullptr, pars);
    c->in_recovery = false; }
  if (rc || ms == c->model_serial) { c->n_integrand_rounds = 3; return status_check(c, 2); }
  return 0;
}

static int repeat_pass(gfh_ctx* c, int rc, const double* pars) {      // 0: repeat the pass; 1: failed
  if (rc == kUnseen) return recover_unseen(c, pars);
  if (rc == kGrowWs) return grow_workspace(c);
  if (rc == kIntegrandPath) return recover_integrand_path(c, pars);
  return 1;
}

static int sweep_pass(gfh_ctx* c, const double* pars, const int32_t* active, int na, const int32_t* jac, int dim,
                      double* JTJ, double* JTres, double* chi2);

int gfh_sweep(gfh_ctx* c, const double* pars, const int32_t* active, int na, const int32_t* jac, int dim,
              double* JTJ, double* JTres, double* chi2) {
  // device group: every member holds the same sums afterwards; member 0 writes the caller's arrays
  GROUP(c, gfh_sweep(k, pars, active, na, jac, dim, r ? nullptr : JTJ, r ? nullptr : JTres, r ? nullptr : chi2));
  NEED_GPU(c);
  for (;;) {
    const int rc = sweep_pass(c, pars, active, na, jac, dim, JTJ, JTres, chi2);
    if (rc != kUnseen && rc != kGrowWs && rc != kIntegrandPath) return rc;
    if (repeat_pass(c, rc, pars)) return 1;
  }
}

static int sweep_pass(gfh_ctx* c, const double* pars, const int32_t* active, int na, const int32_t* jac, int dim,
                      double* JTJ, double* JTres, double* chi2) {
  gfh::Range range("gadfit sweep (STEP 1 + STEP 2)");
  harvest_events(c);
  if (!c->nd) return fail(c, "no data set (gfh_set_data)");
  if (prepare_active(c, active, na, jac, dim)) return 1;
  if (c->gen.finite_diff && c->gen.fd_col_sets && c->has_model && c->n_aux < c->model.n_aux * (1 + na))
    return fail(c, "use_ad = 0 with column sets (gfh_set_fd_column_sets): the model reads " + std::to_string(c->model.n_aux) + " column(s), " +
                std::to_string(na) + " parameter(s) are active, so gfh_set_aux must hold " + std::to_string(c->model.n_aux * (1 + na)) +
                " columns; it holds " + std::to_string(c->n_aux));
  if (c->gen.finite_diff)                              // grad_finite's own check (fitfunction.F90:164-167)
    for (int d = 0; d < c->nd; d++)
      for (int j = 0; j < na; j++) {
        const double step = 0x1p-26 * pars[(size_t)d * c->model.n_pars + active[j]];
        if (!(std::fabs(step) > 2.2250738585072014e-308))
          return fail(c, "Absolute value of parameter " + std::to_string(active[j] + 1) + " is too small.");
      }
  if (upload_pars(c, pars)) return 1;
  // an event record costs ~5 us of stream time: only the model kernel is bracketed by default
  const bool fused = use_fused(c);
  // Small assemblies: the fused kernel's own tail reduces the workgroup partials, assembles the packed
  // normal equations and (single rank) writes the host mailbox -- no reduce/assemble/publish launches.
  const bool small = (int64_t)dim * dim * c->nd <= 65536;
  // (a single workgroup hands nothing over to anybody: the tail is always safe then -- the tiny fits)
  const bool tail = c->tail && fused && c->n_gb > 0 && small && (tail_one_workgroup_per_cu(c) || c->n_gb <= 256);
  // global fits beyond the tail's reach travel pattern-only: [nnz | JTres | chi2].  The layout of `packed` is what the
  // ranks all-reduce, so it may only depend on quantities every rank shares (not on whether THIS rank has points).
  const bool sparse = c->sparse && !small;
  const size_t packed_n = sparse ? (size_t)c->nnz + dim + 1 : (size_t)dim * dim + dim + 1;
  // (level 1 samples: every 8th launch since gfh_reset_timers is bracketed)
  const int tl_ = timed_launch(c, c->n_sweep) ? c->timer_detail : 0;
  const int td = fused ? tl_ : (tl_ ? 2 : 0);
  unsigned long long seq = 0;
  if (tail) {
    if (pinned_reserve(c, sizeof(double) * std::max<size_t>(packed_n + 1, 4096)) || update_tail(c)) return 1;
    if (!c->comm) seq = ++c->mail_seq;
  }
  // (a sweep writes every column of J anew: moving the buffer between two sweeps loses nothing)
  if (c->placement_pending && c->gen.store_j && c->J.p && c->sweeps_on_J >= c->placement_after && place_jacobian_now(c, fused)) return 1;
  if (c->gen.store_j && c->J.p) c->sweeps_on_J++;
  if (td >= 1) HIPCHK(c, hipEventRecord(c->ev[0], c->stream));
  if (fused ? launch_model_sweep_gram(c, tail ? (c->comm ? 1 : 2) : 0, seq, tail ? tail_lds_pad(c) : 0u) : launch_model_sweep(c, mesh_mode_for(c, pars, true))) return 1;
  if (td >= 1) HIPCHK(c, hipEventRecord(c->ev[1], c->stream));
  if (tail) {
    // reduction, assembly and (single rank) the mailbox write happened in the fused kernel's tail
    if (td >= 2) { HIPCHK(c, hipEventRecord(c->ev[2], c->stream)); HIPCHK(c, hipEventRecord(c->ev[3], c->stream)); }
    if (c->comm) {
      if (allreduce_sum(c, c->packed.as<double>(), packed_n, true)) return 1;
      if (td >= 2) HIPCHK(c, hipEventRecord(c->ev[4], c->stream));
      PASS(fetch_result(c, c->packed.as<double>(), packed_n, true));
    } else {
      if (td >= 2) HIPCHK(c, hipEventRecord(c->ev[4], c->stream));
      PASS(await_result(c, seq, packed_n));
    }
  } else {
    // single rank + pattern-only image: k_gather_sum posts the mailbox itself (no k_publish launch)
    const bool self_publish = c->gs_meta.p && c->gs_n && c->gs_sparse == sparse && !c->comm;
    unsigned long long pseq = 0;
    if (self_publish) {
      if (pinned_reserve(c, sizeof(double) * std::max<size_t>(packed_n + 1, 4096))) return 1;
      pseq = ++c->mail_seq;
    }
    if (launch_gram_chain(c, td >= 2, !fused, sparse, pseq)) return 1;
    if (td >= 2) HIPCHK(c, hipEventRecord(c->ev[3], c->stream));
    if (c->comm && allreduce_sum(c, c->packed.as<double>(), packed_n)) return 1;
    if (td >= 2) HIPCHK(c, hipEventRecord(c->ev[4], c->stream));
    PASS(self_publish ? await_result(c, pseq, packed_n) : fetch_result(c, c->packed.as<double>(), packed_n, c->comm != nullptr));
  }
  // with the in-kernel tail the host holds the result before the kernel has formally completed:
  // the events are read when the next call (or gfh_get_timers) needs them
  c->ev_pending = td;
  if (!(tail && !c->comm)) harvest_events(c);
  c->n_sweep++;
  if (sparse) {
    if (JTJ) {
      if (!c->jtj_prezeroed) memset(JTJ, 0, sizeof(double) * (size_t)dim * dim);
      const int* nr = c->h_nz_row.data(); const int* nc = c->h_nz_col.data();
      for (int k = 0; k < c->nnz; k++) {
        const double v = c->h_pinned[k];
        JTJ[(size_t)nc[k] * dim + nr[k]] = v; JTJ[(size_t)nr[k] * dim + nc[k]] = v;      // both triangles, as the dense path
      }
    }
    if (JTres) memcpy(JTres, c->h_pinned + c->nnz, sizeof(double) * dim);
    if (chi2) *chi2 = c->h_pinned[(size_t)c->nnz + dim];
  } else {
    if (JTJ) memcpy(JTJ, c->h_pinned, sizeof(double) * (size_t)dim * dim);
    if (JTres) memcpy(JTres, c->h_pinned + (size_t)dim * dim, sizeof(double) * dim);
    if (chi2) *chi2 = c->h_pinned[(size_t)dim * dim + dim];
  }
  c->have_sweep = true; c->j_valid = c->gen.store_j; c->res_valid = true;
  if (c->order_measured && build_orders(c)) return 1;
  return 0;
}

static int chi2_pass(gfh_ctx* c, const double* pars, double* chi2);

int gfh_chi2(gfh_ctx* c, const double* pars, double* chi2) {
  if (c && c->grp) return gfh::group_run(c, [&](gfh_ctx* k, int r) -> int { double mine = 0.0; return gfh_chi2(k, pars, r ? &mine : chi2); });
  NEED_GPU(c);
  for (;;) {
    // (a recovery replaces the model: the pass then reloads the kernels of the active set the fit is using)
    const std::vector<int32_t> act = c->cur_active, jac = c->cur_jac; const int dim = c->cur_dim;
    const bool had = c->have_sweep, jv = c->j_valid;
    const int rc = chi2_pass(c, pars, chi2);
    if (rc != kUnseen && rc != kGrowWs && rc != kIntegrandPath) return rc;
    if (repeat_pass(c, rc, pars)) return 1;
    if (!act.empty() && prepare_active(c, act.data(), (int)act.size(), jac.data(), dim)) return 1;
    // the new model keeps what the sweep before this chi2() left: its active set, column map and Jacobian in HBM (gfh_omega,
    // gfh_get_points and gfh_time_kernel after a recovery inside chi2() build on them, as gfh_omega's own loop does)
    if (!act.empty()) { c->have_sweep = had; c->j_valid = jv; }
  }
}

static int chi2_pass(gfh_ctx* c, const double* pars, double* chi2) {
  gfh::Range range("gadfit chi2");
  harvest_events(c);
  if (!c->nd) return fail(c, "no data set (gfh_set_data)");
  if (check_aux(c) || ensure_gb_partition(c)) return 1;
  if (!c->cur) {   // chi2 before any sweep: kernels for "no active parameter" are the same TU
    std::vector<int32_t> none;
    if (get_kernels(c, none, true)) return 1;
  }
  // the partial buffer follows the data set (gfh_set_data may have changed it)
  if (dev_alloc(c, c->chi2_partial, sizeof(double) * (size_t)std::max(1, c->n_gb)) || dev_alloc(c, c->vec, sizeof(double) * 64) ||
      pinned_reserve(c, 4096) || ensure_mesh(c)) return 1;
  if (upload_pars(c, pars)) return 1;
  const bool timed = c->n_gb && timed_launch(c, c->n_chi2);
  if (timed) HIPCHK(c, hipEventRecord(c->ev[0], c->stream));
  if (!c->n_gb) {                                       // a rank without points contributes an exact zero
    HIPCHK(c, hipMemsetAsync(c->vec.p, 0, sizeof(double), c->stream));
    if (c->comm && allreduce_sum(c, c->vec.as<double>(), 1)) return 1;
    PASS(fetch_result(c, c->vec.as<double>(), 1, c->comm != nullptr));
  } else if (!c->comm) {                                // single rank (or member of a host-summed group): the kernel's last workgroup posts the mailbox
    const unsigned long long seq = ++c->mail_seq;
    if (launch_model_chi2(c, 2, seq, mesh_mode_for(c, pars, true))) return 1;
    if (timed) HIPCHK(c, hipEventRecord(c->ev[1], c->stream));
    PASS(await_result(c, seq, 1));
  } else {
    if (launch_model_chi2(c, 1, 0, mesh_mode_for(c, pars, true))) return 1;
    if (timed) HIPCHK(c, hipEventRecord(c->ev[1], c->stream));
    if (allreduce_sum(c, c->vec.as<double>(), 1, true)) return 1;
    PASS(fetch_result(c, c->vec.as<double>(), 1, true));
  }
  if (timed) { c->t_chi2 += 1e-3 * ev_ms(c->ev[0], c->ev[1]); c->n_chi2_timed++; }
  c->n_chi2++;
  c->res_valid = c->gen.store_res;
  *chi2 = c->h_pinned[0];
  return 0;
}

// Adaptive parallelism, re_initialize STEP 1 (gadfit.F90:940-975): every rank's device time in the parallel parts
// (STEP 1+2, chi2, STEP 3) since the last call gives new image weights w = old - (1/n - (1/t)/sum(1/t)); the ranges are
// re-cut when that moves some rank's share by more than 1 % of an even share (the reference re-cuts every iteration at
// no cost because every image holds all data; here a move re-uploads the rank's points).  Collective.
int gfh_rebalance(gfh_ctx* c, int* moved) {
  GROUP(c, gfh_rebalance(k, r ? nullptr : moved));
  NEED_GPU(c);
  if (moved) *moved = 0;
  if (!c->load_balancing || c->nranks < 2 || c->hx.empty()) return 0;      // (switched on after gfh_set_data: nothing to cut from)
  harvest_events(c);
  const int n = c->nranks;
  const double total = scaled_time(c->t_sweep, c->n_sweep, c->n_sweep_timed) + scaled_time(c->t_gram, c->n_sweep, c->n_chain_timed) + scaled_time(c->t_chi2, c->n_chi2, c->n_chi2_timed) +
                       scaled_time(c->t_omega, c->n_omega, c->n_omega_timed);
  std::vector<double> t((size_t)n, 0.0);
  t[(size_t)c->rank] = total - c->lb_t_prev;
  c->lb_t_prev = total;
  if (c->comm) {
    if (dev_alloc(c, c->vec, sizeof(double) * (size_t)std::max(64, n + 1))) return 1;
    HIPCHK(c, hipMemcpyAsync(c->vec.p, t.data(), sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
    if (allreduce_sum(c, c->vec.as<double>(), (size_t)n)) return 1;
    if (fetch_result(c, c->vec.as<double>(), n, true)) return 1;
    for (int i = 0; i < n; i++) t[(size_t)i] = c->h_pinned[i];
  } else if (c->member_of) {
    int st = 0;
    if (gfh::group_allreduce(c, t.data(), (size_t)n, &st)) return 1;
  } else return 0;                                          // pseudo-ranks (gfh_debug_set_rank): nobody to exchange with
  std::vector<double> old_w = c->part_w;
  if ((int)old_w.size() != n) old_w.assign((size_t)n, 1.0 / n);
  double tmin = t[0];
  for (double v : t) tmin = std::min(tmin, v);
  if (!(tmin > 2.220446049250313e-16)) {                    // "too fast for load balancing to be effective" (gadfit.F90:964-970)
    c->load_balancing = false;
    return 0;
  }
  std::vector<double> w((size_t)n);
  double sum = 0.0;
  for (int i = 0; i < n; i++) { w[(size_t)i] = 1.0 / t[(size_t)i]; sum += w[(size_t)i]; }
  for (int i = 0; i < n; i++) {
    w[(size_t)i] = old_w[(size_t)i] - (1.0 / n - w[(size_t)i] / sum);     // gadfit.F90:974-975
    if (w[(size_t)i] < 0.0) w[(size_t)i] = 0.0;
  }
  double shift = 0.0;
  for (int i = 0; i < n; i++) shift = std::max(shift, std::fabs(w[(size_t)i] - old_w[(size_t)i]));
  if (shift * n < 0.01) return 0;
  if (gfh_repartition(c, w.data())) return 1;
  if (moved) *moved = 1;
  return 0;
}

// scatter a dim-vector into per-dataset blocks through Jacobian_indices (gadfit.F90:719)
static void scatter_delta(gfh_ctx* c, const double* delta, std::vector<double>& by_par, std::vector<double>& by_act) {
  const int na = (int)c->cur_active.size(), np = c->model.n_pars;
  by_par.assign((size_t)c->nd * np, 0.0); by_act.assign((size_t)c->nd * na, 0.0);
  for (int d = 0; d < c->nd; d++)
    for (int k = 0; k < na; k++) {
      const double v = delta[c->cur_jac[(size_t)d * na + k]];
      by_par[(size_t)d * np + c->cur_active[k]] = v; by_act[(size_t)d * na + k] = v;
    }
}

// per-gram-block partials [b][a] of a J^T v product -> out[dim], summed over ranks
static int jtv_finish(gfh_ctx* c, double* out) {
  const int na = (int)c->cur_active.size(), dim = c->cur_dim;
  const int ps = gram_partial_stride(c->cur_T);
  if ((int64_t)c->nd * na <= 4096 && c->merge_small) {          // one single-workgroup launch instead of three
    if (pinned_reserve(c, sizeof(double) * std::max<size_t>((size_t)dim + 1, 4096))) return 1;
    if (c->comm) {
      HIPCHK(c, launch_jtv_finish(c->stream, c->partial.as<double>(), ps, na, c->ds_first_gb.as<int>(), c->nd, dim, c->inv.as<int>(),
                                  c->vec.as<double>(), c->status.as<int>(), nullptr, nullptr, 0));
      if (allreduce_sum(c, c->vec.as<double>(), (size_t)dim)) return 1;
      PASS(fetch_result(c, c->vec.as<double>(), dim, true));
    } else {
      const unsigned long long seq = ++c->mail_seq;
      HIPCHK(c, launch_jtv_finish(c->stream, c->partial.as<double>(), ps, na, c->ds_first_gb.as<int>(), c->nd, dim, c->inv.as<int>(),
                                  c->vec.as<double>(), c->status.as<int>(), c->h_pinned, c->h_flag, seq));
      PASS(await_result(c, seq, dim));
    }
    memcpy(out, c->h_pinned, sizeof(double) * dim);
    return 0;
  }
  HIPCHK(c, launch_reduce_partials(c->stream, c->partial.as<double>(), ps, na, c->ds_first_gb.as<int>(), c->nd, c->G.as<double>()));
  HIPCHK(c, launch_assemble_vec(c->stream, c->G.as<double>(), na, c->nd, dim, c->inv.as<int>(), c->vec.as<double>()));
  if (c->comm && allreduce_sum(c, c->vec.as<double>(), (size_t)dim)) return 1;
  PASS(fetch_result(c, c->vec.as<double>(), dim, c->comm != nullptr));
  memcpy(out, c->h_pinned, sizeof(double) * dim);
  return 0;
}

static int jtv_to_host(gfh_ctx* c, const double* v_dev, double* out) {
  const int na = (int)c->cur_active.size();
  const int ps = gram_partial_stride(c->cur_T);
  if (c->n_gb) HIPCHK(c, launch_jtv(c->stream, c->J.as<double>(), c->ldj, na, v_dev, c->gb_start.as<i64>(), c->gb_slots.as<int>(),
                                     c->n_gb, c->partial.as<double>(), ps));
  return jtv_finish(c, out);
}

// STEP 3 without the stored Jacobian: gfh_k_omega_jt (generated) recomputes each point's Jacobian row
static int launch_model_omega_jt(gfh_ctx* c) {
  if (!c->n_gb) return 0;
  void* x = c->x.p; void* w = c->w.p; void* pars = c->pars.p; void* parg = c->cur->kernarg_pars ? (void*)c->h_pars : (void*)&pars;
  void* dpp = c->dpars.p; void* dp = c->cur->kernarg_pars ? (void*)c->h_dpars : (void*)&dpp;
  void* gs = c->gb_start.p; void* gn = c->gb_slots.p; void* gd = c->gb_ds.p; void* om = c->omega.p;
  void* part = c->partial.p; int ps = gram_partial_stride(c->cur_T); void* stp = c->status.p;
  void* ax = c->aux.p; long long lda = c->n_slots;
  void* args[] = {&x, &w, parg, dp, &gs, &gn, &gd, &om, &part, &ps, &stp, &ax, &lda};
  HIPCHK(c, hipModuleLaunchKernel(c->cur->omega_jt, c->n_gb, 1, 1, 256, 1, 1, 0, c->stream, args, nullptr));
  return 0;
}

static int omega_pass(gfh_ctx* c, const double* pars, const double* delta1, double* JTomega);

int gfh_omega(gfh_ctx* c, const double* pars, const double* delta1, double* JTomega) {
  if (c && c->grp) return gfh::group_run(c, [&](gfh_ctx* k, int r) -> int {
    std::vector<double> mine(r ? (size_t)std::max(1, k->cur_dim) : 0);
    return gfh_omega(k, pars, delta1, r ? mine.data() : JTomega); });
  NEED_GPU(c);
  for (;;) {
    const std::vector<int32_t> act = c->cur_active, jac = c->cur_jac; const int dim = c->cur_dim;
    const bool jv = c->j_valid;
    const int rc = omega_pass(c, pars, delta1, JTomega);
    if (rc != kUnseen && rc != kGrowWs && rc != kIntegrandPath) return rc;
    if (repeat_pass(c, rc, pars)) return 1;
    // the new model keeps the state STEP 3 builds on: the active set and column map of the sweep before it (and its Jacobian in HBM)
    if (act.empty() || prepare_active(c, act.data(), (int)act.size(), jac.data(), dim)) return act.empty() ? fail(c, "gfh_omega needs a preceding gfh_sweep") : 1;
    c->have_sweep = true; c->j_valid = jv;
  }
}

static int omega_pass(gfh_ctx* c, const double* pars, const double* delta1, double* JTomega) {
  gfh::Range range("gadfit omega (STEP 3)");
  harvest_events(c);
  if (!c->have_sweep) return fail(c, "gfh_omega needs a preceding gfh_sweep (active set, column map)");
  if (c->gen.finite_diff && c->gen.fd_col_sets)
    return fail(c, "gfh_omega: the central difference of use_ad = 0 (fitfunction.F90:188-203) has no column sets at p +- h*delta (gfh_set_fd_column_sets)");
  const bool recompute = c->cur && c->cur->omega_jt && !omega_needs_jacobian(c, (int)c->cur_active.size());
  if (!recompute && !c->j_valid) return fail(c, "gfh_omega: the Jacobian was not kept (gfh_set_keep_jacobian)");
  if (ensure_tile_table(c)) return 1;
  std::vector<double> by_par, by_act;
  scatter_delta(c, delta1, by_par, by_act);
  if (upload_pars(c, pars)) return 1;
  // delta1 scattered per dataset: pinned staging; by value with the kernel arguments for single-dataset
  // fits (as the parameter block), else an asynchronous copy in front of the kernel
  if (c->h_dpars_bytes < sizeof(double) * by_par.size()) {
    if (c->h_dpars) hipHostFree(c->h_dpars);
    c->h_dpars = nullptr; c->h_dpars_bytes = 0;
    HIPCHK(c, hipHostMalloc((void**)&c->h_dpars, sizeof(double) * by_par.size(), hipHostMallocDefault));
    c->h_dpars_bytes = sizeof(double) * by_par.size();
  }
  memcpy(c->h_dpars, by_par.data(), sizeof(double) * by_par.size());
  if (dev_alloc(c, c->dpars, sizeof(double) * by_par.size())) return 1;
  if (!(c->cur && c->cur->kernarg_pars))
    HIPCHK(c, hipMemcpyAsync(c->dpars.p, c->h_dpars, sizeof(double) * by_par.size(), hipMemcpyHostToDevice, c->stream));
  const bool timed = timed_launch(c, c->n_omega);
  if (timed) HIPCHK(c, hipEventRecord(c->ev[0], c->stream));
  if (recompute ? launch_model_omega_jt(c) : launch_model_omega(c, mesh_mode_for(c, pars, false))) return 1;
  if (timed) HIPCHK(c, hipEventRecord(c->ev[1], c->stream));
  PASS(recompute ? jtv_finish(c, JTomega) : jtv_to_host(c, c->omega.as<double>(), JTomega));
  if (timed) { c->t_omega += 1e-3 * ev_ms(c->ev[0], c->ev[1]); c->n_omega_timed++; }
  c->n_omega++;
  return 0;
}

int gfh_aux(gfh_ctx* c, int what, const double* delta1, double* out) {
  if (c && c->grp) return gfh::group_run(c, [&](gfh_ctx* k, int r) -> int {
    std::vector<double> mine(r ? (size_t)std::max(3, k->cur_dim) : 0);
    return gfh_aux(k, what, delta1, r ? mine.data() : out); });
  NEED_GPU(c);
  if (!c->have_sweep) return fail(c, "gfh_aux needs the Jacobian of a preceding gfh_sweep");
  if (!c->j_valid) return fail(c, "gfh_aux: the Jacobian was not kept (gfh_set_keep_jacobian)");
  if (!c->res_valid) return fail(c, "gfh_aux: the residual vector was not kept (gfh_set_keep_jacobian)");
  if (what == 0) return jtv_to_host(c, c->res.as<double>(), out);
  if (what != 1) return fail(c, "gfh_aux: unknown request");
  std::vector<double> by_par, by_act;
  scatter_delta(c, delta1, by_par, by_act);
  const int na = (int)c->cur_active.size(), ps = gram_partial_stride(c->cur_T);
  if (dev_alloc(c, c->dl, sizeof(double) * by_act.size())) return 1;
  HIPCHK(c, hipMemcpy(c->dl.p, by_act.data(), sizeof(double) * by_act.size(), hipMemcpyHostToDevice));
  if (c->n_gb) HIPCHK(c, launch_cosphi(c->stream, c->J.as<double>(), c->ldj, na, c->res.as<double>(), c->dl.as<double>(),
                                        c->gb_start.as<i64>(), c->gb_slots.as<int>(), c->gb_ds.as<int>(), c->n_gb, c->partial.as<double>(), ps));
  // sum over all workgroups regardless of dataset: reuse reduce with a 2-entry "dataset" table
  std::vector<int> all = {0, c->n_gb};
  DevBuf tmp; if (dev_alloc(c, tmp, sizeof(int) * 2)) return 1;
  HIPCHK(c, hipMemcpy(tmp.p, all.data(), sizeof(int) * 2, hipMemcpyHostToDevice));
  HIPCHK(c, launch_reduce_partials(c->stream, c->partial.as<double>(), ps, 3, tmp.as<int>(), 1, c->vec.as<double>()));
  if (c->comm && allreduce_sum(c, c->vec.as<double>(), 3)) { dev_free(tmp); return 1; }
  if (fetch_result(c, c->vec.as<double>(), 3, c->comm != nullptr)) { dev_free(tmp); return 1; }
  dev_free(tmp);
  memcpy(out, c->h_pinned, sizeof(double) * 3);
  return 0;
}

// ------------------------------------------------------------------------- timers / bench hooks
int gfh_get_timers(gfh_ctx* c, double* o) {
  if (!c) return 1;
  if (c->grp) {      // the slowest member of a device group (the counts are the same on all)
    for (int i = 0; i < 8; i++) o[i] = 0.0;
    for (int r = 0; r < gfh::group_size(c); r++) {
      double t[8];
      if (gfh_get_timers(gfh::group_member(c, r), t)) return 1;
      for (int i = 0; i < 8; i++) o[i] = std::max(o[i], t[i]);
    }
    return 0;
  }
  if (c->device >= 0) harvest_events(c);
  // (level 1 brackets every 8th sweep; the Gram / reduce / all-reduce stages are bracketed on those of the sampled sweeps that run
  // at level 2 -- every sampled one on the two-kernel path -- and scaled to all sweeps like the model kernels)
  o[0] = scaled_time(c->t_sweep, c->n_sweep, c->n_sweep_timed); o[1] = scaled_time(c->t_gram, c->n_sweep, c->n_chain_timed);
  o[2] = scaled_time(c->t_reduce, c->n_sweep, c->n_chain_timed); o[3] = scaled_time(c->t_allreduce, c->n_sweep, c->n_chain_timed);
  o[4] = scaled_time(c->t_chi2, c->n_chi2, c->n_chi2_timed); o[5] = scaled_time(c->t_omega, c->n_omega, c->n_omega_timed);
  o[6] = (double)c->n_sweep; o[7] = (double)c->n_chi2;
  return 0;
}
void gfh_reset_timers(gfh_ctx* c) {
  if (!c) return;
  if (c->grp) { for (int r = 0; r < gfh::group_size(c); r++) gfh_reset_timers(gfh::group_member(c, r)); return; }
  if (c->device >= 0) harvest_events(c);
  c->t_sweep = c->t_gram = c->t_reduce = c->t_allreduce = c->t_chi2 = c->t_omega = 0; c->n_sweep = c->n_chi2 = 0; c->n_allreduce = 0;
  c->t_sweep_min = c->t_sweep_max = c->t_sweep_last = 0; c->n_sweep_timed = c->n_chi2_timed = c->n_omega = c->n_omega_timed = c->n_chain_timed = 0;
}
int gfh_get_timer_spread(gfh_ctx* c, double* o) {
  if (!c) return 1;
  if (c->grp) return gfh_get_timer_spread(gfh::group_member(c, 0), o);
  if (c->device >= 0) harvest_events(c);
  o[0] = c->t_sweep_min; o[1] = c->t_sweep_max; o[2] = c->t_sweep_last; o[3] = (double)c->n_sweep_timed;
  return 0;
}

int gfh_launch_sweep(gfh_ctx* c) { GROUP(c, gfh_launch_sweep(k)); NEED_GPU(c); if (!c->have_sweep) return fail(c, "call gfh_sweep once first"); return launch_model_sweep(c); }
int gfh_launch_gram(gfh_ctx* c) { GROUP(c, gfh_launch_gram(k)); NEED_GPU(c); if (!c->have_sweep) return fail(c, "call gfh_sweep once first"); return launch_gram_chain(c, false); }
int gfh_launch_chi2(gfh_ctx* c) {
  GROUP(c, gfh_launch_chi2(k));
  NEED_GPU(c); if (!c->have_sweep) return fail(c, "call gfh_sweep once first");
  return launch_model_chi2(c, 1, 0);
}
int gfh_sync(gfh_ctx* c) { GROUP(c, gfh_sync(k)); NEED_GPU(c); HIPCHK(c, hipStreamSynchronize(c->stream)); return 0; }
void* gfh_stream(gfh_ctx* c) { if (c && c->grp) c = gfh::group_member(c, 0); return c ? (void*)c->stream : nullptr; }

int gfh_time_kernel(gfh_ctx* c, int which, int reps, double* avg_ms) {
  if (c && c->grp) {      // all members launch together; the slowest member's average
    std::vector<double> ms((size_t)gfh::group_size(c), 0.0);
    if (gfh::group_run(c, [&](gfh_ctx* k, int r) -> int { return gfh_time_kernel(k, which, reps, &ms[(size_t)r]); })) return 1;
    *avg_ms = *std::max_element(ms.begin(), ms.end());
    return 0;
  }
  NEED_GPU(c);
  harvest_events(c);
  if (!c->have_sweep) return fail(c, "call gfh_sweep once first");
  if (reps < 1) reps = 1;
  if ((which == 3 || which == 6 || which == 9) && !c->dpars.p) return fail(c, "call gfh_omega once first");
  HIPCHK(c, hipEventRecord(c->ev[0], c->stream));
  for (int r = 0; r < reps; r++) {
    int rc = 0;
    switch (which) {
      case 0: rc = use_fused(c) ? launch_model_sweep_gram(c) : launch_model_sweep(c); break;
      case 4: rc = launch_model_sweep(c); break;
      case 5: if (!use_fused(c)) return fail(c, "no fused kernel for this active set"); rc = launch_model_sweep_gram(c); break;
      case 1: if (c->n_gb) { hipError_t e = launch_gram(c->stream, c->cur_T, c->J.as<double>(), c->ldj, (int)c->cur_active.size(),
                                 c->res.as<double>(), c->gb_start.as<i64>(), c->gb_slots.as<int>(), c->n_gb, c->partial.as<double>());
                             if (e != hipSuccess) return fail(c, hipGetErrorString(e)); } break;
      case 2: rc = launch_model_chi2(c, 1, 0); break;
      case 3: rc = launch_model_omega(c); break;
      // (8, 9: STEP 1 / STEP 3 replaying the recorded quadrature meshes -- valid after a pass at the parameters still in the staging block)
      case 8: if (!c->mesh_valid) return fail(c, "no recorded quadrature mesh to replay"); rc = launch_model_sweep(c, 2); break;
      case 9: if (!c->mesh_valid) return fail(c, "no recorded quadrature mesh to replay"); rc = launch_model_omega(c, 2); break;
      case 6: if (!c->cur->omega_jt) return fail(c, "gfh_k_omega_jt is not available for this model"); rc = launch_model_omega_jt(c); break;
      case 7: if (!c->j_valid) return fail(c, "the Jacobian was not kept (gfh_set_keep_jacobian)");
              if (c->n_gb) { hipError_t e = launch_jtv(c->stream, c->J.as<double>(), c->ldj, (int)c->cur_active.size(), c->res.as<double>(),
                                 c->gb_start.as<i64>(), c->gb_slots.as<int>(), c->n_gb, c->partial.as<double>(), gram_partial_stride(c->cur_T));
                             if (e != hipSuccess) return fail(c, hipGetErrorString(e)); } break;
      default: return fail(c, "unknown kernel id");
    }
    if (rc) return rc;
  }
  HIPCHK(c, hipEventRecord(c->ev[1], c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  *avg_ms = ev_ms(c->ev[0], c->ev[1]) / reps;
  return 0;
}

// ------------------------------------------------------------------------- debug read-back
static int unpad(gfh_ctx* c, const double* dev, double* out) {
  std::vector<double> h((size_t)c->n_slots);
  if (c->n_slots) HIPCHK(c, hipMemcpy(h.data(), dev, sizeof(double) * (size_t)c->n_slots, hipMemcpyDeviceToHost));
  for (int d = 0; d < c->nd; d++) {
    const int64_t len = c->lb[d + 1] - c->lb[d];
    if (len) memcpy(out + c->lb[d], &h[(size_t)c->ds_slot[d]], sizeof(double) * (size_t)len);
  }
  return 0;
}
int gfh_get_residuals(gfh_ctx* c, double* out) {
  GROUP(c, gfh_get_residuals(k, out + k->begin));
  NEED_GPU(c);
  if (!c->res_valid) return fail(c, "the residual vector of the last chi2 pass was not kept (gfh_set_keep_jacobian mode 2)");
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return unpad(c, c->res.as<double>(), out);
}
int gfh_get_weights(gfh_ctx* c, double* out) { GROUP(c, gfh_get_weights(k, out + k->begin)); NEED_GPU(c); HIPCHK(c, hipStreamSynchronize(c->stream)); return unpad(c, c->w.as<double>(), out); }
int gfh_get_omega(gfh_ctx* c, double* out) { GROUP(c, gfh_get_omega(k, out + k->begin)); NEED_GPU(c); HIPCHK(c, hipStreamSynchronize(c->stream)); return unpad(c, c->omega.as<double>(), out); }
int gfh_get_jacobian(gfh_ctx* c, double* out) {
  GROUP(c, gfh_get_jacobian(k, out + (size_t)k->begin * k->cur_active.size()));
  NEED_GPU(c);
  if (!c->have_sweep) return fail(c, "no Jacobian on the device yet");
  if (!c->j_valid) return fail(c, "the Jacobian was not kept (gfh_set_keep_jacobian)");
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const int na = (int)c->cur_active.size();
  std::vector<double> col((size_t)c->count);
  for (int a = 0; a < na; a++) {
    if (unpad(c, c->J.as<double>() + (size_t)a * c->ldj, col.data())) return 1;
    for (int64_t i = 0; i < c->count; i++) out[(size_t)i * na + a] = col[(size_t)i];
  }
  return 0;
}

// Read-back of single points (local indices into this rank's range): the residual and the Jacobian row [n][n_act] of each -- for
// checks at sizes where the whole Jacobian (28.8 GB at 1e8 points x 32 parameters) does not belong on the host.
int gfh_get_points(gfh_ctx* c, int n, const int64_t* index, double* res_out, double* jac_out) {
  NOT_FOR_GROUP(c, "gfh_get_points");
  NEED_GPU(c);
  if (!c->have_sweep) return fail(c, "no sweep on the device yet");
  if (jac_out && !c->j_valid) return fail(c, "the Jacobian was not kept (gfh_set_keep_jacobian)");
  if (res_out && !c->res_valid) return fail(c, "the residual vector of the last pass was not kept");
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const int na = (int)c->cur_active.size();
  for (int k = 0; k < n; k++) {
    const int64_t i = index[k];
    if (i < 0 || i >= c->count) return fail(c, "gfh_get_points: index outside this rank's range");
    int d = 0;
    while (d + 1 < c->nd && i >= c->lb[(size_t)d + 1]) d++;
    const int64_t slot = c->ds_slot[(size_t)d] + (i - c->lb[(size_t)d]);
    if (res_out) HIPCHK(c, hipMemcpy(res_out + k, c->res.as<double>() + slot, sizeof(double), hipMemcpyDeviceToHost));
    if (jac_out) HIPCHK(c, hipMemcpy2D(jac_out + (size_t)k * na, sizeof(double), c->J.as<double>() + slot, sizeof(double) * (size_t)c->ldj,
                                       sizeof(double), (size_t)na, hipMemcpyDeviceToHost));
  }
  return 0;
}

// The abscissas as they lie on the device, back into the caller's concatenated array: this rank's range [begin, begin + count) of
// x_out[n_total] (a device group: every member's range, so the whole array).
int gfh_get_abscissas(gfh_ctx* c, double* x_out) {
  GROUP(c, gfh_get_abscissas(k, x_out));
  NEED_GPU(c);
  if (!x_out) return fail(c, "gfh_get_abscissas: null argument");
  if (!c->nd) return fail(c, "no data set (gfh_set_data)");
  HIPCHK(c, hipStreamSynchronize(c->stream));
  for (int d = 0; d < c->nd; d++) {
    const int64_t len = c->lb[(size_t)d + 1] - c->lb[(size_t)d];
    if (len > 0) HIPCHK(c, hipMemcpy(x_out + c->begin + c->lb[(size_t)d], c->x.as<double>() + c->ds_slot[(size_t)d], sizeof(double) * (size_t)len,
                                     hipMemcpyDeviceToHost));
  }
  return 0;
}

int gfh_jacobian_indices(int nd, int na, const int32_t* active, const int32_t* is_global, int32_t* jac) {
  int shift = 0;   // gadfit.F90:618-628
  for (int i = 0; i < nd; i++)
    for (int j = 0; j < na; j++) {
      if (is_global[active[j]]) { jac[i * na + j] = j; if (i > 0) shift++; }
      else jac[i * na + j] = j + i * na - shift;
    }
  return nd * na - shift;
}

}  // extern "C"

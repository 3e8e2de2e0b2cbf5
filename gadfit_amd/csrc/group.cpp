// group.cpp -- single-process device group (see group.h).  One host thread per member context (the caller's own
// for member 0, a persistent one for each of the others); every C-ABI call on the group handle is run on all members at once (the reference's
// "same program on every image", gadfit.F90:977-1002 for the split, misc.F90:133-170 for the sum).
#include "group.h"
#include "context.h"
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>

namespace gfh {

struct Group {
  std::vector<gfh_ctx*> kids;
  std::vector<std::thread> workers;
  // dispatch: the caller publishes a task and bumps `gen`; workers spin briefly, then sleep on the condvar
  std::mutex m;
  std::condition_variable cv;
  std::atomic<unsigned long long> gen{0};
  std::atomic<int> done{0};
  bool stop = false;
  const std::function<int(gfh_ctx*, int)>* task = nullptr;
  std::vector<int> rc;
  // Host all-reduce (what replaces the two barriers + P gets of misc.F90:133-170 when the members' sums do not travel by RCCL): every
  // member copies its numbers into its own row of a published image and raises its own sequence word (one cache line per member, no
  // shared counter), then adds the rows in rank order as their owners arrive.  Two images, used alternately: a member can only
  // publish sum k + 2 into the image of sum k after every member has finished reading sum k (it has seen their rows of sum k + 1).
  // ONE synchronisation per sum.  A phase-counting spin barrier remains for the rare growth of the images.
  struct alignas(64) Mail { std::atomic<unsigned long long> seq{0}; unsigned long long mine = 0; int st[2] = {0, 0}; std::vector<double> sum; };
  std::vector<Mail> mail;
  std::vector<double> image[2];
  size_t cap = 0;                      // doubles per row (rows start on cache lines)
  std::atomic<int> bar_count{0};
  std::atomic<unsigned long long> bar_phase{0};
  std::atomic<int> abort{0};
};

static inline void cpu_relax() { __builtin_ia32_pause(); }

static void worker_main(Group* g, int r) {
  unsigned long long seen = 0;
  for (;;) {
    bool go = false;
    for (int spin = 0; spin < 20000; spin++) {
      if (g->gen.load(std::memory_order_acquire) != seen) { go = true; break; }
      cpu_relax();
    }
    if (!go) {
      std::unique_lock<std::mutex> lk(g->m);
      g->cv.wait(lk, [&] { return g->stop || g->gen.load(std::memory_order_acquire) != seen; });
      if (g->stop) return;
    }
    {
      std::lock_guard<std::mutex> lk(g->m);
      if (g->stop) return;
    }
    seen = g->gen.load(std::memory_order_acquire);
    int rc = 1;
    try { rc = (*g->task)(g->kids[r], r); } catch (...) { rc = fail(g->kids[r], "exception in a device-group member"); }
    g->rc[r] = rc;
    if (rc) g->abort.store(1, std::memory_order_release);
    g->done.fetch_add(1, std::memory_order_acq_rel);
  }
}

int group_run(gfh_ctx* h, const std::function<int(gfh_ctx*, int)>& fn) {
  Group* g = h->grp;
  const int n = (int)g->kids.size();
  g->task = &fn;
  g->abort.store(0, std::memory_order_relaxed);
  g->bar_count.store(0, std::memory_order_relaxed);
  g->done.store(0, std::memory_order_relaxed);
  // (no member is running: the sums of the task about to start count from 1 again, whatever a failed task left behind)
  for (auto& m : g->mail) { m.seq.store(0, std::memory_order_relaxed); m.mine = 0; }
  {
    std::lock_guard<std::mutex> lk(g->m);
    g->gen.fetch_add(1, std::memory_order_release);
  }
  g->cv.notify_all();
  // member 0 is the calling thread itself: N threads for N members (a caller that only waited would take a core from a member
  // where there are no more cores than members, and the hand-off to a ninth thread is the slowest one)
  {
    int rc0 = 1;
    try { rc0 = fn(g->kids[0], 0); } catch (...) { rc0 = fail(g->kids[0], "exception in a device-group member"); }
    g->rc[0] = rc0;
    if (rc0) g->abort.store(1, std::memory_order_release);
  }
  for (unsigned spin = 0; g->done.load(std::memory_order_acquire) != n - 1; spin++) {
    if (spin < 50000) cpu_relax();
    else std::this_thread::sleep_for(std::chrono::microseconds(20));      // a whole gfh_fit runs inside one task
  }
  g->task = nullptr;
  for (int r = 0; r < n; r++)
    if (g->rc[r]) {
      // prefer the message of a member that failed by itself over "another device failed"
      int src = r;
      for (int q = 0; q < n; q++) if (g->rc[q] && g->kids[q]->err.find("another device of the group") == std::string::npos) { src = q; break; }
      return fail(h, g->kids[src]->err.empty() ? std::string("a device-group member failed") : g->kids[src]->err);
    }
  return 0;
}

static int barrier(Group* g) {
  const int n = (int)g->kids.size();
  const unsigned long long ph = g->bar_phase.load(std::memory_order_acquire);
  if (g->bar_count.fetch_add(1, std::memory_order_acq_rel) + 1 == n) {
    g->bar_count.store(0, std::memory_order_relaxed);
    g->bar_phase.store(ph + 1, std::memory_order_release);
    return 0;
  }
  while (g->bar_phase.load(std::memory_order_acquire) == ph) {
    if (g->abort.load(std::memory_order_acquire)) return 1;
    cpu_relax();
  }
  return 0;
}

int group_allreduce(gfh_ctx* c, double* buf, size_t n, int* status) {
  Group* g = c->member_of;
  const int N = (int)g->kids.size(), r = c->rank;
  if (N == 1) return 0;
  if (n > g->cap) {                     // (every member makes the same call with the same n: all of them come here together)
    if (barrier(g)) return fail(c, "another device of the group failed");
    if (r == 0) {
      const size_t cap = ((std::max<size_t>(2 * n, 2048) + 7) / 8) * 8;
      try { g->image[0].assign((size_t)N * cap, 0.0); g->image[1].assign((size_t)N * cap, 0.0); g->cap = cap; }
      catch (const std::exception&) { g->abort.store(1, std::memory_order_release); }
    }
    if (barrier(g) || n > g->cap) return fail(c, "device group: no memory for the image of the members' sums");
  }
  Group::Mail& me = g->mail[(size_t)r];
  const unsigned long long k = ++me.mine;
  const int side = (int)(k & 1);
  const size_t cap = g->cap;
  double* img = g->image[side].data();
  memcpy(img + (size_t)r * cap, buf, sizeof(double) * n);
  me.st[side] = *status;
  me.seq.store(k, std::memory_order_release);
  // every member forms the same sum in rank order: identical bits everywhere, so the replicated host logic (accept / reject,
  // lambda) takes the same decisions on every member; a row is added as soon as its owner has published it
  // (into the member's own scratch: a sum that another member's failure cuts short leaves the caller's numbers as they were)
  if (me.sum.size() < n) me.sum.resize(std::max<size_t>(n, cap));
  double* acc = me.sum.data();
  int stmax = 0;
  for (int q = 0; q < N; q++) {
    Group::Mail& m = g->mail[(size_t)q];
    for (unsigned spin = 0; m.seq.load(std::memory_order_acquire) < k; spin++) {
      if (g->abort.load(std::memory_order_acquire)) return fail(c, "another device of the group failed");
      if (spin < 4096) cpu_relax(); else std::this_thread::yield();          // (more members than cores: let the owner run)
    }
    const double* row = img + (size_t)q * cap;
    if (q == 0) memcpy(acc, row, sizeof(double) * n);
    else for (size_t i = 0; i < n; i++) acc[i] += row[i];
    stmax = m.st[side] > stmax ? m.st[side] : stmax;
  }
  memcpy(buf, acc, sizeof(double) * n);
  *status = stmax;
  return 0;
}

int group_size(const gfh_ctx* h) { return h && h->grp ? (int)h->grp->kids.size() : 0; }
gfh_ctx* group_member(const gfh_ctx* h, int r) { return h->grp->kids[r]; }

void group_destroy(gfh_ctx* h) {
  Group* g = h->grp;
  {
    std::lock_guard<std::mutex> lk(g->m);
    g->stop = true;
    g->gen.fetch_add(1, std::memory_order_release);
  }
  g->cv.notify_all();
  for (auto& t : g->workers) if (t.joinable()) t.join();
  for (gfh_ctx* k : g->kids) { k->member_of = nullptr; gfh_destroy(k); }
  delete g;
  h->grp = nullptr;
}

int group_create(int n_devices, const int* devices, gfh_ctx** out) {
  if (!out) return 1;
  *out = nullptr;
  // members on device -1 are compile-only contexts (no GPU bound, nothing can run on them): a group made of
  // such members exercises the fan-out, the barrier and the ordered host sum without a card (CPU tests)
  bool dry = devices != nullptr && n_devices > 0;
  for (int i = 0; dry && i < n_devices; i++) dry = devices[i] == -1;
  int visible = 0;
  if (!dry && (hipGetDeviceCount(&visible) != hipSuccess || visible <= 0)) {
    set_global_error("no HIP device available (libgadfit_hip has no CPU fallback)");
    return 1;
  }
  if (n_devices <= 0) { n_devices = visible; devices = nullptr; }
  // GADFIT_HIP_GROUP_WRAP=1: member i of a counted group (devices == NULL) sits on device i modulo the visible
  // ones -- more images than cards, e.g. to rehearse an 8-member run on a one-GPU machine
  bool wrap = false;
  if (const char* e = getenv("GADFIT_HIP_GROUP_WRAP")) wrap = atoi(e) != 0;
  std::vector<int> dev(n_devices);
  for (int i = 0; i < n_devices; i++) {
    dev[i] = devices ? devices[i] : (wrap ? i % visible : i);
    if (!dry && (dev[i] < 0 || dev[i] >= visible)) { set_global_error("device group: device index out of range"); return 1; }
  }
  // The members' sums travel by RCCL all-reduce over xGMI (ncclCommInitAll: one communicator per member) wherever every
  // member has a card of its own; GADFIT_HIP_GROUP_REDUCE=host asks for the ordered host sum instead, which is also what
  // members that share a card get (rehearsals on a one-GPU machine: RCCL cannot put two ranks on one device).
  bool distinct = true;
  for (int i = 0; i < n_devices; i++) for (int j = 0; j < i; j++) if (dev[i] == dev[j]) distinct = false;
  bool rccl = !dry && distinct && n_devices > 1;
  if (const char* e = getenv("GADFIT_HIP_GROUP_REDUCE")) {
    if (!strcmp(e, "host")) rccl = false;
    else if (!strcmp(e, "rccl")) {
      if (dry || !distinct) { set_global_error("device group: RCCL needs one device per member (GADFIT_HIP_GROUP_REDUCE=rccl)"); return 1; }
      rccl = true;
    } else { set_global_error("GADFIT_HIP_GROUP_REDUCE must be rccl or host"); return 1; }
  }
  gfh_ctx* h = nullptr;
  if (gfh_create(-1, &h)) return 1;
  Group* g = new Group();
  h->grp = g;
  for (int i = 0; i < n_devices; i++) {
    gfh_ctx* k = nullptr;
    if (gfh_create(dev[i], &k)) { for (gfh_ctx* q : g->kids) gfh_destroy(q); delete g; h->grp = nullptr; gfh_destroy(h); return 1; }
    k->nranks = n_devices; k->rank = i; k->member_of = g;
    g->kids.push_back(k);
  }
  if (rccl) {
    std::vector<ncclComm_t> comms(n_devices);
    ncclResult_t r = ncclCommInitAll(comms.data(), n_devices, dev.data());
    if (r != ncclSuccess) {
      set_global_error(std::string("ncclCommInitAll: ") + ncclGetErrorString(r));
      for (gfh_ctx* q : g->kids) { q->member_of = nullptr; gfh_destroy(q); }
      delete g; h->grp = nullptr; gfh_destroy(h);
      return 1;
    }
    for (int i = 0; i < n_devices; i++) g->kids[i]->comm = comms[i];
  }
  g->rc.assign(n_devices, 0);
  g->mail = std::vector<Group::Mail>((size_t)n_devices);
  for (int i = 1; i < n_devices; i++) g->workers.emplace_back(worker_main, g, i);      // (member 0 runs on the caller's thread: group_run)
  *out = h;
  return 0;
}

}  // namespace gfh

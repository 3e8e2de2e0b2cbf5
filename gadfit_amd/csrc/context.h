// context.h -- per-GPU state of the hot path (one context = one "image" of the reference).
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <map>
#include <atomic>
#include <thread>
#include <string>
#include <vector>
#include "../../include/gadfit_hip.h"
#include "kernels.h"
#include "model.h"
#include "rtc.h"

namespace gfh {

constexpr int kPadGranule = 512;    // dataset segments are padded to this many slots (= one pass of the widest workgroup, 8 waves)

// The status buffer of a context: [0] the kernels' status word (0 ok, 1 quadrature workspace exhausted, 2 not lowered, 3 a data point
// took a branch of eval() no recorded variant covers), [16] and [24] arrival counters of k_publish / gfh_k_chi2, and from byte 64 the
// report of unseen branches: an arrival counter, then from byte 128 up to kUnseenCap entries {slot, guard outcomes so far (bit k = guard k
// taken), their number} -- layout shared with the generated source (codegen.cpp, gfh_report_unseen).
constexpr int kUnseenCap = 120;
constexpr size_t kStatusBytes = 128 + (size_t)kUnseenCap * 24;
struct UnseenEntry { long long slot; unsigned long long path; int n_guards; int pad; };

struct Group;                       // group.h: single-process device group

struct DevBuf {
  void* p = nullptr; size_t bytes = 0;
  template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

}  // namespace gfh

struct gfh_ctx {
  int device = -1;                 // -1: compile-only context (no GPU bound)
  hipStream_t stream = nullptr;
  std::string err;

  // communicator (replaces coarray images)
  ncclComm_t comm = nullptr;
  int nranks = 1, rank = 0;
  gfh::Group* grp = nullptr;        // this context is the handle of a device group (gfh_create_group): calls fan out to the members
  gfh::Group* member_of = nullptr;  // this context is member `rank` of that group: results are summed over the members on the host

  // adaptive parallelism (load_balancing of gadf_fit, gadfit.F90:672-673, 935-983): image weights of the current
  // partition, and a host copy of the whole point array (every image of the reference holds it) to cut new ranges from
  bool load_balancing = false;
  std::vector<double> part_w;       // empty = 1/nranks each
  std::vector<double> hx, hy, hw, haux; int h_n_aux = 0; int weights_type = -1;
  double lb_t_prev = 0.0; long lb_moves = 0;
  // data partition
  int64_t n_total = 0, begin = 0, count = 0;
  int nd = 0;
  std::vector<int64_t> dp;          // global data_positions (nd+1)
  std::vector<int64_t> lb;          // local img_bounds relative to `begin` (nd+1)
  std::vector<int64_t> ds_slot;     // first slot of each dataset (nd+1)
  int64_t n_slots = 0;
  int64_t ldj = 0;                  // column stride of J in doubles (= n_slots)
  int n_gb = 0;
  int gb_target = 0;                // number of gram workgroups the partition was built for (follows the model kind)
  std::vector<int64_t> h_gb_start; std::vector<int> h_gb_slots, h_gb_ds, h_ds_first_gb;
  gfh::DevBuf x, y, w, res, omega, is_pad, J, tile_ds, gb_start, gb_slots, gb_ds, ds_first_gb;
  // pattern-only assembly/transfer of global fits: upper-triangle entries (row <= col) some dataset touches
  bool sparse_ok = true, sparse = false; int nnz = 0;   // GADFIT_HIP_SPARSE
  gfh::DevBuf nz_row, nz_col; std::vector<int> h_nz_row, h_nz_col;
  gfh::DevBuf gs_meta, gs_list; int gs_n = 0; bool gs_sparse = false;   // source lists of the packed (pattern) image for k_gather_sum
  bool jtj_prezeroed = false;       // gfh_fit: the caller's JTJ buffer holds zeros off the pattern already
  gfh::DevBuf owner;                // [dim] the one dataset using a column, or -1 (k_assemble)
  gfh::DevBuf aux; int n_aux = 0;   // auxiliary per-point columns [n_aux][n_slots] (gfh_set_aux)
  // Mesh hand-over of quadrature models (codegen.cpp, mesh_build): per slot and outermost integrate() call site the record of the
  // bisections the last recording pass made, and the parameter block it made them at.  A pass at exactly those parameters replays
  // them instead of bisecting again: the sweep of an accepted step after the trial chi2() there, STEP 3 after the sweep.
  gfh::DevBuf tile_cost, tile_order, gb_order;   // models with integrate(): measured cost per tile, tiles / gram blocks expensive first (build_orders)
  int n_integrand_rounds = 0;          // passes repeated in a row because an integrand met an unrecorded path (recover_integrand_path)
  bool order_on = true, order_ready = false, order_want = false, order_measured = false; int order_age = 0;
  gfh::DevBuf mesh; int mesh_stride = 0;
  std::vector<double> mesh_pars; bool mesh_valid = false, mesh_on = true;   // GADFIT_HIP_MESH (0: every pass bisects)
  long n_mesh_replays = 0;
  gfh::DevBuf partial, G, chi2_partial, packed, pars, dpars, inv, dl, vec, status;
  int* h_status = nullptr;          // pinned, host-coherent 64 B: the result mailbox's flag lives at byte 8
  unsigned long long* h_flag = nullptr;   // sequence number of the last published result (k_publish)
  unsigned long long mail_seq = 0;
  int tile = 0, n_tiles = 0;        // tile of the loaded kernels (tile_ds is built for it)
  double* h_pinned = nullptr; size_t h_pinned_bytes = 0;   // results (D2H)
  double* h_pars = nullptr; size_t h_pars_bytes = 0;       // parameter block (H2D)
  double* h_dpars = nullptr; size_t h_dpars_bytes = 0;     // delta1 per dataset for STEP 3 (H2D / kernel argument)

  // model + kernels
  gfh::Model model; bool has_model = false;
  long model_serial = 0, aux_serial = 0;   // bumped by gfh_set_model* / gfh_set_aux*: did an unseen-branch handler change anything?
  gfh_unseen_handler unseen_fn = nullptr; void* unseen_user = nullptr;
  gfh_pars_hook pars_fn = nullptr; void* pars_user = nullptr;      // gfh_set_pars_hook: passive entries refreshed before every pass
  bool in_pars_hook = false;                                         // ... the hook is running (on this context's own thread)
  // Quadrature workspaces beyond the scratch budget (GenConfig::ws_global): the context's pool in global memory, one slot of
  // wsg_wave_doubles per wave of a launch; the launchers cap their grids at the slots there are.  Allocated at the first launch that
  // needs it (hipMalloc: a failure is an error code, not the runtime's abort), freed by gfh_destroy / when the model changes its sizes.
  gfh::DevBuf wsg; int64_t wsg_waves = 0, wsg_wave_doubles = 0;
  int64_t wsg_tried = 0;             // the largest number of slots the pool was last ASKED for (the card may have granted fewer: wsg_waves): not asked again until more are wanted
  bool ws_grown = false;            // a pass has exhausted the fast workspaces: the kernels carry the user's sizes (kept through a recovery's new model)
  bool in_recovery = false;         // the unseen-branch handler is running (gfh_set_model_variants then keeps ws_grown)
  int ws_fast = 100;                // quadrature workspace the kernels carry first (GADFIT_HIP_WS_FAST; 0: the user's size from the start)
  std::thread pending;              // gfh_set_data_begin: the upload in flight (joined by the next call on this context)
  std::vector<int32_t> pending_hint_cols;   // gfh_set_variant_hint_columns: consumed by the next gfh_set_model_variants
  bool creating = false;      // `pending` is the device part of gfh_create_begin (not an upload): gfh_set_data_begin chains its upload behind it
  bool create_failed = false; std::string create_err;      // ... and it failed: every call that needs the device fails with its message
  int pending_rc = 0;
  std::atomic<bool> stop_warm{false};   // set by join_pending: the upload thread stops keeping the part busy
  bool keep_warm = false;           // GADFIT_HIP_KEEP_WARM=1: the upload thread keeps the part busy until the caller is back.  Off by default since round 5: it
                                    // shortens the 10 iterations of the first fit by 0.7 ms (6.7 -> 6.0) and the first gadf_fit by nothing that can be
                                    // measured (303 ms either way, profiles/r05_keep_warm.md) for ~120 ms of dummy launches
  double warm_ms = 0;               // how long the last upload thread kept the part busy after its upload
  void* hc_dst = nullptr; const void* hc_src = nullptr; size_t hc_bytes = 0;   // gfh_queue_host_copy: a host-side copy made beside the next upload
  std::thread host_copy;               // ... on a thread of its own (gfh_wait_host_copy joins it)
  long n_unseen_rounds = 0;         // passes repeated because a point left the recorded decision tree (since the model was set)
  gfh::GenConfig gen;
  std::map<std::vector<int32_t>, gfh::ModelKernels> kernel_cache;
  gfh::ModelKernels* cur = nullptr;
  std::vector<int32_t> cur_active, cur_jac;
  int cur_dim = 0, cur_T = 0;
  const gfh::ModelKernels* prepared_cur = nullptr;
  bool prepared = false, prepared_store_j = true;   // prepare_active's work is valid for (cur, cur_active, cur_jac, cur_dim)
  bool have_sweep = false;          // a sweep ran with the current active set (res valid on device)
  bool j_valid = false;             // the Jacobian of that sweep is in HBM
  bool res_valid = false;           // the residual vector of the last pass is in HBM
  int keep_jacobian = 1;            // 0 never, 1 always (reference behaviour), 2 gfh_fit decides (GADFIT_HIP_KEEP_J)
  int lookahead = 1;                // gfh_fit / gfh_lm_iterate: first trial chi2 from a sweep at the trial point (GADFIT_HIP_LOOKAHEAD)
  bool kernarg = true;              // one dataset: parameters as a by-value kernel argument instead of an H2D copy per pass (GADFIT_HIP_KERNARG)
  bool merge_small = true;          // J^T v: reduce + assemble + publish as one single-workgroup launch when small (GADFIT_HIP_MERGE_SMALL)
  bool tail = true;                 // fused kernel reduces/assembles/publishes in its own tail for small dim^2*n_datasets (GADFIT_HIP_TAIL)
  gfh::DevBuf slice, counters, tail_dev; std::vector<char> tail_host;
  bool fused = true;                // STEP 1+2 in one kernel (GADFIT_HIP_FUSED=0: separate sweep and Gram kernels)

  // timers (seconds) + counters
  double t_sweep = 0, t_gram = 0, t_reduce = 0, t_allreduce = 0, t_chi2 = 0, t_omega = 0;
  long n_sweep = 0, n_chi2 = 0, n_allreduce = 0;
  double t_sweep_min = 0, t_sweep_max = 0, t_sweep_last = 0; long n_sweep_timed = 0, n_chi2_timed = 0, n_omega = 0, n_omega_timed = 0, n_chain_timed = 0;
  // 0: no events; 1: events around every 8th launch of each model kernel (an event record costs ~4 us of stream time: two per
  // launch were 8 us of a 35 us small-fit iteration), the sums scaled to all launches; 2: every launch, also reduce/all-reduce
  int timer_detail = 1;
  int ev_pending = 0;               // timer level of a sweep whose events have not been read yet
  int placement_tries = 16;         // candidate allocations of a large Jacobian buffer that are timed (gfh_set_placement_tries; 1: take the first)
  double placement_ms[8] = {0};     // the candidates' store-stream times of the last placement, [0] = the one kept
  int placement_n = 0;
  int placement_data_n = 0; double placement_data_ms = 0.0;   // round 6: re-placements of {x, y, w, res} tried behind the Jacobian's, the kernel's time on the set kept
  int placement_after = 48;         // sweeps that must have written the buffer before candidates are timed (gfh_set_placement_after)
  int64_t sweeps_on_J = 0;          // ... counted since the buffer was (re)allocated
  double placement_copy_rate = 0;   // B/s of a device-to-device copy inside the first candidate (the measure the placement's thresholds scale with)
  bool placement_pending = false;   // the Jacobian buffer was (re)allocated and is large: the next sweep that writes it times candidates first
  hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
};

namespace gfh {
int fail(gfh_ctx* c, const std::string& msg);
void set_global_error(const std::string& msg);
void set_store_j(gfh_ctx* c, bool on);
void set_store_res(gfh_ctx* c, bool on);
bool uses_fused_kernel(const gfh_ctx* c);
bool sweep_chi2_is_bitwise(const gfh_ctx* c);
bool omega_needs_jacobian(const gfh_ctx* c, int n_active);
int join_pending(gfh_ctx* c);     // waits for an upload started by gfh_set_data_begin; its result
// Named ranges for `rocprofv3 --marker-trace` (ROCTX): every pass, a fit and its iterations, an upload, the recovery from an unseen
// branch.  Off unless GADFIT_HIP_ROCTX=1 (the library is looked up at run time: no link dependency, no cost when off).
struct Range {
  explicit Range(const char* name);
  ~Range();
  Range(const Range&) = delete; Range& operator=(const Range&) = delete;
 private:
  bool on_;
};
}  // namespace gfh

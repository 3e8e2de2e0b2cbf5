/* gadfit_hip.h -- C ABI of libgadfit_hip.so, the MI355X hot path of gadfit's LM fit.
 *
 * The reference (raullaasner/gadfit v2.0.1) has no FFI seam for this path: STEP 1 / STEP 2
 * and chi2() are inline code of gadf_fit (fortran/gadfit/gadfit.F90:674-701, 1015-1034).
 * This header IS the seam: each entry point names the reference code it replaces.  Plain
 * pointers and sizes only; all host arrays remain caller-owned; every call returns 0 on
 * success and non-zero on error (message via gfh_last_error) -- the Fortran wrapper turns
 * that into `call error(__FILE__, __LINE__, msg)` (messaging.f90:32-41 convention).
 *
 * Threading: one host thread per context (the reference is single-threaded per image).
 * One context drives one GPU; N processes (one per GPU) form a communicator over RCCL
 * (replaces the coarray images + co_sum of misc.F90:133-170).
 */
#ifndef GADFIT_HIP_H
#define GADFIT_HIP_H
#include <stdint.h>
#include "gadfit_tape.h"

#ifdef __cplusplus
extern "C" {
#endif
/* libgadfit_hip.so is built with -fvisibility=hidden: what this header declares is the library's whole dynamic symbol table
 * (tests/test_cpu_cabi.py compares `nm -D` with it). */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

typedef struct gfh_ctx gfh_ctx;

/* ---- lifetime: replaces ad_init_reverse (AD:272-313) / gadf_close (gadfit.F90:1399-1412).
 * A destroyed context leaves its stream, events, pinned buffers and small device blocks (up to 4 MB each, 64 MB per device) to the
 * next context created on the same device in this process (a batch of small fits, gadf_init ... gadf_close each: 3.6 ms of runtime
 * calls per cycle otherwise); everything larger goes back to the runtime at once.  GADFIT_HIP_POOL=0 switches that off. */
int  gfh_create(int device, gfh_ctx** ctx);
/* The same, returning at once: the device part of the creation -- the HIP runtime's own start-up, 80 ms per process and 240 ms for the
 * first process on a machine, then stream, events, mailbox -- runs on a thread of the context while the caller goes on with host work
 * (gadf_init ... gadf_set, then the recording of eval() in the first gadf_fit: ad_init_reverse, AD:272-313, costs the reference nothing
 * comparable).  The first call that needs the device waits for it and reports its failure, if any ("no HIP device available ...");
 * gfh_set_data_begin queues its upload behind it without waiting.  device < 0 and GADFIT_HIP_ASYNC_INIT=0: plain gfh_create. */
int  gfh_create_begin(int device, gfh_ctx** ctx);
void gfh_destroy(gfh_ctx* ctx);
const char* gfh_last_error(const gfh_ctx* ctx);          /* ctx may be NULL: last global error */

/* ---- single-process device group: num_images() images of the reference as one context per GPU,
 * each driven by its own host thread, behind ONE handle (a plain Fortran program on a multi-GPU node,
 * no launcher).  n_devices <= 0: every visible device; devices == NULL: 0 .. n_devices-1.  The
 * handle takes every call of this header; a call runs on all members at once: gfh_set_data splits the
 * concatenated point array by gfh_partition (gadfit.F90:977-983), passes and whole fits (gfh_fit: the
 * LM loop replicated per member, as per image) sum J^T J / J^T r / chi2 / J^T omega over the members
 * -- co_sum, misc.F90:133-170 -- with RCCL all-reduces over xGMI (ncclCommInitAll; the default wherever every member has a
 * card of its own), or under GADFIT_HIP_GROUP_REDUCE=host (and for members that share a card) on the host in rank order from
 * the members' pinned result mailboxes (identical bits on every member).
 * Outputs are member 0's; gfh_get_residuals / _jacobian / _omega return the whole arrays.  Not available
 * on a group handle: gfh_comm_init, gfh_debug_set_rank, gfh_set_data_local, gfh_set_aux_local. */
int  gfh_create_group(int n_devices, const int* devices, gfh_ctx** ctx);
int  gfh_group_size(const gfh_ctx* ctx);                  /* members of a group handle; 1 for a plain context */
/* Test hook (CPU tests of the group's host sum: a group whose devices are all -1 has compile-only members): member r
 * sums bufs[r][0..n) over the members in rank order, in place; status[r] becomes the maximum over the members;
 * member fail_member (>= 0) fails before the barrier and the others must return an error too. */
int  gfh_debug_group_allreduce(gfh_ctx* ctx, double* bufs, int n, int* status, int fail_member);
/* Host cost of the group's machinery, measured by the library itself (no Python between the calls): out2[0] = microseconds per
 * sum of n doubles over the members inside ONE task, as the passes of a gfh_fit on a group handle pay it (`rounds` sums back to
 * back); out2[1] = microseconds per fan-out of an empty call to the member threads (what every call on the handle pays once).
 * Works on groups of compile-only members too (tools/probes/group_latency.py, profiles/r04_scaling_model.md). */
int  gfh_debug_group_latency(gfh_ctx* ctx, int n, int rounds, double* out2);
/* What ONE cross-rank sum (co_sum, misc.F90:133-170; call sites gadfit.F90:700-701, 735, 1032) of n doubles + the status slot costs on
 * this context's own path, measured by the library: ncclAllReduce on its RCCL communicator (one process per GPU after gfh_comm_init, or
 * the members of a device group under ncclCommInitAll), else the group's ordered host sum.  COLLECTIVE: every rank (or the group handle)
 * calls it with the same n and rounds.  out6[0..3] = median, 95th percentile, shortest, longest microseconds of the sum alone -- between
 * two HIP events on the otherwise idle stream of the context (host sum: host clock) --, out6[4] = median microseconds on the host clock
 * from enqueueing the sum to its numbers lying in the host mailbox (all-reduce + publish + flag: what a pass adds to its kernel),
 * out6[5] = ranks the communicator (or the group) counts.  A context with neither fails.  (bench.py `allreduce_us`.) */
int  gfh_debug_allreduce_latency(gfh_ctx* ctx, int n, int rounds, double* out6);
int  gfh_version(void);

/* ---- communicator: replaces num_images()/this_image() + co_sum (misc.F90:133-170).
 * id is a 128-byte opaque RCCL unique id created by rank 0 and distributed by the caller
 * (torch.distributed store, a file, ...).  Without gfh_comm_init the context is 1 image. */
#define GFH_UNIQUE_ID_BYTES 128
int  gfh_comm_unique_id(void* id);
int  gfh_comm_init(gfh_ctx* ctx, int nranks, int rank, const void* id);
/* Launcher-free variant for plain (Fortran) programs started once per GPU by a shell loop:
 * reads GADFIT_HIP_NRANKS / GADFIT_HIP_RANK and exchanges the id through the file named by
 * GADFIT_HIP_IDFILE (rank 0 writes it atomically, the others poll).  No-op without the variables. */
int  gfh_comm_init_from_env(gfh_ctx* ctx);
/* How this context's cross-rank sums travel: *rccl_nranks = ranks of its RCCL communicator as RCCL reports them
 * (ncclCommCount; 0 = none: one image, or a device group summing on the host); *n_allreduce = all-reduces issued since
 * gfh_reset_timers.  Either pointer may be NULL. */
int  gfh_comm_info(gfh_ctx* ctx, int* rccl_nranks, int64_t* n_allreduce);
/* Test hook, needs no GPU: the share of the points and the layout of the all-reduced image [JTJ | JTres | chi2] (dense, or
 * pattern-only for global fits) as rank `rank` of `nranks` derives them from (n_total, data_positions, Jacobian_indices, dim).
 * out[0] length of the image, out[1] pattern-only (0/1), out[2] nnz, out[3] hash of the pattern and index tables -- the same
 * on every rank, which ncclAllReduce silently requires -- out[4] first point, out[5] point count, out[6] datasets held,
 * out[7] gram workgroups of this rank.  nz_row / nz_col (may be NULL; nz_cap entries): the (row <= col) pattern in the order
 * of the pattern-only image's leading nnz values. */
int  gfh_debug_packed_layout(int nranks, int rank, int64_t n_total, int n_datasets, const int64_t* data_positions, int n_act,
                             const int32_t* jac_idx, int dim, int sparse_ok, int64_t* out, int32_t* nz_row, int32_t* nz_col, int nz_cap);
/* Test hook: give the context the geometry of rank `rank` of `nranks` WITHOUT a communicator, so the
 * sharding (gfh_partition, per-dataset sub-ranges, local layout) can be exercised on one GPU; the
 * caller sums the per-rank results itself.  Must precede gfh_set_data. */
int  gfh_debug_set_rank(gfh_ctx* ctx, int nranks, int rank);

/* ---- partition: re_initialize STEP 2 (gadfit.F90:977-983) with equal image weights:
 * int(N/G) points each, remainder +1 to the first ranks; contiguous in the concatenated
 * [dataset 1 | dataset 2 | ...] order. */
void gfh_partition(int64_t n_total, int nranks, int rank, int64_t* begin, int64_t* count);

/* ---- the Gauss-Kronrod tables (gauss_kronrod_parameters.F90; numerical_integration.F90:139-171): roots[points], wk[points]
 * (Kronrod weights), wg[points / 2] (Gauss weights, for the even 1-based positions); points in {15, 21, 31, 41, 51, 61}, else 1.
 * Host data only -- what the generated kernels carry, handed to the Fortran layer's integrate() for evaluations of eval() outside
 * gadf_fit (gadf_print, gadfit.F90:1255-1397). */
int  gfh_gk_rule(int points, double* roots, double* wg, double* wk);

/* ---- data: replaces read_data (gadfit.F90:401-443) + re_initialize's img_bounds
 * (977-1002).  x, y, w: the GLOBAL concatenated arrays (w = weights as used in
 * (y-f)*w, i.e. after init_weights, gadfit.F90:445-470); data_positions: n_datasets+1
 * 0-based offsets.  The context uploads only its own rank's range. */
int  gfh_set_data(gfh_ctx* ctx, int64_t n_total, const double* x, const double* y,
                  const double* w, int n_datasets, const int64_t* data_positions);
/* The same, returning at once: geometry and small tables are set by the call, the N-sized copies run on a thread of the library
 * while the caller does host work of its own (the Fortran layer records eval() over the data meanwhile).  The arrays must stay
 * valid and unchanged until the NEXT call on this context, which waits for the upload and reports its failure, if any. */
int  gfh_set_data_begin(gfh_ctx* ctx, int64_t n_total, const double* x, const double* y,
                        const double* w, int n_datasets, const int64_t* data_positions);
/* A host-to-host copy of `bytes` bytes that the NEXT gfh_set_data_begin starts on a thread of its own, beside the upload (the
 * Fortran layer's private copy of the abscissas -- gadfit.F90:82-88 keeps x_data for later fits and gadf_print -- which is not needed
 * before this fit returns).  Nothing on the device waits for it; gfh_wait_host_copy does, and src / dst must stay valid until then. */
int  gfh_queue_host_copy(gfh_ctx* ctx, void* dst, const void* src, int64_t bytes);
int  gfh_wait_host_copy(gfh_ctx* ctx);
/* Same, but the caller passes only this rank's slice [begin, begin+count) as given by
 * gfh_partition (avoids materialising 1e8-point arrays on every rank). */
int  gfh_set_data_local(gfh_ctx* ctx, int64_t n_total, int n_datasets,
                        const int64_t* data_positions, int64_t begin, int64_t count,
                        const double* x_local, const double* y_local, const double* w_local);
/* Auxiliary per-point real inputs of the model (GFH_AUX nodes of the tape, gadfit_tape.h): n_aux columns,
 * column k = aux[k*n_total .. k*n_total + n_total) in the GLOBAL point order of gfh_set_data (the _local
 * variant takes this rank's slice, column stride = its count).  This is how real(kp) arithmetic on the
 * abscissa inside a Fortran eval() -- invisible to operator overloading, fitfunction.F90:59-63 -- reaches
 * the device: the recorder tabulates such values once per data point.  Call after gfh_set_data; new data
 * drops the columns.  A model with n_aux > 0 refuses to run until they are set. */
int  gfh_set_aux(gfh_ctx* ctx, int n_aux, const double* aux);
int  gfh_set_aux_local(gfh_ctx* ctx, int n_aux, const double* aux_local);
/* init_weights on the device from y (and sigma for USER), gadfit.F90:445-470.
 * error_type: 0 NONE, 1 SQRT_Y, 2 PROPTO_Y, 3 INVERSE_Y, 4 USER (w currently holds sigma). */
int  gfh_init_weights(gfh_ctx* ctx, int error_type);

/* ---- model: replaces the dynamic dispatch to fitfunc%eval (fitfunction.F90:47, 59-63).
 * The tape is copied.  Kernels are generated per (tape, active set) on first use, compiled
 * with hiprtc for gfx950 and cached on disk (GADFIT_HIP_CACHE or <libdir>/kcache). */
int  gfh_set_model(gfh_ctx* ctx, const gfh_tape* tape);
/* A model whose eval() BRANCHES.  The reference's `ad` module exports `>` and `<` on advar (automatic_differentiation.F90:
 * 315-395, values only) and gadf_fit evaluates eval() afresh at every point (gadfit.F90:679-690), so a piecewise model
 * (`if (x < pars(2)) ...`, max(p1, p2 x), a clipped term) simply takes its branch per point and per parameter set.  Here every
 * path through eval() that a recording has taken is one tape -- a VARIANT -- whose comparisons are guard nodes carrying the
 * outcome on that path (gadfit_tape.h, GFH_GUARD_*).  The device walks the variants' common decision tree per data point at the
 * CURRENT parameters and runs the body of the variant it arrives at; a breakpoint that is an active parameter moves points from
 * one variant to another between iterations without the host being involved.
 *   tapes[0 .. n_variants)  independent, complete tapes (same n_pars and quadrature settings); copied.  Tapes that take the same
 *                           path through eval() and differ only inside an INTEGRAND (an integrand comparing AD variables,
 *                           recorded with the integration variable at several places of its range) are pooled: one variant
 *                           whose integrand picks its recording per evaluation on the device (gfh_model_n_tapes counts them all).
 *   hint_aux                -1, or the auxiliary column (gfh_set_aux) holding, per data point, the index of the variant the point
 *                           took when the columns were tabulated.  Needed only when two variants part ways WITHOUT a guard
 *                           (a Fortran eval() branching on the plain real x, which no recorder sees): gfh_model_needs_hint.
 * gfh_set_model(ctx, t) == gfh_set_model_variants(ctx, 1, &t, -1). */
int  gfh_set_model_variants(gfh_ctx* ctx, int n_variants, const gfh_tape* const* tapes, int hint_aux);
/* One per-point variant column PER SET OF OUTCOMES (round 5), for the NEXT gfh_set_model_variants on this context (which must hand
 * over n_tapes tapes and a hint_aux >= 0; otherwise the call has no effect): cols[t] = the auxiliary column that holds, per data
 * point, the index of the tape the point follows when the comparisons of AD variables met on tape t's path come out as on tape t.
 * With the comparisons given, what is left to decide is plain-real control flow -- a function of the abscissa alone, whatever the
 * parameters: such a column is tabulated once, and a point whose comparison flips during a fit (a fitted breakpoint moving across it)
 * finds its leaf behind a fork without the host.  At a fork the walk reads the column of any recording that passes there, at a leaf
 * the column of the leaf's own outcomes, which must name the leaf (else the point is reported, as with the single column).  Tapes that
 * hold the same outcomes share a column. */
int  gfh_set_variant_hint_columns(gfh_ctx* ctx, int n_tapes, const int32_t* cols);
int  gfh_model_needs_hint(gfh_ctx* ctx);       /* 1: the current variants fork without a comparison; 0: they do not; -1: no model */
int  gfh_model_n_variants(gfh_ctx* ctx);
/* recordings the current model was made of (variants of eval() + further recordings of integrands that compare AD variables, which are
 * pooled into their call sites and do not count as variants) */
int  gfh_model_n_tapes(gfh_ctx* ctx);
/* A data point may take a turn through eval() that no recorded variant covers (a guard comes out the other way for the first time:
 * the parameters have moved, or the recordings sampled the data).  The kernels then raise status 3 and report the points (up to 120
 * per pass); the library calls `fn` -- on the calling thread (device groups: on the member's thread, one call at a time) --
 * with those points, expects it to extend the model (gfh_set_model_variants on `target`, and gfh_set_aux if columns change) and
 * repeats the pass.  Without a handler, or when the handler changes nothing, the call fails with a message naming the point.
 *   index[k]     global index of the point in the concatenated data (gadfit.F90:82)      dataset[k]  its dataset (0-based)
 *   x[k]         its abscissa
 *   path[k], n_guards[k]   outcomes of the comparisons the device evaluated before it left the recorded tree (bit j = outcome of
 *                the j-th comparison met): a recording of eval(x[k]) that FORCES these outcomes and decides the comparisons
 *                after them naturally yields the missing variant even where host and device values differ in the last bit
 *   pars         [n_datasets][n_pars] parameters of the pass.
 * n_points = 0 (all arrays NULL but pars): an INTEGRAND met a path through its comparisons of AD variables that no recording of it has
 * (status 2): the handler records eval() over a sample of the data again at these parameters -- the integration variable at several
 * places of its range -- and hands the extended model over; returning non-zero, or a model that gained nothing, ends in the error. */
typedef int (*gfh_unseen_handler)(void* user, gfh_ctx* target, int n_points, const int64_t* index, const int32_t* dataset,
                                  const double* x, const uint64_t* path, const int32_t* n_guards, const double* pars);
int  gfh_set_unseen_handler(gfh_ctx* ctx, gfh_unseen_handler fn, void* user);
/* A hook called before EVERY pass (gfh_sweep / gfh_chi2 / gfh_omega / gfh_aux, also those inside gfh_fit and gfh_lm_iterate) with the
 * [n_datasets][n_pars] parameter block the pass is about to use; it may rewrite PASSIVE entries in place.  This is how a real that a
 * Fortran eval() forms from the %val of a fitted parameter follows the parameters: the reference recomputes it whenever eval() runs
 * (gadfit.F90:679-690), the recorder cannot see inside plain real arithmetic, so the layer declares one passive pseudo-parameter per
 * such real (the tape reads it as GFH_VAL(GFH_PARAM(n)), gadfit_tape.h) and sets it here by calling eval() once per dataset at the
 * parameters of the pass.  Non-zero return: the pass fails.  Members of a device group call it one at a time. */
typedef int (*gfh_pars_hook)(void* user, gfh_ctx* target, double* pars);
int  gfh_set_pars_hook(gfh_ctx* ctx, gfh_pars_hook fn, void* user);
/* out4[0] = passes repeated because a point left the recorded decision tree, out4[1] = passes of quadrature models that replayed the
 * recorded meshes of the pass before them instead of bisecting again (same parameters: the sweep of an accepted step after the trial
 * chi2() there, STEP 3 after the sweep; bitwise the same results; GADFIT_HIP_MESH=0 switches the hand-over off), out4[2] = variants
 * of the model, out4[3] = the outer quadrature workspace the kernels currently carry (intervals) in the high 32 bits, the inner one in the low 32. */
int  gfh_get_counters(gfh_ctx* ctx, int64_t* out4);
/* Work count of the last recording pass of a model with integrate(), read back from the device's own mesh records: out4[0] = adaptive
 * integrals (data point x outermost call site) with a record, out4[1] = the bisections they made in all -- an integral that ends on n
 * intervals made n - 1 bisections: 2n - 1 Gauss-Kronrod panels on values, then n with the gradient in the reference
 * (numerical_integration.F90:236-284) --, out4[2] = integrals without a record (more than 63 bisections), out4[3] = call sites per point.
 * bench.py's algorithmic floor of BASELINE config 4 is formed from this.  Needs a GPU and a pass that recorded (gfh_sweep / gfh_chi2). */
int  gfh_debug_mesh_stats(gfh_ctx* ctx, int64_t* out4);
/* Device memory as this context sees it: out3[0] = free and out3[1] = total bytes of its card (hipMemGetInfo), out3[2] = bytes of the
 * context's pool of quadrature workspaces -- the user-sized interval workspaces of integrate() (numerical_integration.F90:40-51,
 * 128-134: heap arrays there) live in ONE allocation of the context, a slot per wave of a launch, once they exceed 8 KB of scratch per
 * lane; it is cut at the first pass that needs it, an error code if the card cannot provide it, and freed by gfh_destroy.  A group
 * handle reports member 0's card and the sum of the members' pools. */
int  gfh_device_memory(gfh_ctx* ctx, int64_t* out3);
/* Generated HIP source for the current model and an active set (debug / AOT builds).
 * Returns bytes needed (including NUL); copies at most cap bytes. */
int64_t gfh_model_source(gfh_ctx* ctx, int n_act, const int32_t* active_pars, char* buf, int64_t cap);
/* Compile (or load from cache) without launching; usable without a GPU. */
int  gfh_model_prepare(gfh_ctx* ctx, int n_act, const int32_t* active_pars);

/* Selects the active set / column map ahead of the first sweep (gadfit.F90:586-631): loads
 * the kernels for it and sizes the device work arrays (re_initialize STEP 3, 1004-1011). */
int  gfh_set_active(gfh_ctx* ctx, const int32_t* active_pars, int n_act, const int32_t* jac_idx, int dim);

/* ---- STEP 1 + STEP 2 + co_sum (gadfit.F90:675-701): residuals, Jacobian, JTJ, JTres.
 * pars [n_datasets][n_pars]; active_pars: n_act 0-based parameter indices;
 * jac_idx [n_datasets][n_act] 0-based columns (Jacobian_indices, gadfit.F90:615-628);
 * JTJ dim*dim column-major (symmetric, both triangles filled), JTres dim, chi2 = sum res^2
 * at these parameters (free by-product).  Blocking; outputs are already summed over ranks.
 * J and res stay device-resident for gfh_omega / gfh_aux. */
int  gfh_sweep(gfh_ctx* ctx, const double* pars, const int32_t* active_pars, int n_act,
               const int32_t* jac_idx, int dim, double* JTJ, double* JTres, double* chi2);
/* ---- chi2() (gadfit.F90:1015-1034): all parameters passive; refreshes device res (except inside a gfh_fit under
 * keep_jacobian mode 2 whose options never read it, see gfh_set_keep_jacobian).  Same partition, thread-to-point map and
 * order of additions as the fused STEP 1+2 kernel: at the parameters of a sweep it returns bitwise that sweep's sum r^2. */
int  gfh_chi2(gfh_ctx* ctx, const double* pars, double* chi2);
/* ---- STEP 3 (gadfit.F90:715-735): omega_i = -f''_{delta1}(x_i) w_i in forward mode,
 * JTomega = J^T omega; delta1 length dim; pars = the parameters of the last gfh_sweep.  One kernel
 * (gfh_k_omega_jt) computes omega AND J^T omega, recomputing each point's Jacobian row in registers --
 * the stored J is not read (8*n_act B/point less traffic) and need not exist (gfh_set_keep_jacobian).
 * Models with integrate() and robust losses take the two-kernel path that reads the J of the last
 * gfh_sweep (env GADFIT_HIP_OMEGA_JT=0 forces it); both give bitwise the same J^T omega. */
int  gfh_omega(gfh_ctx* ctx, const double* pars, const double* delta1, double* JTomega);
/* ---- convergence reductions with the J of the last sweep and the res of the last
 * chi2/sweep: what=0: out[dim] = J^T res (gadfit.F90:849-850);
 * what=1: out[3] = {res.Jdelta, res.res, Jdelta.Jdelta} (gadfit.F90:865-874). */
int  gfh_aux(gfh_ctx* ctx, int what, const double* delta1, double* out);

/* ---- the LM driver gadf_fit (gadfit.F90:502-1035) on top of the calls above; the damped
 * solve (potr_f08, gadfit_linalg.F90:36-57) and the lambda logic run on the host. */
typedef struct gfh_fit_options {
  double lambda, lam_up, lam_down, accth, grad_chi2, cos_phi, rel_error, rel_error_global,
         chi2_rel, chi2_abs;
  int has_lambda, has_lam_up, has_lam_down, has_accth, has_grad_chi2, has_cos_phi,
      has_rel_error, has_rel_error_global, has_chi2_rel, has_chi2_abs;
  const double* DTD_min;   /* NULL = absent; length dim */
  int lam_incs, has_lam_incs;
  int uphill, has_uphill;
  int max_iter, has_max_iter;
  int damp_max, has_damp_max;
  int nielsen, has_nielsen;
  int umnigh, has_umnigh;
  int verbosity;           /* 0 silent, 1 one line per iteration on stdout */
  double umnigh_a;         /* in/out: the SAVEd local of gadfit.F90:515 */
} gfh_fit_options;

typedef struct gfh_fit_result {
  int iterations, dim, dof, exit_reason;
  double lambda, chi2;
  int n_sweeps, n_chi2, n_omega;   /* STEP 1+2 passes consumed, chi2 values requested, STEP 3 passes (the reference's counts) */
  int n_lookahead;                 /* how many of the n_chi2 trial values came from a look-ahead sweep */
  double seconds;          /* wall time of the main loop */
} gfh_fit_result;

/* Robust cost function of the C++ solver (c++/gadfit/lm_solver.h:76-83 `enum class Loss`,
 * lm_solver.cpp:255-284): STEP 1 scales the weighted residual and its Jacobian row by
 * sqrt(rho'(res^2)) (lm_solver.cpp:303-317); chi2() stays the plain sum (lm_solver.cpp:513-529).
 * The Fortran reference has no such option (= GFH_LOSS_LINEAR, the default). */
enum { GFH_LOSS_LINEAR = 0, GFH_LOSS_CAUCHY = 1, GFH_LOSS_HUBER = 2 };
int  gfh_set_loss(gfh_ctx* ctx, int loss);

/* Whether STEP 1 keeps the Jacobian in HBM.  The reference stores JacobianT because its STEP 2
 * is a separate matmul (gadfit.F90:689-698); the fused kernel forms J^T J / J^T r from registers,
 * so J is only read back by the grad_chi2 / cos_phi tests (849-850, 865-873), gfh_get_jacobian and --
 * for models with integrate() or a robust loss -- STEP 3 (J^T omega, gadfit.F90:734).  mode 1
 * (default): always written, as the reference.  mode 0: never (those calls then fail with a clear
 * message).  mode 2: gfh_fit writes it only when its options read it back, and likewise lets its chi2() passes skip the
 * residual store (res is read by the grad_chi2 / cos_phi tests and gfh_get_residuals only; a read-back of residuals that were
 * not kept fails with a clear message).  Without the store the sweep is bound by
 * the FP64 pipe instead of HBM writes and needs 8*n_act bytes per point less memory.
 * Env GADFIT_HIP_KEEP_J.  Results (J^T J, J^T r, chi2, res) are bitwise the same in all modes. */
int  gfh_set_keep_jacobian(gfh_ctx* ctx, int mode);

/* load_balancing of gadf_fit -- "adaptive parallelism" (gadfit.F90:672-673, re_initialize 935-983): with more than
 * one rank (processes with a communicator, or the members of a device group) gfh_fit re-cuts the contiguous ranges
 * before every iteration from the device time each rank spent in the parallel parts (STEP 1+2, chi2, STEP 3) since the
 * last cut, by the reference's weight update; it pays when the cost per point varies along the array (adaptive
 * quadrature).  Every image of the reference holds all data, so there a new cut is free; here the library keeps a host
 * copy of the arrays given to gfh_set_data / gfh_set_aux (made by those calls while the option is on) and a cut that moves a share
 * by more than 1 % re-uploads the rank's points.  The look-ahead schedule is off while it is on.
 * gfh_repartition: the cut for given image weights [nranks] (same on every rank); gfh_rebalance: one measurement +
 * weight update + cut (collective; *moved = 1 when the ranges changed). */
int  gfh_set_load_balancing(gfh_ctx* ctx, int on);
int  gfh_repartition(gfh_ctx* ctx, const double* weights);
int  gfh_rebalance(gfh_ctx* ctx, int* moved);
/* current ranges [begin, begin+count) of the members of a group handle ([gfh_group_size] entries; one entry for a plain context) */
int  gfh_group_ranges(gfh_ctx* ctx, int64_t* begins, int64_t* counts);

/* use_ad of gadf_fit (gadfit.F90:501, 583-584; default 1).  0: every parameter stays passive; STEP 1 takes the
 * gradient by the reference's forward differences (grad_finite, fitfunction.F90:155-174: step = sqrt(epsilon)*p
 * as (p+step)-p, one extra value evaluation per active parameter; gadfit.F90:686-687) and STEP 3 the second
 * directional derivative by its central difference (dir_deriv_2nd_finite, fitfunction.F90:188-203, h =
 * epsilon**(1/4); gadfit.F90:725-728).  Same kernels otherwise (fused sweep + Gram, chi2, omega); a parameter
 * whose step underflows raises the reference's error. */
int  gfh_set_use_ad(gfh_ctx* ctx, int on);
/* use_ad = 0 for a model whose per-point columns (gfh_set_aux) follow the PARAMETERS -- reals a host eval() forms from parameter
 * values in plain arithmetic, up to a whole black-box function (what use_ad=.false. exists for, gadfit.F90:583-584): the reference's
 * forward differences call eval() at p + step e_j (fitfunction.F90:155-174), where those reals have moved.  on = 1: gfh_set_aux holds
 * 1 + n_active SETS of the model's n_aux columns, [set][column][point]; set 0 belongs to the parameters of the pass, set 1 + j to
 * p + step e_j for the j-th active parameter of gfh_sweep's list, step = sqrt(epsilon) p as (p + step) - p per dataset; evaluation j of
 * the differences reads set 1 + j.  The host fills the sets from a gfh_pars_hook.  gfh_chi2 reads set 0; gfh_omega is refused (its
 * central difference would need sets at p +- h delta).  Default 0. */
int  gfh_set_fd_column_sets(gfh_ctx* ctx, int on);

/* Look-ahead schedule of gfh_fit / gfh_lm_iterate (default 1; env GADFIT_HIP_LOOKAHEAD).
 * The reference evaluates chi2() at the trial parameters (gadfit.F90:753) and, after accepting,
 * sweeps the same parameters again for the Jacobian (675-701).  The fused sweep kernel returns
 * sum r^2 with J^T J / J^T r, so with look-ahead the first trial of an iteration runs the sweep
 * instead of chi2() and an accepted step hands J^T J / J^T r to the next iteration: one N-sized
 * pass per accepted iteration instead of two, same numbers; the chi2() before the loop (gadfit.F90:670)
 * likewise is the sum r^2 of the first iteration's sweep.  Armed from the start; a rejected first trial (its sweep
 * is thrown away) disarms it until 4 iterations in a row have accepted their first trial (8, 16, ... 64 after each further rejection
 * while armed: a fit that keeps rejecting runs the reference's schedule); retrials after a
 * rejection use chi2().  The sweep's sum r^2 is bitwise what chi2() returns at those parameters, so both schedules
 * see the same chi2 values, take the same decisions and return the same bits.  Not used together with the grad_chi2 /
 * cos_phi tests, which read the device's (old J, new res) pair, nor with a robust loss (the sweep's sum is then the
 * robust one), nor where that bitwise identity does not hold: the two-kernel STEP 1+2 path (models with integrate(), more
 * than 80 active parameters, GADFIT_HIP_FUSED=0) and GADFIT_HIP_FAST_DIV=0.  0 = the reference's schedule. */
int  gfh_set_lookahead(gfh_ctx* ctx, int on);

/* pars [n_datasets][n_pars] in/out; is_global [n_pars]. */
int  gfh_fit(gfh_ctx* ctx, double* pars, int n_act, const int32_t* active_pars,
             const int32_t* is_global, gfh_fit_options* opt, gfh_fit_result* res);

/* n_iter iterations of the basic LM scheme without convergence exits (bench / profiling):
 * each = gfh_sweep + damped solve + parameter update + chi2 at the trial parameters (gfh_chi2,
 * or with look-ahead the sweep at the trial point, which an accepted step hands to the next
 * iteration of the same call) + accept (lambda /= 10) or reject (restore, lambda *= 10),
 * damp_max DTD as gadfit.F90:702-710.
 * state3 in/out = {lambda, old_chi2 (<0: evaluate first), accepted count}; DTD in/out [dim]. */
int  gfh_lm_iterate(gfh_ctx* ctx, double* pars, int n_act, const int32_t* active_pars,
                    const int32_t* is_global, int n_iter, double* state3, double* DTD);

/* ---- Jacobian_indices / dim (gadfit.F90:615-631) as a helper for callers */
int  gfh_jacobian_indices(int n_datasets, int n_act, const int32_t* active_pars,
                          const int32_t* is_global, int32_t* jac_idx);
/* The damped normal-equation solve (JTJ + lambda DTD) out = rhs of gadfit.F90:711-713 as gfh_fit performs
 * it on the host.  jac_idx [n_datasets][n_act] as above; DTD is the diagonal [dim].  With
 * use_structure != 0 and several datasets the block-arrow structure of a global fit is used: the local
 * parameters of different datasets do not couple, so their blocks are factorised one by one and only
 * the Schur complement of the global parameters is dense (the reference factorises the whole matrix
 * densely, doc/user_guide.tex:222-235) -- the same solution up to rounding at a fraction of the host
 * time (dim 259: 0.55 ms -> 20 us).  use_structure = 0: dense Cholesky (blocked above dim 64). */
int  gfh_solve_damped(int n_datasets, int n_act, const int32_t* jac_idx, int dim, const double* JTJ,
                      const double* DTD, double lambda, const double* rhs, double* out, int use_structure);
/* potr_f08 (gadfit_linalg.F90:36-57): a n*n column-major (destroyed), b rhs -> solution */
int  gfh_potr(int n, double* a, double* b);

/* ---- timers (Jacobian_timer, linalg_timer, chi2_timer, omega_timer; gadfit.F90:109-110),
 * device time from HIP events, seconds accumulated since creation or gfh_reset_timers.
 * out[8] = {sweep kernel, gram kernel, reduce+assemble, allreduce, chi2 kernel,
 *           omega kernel, n_sweep_launches, n_chi2_launches}
 * An event record costs ~4 us of stream time (two per launch: 8 us of the 35 us a small fit's LM iteration takes), so the
 * level is selectable: 0 = none; 1 (default) = events around every 8th launch of each model kernel (sweep[+gram], chi2,
 * omega; every launch under adaptive load balancing), the reported sums scaled to the number of launches; 2 = every launch,
 * and also reduce+assemble and all-reduce (env GADFIT_HIP_TIMERS).  gfh_get_timer_spread counts the launches that were timed. */
int  gfh_get_timers(gfh_ctx* ctx, double* out8);
/* Placement of the Jacobian buffer (no reference counterpart: gadfit.F90:632-640 allocates JacobianT once per image).  The sweep is
 * bound by `n_act` concurrent column streams into this buffer, and how fast the part absorbs them depends on the physical pages
 * behind the allocation (0.41 ... 0.47 ms for the same 2.6 GB buffer over a row of fresh allocations).  Once `after` sweeps
 * (gfh_set_placement_after, default 48, 0 = at the first sweep) have written a (re)allocated buffer of 256 MB or more -- the
 * search costs 25-150 ms at the headline size, as much as 50-300 sweeps, and gains 5-10 % per sweep: a job that has run that long
 * is taken to run on; one ten-iteration fit never pays for it --
 * it is allocated up to `tries` times (default 16 = the most, 1 = take the first; all held at once, never beyond half of the card's
 * memory), each candidate timed with four launches of the kernel that is about to run, until one runs on the fast side -- judged
 * against this card's own device-to-device copy rate, measured inside the first candidate, and (round 6) never below 6.3 TB/s of
 * algorithmic traffic: slow pages under that copy would lower the bar for themselves; four candidates within 1.5 % of each other end
 * the search (a kernel bound by its arithmetic) --; the fastest is kept.  Round 6: while the kernel is still on the slow side the
 * four DATA arrays x, y, w, res are then re-placed as a set the same way (allocated anew, copied device to device, timed; up to
 * `tries` sets, four in a row without a gain end it): they are 32 of the kernel's 32 + 8 p bytes per point, and their pages decide
 * as much (GADFIT_HIP_PLACE_DATA=0: not).  Models with integrate() are not placed (their sweeps are bound by the quadrature).
 * gfh_get_placement: out8[0] = kernel time (ms) on the buffer in use, out8[1..6] = on the candidates that were freed (0 = none /
 * no placement ran), out8[7] = the measured copy rate in GB/s. */
int  gfh_set_placement_tries(gfh_ctx* ctx, int tries);
int  gfh_set_placement_after(gfh_ctx* ctx, int sweeps);
int  gfh_get_placement(gfh_ctx* ctx, double* out8);
int  gfh_set_timer_detail(gfh_ctx* ctx, int level);
/* Spread of the STEP 1(+2) kernel's launches since gfh_reset_timers: out[4] = {shortest, longest, last
 * duration in seconds, launches counted}.  Under sustained back-to-back launches an MI355X slows the
 * HBM-write-bound kernel by ~20 % against its first launches (power management): the shortest duration
 * is the kernel on a cool chip, the average (gfh_get_timers) is what a long fit sees. */
int  gfh_get_timer_spread(gfh_ctx* ctx, double* out4);
void gfh_reset_timers(gfh_ctx* ctx);

/* ---- bench / profiling hooks: launch kernels without the host round trip.
 * gfh_launch_sweep: the AD sweep kernel only (STEP 1) at the parameters already uploaded by
 * the last gfh_sweep; gfh_launch_gram: the JTJ/JTres kernel chain only (STEP 2).
 * Asynchronous on the context's stream; gfh_sync waits.  gfh_stream returns the hipStream_t. */
int  gfh_launch_sweep(gfh_ctx* ctx);
int  gfh_launch_gram(gfh_ctx* ctx);
int  gfh_launch_chi2(gfh_ctx* ctx);
int  gfh_sync(gfh_ctx* ctx);
void* gfh_stream(gfh_ctx* ctx);
/* event-timed repetition: runs `reps` launches of kernel `which` (0 sweep, 1 gram, 2 chi2,
 * 3 omega, 4 plain sweep, 5 fused sweep+gram, 6 omega + J^T omega, 7 J^T v from the stored Jacobian) on the context stream
 * between two HIP events; returns average ms per launch. */
int  gfh_time_kernel(gfh_ctx* ctx, int which, int reps, double* avg_ms);

/* ---- debug read-back (tests): local un-padded residuals (count) and Jacobian
 * [count][n_act] row-major (= JacobianT(dim,N) restricted to the active columns). */
int  gfh_get_residuals(gfh_ctx* ctx, double* res_out);
int  gfh_get_jacobian(gfh_ctx* ctx, double* jac_out);
/* The same for single points: index[k] = local index into this rank's range (gfh_local_begin / gfh_local_count); res_out[n] and/or
 * jac_out[n][n_act] (either may be NULL).  For checks at sizes where the whole Jacobian does not belong on the host. */
int  gfh_get_points(gfh_ctx* ctx, int n, const int64_t* index, double* res_out, double* jac_out);
/* Text data files of gadf_add_dataset(path) (gadfit.F90:212-215 counts the records that begin with a number with one list-directed read
 * each, gadfit.F90:422-437 then reads the first two or three numbers of every record; records that do not begin with a number are
 * skipped, blank ones are transparent).  gfh_read_columns parses the whole file once -- mapped, cut at line ends, a thread per piece;
 * Fortran's D/Q exponent letters, commas and r*c repeat counts understood -- and returns the number of data records; gfh_take_columns
 * copies the columns into the caller's arrays (w may be NULL; it is only written for n_columns = 3) and frees the handle.  A record
 * that begins with a number but holds fewer than n_columns values is an error (the reference would mis-align silently).  Host code:
 * needs no GPU and no context; errors through gfh_last_error(NULL).  GADFIT_HIP_READ_THREADS caps the threads. */
typedef struct gfh_columns gfh_columns;
int  gfh_read_columns(const char* path, int n_columns, gfh_columns** out, int64_t* n_points);
int  gfh_take_columns(gfh_columns* cols, double* x, double* y, double* w);
void gfh_free_columns(gfh_columns* cols);
/* The abscissas as uploaded, back into x_out[n_total] (the concatenated array of gfh_set_data): this rank's range; on a device-group
 * handle every member's, i.e. all of it.  (gadfit.F90:82-88 keeps x_data on the host; a host layer that prefers not to can read
 * the abscissas back from here.) */
int  gfh_get_abscissas(gfh_ctx* ctx, double* x_out);
int  gfh_get_omega(gfh_ctx* ctx, double* omega_out);
int  gfh_get_weights(gfh_ctx* ctx, double* w_out);                /* [local_count] the weights as the kernels use them (after gfh_init_weights, gadfit.F90:445-470) */
int64_t gfh_local_count(gfh_ctx* ctx);
int64_t gfh_local_begin(gfh_ctx* ctx);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif

/* ad_tls.c -- per-thread state of the Fortran recorder's checking mode (ad.F90: ad_thread_check; gadfit.F90: discover).
 *
 * gadf_fit records eval() over up to 2^17 abscissas of the data to learn (and verify) what the model is; once the paths
 * through eval() are known, each further recording only has to AGREE with one of them, node by node as it is made, and
 * those checks are independent of each other: they run on several OpenMP threads.  What a check writes per node -- a
 * node counter, two flags -- must then be per thread.  A Fortran module variable is shared by all threads, and flang
 * reaches an OpenMP threadprivate one through a call into the OpenMP runtime at every access (measured: 30 x slower than a
 * plain variable), so this state lives here in native thread-local storage (initial-exec model: one %fs-relative access),
 * linked statically into libgadfit_f.a.  The known recording is shared and read-only while threads run.  Host code only.
 */
#define _GNU_SOURCE
#include <math.h>
#include <sched.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

enum { GFH_CONST_OP = 0 };   /* enum gfh_op GFH_CONST (include/gadfit_tape.h) */

/* up to GFH_ADCHK_PATHS known recordings at once (the paths of an eval() that branches on the plain real x); slot 0 is what
 * gfh_adchk_load fills; a thread picks the one its next recording is compared with (gfh_adchk_use; default 0) */
enum { GFH_ADCHK_PATHS = 16 };
/* ... with, for a path that calls integrate() (gfh_adchk_load_ints; absent: nsub = 0): the sub-tape of every node (0 = eval(),
 * 1.. = integrands in the order their recordings begin), the nodes bound to the integrands' pars(:), the call sites and the result
 * node of every sub-tape */
/* what a node is compared with, packed into 16 bytes (one load): operation | flags << 8 | sub-tape << 16, the two operands, the
 * literal class -- built by the load calls from the caller's arrays */
typedef struct { int32_t opfs, a, b, cls; } nodekey_t;
typedef struct {
  const int32_t *op, *a, *b, *fl, *cls; const double *c, *alpha, *beta; int n;
  nodekey_t* key; int key_cap;
  int nsub, nint, nip;
  const int32_t *sub, *ipar, *i_integrand, *i_lower, *i_upper, *i_linf, *i_uinf, *i_nip, *sub_result;
  const double *i_rel, *i_abs;
} known_t;
static known_t g_paths[GFH_ADCHK_PATHS];
static void pack_keys(known_t* g) {
  if (g->key_cap < g->n) { free(g->key); g->key = (nodekey_t*)malloc(sizeof(nodekey_t) * (size_t)(g->n > 0 ? g->n : 1)); g->key_cap = g->key ? g->n : 0; }
  if (!g->key) { g->n = 0; return; }          /* (no memory: nothing checks out, every point takes the serial path) */
  for (int j = 0; j < g->n; j++) {
    g->key[j].opfs = (g->op[j] & 0xff) | ((g->fl[j] & 0xff) << 8) | ((g->sub ? g->sub[j] : 0) << 16);
    g->key[j].a = g->a[j]; g->key[j].b = g->b[j]; g->key[j].cls = g->cls[j];
  }
}
static __thread int t_use;
#define g_known (g_paths[t_use])
enum { GFH_ADCHK_AUX_MAX = 64 };
enum { GFH_ADCHK_MAX_SUB = 16 };        /* AD_MAX_SUB of module ad */
static __thread struct { int n, diverged, litfail, n_aux; double x; double aux[GFH_ADCHK_AUX_MAX]; int auxk[GFH_ADCHK_AUX_MAX];
                         int cur, nsub, depth, n_int, n_ipar, sub_n[GFH_ADCHK_MAX_SUB + 1], parent[4]; } t_chk;
/* outcomes forced on the first comparisons of this thread's recordings (bit k = outcome of the k-th comparison met), as the
 * recorder's ad_set_script does for the serial recordings; comparisons met so far in the current recording */
static __thread struct { int n, count; uint64_t bits; } t_script;

/* the known recording: arrays of n nodes, kept alive and unchanged by the caller while threads run
 * (cls: 1 = constant literal c, 2 = affine literal alpha x + beta, 3 = per-point input, other = anything) */
void gfh_adchk_load_path(int k, int n, const int32_t* op, const int32_t* a, const int32_t* b, const int32_t* fl, const int32_t* cls,
                         const double* c, const double* alpha, const double* beta) {
  if (k < 0 || k >= GFH_ADCHK_PATHS) return;
  known_t* g = &g_paths[k];
  g->op = op; g->a = a; g->b = b; g->fl = fl; g->cls = cls; g->c = c; g->alpha = alpha; g->beta = beta; g->n = n;
  g->nsub = g->nint = g->nip = 0; g->sub = 0;
  pack_keys(g);
}
/* known recording k calls integrate(): see known_t (arrays kept alive and unchanged by the caller while threads run) */
void gfh_adchk_load_ints(int k, int nsub, int nint, int nip, const int32_t* sub, const int32_t* ipar, const int32_t* sub_result,
                         const int32_t* i_integrand, const int32_t* i_lower, const int32_t* i_upper, const int32_t* i_linf,
                         const int32_t* i_uinf, const int32_t* i_nip, const double* i_rel, const double* i_abs) {
  if (k < 0 || k >= GFH_ADCHK_PATHS) return;
  known_t* g = &g_paths[k];
  g->nsub = nsub; g->nint = nint; g->nip = nip; g->sub = sub; g->ipar = ipar; g->sub_result = sub_result;
  g->i_integrand = i_integrand; g->i_lower = i_lower; g->i_upper = i_upper; g->i_linf = i_linf; g->i_uinf = i_uinf; g->i_nip = i_nip;
  g->i_rel = i_rel; g->i_abs = i_abs;
  pack_keys(g);
}
void gfh_adchk_load(int n, const int32_t* op, const int32_t* a, const int32_t* b, const int32_t* fl, const int32_t* cls,
                    const double* c, const double* alpha, const double* beta) {
  gfh_adchk_load_path(0, n, op, a, b, fl, cls, c, alpha, beta);
}
/* the calling thread's next recordings take the first n comparisons as prescribed (n = 0: as their values decide) */
void gfh_adchk_script(int n, uint64_t bits) { t_script.n = n < 0 ? 0 : (n > 64 ? 64 : n); t_script.bits = bits; }
/* one comparison met while recording on this thread: the outcome the recording continues with */
int gfh_adchk_guard(int natural) {
  const int k = t_script.count++;
  return k < t_script.n ? (int)((t_script.bits >> k) & 1u) : (natural != 0);
}
/* the calling thread's recordings are compared with known recording k from now on */
void gfh_adchk_use(int k) { t_use = k >= 0 && k < GFH_ADCHK_PATHS ? k : 0; }

/* a recording at abscissa x begins; its first n_params nodes (the parameters) are what the known recording begins with */
void gfh_adchk_begin(double x, int n_params) {
  t_chk.n = n_params; t_chk.diverged = n_params > g_known.n; t_chk.litfail = 0; t_chk.x = x; t_chk.n_aux = 0; t_script.count = 0;
  t_chk.cur = 0; t_chk.nsub = 0; t_chk.depth = 0; t_chk.n_int = 0; t_chk.n_ipar = 0; t_chk.sub_n[0] = n_params;
}
/* nesting depth of integrate() the recording is in (0: eval() itself) */
int gfh_adchk_depth(void) { return t_chk.depth; }
/* integrate() is being recorded: the nodes bound to its pars(:), in the enclosing sub-tape */
void gfh_adchk_ipar(int n, const int32_t* nodes) {
  if (!t_chk.diverged) {
    if (t_chk.n_ipar + n > g_known.nip) t_chk.diverged = 1;
    else for (int j = 0; j < n; j++) if (nodes[j] != g_known.ipar[t_chk.n_ipar + j]) { t_chk.diverged = 1; break; }
  }
  t_chk.n_ipar += n;
}
/* ... its integrand's recording begins: returns the new sub-tape */
int gfh_adchk_sub_enter(void) {
  if (t_chk.nsub >= GFH_ADCHK_MAX_SUB || t_chk.depth >= 3) { t_chk.diverged = 1; return t_chk.nsub; }
  t_chk.parent[t_chk.depth] = t_chk.cur;
  t_chk.nsub++; t_chk.cur = t_chk.nsub; t_chk.depth++; t_chk.sub_n[t_chk.cur] = 0;
  if (t_chk.nsub > g_known.nsub) t_chk.diverged = 1;
  return t_chk.cur;
}
/* ... and ends with this result node */
void gfh_adchk_sub_leave(int result) {
  if (t_chk.depth <= 0) { t_chk.diverged = 1; return; }
  if (!t_chk.diverged && g_known.sub_result[t_chk.cur] != result) t_chk.diverged = 1;
  t_chk.depth--; t_chk.cur = t_chk.parent[t_chk.depth];
}
/* ... the call site itself: returns its 1-based index */
int gfh_adchk_integral(int integrand, int lower, int upper, int linf, int uinf, int nip, double rel, double abs_) {
  const int i = t_chk.n_int++;
  if (!t_chk.diverged) {
    if (i >= g_known.nint || g_known.i_integrand[i] != integrand || g_known.i_lower[i] != lower || g_known.i_upper[i] != upper ||
        g_known.i_linf[i] != linf || g_known.i_uinf[i] != uinf || g_known.i_nip[i] != nip || g_known.i_rel[i] != rel || g_known.i_abs[i] != abs_)
      t_chk.diverged = 1;
  }
  return i + 1;
}

/* one node; returns its index in the sub-tape it belongs to (eval()'s tape: its position in the recording) */
static inline int emit(int op, int a, int b, int flags, double c);
int gfh_adchk_emit(int op, int a, int b, int flags, double c) { return emit(op, a, b, flags, c); }
/* The fast forms (module ad, ad_fast_check): what an elemental emits, in one call.  op2: the operation on two nodes (b = -1: a unary
 * one; GFH_POWI: b is the exponent).  op_lit: a real operand -- its literal node, then the operation on it and node a (lit_first: the
 * literal is the first operand).  lift: a real assigned to an AD variable -- its literal node, then GFH_LIFT of it. */
enum { GFH_LIFT_OP = 3, GFH_F_REAL_FLAG = 1 };
int gfh_adchk_op2(int op, int a, int b) { return emit(op, a, b, 0, 0.0); }
int gfh_adchk_op_lit(int op, int a, double r, int lit_first) {
  const int kc = emit(GFH_CONST_OP, -1, -1, GFH_F_REAL_FLAG, r);
  return lit_first ? emit(op, kc, a, 0, 0.0) : emit(op, a, kc, 0, 0.0);
}
int gfh_adchk_lift(double r) { return emit(GFH_LIFT_OP, emit(GFH_CONST_OP, -1, -1, GFH_F_REAL_FLAG, r), -1, 0, 0.0); }
static inline int emit(int op, int a, int b, int flags, double c) {
  const int j = t_chk.n++;
  const int k = t_chk.sub_n[t_chk.cur]++;
  if (t_chk.diverged) return k;
  const known_t* g = &g_known;
  if (j >= g->n) { t_chk.diverged = 1; return k; }
  const nodekey_t key = g->key[j];
  if (key.opfs != ((op & 0xff) | ((flags & 0xff) << 8) | (t_chk.cur << 16)) || key.a != a || key.b != b || (op | flags) > 0xff) { t_chk.diverged = 1; return k; }
  if (op == GFH_CONST_OP) {
    const int cls = key.cls;
    if (cls == 1) { if (c != g->c[j] && !(c != c && g->c[j] != g->c[j])) t_chk.litfail = 1; }
    else if (cls == 2) {
      const double al = g->alpha[j], be = g->beta[j], want = al * t_chk.x + be;
      if (!(fabs(want - c) <= 1e-11 * (fabs(c) + fabs(al * t_chk.x) + fabs(be)))) t_chk.litfail = 1;
    }
    else if (cls == 3) {      /* a per-point input (auxiliary column): its value at this abscissa is what the tabulation wants */
      if (t_chk.n_aux < GFH_ADCHK_AUX_MAX) { t_chk.aux[t_chk.n_aux] = c; t_chk.auxk[t_chk.n_aux] = j; }
      t_chk.n_aux++;
    }
  }
  return k;
}

/* the class-3 literals the recording met, in node order: values and 0-based node indices (cap entries at most); returns how many
 * there were */
int gfh_adchk_aux(int cap, double* vals, int32_t* nodes) {
  const int n = t_chk.n_aux < GFH_ADCHK_AUX_MAX ? t_chk.n_aux : GFH_ADCHK_AUX_MAX;
  for (int k = 0; k < n && k < cap; k++) { vals[k] = t_chk.aux[k]; nodes[k] = t_chk.auxk[k]; }
  return t_chk.n_aux;
}

/* nodes emitted; whether the operations differed; whether a literal was not what it was taken for */
void gfh_adchk_end(int* n, int* diverged, int* litfail) {
  if (t_chk.nsub != g_known.nsub || t_chk.n_int != g_known.nint || t_chk.n_ipar != g_known.nip || t_chk.depth != 0) t_chk.diverged = 1;
  *n = t_chk.n; *diverged = t_chk.diverged; *litfail = t_chk.litfail;
}

/* Host CPUs this process may really keep busy: its affinity mask, cut down to the cgroup's CPU quota where there is one (a GPU box
 * hands a job the share of its card -- 16 of 256 hardware threads on the pool this was measured on -- and a process that runs more
 * busy threads than its quota is THROTTLED as a whole for the rest of the scheduler period: with 16 recorder threads + the upload
 * thread + the runtime's own, the capture of 1e7 points took 250 or 430 ms from run to run, with 14 it takes 270-310,
 * profiles/r05_record_threads.txt).  gadfit.F90 sizes its recorder threads by this less two. */
int gfh_host_cpu_budget(void) {
  int cpus = 0;
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof set, &set) == 0) cpus = CPU_COUNT(&set);
  if (cpus < 1) cpus = 1;
  double quota = 0.0;
  FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r");
  if (f) {
    char q[64]; double per = 0.0;
    if (fscanf(f, "%63s %lf", q, &per) == 2 && q[0] != 'm' && per > 0.0) quota = atof(q) / per;
    fclose(f);
  } else {
    double q = 0.0, per = 0.0;
    f = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r");
    if (f) { if (fscanf(f, "%lf", &q) != 1) q = 0.0; fclose(f); }
    f = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r");
    if (f) { if (fscanf(f, "%lf", &per) != 1) per = 0.0; fclose(f); }
    if (q > 0.0 && per > 0.0) quota = q / per;
  }
  if (quota > 0.0 && quota + 0.5 < cpus) cpus = (int)(quota + 0.5);
  return cpus < 1 ? 1 : cpus;
}

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
#include <math.h>
#include <stdint.h>

enum { GFH_CONST_OP = 0 };   /* enum gfh_op GFH_CONST (include/gadfit_tape.h) */

/* up to GFH_ADCHK_PATHS known recordings at once (the paths of an eval() that branches on the plain real x); slot 0 is what
 * gfh_adchk_load fills; a thread picks the one its next recording is compared with (gfh_adchk_use; default 0) */
enum { GFH_ADCHK_PATHS = 16 };
typedef struct { const int32_t *op, *a, *b, *fl, *cls; const double *c, *alpha, *beta; int n; } known_t;
static known_t g_paths[GFH_ADCHK_PATHS];
static __thread int t_use;
#define g_known (g_paths[t_use])
enum { GFH_ADCHK_AUX_MAX = 64 };
static __thread struct { int n, diverged, litfail, n_aux; double x; double aux[GFH_ADCHK_AUX_MAX]; int auxk[GFH_ADCHK_AUX_MAX]; } t_chk;
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
}

/* one node; returns its index in eval()'s tape */
int gfh_adchk_emit(int op, int a, int b, int flags, double c) {
  const int j = t_chk.n++;
  if (t_chk.diverged) return j;
  if (j >= g_known.n || op != g_known.op[j] || a != g_known.a[j] || b != g_known.b[j] || flags != g_known.fl[j]) { t_chk.diverged = 1; return j; }
  if (op == GFH_CONST_OP) {
    const int cls = g_known.cls[j];
    if (cls == 1) { if (c != g_known.c[j] && !(c != c && g_known.c[j] != g_known.c[j])) t_chk.litfail = 1; }
    else if (cls == 2) {
      const double al = g_known.alpha[j], be = g_known.beta[j], want = al * t_chk.x + be;
      if (!(fabs(want - c) <= 1e-11 * (fabs(c) + fabs(al * t_chk.x) + fabs(be)))) t_chk.litfail = 1;
    }
    else if (cls == 3) {      /* a per-point input (auxiliary column): its value at this abscissa is what the tabulation wants */
      if (t_chk.n_aux < GFH_ADCHK_AUX_MAX) { t_chk.aux[t_chk.n_aux] = c; t_chk.auxk[t_chk.n_aux] = j; }
      t_chk.n_aux++;
    }
  }
  return j;
}

/* the class-3 literals the recording met, in node order: values and 0-based node indices (cap entries at most); returns how many
 * there were */
int gfh_adchk_aux(int cap, double* vals, int32_t* nodes) {
  const int n = t_chk.n_aux < GFH_ADCHK_AUX_MAX ? t_chk.n_aux : GFH_ADCHK_AUX_MAX;
  for (int k = 0; k < n && k < cap; k++) { vals[k] = t_chk.aux[k]; nodes[k] = t_chk.auxk[k]; }
  return t_chk.n_aux;
}

/* nodes emitted; whether the operations differed; whether a literal was not what it was taken for */
void gfh_adchk_end(int* n, int* diverged, int* litfail) { *n = t_chk.n; *diverged = t_chk.diverged; *litfail = t_chk.litfail; }

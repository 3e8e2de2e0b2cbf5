// model.h -- host-side copy of a gfh_tape plus the device code generator interface.
#pragma once
#include <algorithm>
#include <cstdint>
#include <string>
#include <vector>
#include "../../include/gadfit_tape.h"

namespace gfh {

struct Node { int32_t op, a, b, flags; double c; };
struct SubTape { std::vector<Node> nodes; int32_t result; };
struct Integral {
  int32_t integrand, lower, upper, lower_inf, upper_inf, n_ipars, ipar_off, depth;
  double rel_error, abs_error;
};

// A model is one or more recorded paths ("variants") through the user's eval() (gadfit_tape.h, guard nodes).  sub[0] is eval() of
// variant 0, sub[1..] the integrand sub-tapes of ALL variants (pooled; identical ones are shared), more_evals[v-1] eval() of variant v.
struct Model {
  int32_t n_pars = 0;
  std::vector<SubTape> sub;
  std::vector<SubTape> more_evals;
  std::vector<Integral> integrals;
  std::vector<int32_t> ipar_nodes;
  // alts[I]: further recordings (sub-tape indices) of the integrand of call site I -- an integrand that compares AD variables takes
  // its branch anew at every abscissa of the quadrature (AD:315-395), a recording follows one path through it; the recordings of a
  // call site that differ only in that path are pooled here and the device picks per evaluation (codegen.cpp, emit_family)
  std::vector<std::vector<int32_t>> alts;
  int32_t n_tapes = 1;          // recordings handed over (gfh_set_model_variants): variants of eval() + further recordings of integrands
  int32_t gk_points = 15;
  int32_t n_aux = 0;            // auxiliary per-point columns read by eval() (GFH_AUX)
  int32_t hint_aux = -1;        // the auxiliary column that names, per data point, the variant it took when the columns were tabulated
                                // (needed only where variants part ways WITHOUT a comparison: plain-real control flow on x)
  // Round 5: one such column PER SET OF OUTCOMES.  hint_cols[t] (per tape; empty: hint_aux for all): the column that holds, per data
  // point, the tape the point follows when the comparisons of AD variables on tape t's own path come out as on tape t -- a function of
  // the abscissa alone (the comparisons are given, only plain-real control flow is left), so it is tabulated once and a point that
  // changes sides at a comparison finds its leaf behind a fork without the host (gfh_set_variant_hint_columns, codegen.cpp emit_selector).
  std::vector<int32_t> hint_cols;
  std::vector<int32_t> tape_variant;      // tape t of the hand-over -> the variant of eval() it became (pooled integrand recordings join an earlier one)
  int32_t ws_size = 1000, ws_size_inner = 1000;   // quadrature workspaces the user asked for (NI:40, 128-134)
  double rel_error_outer = 0, rel_error_inner = 0;

  // copies and validates; returns false and sets err on malformed tapes
  bool load(const gfh_tape* t, std::string* err);
  bool load_variants(int n, const gfh_tape* const* t, int hint_aux, std::string* err, const std::vector<int32_t>* hint_cols = nullptr);
  int hint_col_of_variant(int v) const;      // the column a walk reads for variant v (the first tape that became v)
  std::vector<int> tapes_of_variant(int v) const;
  bool has_integrals() const { return !integrals.empty(); }
  int n_variants() const { return 1 + (int)more_evals.size(); }
  const SubTape& eval(int v) const { return v == 0 ? sub[0] : more_evals[(size_t)v - 1]; }
  bool has_guards() const;
  bool branching() const { return n_variants() > 1 || has_guards(); }
  // do the variants part ways somewhere without a comparison (so that only a per-point column can tell them apart)?
  bool needs_hint() const;
};

inline bool is_guard_op(int op) { return op == GFH_GUARD_GT || op == GFH_GUARD_LT; }

// Options of the generated kernels (kept in the source text so the cache key sees them).
struct GenConfig {
  int block = 256;        // threads per workgroup of the plain sweep / chi2 / omega kernels = slots per tile
  bool fd_col_sets = false; // ... with the auxiliary columns in 1 + n_active sets (gfh_set_fd_column_sets): evaluation j of the forward differences reads set 1 + j
  bool finite_diff = false; // use_ad = .false.: gradient / second directional derivative by the reference's finite differences (fitfunction.F90:155-203)
  bool omega_jt = true;   // STEP 3 kernel that recomputes the Jacobian row instead of reading J (gfh_k_omega_jt)
  int kernarg_pars = 0;   // > 0: the parameter block (this many doubles) is a by-value kernel argument
  int ws_size = 100;      // per-lane quadrature workspace (intervals) per nesting level that is compiled in: the fast form carries 100; a
                          // pass that exhausts it is repeated with the user's size (Model::ws_size, reference default 1000) before the reference's error is raised
  int ws_size_inner = 100;
  bool ws_global = false; // the interval workspaces live in a pool in GLOBAL memory, [wave slot][level][interval][lo|hi|err|sum][64 lanes] (a
                          // wave's access to one field of one interval is one coalesced 512 B row), handed to the kernels as an argument, instead
                          // of per-lane scratch: every workspace too large for kScratchBudget bytes of scratch per lane (plan_workspaces)
  bool store_j = true;    // fused kernel writes the Jacobian to HBM (gfh_set_keep_jacobian)
  bool store_res = true;  // chi2 kernel writes the residual vector (the reference's chi2() side effect, gadfit.F90:1024-1026)
  int loss = 0;           // robust cost (gfh_set_loss): 0 linear, 1 cauchy, 2 huber
  bool fast_div = true;   // share one reciprocal per denominator (<= 1 ulp from the reference's r/v)
  int ablate = 0;         // TIMING EXPERIMENTS ONLY (GADFIT_HIP_ABLATE, wrong results): fused matrix path without 1 its stage writes, 2 its fragment reads, 4 its matrix instructions, 8 the AD body
  int matrix_prio = -3;   // fused kernel without the Jacobian store: s_setprio of a wave while it is in its matrix phase (> 0) or in its AD phase (< 0: priority
                          // -matrix_prio there, 0 in the matrix phase).  The two waves of a SIMD share one FP64 pipe; the wave in its AD phase issues short
                          // instructions between the other wave's 64- and 17-cycle matrix instructions when it goes first: 0.327 -> 0.314 ms at the headline
                          // size (profiles/r04_nostore.md); the stored form (waves of a workgroup in phase, store-bound) does not move and is left alone
  int frag_ahead = 1;     // fused kernel, matrix phase: the LDS fragment reads of k-step s + frag_ahead are issued before the matrix instructions of step s
  int waves_per_eu = 0;   // > 0: the plain sweep / chi2 / omega kernels are compiled for at least this many waves per SIMD (register cap)
  int half_stage = -1;    // fused kernel, matrix path: the wave's LDS stage holds 32 points instead of 64 and a pass feeds the matrix cores in two
                          // half-passes (lanes 0-31, then lanes 32-63: the k-steps in their old order, bitwise the same sums).  Half the LDS
                          // per wave = more waves per SIMD.  -1: where it pays (fused_half_stage); 0 / 1 force (GADFIT_HIP_HALF_STAGE)
  int fused_waves = 0;    // > 0: cap on the waves per workgroup of the fused kernel and gfh_k_chi2 (GADFIT_HIP_FUSED_WAVES; experiments)
  int single_image = 0;   // > 0: the single cross-wave reduction image also below 5 tiles (GADFIT_HIP_SINGLE_IMAGE; experiments)
  int frag_late = 1;      // fused kernel, matrix phase: the fragment reads of the next k-step are issued behind the 16x16x4 matrix instructions of this one
                          // instead of in front of them: the wave's LDS instructions then take no issue slots from the FP64 pipe its SIMD's waves share
                          // (round 5: no-store 0.3135 -> 0.299 ms, stored 0.209 -> 0.178 ms at N = 4e6; bitwise; GADFIT_HIP_FRAG_LATE=0: the old order)
  int fused_wpe = 0;      // > 0: the fused kernel is compiled for this many waves per SIMD (register cap; GADFIT_HIP_FUSED_WPE)
  int coop = 1;           // fused kernel from 6 tiles (81 active parameters) on: the workgroup-cooperative Gram (GADFIT_HIP_COOP; 0: off -- two kernels beyond 80; 2: from 5 tiles on)
  int valu_ahead = 0;     // fused kernel, VALU form (<= 8 active parameters): passes its x, y, w loads run ahead: 2 = two (GADFIT_HIP_VALU_AHEAD; experiment, see valu_ahead_for), else one
};

// Where the quadrature workspaces of a translation unit live (numerical_integration.F90:40-51, 128-134: the reference's are heap arrays
// of the user's size).  Up to kScratchBudget bytes per lane they are private scratch (the fast form: the user's sizes capped at `fast`
// intervals and at what the budget holds once the per-interval gradients the bisection carries are counted); anything larger is the
// global pool (GenConfig::ws_global), which the context allocates with hipMalloc and frees with itself -- private scratch of tens of
// KB per lane is reserved by the runtime for the whole device until the process ends, and two queues asking for it at once can end
// the process (HSA_STATUS_ERROR_OUT_OF_RESOURCES).  grown: a pass has exhausted the fast form, carry the user's sizes.
constexpr int kScratchBudget = 7936;      // 8 KB per lane less 256 B for the register spills of the bodies
struct WsPlan { int ws_size, ws_size_inner; bool global; };
WsPlan plan_workspaces(const Model& m, int fast, bool grown);
bool carries_gradients(int n_ipars, int ws, bool global);      // does the bisection of a call site keep its panels' gradients per interval?
// The global pool's row of one interval at nesting level 1 or 2: [lo|hi|err|sum] and, where the level's call sites carry their panels'
// gradients (round 5: the pool form as well -- up to kWsgCarryMax integrand parameters), one more field per parameter; x 64 lanes.
constexpr int kWsgCarryMax = 8;
int wsg_row_fields(const Model& m, int level);
inline long wsg_wave_doubles(const Model& m, int ws1, int ws2) {
  bool nested = false; for (const Integral& in : m.integrals) if (in.depth >= 2) nested = true;
  return 64L * wsg_row_fields(m, 1) * ws1 + (nested ? 64L * wsg_row_fields(m, 2) * ws2 : 0L);
}
inline bool nested_integrals(const Model& m) { for (const Integral& in : m.integrals) if (in.depth >= 2) return true; return false; }

// Mesh hand-over between passes at the same parameters (codegen.cpp, emit_integral_site): per data point and outermost
// integrate() call site one record of kMeshRecord bytes -- [0] the number of bisections (255: not recorded), [1..] which interval
// each bisection split.  mesh_sites: records per point the model's kernels use (0: none).
constexpr int kMeshRecord = 64, kMeshSitesMax = 4;
int mesh_sites(const Model& m);

// Up to this many active parameters the fused kernel forms J^T J / J^T r in per-lane VALU accumulators
// (n (n + 1) / 2 + n + 1 of them) instead of on the matrix cores (codegen.cpp, GFH_K_SWEEP_GRAM).  Measured at N = 1e7: 8 parameters
// 0.166 ms against 0.179 ms with one matrix tile; 12 parameters 0.258 ms (252 VGPRs) against 0.188 ms: the boundary stays at 8.
constexpr int kValuGramMax = 8;
// ... and how many passes ahead that form loads its inputs: ONE.  Two (three rotating register sets, GADFIT_HIP_VALU_AHEAD=2) were built
// and measured in round 6 on the theory that two waves per SIMD leave too few bytes in flight: in-process A/B over six fresh contexts
// each, configs 2 and 3 -- 0.1496 / 0.0925 ms against 0.1498 / 0.0896 (profiles/r06_valu_form_ab.txt): nothing at 8 parameters, a
// loss at 7 (126 -> 136 VGPRs costs the fourth wave per SIMD).  What the short kernels had been losing was their epilogue.
inline int valu_ahead_for(int n_active, const GenConfig& cfg) { (void)n_active; return cfg.valu_ahead > 1 ? 2 : 1; }

// The fused STEP 1 + STEP 2 kernel exists for up to this many active parameters (8 tiles of 16); beyond it gfh_k_sweep writes J and
// k_gram_block forms the Gram image from it.  Up to 4 tiles every wave keeps all tile pairs of its own points; round 5 took 5 tiles
// that way on a half stage (21 accumulator tiles at 6 spilled 150-200 registers and lost to the two-kernel path: profiles/r05_fused_tiles.md);
// round 6: from 6 tiles on (81 ... 128 parameters) the waves of a workgroup share the tile pairs out and read each other's stages
// (GFH_COOP, codegen.cpp; GADFIT_HIP_COOP=0: the two-kernel path beyond 80 parameters as in round 5).
// gfh_k_omega_jt (STEP 3 without the stored Jacobian) stops at kOmegaJtMaxActive.
constexpr int kFusedMaxActive = 128, kFusedMaxActiveNoCoop = 80, kOmegaJtMaxActive = 64;
// Does the fused kernel's matrix path stage 32 points per wave (two half-passes) instead of 64?  Measured at 32 parameters
// (profiles/r05_halfstage.md): the half-passes themselves cost 18 % at equal occupancy (twice the stage-write instructions, a
// bubble at the half boundary) and the gradient that waits in registers keeps three waves per SIMD out of reach (55 spills at 168
// registers), so up to 4 tiles the full stage stays.  With 5 and 6 tiles four full stages do not fit the 160 KB of a CU: there
// the half stage is what makes the fused kernel possible at all (against a Jacobian written and read back: 4.6 x the traffic).
// (5 tiles stay with round 5's per-wave form: 0.76 against 0.89 ms at p = 80, N = 4e6 -- there the 15 accumulator tiles still fit and
// its diagonal tiles run as 4x4x4 blocks; GADFIT_HIP_COOP=2 forces the cooperative form from 65 parameters on: profiles/r06_coop.md)
inline bool fused_coop(int n_active, const GenConfig& cfg) { return cfg.coop != 0 && n_active > (cfg.coop >= 2 ? 64 : kFusedMaxActiveNoCoop); }
// (the largest active set the fused kernel takes under this configuration)
inline int fused_max_active(const GenConfig& cfg) { return cfg.coop != 0 ? kFusedMaxActive : kFusedMaxActiveNoCoop; }
inline bool fused_half_stage(int n_active, const GenConfig& cfg) {
  if (n_active <= kValuGramMax) return false;
  if (n_active > 64) return true;
  if (cfg.half_stage >= 0) return cfg.half_stage != 0;
  return false;
}
// 5 and 6 tiles: ONE cross-wave reduction image per workgroup that the waves add into in order (the same order of additions as one
// image per wave, a quarter of the LDS), laid over the stages once they are dead.
inline bool fused_single_image(int n_active, const GenConfig& cfg) { return !fused_coop(n_active, cfg) && (n_active > 64 || (cfg.single_image > 0 && n_active > kValuGramMax)); }
inline int fused_stage_stride(int n_active, const GenConfig& cfg) { return fused_half_stage(n_active, cfg) ? 34 : 66; }
// LDS of one workgroup of the fused kernel's matrix path with fw waves (the generated source declares exactly this: GFH_LDS_DOUBLES)
inline long fused_lds_bytes_for(int n_active, int fw, const GenConfig& cfg) {
  const long T = (n_active + 15) / 16, npair = T * (T + 1) / 2;
  const long stage = (16 * T + 1) * fused_stage_stride(n_active, cfg);
  const long img = npair * 256 + 16 * T + 1;                          // the workgroup's own sums, kept for the single-workgroup tail
  if (fused_coop(n_active, cfg)) return std::max(fw * stage, T * 64 + 8 + img) * 8;      // (the epilogue lies over the stages)
  if (fused_single_image(n_active, cfg)) return std::max(fw * stage, npair * 256 + fw * (T * 64 + 4) + img) * 8;
  const long red = npair * 256 + T * 64 + 4;                          // cross-wave reduction image, one per wave, shares the stages' buffer
  return fw * std::max(stage, red) * 8 + img * 8 + 64;
}
// Waves per workgroup of the fused kernel: 8 (one workgroup per CU at 32 parameters), fewer where 8 stages of
// [(16T+1) rows][stride] fp64 do not fit the 160 KB LDS.
inline int fused_waves_for(int n_active, const GenConfig& cfg) {
  int fw = fused_coop(n_active, cfg) ? 4 : 8;      // (cooperative form: one wave per SIMD -- the gradient alone is 2 NA registers)
  while (fw > 1 && fused_lds_bytes_for(n_active, fw, cfg) > 160L * 1024) fw /= 2;
  if (cfg.fused_waves > 0 && n_active > kValuGramMax) fw = std::min(fw, cfg.fused_waves);
  return fw;
}

// Generates one HIP translation unit with three kernels for (model, active set):
//   gfh_k_sweep : residual + Jacobian columns (reverse mode, statically unrolled)
//   gfh_k_chi2  : residual only, all parameters passive, per-block sum of squares
//   gfh_k_omega : second directional derivative (forward mode val/d/dd)
bool generate_source(const Model& m, const std::vector<int32_t>& active, const GenConfig& cfg,
                     std::string* src, std::string* err);

}  // namespace gfh

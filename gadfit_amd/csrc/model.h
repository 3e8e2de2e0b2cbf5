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

struct Model {
  int32_t n_pars = 0;
  std::vector<SubTape> sub;
  std::vector<Integral> integrals;
  std::vector<int32_t> ipar_nodes;
  int32_t gk_points = 15;
  int32_t n_aux = 0;            // auxiliary per-point columns read by eval() (GFH_AUX)
  double rel_error_outer = 0, rel_error_inner = 0;

  // copies and validates; returns false and sets err on malformed tapes
  bool load(const gfh_tape* t, std::string* err);
  bool has_integrals() const { return !integrals.empty(); }
};

// Options of the generated kernels (kept in the source text so the cache key sees them).
struct GenConfig {
  int block = 256;        // threads per workgroup of the plain sweep / chi2 / omega kernels = slots per tile
  bool finite_diff = false; // use_ad = .false.: gradient / second directional derivative by the reference's finite differences (fitfunction.F90:155-203)
  bool omega_jt = true;   // STEP 3 kernel that recomputes the Jacobian row instead of reading J (gfh_k_omega_jt)
  int kernarg_pars = 0;   // > 0: the parameter block (this many doubles) is a by-value kernel argument
  int ws_size = 100;      // per-lane quadrature workspace (intervals) per nesting level
  bool store_j = true;    // fused kernel writes the Jacobian to HBM (gfh_set_keep_jacobian)
  bool store_res = true;  // chi2 kernel writes the residual vector (the reference's chi2() side effect, gadfit.F90:1024-1026)
  int loss = 0;           // robust cost (gfh_set_loss): 0 linear, 1 cauchy, 2 huber
  bool fast_div = true;   // share one reciprocal per denominator (<= 1 ulp from the reference's r/v)
};

// Up to this many active parameters the fused kernel forms J^T J / J^T r in per-lane VALU accumulators
// (n (n + 1) / 2 + n + 1 of them) instead of on the matrix cores (codegen.cpp, GFH_K_SWEEP_GRAM).  Measured at N = 1e7: 8 parameters
// 0.166 ms against 0.179 ms with one matrix tile; 12 parameters 0.258 ms (252 VGPRs) against 0.188 ms: the boundary stays at 8.
constexpr int kValuGramMax = 8;

// Waves per workgroup of the fused kernel: 8 (one workgroup per CU at 32 parameters), fewer where 8 stages of
// [(16T+1) rows][66] fp64 do not fit the 160 KB LDS.
inline int fused_waves_for(int n_active) {
  const int T = (n_active + 15) / 16;
  const long red = (T * (T + 1) / 2 * 256L + T * 64 + 4) * 8;      // cross-wave reduction image shares the buffer
  const long stage = std::max((16L * T + 1) * 66 * 8, red);
  const long tail = (T * (T + 1) / 2 * 256L + 16 * T + 1) * 8 + 64;   // the workgroup's own sums, kept for the single-workgroup tail
  int fw = 8;
  while (fw > 1 && fw * stage + tail > 160L * 1024) fw /= 2;
  return fw;
}

// Generates one HIP translation unit with three kernels for (model, active set):
//   gfh_k_sweep : residual + Jacobian columns (reverse mode, statically unrolled)
//   gfh_k_chi2  : residual only, all parameters passive, per-block sum of squares
//   gfh_k_omega : second directional derivative (forward mode val/d/dd)
bool generate_source(const Model& m, const std::vector<int32_t>& active, const GenConfig& cfg,
                     std::string* src, std::string* err);

}  // namespace gfh

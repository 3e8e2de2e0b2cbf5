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

// Tunables of the generated kernels (kept in the source text so the cache key sees them).
struct GenConfig {
  int block = 256;        // threads per workgroup
  int ppl = 1;            // data points per lane
  bool wave_spec = false; // fused kernel with dedicated store waves (gfh_k_sweep_gram_ws)
  int ws_compute_waves = 8; // compute waves per workgroup of that variant (plus 4 store waves)
  int fused_waves = 8;    // waves per workgroup of the fused sweep+Gram kernel
  bool finite_diff = false; // use_ad = .false.: gradient / second directional derivative by the reference's finite differences (fitfunction.F90:155-203)
  bool lazy_forward = false; // gfh_point_grad: forward values emitted just before their first use in the reverse sweep (GADFIT_HIP_LAZY=1): same values, other
                             // live ranges; no measurable difference on the fused kernel or the plain sweep (run-to-run spread)
  bool omega_jt = true;   // STEP 3 kernel that recomputes the Jacobian row instead of reading J (gfh_k_omega_jt)
  int kernarg_pars = 0;   // > 0: the parameter block (this many doubles, one dataset) is a by-value kernel argument
  bool vm_wait_fix = true; // fused kernel: first pass's loads consumed before the loop (no store-queue drain per pass)
  bool half_stage = false;// fused kernel: 32-point LDS stage per wave, filled twice per pass (twice the waves per CU)
  bool fused_sync = true; // keep a workgroup's waves in phase (AD phase | matrix phase)
  bool spread_stores = true; // fused kernel: J stores interleaved with the k-steps
  bool pair_store = false;// fused kernel: 16-byte stores of column pairs from the LDS stage (needs ldj*8 < 2^31)
  int store_aux = 2;      // cache-policy bits of the J/res buffer stores (2 = nt: written once, streamed)
  int ws_size = 100;      // per-lane quadrature workspace (intervals) per nesting level
  int ablate = 0;         // timing experiments only: 1 = no J stores, 2 = no MFMA (results wrong)
  bool store_j = true;          // fused kernel writes the Jacobian to HBM (gfh_set_keep_jacobian)
  int loss = 0;                 // robust cost (gfh_set_loss): 0 linear, 1 cauchy, 2 huber
  bool fast_div = true;   // share one reciprocal per denominator (<= 1 ulp from the reference's r/v)
};

// Waves per workgroup of the fused kernel that fit the 160 KB LDS: each wave owns a
// [(16T+1) rows][66] fp64 stage.
inline int fused_waves_for(int n_active, int requested, bool half_stage = false) {
  const int T = (n_active + 15) / 16;
  const long red = (T * (T + 1) / 2 * 256L + T * 64 + 4) * 8;      // cross-wave reduction image shares the buffer
  const long stage = std::max((16L * T + 1) * (half_stage ? 34 : 66) * 8, red);
  int fw = requested;
  while (fw > 1 && fw * stage > 160L * 1024) fw /= 2;
  return fw;
}

inline int ws_compute_waves_for(int n_active, int requested) {
  const int T = (n_active + 15) / 16;
  const long stage = (16L * T + 1) * 66 * 8;
  int nc = requested;
  while (nc > 4 && nc * stage > 160L * 1024) nc -= 4;
  return nc;
}

// Generates one HIP translation unit with three kernels for (model, active set):
//   gfh_k_sweep : residual + Jacobian columns (reverse mode, statically unrolled)
//   gfh_k_chi2  : residual only, all parameters passive, per-block sum of squares
//   gfh_k_omega : second directional derivative (forward mode val/d/dd)
bool generate_source(const Model& m, const std::vector<int32_t>& active, const GenConfig& cfg,
                     std::string* src, std::string* err);

}  // namespace gfh

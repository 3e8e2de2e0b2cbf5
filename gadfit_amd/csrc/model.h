// model.h -- host-side copy of a gfh_tape plus the device code generator interface.
#pragma once
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
  double rel_error_outer = 0, rel_error_inner = 0;

  // copies and validates; returns false and sets err on malformed tapes
  bool load(const gfh_tape* t, std::string* err);
  bool has_integrals() const { return !integrals.empty(); }
};

// Tunables of the generated kernels (kept in the source text so the cache key sees them).
struct GenConfig {
  int block = 256;        // threads per workgroup
  int ppl = 1;            // data points per lane
};

// Generates one HIP translation unit with three kernels for (model, active set):
//   gfh_k_sweep : residual + Jacobian columns (reverse mode, statically unrolled)
//   gfh_k_chi2  : residual only, all parameters passive, per-block sum of squares
//   gfh_k_omega : second directional derivative (forward mode val/d/dd)
bool generate_source(const Model& m, const std::vector<int32_t>& active, const GenConfig& cfg,
                     std::string* src, std::string* err);

}  // namespace gfh

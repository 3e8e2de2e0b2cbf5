// codegen.cpp -- lowers a model tape to HIP source for gfx950.
//
// What the reference does per data point (gadfit.F90:679-690): run the user's eval() with
// operator overloading, which appends every elemental to a run-time tape
// (automatic_differentiation.F90:451-479), then walk that tape backwards in ad_grad
// (AD:1476-1659).  The tape's STRUCTURE is the same for every point, so here it is unrolled
// at code-generation time: forward values and adjoints become named doubles that the
// compiler keeps in VGPRs (one lane = one data point), the op dispatch disappears, and the
// (advar,advar)/(advar,real)/(real,advar) variant of each elemental is chosen statically
// from the operands' static type and activity -- the same choice the reference makes at run
// time from `index /= 0` (AD:454-479 pattern).  Formulas follow the reference line by line
// (cited below) so results agree to rounding.
//
// Three kernels per (model, active set):
//   gfh_k_sweep  STEP 1 (gadfit.F90:675-693): res_i = (y_i-f)*w_i, J[a][i] = df/dp_a * w_i
//   gfh_k_chi2   chi2() (gadfit.F90:1015-1034): all parameters passive, res_i and sum res^2
//   gfh_k_omega  STEP 3 (gadfit.F90:715-731): omega_i = -f''_delta(x_i) * w_i, forward mode
#include "model.h"
#include <cmath>
#include <cstdio>
#include <cstring>
#include <sstream>

namespace gfh {

bool Model::load(const gfh_tape* t, std::string* err) {
  if (!t || t->n_subtapes < 1 || !t->sub) { *err = "empty tape"; return false; }
  n_pars = t->n_pars;
  sub.clear(); integrals.clear(); ipar_nodes.clear();
  int n_bind = 0;
  for (int i = 0; i < t->n_integrals; i++) {
    const gfh_integral& g = t->integrals[i];
    integrals.push_back({g.integrand, g.lower, g.upper, g.lower_inf, g.upper_inf, g.n_ipars,
                         g.ipar_off, g.depth, g.rel_error, g.abs_error});
    if (g.ipar_off + g.n_ipars > n_bind) n_bind = g.ipar_off + g.n_ipars;
    if (g.integrand < 1 || g.integrand >= t->n_subtapes) { *err = "integral refers to a missing sub-tape"; return false; }
  }
  for (int i = 0; i < n_bind; i++) ipar_nodes.push_back(t->ipar_nodes[i]);
  for (int s = 0; s < t->n_subtapes; s++) {
    const gfh_subtape& st = t->sub[s];
    SubTape o; o.result = st.result;
    if (st.n_nodes < 1 || st.result < 0 || st.result >= st.n_nodes) { *err = "malformed sub-tape"; return false; }
    for (int k = 0; k < st.n_nodes; k++) {
      const gfh_node& n = st.nodes[k];
      Node d{n.op, n.a, n.b, n.flags, n.c};
      auto bad_ref = [&](int r) { return r < 0 || r >= k; };
      switch (n.op) {
        case GFH_CONST: case GFH_X: case GFH_IVAR: break;
        case GFH_PARAM: if (n.a < 0 || n.a >= n_pars) { *err = "parameter index out of range"; return false; } break;
        case GFH_IPARAM: if (n.a < 0) { *err = "bad integrand parameter"; return false; } break;
        case GFH_LIFT: case GFH_NEG: case GFH_POWI:
          if (bad_ref(n.a)) { *err = "operand refers forward"; return false; } break;
        case GFH_ADD: case GFH_SUB: case GFH_MUL: case GFH_DIV: case GFH_POW:
          if (bad_ref(n.a) || bad_ref(n.b)) { *err = "operand refers forward"; return false; } break;
        case GFH_INTEGRATE: if (n.a < 0 || n.a >= t->n_integrals) { *err = "bad integral index"; return false; } break;
        default:
          if (n.op >= GFH_ABS && n.op <= GFH_ERF) { if (bad_ref(n.a)) { *err = "operand refers forward"; return false; } }
          else { *err = "unknown op code " + std::to_string(n.op); return false; }
      }
      o.nodes.push_back(d);
    }
    sub.push_back(std::move(o));
  }
  gk_points = t->gk_points ? t->gk_points : 15;
  rel_error_outer = t->rel_error_outer; rel_error_inner = t->rel_error_inner;
  return true;
}

namespace {

std::string lit(double c) {
  char b[64];
  if (std::isnan(c)) return "__builtin_nan(\"\")";
  if (std::isinf(c)) return c > 0 ? "__builtin_inf()" : "(-__builtin_inf())";
  snprintf(b, sizeof b, "%a", c);   // hex float: exact
  return b;
}

const char* fn_name(int op) {
  switch (op) {
    case GFH_ABS: return "fabs"; case GFH_EXP: return "exp"; case GFH_SQRT: return "sqrt";
    case GFH_LOG: return "log"; case GFH_SIN: return "sin"; case GFH_COS: return "cos";
    case GFH_TAN: return "tan"; case GFH_ASIN: return "asin"; case GFH_ACOS: return "acos";
    case GFH_ATAN: return "atan"; case GFH_SINH: return "sinh"; case GFH_COSH: return "cosh";
    case GFH_TANH: return "tanh"; case GFH_ASINH: return "asinh"; case GFH_ACOSH: return "acosh";
    case GFH_ATANH: return "atanh"; case GFH_ERF: return "erf";
  }
  return "?";
}

// 2/sqrt(pi) as the reference forms it: 2.0_kp/sqrtpi (AD:1633, gadf_constants.F90:29-33)
const char* TWO_OVER_SQRTPI = "(2.0/0x1.c5bf891b4ef6bp+0)";

struct Gen {
  const Model& m;
  const SubTape& st;
  std::vector<char> is_real, act;   // per node
  std::ostringstream o;
  std::string ind = "  ";

  Gen(const Model& mm, const SubTape& s) : m(mm), st(s) {}

  std::string v(int k) const { return "v" + std::to_string(k); }
  std::string b(int k) const { return "b" + std::to_string(k); }
  std::string d(int k) const { return "d" + std::to_string(k); }
  std::string dd(int k) const { return "e" + std::to_string(k); }

  void analyse(const std::vector<char>& par_active) {
    int n = (int)st.nodes.size();
    is_real.assign(n, 0); act.assign(n, 0);
    for (int k = 0; k < n; k++) {
      const Node& nd = st.nodes[k];
      is_real[k] = (nd.flags & GFH_F_REAL) ? 1 : 0;
      switch (nd.op) {
        case GFH_PARAM: act[k] = par_active[nd.a]; break;
        case GFH_CONST: case GFH_X: act[k] = 0; break;
        case GFH_LIFT: act[k] = 0; break;
        case GFH_ADD: case GFH_SUB: case GFH_MUL: case GFH_DIV: case GFH_POW:
          act[k] = (act[nd.a] || act[nd.b]) && !is_real[k]; break;
        default: act[k] = act[nd.a] && !is_real[k]; break;
      }
    }
  }

  // integer power by repeated squaring, same multiplication order as oracle powi()
  std::string powi_expr(const std::string& x, int n, const std::string& tmp) {
    if (n == 0) return "1.0";
    unsigned mag = n < 0 ? (unsigned)(-(long)n) : (unsigned)n;
    std::string r; std::string base = x; int lvl = 0;
    // emit temporaries for the squarings
    while (mag) {
      if (mag & 1u) r = r.empty() ? base : "(" + r + "*" + base + ")";
      mag >>= 1;
      if (mag) {
        std::string nb = tmp + "_s" + std::to_string(lvl++);
        o << ind << "const double " << nb << " = " << base << "*" << base << ";\n";
        base = nb;
      }
    }
    return n < 0 ? "(1.0/" + r + ")" : r;
  }

  // ---------------------------------------------------------------- forward values
  // variant of a binary op: 0 plain(real or both passive advar), 1 aa, 2 ar, 3 ra
  int variant(const Node& nd, int k) const {
    if (is_real[k]) return 0;
    bool ra_ = is_real[nd.a], rb_ = is_real[nd.b];
    if (!ra_ && !rb_) {
      if (act[nd.a] && act[nd.b]) return 1;
      if (act[nd.a]) return 2;
      if (act[nd.b]) return 3;
      return 0;
    }
    return ra_ ? 3 : 2;   // static overload (advar,real) / (real,advar), also when passive
  }

  void emit_values(bool with_aux) {
    int n = (int)st.nodes.size();
    for (int k = 0; k < n; k++) {
      const Node& nd = st.nodes[k];
      std::string lhs = ind + "const double " + v(k) + " = ";
      switch (nd.op) {
        case GFH_CONST: o << lhs << lit(nd.c) << ";\n"; break;
        case GFH_X: o << lhs << "X;\n"; break;
        case GFH_PARAM: o << lhs << "P[" << nd.a << "];\n"; break;
        case GFH_LIFT: o << lhs << v(nd.a) << ";\n"; break;
        case GFH_NEG: o << lhs << "-" << v(nd.a) << ";\n"; break;
        case GFH_ADD: o << lhs << v(nd.a) << " + " << v(nd.b) << ";\n"; break;
        case GFH_SUB: o << lhs << v(nd.a) << " - " << v(nd.b) << ";\n"; break;
        case GFH_MUL: o << lhs << v(nd.a) << " * " << v(nd.b) << ";\n"; break;
        case GFH_DIV: {
          int var = variant(nd, k);
          if (var == 1 || var == 2) {   // AD:814-841, 843-866: multiply by the reciprocal
            o << ind << "const double i" << k << " = 1.0 / " << v(nd.b) << ";\n";
            o << lhs << v(nd.a) << " * i" << k << ";\n";
          } else o << lhs << v(nd.a) << " / " << v(nd.b) << ";\n";   // AD:892-913, 839
          break;
        }
        case GFH_POW: o << lhs << "pow(" << v(nd.a) << ", " << v(nd.b) << ");\n"; break;
        case GFH_POWI: { std::string e = powi_expr(v(nd.a), nd.b, "q" + std::to_string(k)); o << lhs << e << ";\n"; break; }
        default: o << lhs << fn_name(nd.op) << "(" << v(nd.a) << ");\n"; break;
      }
    }
    (void)with_aux;
  }

  // ---------------------------------------------------------------- reverse sweep, AD:1476-1659
  void emit_reverse() {
    int n = (int)st.nodes.size();
    for (int k = 0; k < n; k++) if (act[k]) o << ind << "double " << b(k) << " = 0.0;\n";
    if (!act[st.result]) return;
    o << ind << b(st.result) << " = 1.0;\n";                                   // AD:1490
    auto acc = [&](int tgt, const std::string& sign, const std::string& expr) {
      o << ind << b(tgt) << " = " << b(tgt) << " " << sign << " " << expr << ";\n";
    };
    for (int k = n - 1; k >= 0; k--) {
      if (!act[k]) continue;
      const Node& nd = st.nodes[k];
      const std::string bk = b(k);
      switch (nd.op) {
        case GFH_PARAM: break;
        case GFH_ADD: {
          int var = variant(nd, k);
          if (var == 1) { acc(nd.a, "+", bk); acc(nd.b, "+", bk); }          // AD:1496-1499
          else acc(var == 2 ? nd.a : nd.b, "+", bk);                            // AD:1500-1502
          break;
        }
        case GFH_SUB: {
          int var = variant(nd, k);
          if (var == 1) { acc(nd.a, "+", bk); acc(nd.b, "-", bk); }          // AD:1503-1506
          else if (var == 2) acc(nd.a, "+", bk);                               // AD:1500-1502
          else acc(nd.b, "-", bk);                                             // AD:1507-1509
          break;
        }
        case GFH_MUL: {
          int var = variant(nd, k);
          if (var == 1) { acc(nd.a, "+", bk + "*" + v(nd.b)); acc(nd.b, "+", bk + "*" + v(nd.a)); }  // AD:1510-1515
          else if (var == 2) acc(nd.a, "+", bk + "*" + v(nd.b));               // AD:1516-1520
          else acc(nd.b, "+", bk + "*" + v(nd.a));
          break;
        }
        case GFH_DIV: {
          int var = variant(nd, k);
          if (var == 1) {                                                       // AD:1521-1527
            acc(nd.a, "+", bk + "/" + v(nd.b));
            acc(nd.b, "-", bk + "*" + v(k) + "/" + v(nd.b));
          } else if (var == 2) acc(nd.a, "+", bk + "*i" + std::to_string(k));   // AD:1516-1520, const = inv
          else acc(nd.b, "-", bk + "*" + v(nd.a) + "/" + v(nd.b) + "/" + v(nd.b));  // AD:1528-1533
          break;
        }
        case GFH_POW: {
          int var = variant(nd, k);
          if (var == 1) {                                                       // AD:1534-1541
            acc(nd.a, "+", bk + "*" + v(nd.b) + "*pow(" + v(nd.a) + ", " + v(nd.b) + " - 1.0)");
            acc(nd.b, "+", bk + "*log(" + v(nd.a) + ")*" + v(k));             // x1**x2 recomputed = y
          } else if (var == 2)                                                  // AD:1542-1547
            acc(nd.a, "+", bk + "*" + v(nd.b) + "*pow(" + v(nd.a) + ", " + v(nd.b) + " - 1.0)");
          else                                                                  // AD:1548-1553
            acc(nd.b, "+", bk + "*log(" + v(nd.a) + ")*" + v(k));
          break;
        }
        case GFH_POWI: {                                                        // AD:1554-1558
          std::string e = powi_expr(v(nd.a), nd.b - 1, "r" + std::to_string(k));
          acc(nd.a, "+", bk + "*" + lit((double)nd.b) + "*" + e);
          break;
        }
        case GFH_ABS:                                                           // AD:1559-1567
          o << ind << b(nd.a) << " = (" << v(nd.a) << " < 0.0) ? " << b(nd.a) << " - " << bk << " : " << b(nd.a) << " + " << bk << ";\n";
          break;
        case GFH_EXP: acc(nd.a, "+", bk + "*" + v(k)); break;                  // AD:1568-1571
        case GFH_SQRT: acc(nd.a, "+", bk + "/2.0/" + v(k)); break;             // AD:1572-1575
        case GFH_LOG: acc(nd.a, "+", bk + "/" + v(nd.a)); break;               // AD:1576-1579
        case GFH_SIN: acc(nd.a, "+", bk + "*cos(" + v(nd.a) + ")"); break;     // AD:1581-1584
        case GFH_COS: acc(nd.a, "-", bk + "*sin(" + v(nd.a) + ")"); break;     // AD:1585-1588
        case GFH_TAN: o << ind << "const double c" << k << " = cos(" << v(nd.a) << ");\n";
                      acc(nd.a, "+", bk + "/(c" + std::to_string(k) + "*c" + std::to_string(k) + ")"); break;   // AD:1589-1592
        case GFH_ASIN: acc(nd.a, "+", bk + "/sqrt(1.0 - " + v(nd.a) + "*" + v(nd.a) + ")"); break;  // AD:1593-1596
        case GFH_ACOS: acc(nd.a, "-", bk + "/sqrt(1.0 - " + v(nd.a) + "*" + v(nd.a) + ")"); break;  // AD:1597-1600
        case GFH_ATAN: acc(nd.a, "+", bk + "/(1.0 + " + v(nd.a) + "*" + v(nd.a) + ")"); break;      // AD:1601-1604
        case GFH_SINH: acc(nd.a, "+", bk + "*cosh(" + v(nd.a) + ")"); break;   // AD:1606-1609
        case GFH_COSH: acc(nd.a, "+", bk + "*sinh(" + v(nd.a) + ")"); break;   // AD:1610-1613
        case GFH_TANH: o << ind << "const double c" << k << " = cosh(" << v(nd.a) << ");\n";
                       acc(nd.a, "+", bk + "/(c" + std::to_string(k) + "*c" + std::to_string(k) + ")"); break;  // AD:1614-1617
        case GFH_ASINH: acc(nd.a, "+", bk + "/sqrt(" + v(nd.a) + "*" + v(nd.a) + " + 1.0)"); break; // AD:1618-1621
        case GFH_ACOSH: acc(nd.a, "+", bk + "/sqrt(" + v(nd.a) + "*" + v(nd.a) + " - 1.0)"); break; // AD:1622-1625
        case GFH_ATANH: acc(nd.a, "+", bk + "/(1.0 - " + v(nd.a) + "*" + v(nd.a) + ")"); break;     // AD:1626-1629
        case GFH_ERF: acc(nd.a, "+", bk + "*" + TWO_OVER_SQRTPI + "*exp(-(" + v(nd.a) + "*" + v(nd.a) + "))"); break; // AD:1631-1635
        default: break;
      }
    }
  }

  // ---------------------------------------------------------------- forward mode (val,d,dd)
  // Active nodes carry d<k> and e<k> (= dd).  Formulas: the `else` branches of AD:454-1459.
  void emit_forward_dd() {
    int n = (int)st.nodes.size();
    for (int k = 0; k < n; k++) {
      if (!act[k]) continue;
      const Node& nd = st.nodes[k];
      auto D = [&](const std::string& e) { o << ind << "const double " << d(k) << " = " << e << ";\n"; };
      auto E = [&](const std::string& e) { o << ind << "const double " << dd(k) << " = " << e << ";\n"; };
      const std::string va = nd.a >= 0 && nd.op != GFH_PARAM ? v(nd.a) : "", y = v(k);
      const std::string da = nd.a >= 0 && nd.op != GFH_PARAM ? d(nd.a) : "", ea = nd.a >= 0 && nd.op != GFH_PARAM ? dd(nd.a) : "";
      std::string ks = std::to_string(k);
      switch (nd.op) {
        case GFH_PARAM: D("DP[" + std::to_string(nd.a) + "]"); E("0.0"); break;   // gadfit.F90:719: %d = delta1, dd = 0
        case GFH_ADD: {
          int var = variant(nd, k);
          if (var == 1) { D(d(nd.a) + " + " + d(nd.b)); E(dd(nd.a) + " + " + dd(nd.b)); }   // AD:468-469
          else { int s = var == 2 ? nd.a : nd.b; D(d(s)); E(dd(s)); }                           // AD:495-496, 540-541
          break;
        }
        case GFH_SUB: {
          int var = variant(nd, k);
          if (var == 1) { D(d(nd.a) + " - " + d(nd.b)); E(dd(nd.a) + " - " + dd(nd.b)); }   // AD:585-586
          else if (var == 2) { D(d(nd.a)); E(dd(nd.a)); }                                     // AD:616-617
          else { D("-" + d(nd.b)); E("-" + dd(nd.b)); }                                       // AD:661-662
          break;
        }
        case GFH_MUL: {
          int var = variant(nd, k);
          std::string x1 = v(nd.a), x2 = v(nd.b);
          if (var == 1) {                                                                       // AD:707-708
            D(x1 + "*" + d(nd.b) + " + " + d(nd.a) + "*" + x2);
            E(x1 + "*" + dd(nd.b) + " + 2.0*" + d(nd.a) + "*" + d(nd.b) + " + " + dd(nd.a) + "*" + x2);
          } else if (var == 2) { D(d(nd.a) + "*" + x2); E(dd(nd.a) + "*" + x2); }             // AD:736-737
          else { D(x1 + "*" + d(nd.b)); E(x1 + "*" + dd(nd.b)); }                              // AD:783-784
          break;
        }
        case GFH_DIV: {
          int var = variant(nd, k);
          if (var == 1) {                                                                       // AD:830-831
            D("(" + d(nd.a) + " - " + y + "*" + d(nd.b) + ")*i" + ks);
            E("(" + dd(nd.a) + " - " + y + "*" + dd(nd.b) + " - 2.0*" + d(k) + "*" + d(nd.b) + ")*i" + ks);
          } else if (var == 2) { D(d(nd.a) + "*i" + ks); E(dd(nd.a) + "*i" + ks); }            // AD:861-862
          else {                                                                                // AD:907-908
            D("-" + y + "*" + d(nd.b) + "/" + v(nd.b));
            E("(-" + y + "*" + dd(nd.b) + " - 2.0*" + d(k) + "*" + d(nd.b) + ")/" + v(nd.b));
          }
          break;
        }
        case GFH_POW: {
          int var = variant(nd, k);
          std::string x1 = v(nd.a), x2 = v(nd.b);
          if (var == 1) {                                                                       // AD:975-980
            o << ind << "const double l" << ks << " = log(" << x1 << ");\n";
            D(y + "*" + d(nd.b) + "*l" + ks + " + " + d(nd.a) + "*" + x2 + "*pow(" + x1 + ", " + x2 + " - 1.0)");
            o << ind << "const double i" << ks << " = 1.0 / " << x1 << ";\n";
            E(d(k) + "*" + d(k) + "/" + y + " + " + y + "*(" + dd(nd.b) + "*l" + ks + " + (2.0*" + d(nd.b) + "*" + d(nd.a) +
              " + " + x2 + "*(" + dd(nd.a) + " - " + d(nd.a) + "*" + d(nd.a) + "*i" + ks + "))*i" + ks + ")");
          } else if (var == 2) {                                                                // AD:1005-1008
            D(d(nd.a) + "*" + x2 + "*pow(" + x1 + ", " + x2 + " - 1.0)");
            o << ind << "const double i" << ks << " = 1.0 / " << x1 << ";\n";
            E(d(k) + "*" + d(k) + "/" + y + " + " + y + "*" + x2 + "*(" + dd(nd.a) + " - " + d(nd.a) + "*" + d(nd.a) + "*i" + ks + ")*i" + ks);
          } else {                                                                              // AD:1076-1079
            o << ind << "const double l" << ks << " = log(" << x1 << ");\n";
            D(y + "*" + d(nd.b) + "*l" + ks);
            E(d(k) + "*" + d(k) + "/" + y + " + " + y + "*" + dd(nd.b) + "*l" + ks);
          }
          break;
        }
        case GFH_POWI: {                                                                        // AD:1051-1054
          std::string nn = lit((double)nd.b);
          o << ind << "const double i" << ks << " = 1.0 / " << va << ";\n";
          D(y + "*" + nn + "*" + da + "*i" + ks);
          E(d(k) + "*" + d(k) + "/" + y + " + " + y + "*" + nn + "*(" + ea + " - " + da + "*" + da + "*i" + ks + ")*i" + ks);
          break;
        }
        case GFH_ABS:                                                                           // AD:951-953
          o << ind << "const double s" << ks << " = copysign(1.0, " << va << ");\n";
          D(da + "*s" + ks); E(ea + "*s" + ks); break;
        case GFH_EXP: D(da + "*" + y); E(ea + "*" + y + " + " + da + "*" + d(k)); break;      // AD:1123-1124
        case GFH_SQRT:                                                                          // AD:1144-1146
          o << ind << "const double i" << ks << " = 1.0 / " << y << ";\n";
          D(da + "/2.0*i" + ks); E("(" + ea + "*i" + ks + " - " + d(k) + "*" + da + "/" + va + ")/2.0"); break;
        case GFH_LOG:                                                                           // AD:1166-1168
          o << ind << "const double i" << ks << " = 1.0 / " << va << ";\n";
          D(da + "*i" + ks); E("(" + ea + " - " + da + "*" + d(k) + ")*i" + ks); break;
        case GFH_SIN:                                                                           // AD:1188-1190
          o << ind << "const double c" << ks << " = cos(" << va << ");\n";
          D(da + "*c" + ks); E(ea + "*c" + ks + " - " + da + "*" + da + "*" + y); break;
        case GFH_COS:                                                                           // AD:1210-1212
          o << ind << "const double c" << ks << " = -sin(" << va << ");\n";
          D(da + "*c" + ks); E(ea + "*c" + ks + " - " + da + "*" + da + "*" + y); break;
        case GFH_TAN:                                                                           // AD:1232-1235
          o << ind << "const double t" << ks << " = 1.0 / cos(" << va << ");\n";
          o << ind << "const double c" << ks << " = t" << ks << "*t" << ks << ";\n";
          D(da + "*c" + ks); E(ea + "*c" + ks + " + 2.0*" + da + "*" + y + "*" + d(k)); break;
        case GFH_ASIN:                                                                          // AD:1255-1257
          o << ind << "const double t" << ks << " = 1.0 / sqrt(1.0 - " << va << "*" << va << ");\n";
          D(da + "*t" + ks); E("t" + ks + "*(" + ea + " + " + va + "*" + d(k) + "*" + d(k) + ")"); break;
        case GFH_ACOS:                                                                          // AD:1277-1279
          o << ind << "const double t" << ks << " = -1.0 / sqrt(1.0 - " << va << "*" << va << ");\n";
          D(da + "*t" + ks); E("t" + ks + "*(" + ea + " + " + va + "*" + d(k) + "*" + d(k) + ")"); break;
        case GFH_ATAN:                                                                          // AD:1299-1301
          o << ind << "const double t" << ks << " = 1.0 / (1.0 + " << va << "*" << va << ");\n";
          D(da + "*t" + ks); E(ea + "*t" + ks + " - 2.0*" + va + "*" + d(k) + "*" + d(k)); break;
        case GFH_SINH:                                                                          // AD:1321-1323
          o << ind << "const double c" << ks << " = cosh(" << va << ");\n";
          D(da + "*c" + ks); E(ea + "*c" + ks + " + " + y + "*" + da + "*" + da); break;
        case GFH_COSH:                                                                          // AD:1343-1345
          o << ind << "const double c" << ks << " = sinh(" << va << ");\n";
          D(da + "*c" + ks); E(ea + "*c" + ks + " + " + y + "*" + da + "*" + da); break;
        case GFH_TANH:                                                                          // AD:1365-1367
          o << ind << "const double h" << ks << " = cosh(" << va << ");\n";
          o << ind << "const double t" << ks << " = 1.0 / (h" << ks << "*h" << ks << ");\n";
          D(da + "*t" + ks); E(ea + "*t" + ks + " - 2.0*" + y + "*" + da + "*" + d(k)); break;
        case GFH_ASINH:                                                                         // AD:1387-1389
          o << ind << "const double t" << ks << " = 1.0 / sqrt(1.0 + " << va << "*" << va << ");\n";
          D(da + "*t" + ks); E("t" + ks + "*(" + ea + " - " + va + "*" + d(k) + "*" + d(k) + ")"); break;
        case GFH_ACOSH:                                                                         // AD:1409-1411
          o << ind << "const double t" << ks << " = 1.0 / sqrt(" << va << "*" << va << " - 1.0);\n";
          D(da + "*t" + ks); E("t" + ks + "*(" + ea + " - " + va + "*" + d(k) + "*" + d(k) + ")"); break;
        case GFH_ATANH:                                                                         // AD:1431-1433
          o << ind << "const double t" << ks << " = 1.0 / (1.0 - " << va << "*" << va << ");\n";
          D(da + "*t" + ks); E("(" + ea + " + 2.0*" + va + "*" + da + "*" + d(k) + ")*t" + ks); break;
        case GFH_ERF:                                                                           // AD:1453-1455
          o << ind << "const double t" << ks << " = " << TWO_OVER_SQRTPI << "*exp(-(" << va << "*" << va << "));\n";
          D(da + "*t" + ks); E("(" + ea + " - 2.0*" + da + "*" + da + "*" + va + ")*t" + ks); break;
        default: break;
      }
    }
  }
};

}  // namespace

bool generate_source(const Model& m, const std::vector<int32_t>& active, const GenConfig& cfg,
                     std::string* src, std::string* err) {
  if (m.has_integrals()) { *err = "models with integrate() are not yet lowered to the device"; return false; }
  const SubTape& st = m.sub[0];
  for (const Node& nd : st.nodes)
    if (nd.op == GFH_IVAR || nd.op == GFH_IPARAM) { *err = "integrand node in eval() tape"; return false; }
  const int NA = (int)active.size(), NP = m.n_pars;
  std::vector<char> pa(NP, 0), none(NP, 0);
  for (int a : active) { if (a < 0 || a >= NP) { *err = "active parameter out of range"; return false; } pa[a] = 1; }
  // adjoint source per active parameter: every PARAM node of that parameter
  std::ostringstream s;
  s << "// generated by libgadfit_hip codegen -- model with " << st.nodes.size() << " tape nodes, "
    << NP << " parameters, " << NA << " active\n";
  s << "#define GFH_BLOCK " << cfg.block << "\n#define GFH_PPL " << cfg.ppl << "\n#define GFH_NP " << NP
    << "\n#define GFH_NA " << (NA > 0 ? NA : 1) << "\n";
  s << R"(
typedef long long i64;

// One data point, reverse mode: value F and gradient G[a] = dF/dp_active(a).
static __device__ __forceinline__ void gfh_point_grad(const double X, const double* __restrict__ P,
                                                      double& F, double (&G)[GFH_NA]) {
)";
  {
    Gen g(m, st); g.analyse(pa); g.emit_values(false); g.emit_reverse();
    s << g.o.str();
    s << "  F = " << g.v(st.result) << ";\n";
    for (int j = 0; j < NA; j++) {
      std::string e;
      for (int k = 0; k < (int)st.nodes.size(); k++)
        if (st.nodes[k].op == GFH_PARAM && st.nodes[k].a == active[j] && g.act[k]) e += (e.empty() ? "" : " + ") + g.b(k);
      s << "  G[" << j << "] = " << (e.empty() ? "0.0" : e) << ";\n";
    }
  }
  s << R"(}

// One data point, every parameter passive (chi2 path): value only.
static __device__ __forceinline__ double gfh_point_value(const double X, const double* __restrict__ P) {
)";
  {
    Gen g(m, st); g.analyse(none); g.emit_values(false);
    s << g.o.str();
    s << "  return " << g.v(st.result) << ";\n";
  }
  s << R"(}

// One data point, forward mode: second directional derivative along DP (per-parameter d seeds).
static __device__ __forceinline__ double gfh_point_dd(const double X, const double* __restrict__ P,
                                                      const double* __restrict__ DP) {
)";
  {
    Gen g(m, st); g.analyse(pa); g.emit_values(false); g.emit_forward_dd();
    s << g.o.str();
    if (g.act[st.result]) s << "  return " << g.dd(st.result) << ";\n";
    else s << "  return 0.0;\n";
  }
  s << "}\n";
  // ---- hand-written kernel skeletons (the model body above is the only generated part)
  s << R"(
// Device layout (DESIGN.md "Data layout"): slots are data points padded per dataset to a
// multiple of the tile so every tile is full and belongs to one dataset; pad slots carry
// w = 0.  x, y, w, res, omega: [n_slots]; J: [NA][ldj] (parameter-major: a wave's store of
// one Jacobian column is 64 consecutive doubles = one fully coalesced 512 B write).
#define GFH_TILE (GFH_BLOCK * GFH_PPL)

extern "C" __global__ __launch_bounds__(GFH_BLOCK)
void gfh_k_sweep(const double* __restrict__ x, const double* __restrict__ y, const double* __restrict__ w,
                 const double* __restrict__ pars, const int* __restrict__ tile_ds, const int n_tiles,
                 double* __restrict__ res, double* __restrict__ J, const i64 ldj) {
  for (int t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    const double* __restrict__ P = pars + (i64)tile_ds[t] * GFH_NP;   // wave-uniform: scalar loads
    const i64 base = (i64)t * GFH_TILE + threadIdx.x;
#pragma unroll
    for (int q = 0; q < GFH_PPL; q++) {
      const i64 i = base + (i64)q * GFH_BLOCK;
      const double X = x[i], Y = y[i], W = w[i];
      double F, G[GFH_NA];
      gfh_point_grad(X, P, F, G);
      res[i] = (Y - F) * W;                       // gadfit.F90:682-683
#pragma unroll
      for (int a = 0; a < GFH_NA; a++) J[(i64)a * ldj + i] = G[a] * W;   // gadfit.F90:689-690
    }
  }
}

extern "C" __global__ __launch_bounds__(GFH_BLOCK)
void gfh_k_chi2(const double* __restrict__ x, const double* __restrict__ y, const double* __restrict__ w,
                const double* __restrict__ pars, const int* __restrict__ tile_ds, const int n_tiles,
                double* __restrict__ res, double* __restrict__ partial) {
  double s = 0.0;
  for (int t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    const double* __restrict__ P = pars + (i64)tile_ds[t] * GFH_NP;
    const i64 base = (i64)t * GFH_TILE + threadIdx.x;
#pragma unroll
    for (int q = 0; q < GFH_PPL; q++) {
      const i64 i = base + (i64)q * GFH_BLOCK;
      const double r = (y[i] - gfh_point_value(x[i], P)) * w[i];   // gadfit.F90:1024-1026
      res[i] = r;
      s += r * r;
    }
  }
  // deterministic block reduction: wave shuffle tree, then the wave sums in order
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  __shared__ double ws[GFH_BLOCK / 64];
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double tot = ws[0];
#pragma unroll
    for (int k = 1; k < GFH_BLOCK / 64; k++) tot += ws[k];
    partial[blockIdx.x] = tot;
  }
}

extern "C" __global__ __launch_bounds__(GFH_BLOCK)
void gfh_k_omega(const double* __restrict__ x, const double* __restrict__ w,
                 const double* __restrict__ pars, const double* __restrict__ dpars,
                 const int* __restrict__ tile_ds, const int n_tiles, double* __restrict__ omega) {
  for (int t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    const int ds = tile_ds[t];
    const double* __restrict__ P = pars + (i64)ds * GFH_NP;
    const double* __restrict__ DP = dpars + (i64)ds * GFH_NP;        // delta1 scattered per dataset
    const i64 base = (i64)t * GFH_TILE + threadIdx.x;
#pragma unroll
    for (int q = 0; q < GFH_PPL; q++) {
      const i64 i = base + (i64)q * GFH_BLOCK;
      omega[i] = -gfh_point_dd(x[i], P, DP) * w[i];                  // gadfit.F90:722-723
    }
  }
}
)";
  *src = s.str();
  return true;
}

}  // namespace gfh

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
#include "../../include/gadfit_gk_tables.h"
#include <cmath>
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <functional>
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
        case GFH_AUX:
          // (inside an integrand too: a real of the enclosing eval() that the integrand takes without passing it through pars(:))
          if (n.a < 0 || n.a >= t->n_aux) { *err = "auxiliary column out of range"; return false; }
          break;
        case GFH_PARAM: if (n.a < 0 || n.a >= n_pars) { *err = "parameter index out of range"; return false; } break;
        case GFH_IPARAM: if (n.a < 0) { *err = "bad integrand parameter"; return false; } break;
        case GFH_LIFT: case GFH_NEG: case GFH_POWI: case GFH_VAL:
          if (bad_ref(n.a)) { *err = "operand refers forward"; return false; } break;
        case GFH_ADD: case GFH_SUB: case GFH_MUL: case GFH_DIV: case GFH_POW:
          if (bad_ref(n.a) || bad_ref(n.b)) { *err = "operand refers forward"; return false; } break;
        case GFH_INTEGRATE: if (n.a < 0 || n.a >= t->n_integrals) { *err = "bad integral index"; return false; } break;
        case GFH_GUARD_GT: case GFH_GUARD_LT:
          // (inside an integrand, s != 0: decided per evaluation of the integrand -- Model::alts, emit_family)
          if (bad_ref(n.a) || bad_ref(n.b)) { *err = "operand refers forward"; return false; }
          break;
        default:
          if (n.op >= GFH_ABS && n.op <= GFH_ERF) { if (bad_ref(n.a)) { *err = "operand refers forward"; return false; } }
          else { *err = "unknown op code " + std::to_string(n.op); return false; }
      }
      o.nodes.push_back(d);
    }
    sub.push_back(std::move(o));
  }
  gk_points = t->gk_points ? t->gk_points : 15;
  n_aux = t->n_aux > 0 ? t->n_aux : 0;
  rel_error_outer = t->rel_error_outer; rel_error_inner = t->rel_error_inner;
  ws_size = t->ws_size > 0 ? t->ws_size : 1000;                       // NI:40 DEFAULT_WORKSPACE_SIZE
  ws_size_inner = t->ws_size_inner > 0 ? t->ws_size_inner : 1000;
  if (ws_size < 2 || ws_size_inner < 2) { *err = "quadrature workspace size must be at least 2"; return false; }
  // (any size the device's memory holds: workspaces beyond the scratch budget live in the context's global pool, plan_workspaces)
  if (ws_size > (1 << 22) || ws_size_inner > (1 << 22)) { *err = "quadrature workspace size beyond 4194304 intervals"; return false; }
  more_evals.clear(); hint_aux = -1; hint_cols.clear(); tape_variant.assign(1, 0);
  alts.assign(integrals.size(), {});
  // a guard has no value: nothing may use one as an operand, a bound, a binding or the result
  for (const SubTape& st : sub) {
    auto guard = [&](int k) { return k >= 0 && k < (int)st.nodes.size() && is_guard_op(st.nodes[(size_t)k].op); };
    bool bad = guard(st.result);
    for (const Node& nd : st.nodes) {
      switch (nd.op) {
        case GFH_CONST: case GFH_X: case GFH_AUX: case GFH_PARAM: case GFH_IVAR: case GFH_IPARAM: case GFH_GUARD_GT: case GFH_GUARD_LT: break;
        case GFH_INTEGRATE: {
          const Integral& in = integrals[(size_t)nd.a];
          if ((!in.lower_inf && guard(in.lower)) || (!in.upper_inf && guard(in.upper))) bad = true;
          for (int q = 0; q < in.n_ipars; q++) if (guard(ipar_nodes[(size_t)in.ipar_off + q])) bad = true;
          break;
        }
        case GFH_ADD: case GFH_SUB: case GFH_MUL: case GFH_DIV: case GFH_POW: if (guard(nd.a) || guard(nd.b)) bad = true; break;
        default: if (guard(nd.a)) bad = true; break;
      }
    }
    if (bad) { *err = "a comparison is used as a value"; return false; }
  }
  return true;
}

bool Model::has_guards() const {
  for (int v = 0; v < n_variants(); v++) for (const Node& nd : eval(v).nodes) if (is_guard_op(nd.op)) return true;
  return false;
}

namespace {
bool same_bits(double a, double b) { return memcmp(&a, &b, sizeof a) == 0; }
// same operation (guards: whatever their recorded outcome)
bool same_node(const Node& a, const Node& b) {
  return a.op == b.op && a.a == b.a && a.b == b.b && (a.flags & ~GFH_F_TAKEN) == (b.flags & ~GFH_F_TAKEN) && same_bits(a.c, b.c);
}
bool same_subtape(const SubTape& a, const SubTape& b) {
  if (a.result != b.result || a.nodes.size() != b.nodes.size()) return false;
  for (size_t k = 0; k < a.nodes.size(); k++) if (!same_node(a.nodes[k], b.nodes[k]) || a.nodes[k].flags != b.nodes[k].flags) return false;
  return true;
}
}  // namespace

// Further recorded paths of the same eval().  Their integrand sub-tapes and integrate() call sites join the pool of variant 0
// (sub[1..], integrals, ipar_nodes), identical ones shared -- so one generated device function serves every variant that calls
// it, and an INTEGRATE node of two variants is the same operation exactly when it carries the same pooled index.
bool Model::load_variants(int n, const gfh_tape* const* t, int hint, std::string* err, const std::vector<int32_t>* cols) {
  if (n < 1 || !t || !t[0]) { *err = "no variant"; return false; }
  if (!load(t[0], err)) return false;
  n_tapes = n;
  tape_variant.assign((size_t)n, 0);
  for (int v = 1; v < n; v++) {
    Model o;
    if (!t[v]) { *err = "null variant"; return false; }
    if (!o.load(t[v], err)) { *err = "variant " + std::to_string(v) + ": " + *err; return false; }
    if (o.n_pars != n_pars) { *err = "variants disagree on the number of parameters"; return false; }
    if (o.gk_points != gk_points || o.rel_error_outer != rel_error_outer || o.rel_error_inner != rel_error_inner ||
        o.ws_size != ws_size || o.ws_size_inner != ws_size_inner) { *err = "variants disagree on the quadrature settings"; return false; }
    n_aux = std::max(n_aux, o.n_aux);
    std::vector<int> sub_map(o.sub.size(), -1), int_map(o.integrals.size(), -1);
    std::vector<char> busy(o.integrals.size(), 0);
    bool ok = true;
    // pooled index of the variant's integral i (its integrand pooled first; integrands may nest call sites: depth <= 2, NI:70)
    std::function<int(int)> pool_integral;
    auto pool_sub = [&](int s_) -> int {
      if (sub_map[(size_t)s_] >= 0) return sub_map[(size_t)s_];
      SubTape st = o.sub[(size_t)s_];
      for (Node& nd : st.nodes) if (nd.op == GFH_INTEGRATE) { nd.a = pool_integral(nd.a); if (nd.a < 0) return -1; }
      for (size_t k = 1; k < sub.size(); k++) if (same_subtape(sub[k], st)) return sub_map[(size_t)s_] = (int)k;
      sub.push_back(std::move(st));
      return sub_map[(size_t)s_] = (int)sub.size() - 1;
    };
    pool_integral = [&](int i) -> int {
      if (int_map[(size_t)i] >= 0) return int_map[(size_t)i];
      if (busy[(size_t)i]) { ok = false; *err = "recursive integrate() call site"; return -1; }
      busy[(size_t)i] = 1;
      Integral in = o.integrals[(size_t)i];
      in.integrand = pool_sub(in.integrand);
      busy[(size_t)i] = 0;
      if (in.integrand < 0) return -1;
      const int32_t* binds = o.ipar_nodes.data() + in.ipar_off;
      for (size_t k = 0; k < integrals.size(); k++) {
        const Integral& e = integrals[k];
        if (e.integrand == in.integrand && e.lower == in.lower && e.upper == in.upper && e.lower_inf == in.lower_inf && e.upper_inf == in.upper_inf &&
            e.n_ipars == in.n_ipars && e.depth == in.depth && same_bits(e.rel_error, in.rel_error) && same_bits(e.abs_error, in.abs_error) &&
            std::equal(binds, binds + in.n_ipars, ipar_nodes.begin() + e.ipar_off))
          return int_map[(size_t)i] = (int)k;
      }
      const int off = (int)ipar_nodes.size();
      ipar_nodes.insert(ipar_nodes.end(), binds, binds + in.n_ipars);
      in.ipar_off = off;
      integrals.push_back(in);
      return int_map[(size_t)i] = (int)integrals.size() - 1;
    };
    SubTape ev = o.sub[0];
    for (Node& nd : ev.nodes) if (nd.op == GFH_INTEGRATE) { nd.a = pool_integral(nd.a); if (nd.a < 0 || !ok) { if (err->empty()) *err = "bad integrate() call site"; return false; } }
    alts.resize(integrals.size());
    bool dup = false;
    for (int w = 0; w < n_variants() && !dup; w++) dup = same_subtape(eval(w), ev);
    if (dup) { *err = "variant " + std::to_string(v) + " repeats an earlier one"; return false; }
    // the same path through eval() as an earlier variant, with an integrand that took another path through ITS comparisons (the
    // call sites agree in everything but the integrand's sub-tape): not a variant of eval() but a further recording of that
    // integrand.  An integrand that calls integrate() itself is compared the same way, node by node (so the recordings of an INNER
    // integrand that compares AD variables end up at the inner call site).
    std::function<bool(int, int, std::vector<std::pair<int, int>>&)> same_site = [&](int Ia, int Ib, std::vector<std::pair<int, int>>& add) -> bool {
      if (Ia == Ib) return true;
      const Integral &x = integrals[(size_t)Ia], &y = integrals[(size_t)Ib];
      const bool site = x.lower == y.lower && x.upper == y.upper && x.lower_inf == y.lower_inf && x.upper_inf == y.upper_inf &&
                        x.n_ipars == y.n_ipars && x.depth == y.depth && same_bits(x.rel_error, y.rel_error) && same_bits(x.abs_error, y.abs_error) &&
                        std::equal(ipar_nodes.begin() + x.ipar_off, ipar_nodes.begin() + x.ipar_off + x.n_ipars, ipar_nodes.begin() + y.ipar_off);
      if (!site) return false;
      if (x.integrand == y.integrand) return true;
      const SubTape &sa = sub[(size_t)x.integrand], &sb = sub[(size_t)y.integrand];
      // the same recording of the integrand up to call sites inside it that are themselves the same site?
      if (sa.result == sb.result && sa.nodes.size() == sb.nodes.size()) {
        std::vector<std::pair<int, int>> inner;
        bool same = true, any_int = false;
        for (size_t k = 0; k < sa.nodes.size() && same; k++) {
          const Node &p = sa.nodes[k], &q = sb.nodes[k];
          if (same_node(p, q) && p.flags == q.flags) continue;
          if (p.op == GFH_INTEGRATE && q.op == GFH_INTEGRATE && p.b == q.b && p.flags == q.flags && same_site(p.a, q.a, inner)) { any_int = true; continue; }
          same = false;
        }
        if (same && any_int) { add.insert(add.end(), inner.begin(), inner.end()); return true; }
      }
      add.push_back({Ia, y.integrand});                  // another path through this integrand's own comparisons
      return true;
    };
    bool joined = false;
    for (int w = 0; w < n_variants() && !joined; w++) {
      const SubTape& e = eval(w);
      if (e.result != ev.result || e.nodes.size() != ev.nodes.size()) continue;
      std::vector<std::pair<int, int>> add;
      bool same = true, any_int = false;
      for (size_t k = 0; k < e.nodes.size() && same; k++) {
        const Node &p = e.nodes[k], &q = ev.nodes[k];
        if (same_node(p, q) && p.flags == q.flags) continue;
        if (p.op == GFH_INTEGRATE && q.op == GFH_INTEGRATE && p.b == q.b && p.flags == q.flags && p.a != q.a && same_site(p.a, q.a, add)) { any_int = true; continue; }
        same = false;
      }
      if (!same || !any_int) continue;
      for (auto& d : add) {
        std::vector<int32_t>& f = alts[(size_t)d.first];
        if (d.second != integrals[(size_t)d.first].integrand && std::find(f.begin(), f.end(), (int32_t)d.second) == f.end()) f.push_back((int32_t)d.second);
      }
      joined = true;
      tape_variant[(size_t)v] = w;
    }
    if (joined) continue;
    more_evals.push_back(std::move(ev));
    tape_variant[(size_t)v] = n_variants() - 1;
  }
  alts.resize(integrals.size());
  if (hint >= n_aux) { *err = "the per-point variant column lies outside the auxiliary columns"; return false; }
  hint_aux = hint < 0 ? -1 : hint;
  hint_cols.clear();
  if (cols && hint_aux >= 0 && (int)cols->size() == n) {
    for (int32_t cidx : *cols) if (cidx < 0 || cidx >= n_aux) { *err = "a per-point variant column lies outside the auxiliary columns"; return false; }
    hint_cols = *cols;
  }
  return true;
}

int Model::hint_col_of_variant(int v) const {
  if (hint_cols.empty()) return hint_aux;
  for (size_t t = 0; t < tape_variant.size(); t++) if (tape_variant[t] == v) return hint_cols[t];
  return hint_aux;
}
std::vector<int> Model::tapes_of_variant(int v) const {
  std::vector<int> out;
  for (size_t t = 0; t < tape_variant.size(); t++) if (tape_variant[t] == v) out.push_back((int)t);
  if (out.empty()) out.push_back(v);          // (a model set through gfh_set_model: tape 0 = variant 0)
  return out;
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
    case GFH_ABS: return "fabs"; case GFH_EXP: return "gfh_exp"; case GFH_SQRT: return "sqrt";
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
  bool fast_div = true;             // one reciprocal per distinct denominator, products elsewhere
  std::vector<char> inv_done;

  Gen(const Model& mm, const SubTape& s, bool fast) : m(mm), st(s), fast_div(fast), inv_done(s.nodes.size(), 0) {}

  // n<k> = 1/v<k>, emitted at first use (straight-line code: v<k> is already defined)
  std::string inv(int k) {
    if (!inv_done[k]) { o << ind << "const double n" << k << " = 1.0 / " << v(k) << ";\n"; inv_done[k] = 1; }
    return "n" + std::to_string(k);
  }

  std::string v(int k) const { return "v" + std::to_string(k); }
  std::string b(int k) const { return "b" + std::to_string(k); }
  std::string d(int k) const { return "d" + std::to_string(k); }
  std::string dd(int k) const { return "e" + std::to_string(k); }

  // par_active: activity of PARAM nodes (eval tape) or of IPARAM nodes (integrand tapes)
  bool ivar_active = false;
  // the body of an integrand: the data point's abscissa and auxiliary columns (GFH_X / GFH_AUX nodes inside a sub-tape: a real the
  // user's integrand takes from the enclosing eval() without passing it through pars(:), numerical_integration.F90:195-201 evaluates the
  // integrand afresh in that scope) are not among its arguments -- they are read back from the lane's slot of the LDS stash that
  // the point functions fill (GFH_LANE_STASH)
  bool in_integrand = false;
  int mode = 0;   // 0 value, 1 reverse (grad), 2 forward (val,d,dd)
  // Mesh hand-over (emit_integral_site): the body of a point function hands every outermost single-piece integrate() call
  // site its 64-byte record of the lane's mesh; bodies of integrands and the selector pass none.
  bool mesh_top = false;
  int mesh_next = 0;
  std::string mesh_args(const Integral& in) {
    if (!mesh_top || in.depth > 1 || (in.lower_inf && in.upper_inf) || mesh_next >= kMeshSitesMax) return "nullptr, 0";
    return "MESH ? MESH + " + std::to_string(kMeshRecord * mesh_next++) + " : nullptr, MESH_MODE";
  }
  void analyse(const std::vector<char>& par_active) {
    int n = (int)st.nodes.size();
    is_real.assign(n, 0); act.assign(n, 0);
    for (int k = 0; k < n; k++) {
      const Node& nd = st.nodes[k];
      is_real[k] = (nd.flags & GFH_F_REAL) ? 1 : 0;
      switch (nd.op) {
        case GFH_PARAM: case GFH_IPARAM: act[k] = nd.a < (int)par_active.size() ? par_active[nd.a] : 0; break;
        case GFH_IVAR: act[k] = ivar_active; break;
        case GFH_INTEGRATE: {
          const Integral& in = m.integrals[nd.a];
          char a = 0;
          for (int j = 0; j < in.n_ipars; j++) a |= act[m.ipar_nodes[in.ipar_off + j]];
          if (!in.lower_inf) a |= act[in.lower];
          if (!in.upper_inf) a |= act[in.upper];
          act[k] = a;
          break;
        }
        case GFH_CONST: case GFH_X: case GFH_AUX: act[k] = 0; break;
        case GFH_GUARD_GT: case GFH_GUARD_LT: act[k] = 0; break;      // a comparison of values (AD:315-395): no value, no derivative
        case GFH_LIFT: act[k] = 0; break;
        case GFH_VAL: act[k] = 0; is_real[k] = 1; break;      // the %val of an advar: a plain real whatever the flags of a third-party tape say (no derivative flows through it)
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

  void emit_values(bool) {
    int n = (int)st.nodes.size();
    for (int k = 0; k < n; k++) emit_value_node(k);
  }

  // forward mode: value and (d, dd) of each node together, in tape order (an integrate()
  // call needs the tangents of its bindings when it is evaluated)
  void emit_forward_all() {
    int n = (int)st.nodes.size();
    for (int k = 0; k < n; k++) { emit_value_node(k); if (act[k]) emit_dd_node(k); }
  }

  void emit_value_node(int k) {
    {
      const Node& nd = st.nodes[k];
      std::string lhs = ind + "const double " + v(k) + " = ";
      switch (nd.op) {
        case GFH_CONST: o << lhs << lit(nd.c) << ";\n"; break;
        case GFH_GUARD_GT: case GFH_GUARD_LT: break;                   // decided by gfh_select before this body runs
        case GFH_X: o << lhs << (in_integrand ? "gfh_lane_x[threadIdx.x];\n" : "X;\n"); break;
        case GFH_AUX:
          if (in_integrand) o << lhs << "gfh_lane_axp[threadIdx.x][(i64)" << nd.a << " * gfh_lane_lda];\n";
          else o << lhs << "AXP[(i64)" << nd.a << " * LDA];\n";
          break;
        case GFH_PARAM: o << lhs << "P[" << nd.a << "];\n"; break;
        case GFH_IVAR: o << lhs << "T;\n"; break;
        case GFH_IPARAM: o << lhs << "Q[" << nd.a << "];\n"; break;
        case GFH_INTEGRATE: emit_integrate_call(k); break;
        case GFH_LIFT: case GFH_VAL: o << lhs << v(nd.a) << ";\n"; break;
        case GFH_NEG: o << lhs << "-" << v(nd.a) << ";\n"; break;
        case GFH_ADD: o << lhs << v(nd.a) << " + " << v(nd.b) << ";\n"; break;
        case GFH_SUB: o << lhs << v(nd.a) << " - " << v(nd.b) << ";\n"; break;
        case GFH_MUL: o << lhs << v(nd.a) << " * " << v(nd.b) << ";\n"; break;
        case GFH_DIV: {
          int var = variant(nd, k);
          if (fast_div && !is_real[k]) {
            // reciprocal reuse: every division by v<b> (here and in the derivative code)
            // shares one 1/v<b>; differs from the reference's r/a = r/v (AD:892-913) by <= 1 ulp
            std::string nb = inv(nd.b);
            o << lhs << v(nd.a) << " * " << nb << ";\n";
          } else if (var == 1 || var == 2) {   // AD:814-841, 843-866: multiply by the reciprocal
            o << ind << "const double i" << k << " = 1.0 / " << v(nd.b) << ";\n";
            o << lhs << v(nd.a) << " * i" << k << ";\n";
          } else o << lhs << v(nd.a) << " / " << v(nd.b) << ";\n";   // AD:892-913, 839
          break;
        }
        case GFH_POW:
          if (fast_div && !is_real[k]) {
            // x**y together with ln x (the derivative code wants it: AD:975-980, 1534-1553): one extended-precision logarithm serves both
            o << ind << "double l" << k << ";\n";
            o << lhs << "gfh_pow_ln(" << v(nd.a) << ", " << v(nd.b) << ", l" << k << ");\n";
          } else o << lhs << "pow(" << v(nd.a) << ", " << v(nd.b) << ");\n";
          break;
        case GFH_POWI: { std::string e = powi_expr(v(nd.a), nd.b, "q" + std::to_string(k)); o << lhs << e << ";\n"; break; }
        default: o << lhs << fn_name(nd.op) << "(" << v(nd.a) << ");\n"; break;
      }
    }
  }

  // integrate(f, pars, lower, upper) call site (NI:193-630): the adaptive rule lives in the
  // generated gfh_int<I>_* functions; here the bindings are gathered and, in reverse mode,
  // the partials w.r.t. pars(:) and f at the bounds are kept for the return sweep.
  void emit_integrate_call(int k) {
    const Node& nd = st.nodes[k];
    const Integral& in = m.integrals[nd.a];
    const std::string I = std::to_string(nd.a), ks = std::to_string(k);
    o << ind << "double q" << ks << "[" << (in.n_ipars > 0 ? in.n_ipars : 1) << "];\n";
    for (int j = 0; j < in.n_ipars; j++) o << ind << "q" << ks << "[" << j << "] = " << v(m.ipar_nodes[in.ipar_off + j]) << ";\n";
    const std::string lo = in.lower_inf ? "0.0" : v(in.lower), hi = in.upper_inf ? "0.0" : v(in.upper);
    if (mode == 1 && act[k]) {
      o << ind << "double " << v(k) << ", g" << ks << "[" << (in.n_ipars > 0 ? in.n_ipars : 1) << "], fl" << ks << ", fh" << ks << ";\n";
      o << ind << "gfh_int" << I << "_grad<" << ((!in.lower_inf && act[in.lower]) ? "true" : "false") << ", " << ((!in.upper_inf && act[in.upper]) ? "true" : "false") << ">(" << lo << ", " << hi << ", q" << ks << ", " << v(k) << ", g" << ks << ", fl" << ks << ", fh" << ks << ", STATUS, " << mesh_args(in) << ");\n";
    } else if (mode == 2 && act[k]) {
      // forward mode (NI:425-437, 480-487, 527-534): tangents of pars(:) and of the bounds go in
      const int NQ = in.n_ipars > 0 ? in.n_ipars : 1;
      o << ind << "double qd" << ks << "[" << NQ << "], qe" << ks << "[" << NQ << "];\n";
      for (int j = 0; j < in.n_ipars; j++) {
        int bn = m.ipar_nodes[in.ipar_off + j];
        o << ind << "qd" << ks << "[" << j << "] = " << (act[bn] ? d(bn) : "0.0") << "; qe" << ks << "[" << j << "] = " << (act[bn] ? dd(bn) : "0.0") << ";\n";
      }
      const bool la = !in.lower_inf && act[in.lower], ua = !in.upper_inf && act[in.upper];
      o << ind << "double " << v(k) << ", " << d(k) << ", " << dd(k) << ";\n";
      o << ind << "gfh_int" << I << "_fwd<" << (la ? "true" : "false") << ", " << (ua ? "true" : "false") << ">(" << lo << ", "
        << (la ? d(in.lower) : "0.0") << ", " << (la ? dd(in.lower) : "0.0") << ", " << hi << ", " << (ua ? d(in.upper) : "0.0") << ", "
        << (ua ? dd(in.upper) : "0.0") << ", q" << ks << ", qd" << ks << ", qe" << ks << ", " << v(k) << ", " << d(k) << ", " << dd(k) << ", STATUS, " << mesh_args(in) << ");\n";
    } else {
      o << ind << "const double " << v(k) << " = gfh_int" << I << "_val(" << lo << ", " << hi << ", q" << ks << ", STATUS, " << mesh_args(in) << ");\n";
    }
  }

  // ---------------------------------------------------------------- reverse sweep, AD:1476-1659
  void emit_reverse() {
    int n = (int)st.nodes.size();
    for (int k = 0; k < n; k++) if (act[k]) o << ind << "double " << b(k) << " = 0.0;\n";
    if (!act[st.result]) return;
    o << ind << b(st.result) << " = 1.0;\n";                                   // AD:1490
    // The reference zeroes the adjoints and accumulates (AD:1482-1490).  The code is straight-line, so which contribution
    // to an adjoint is the first one is known here: it is assigned instead of added to 0.0 -- the same value (0.0 + t = t
    // for every t except t = -0.0, which it turns into +0.0: a Jacobian entry that is a zero keeps the sign of its last
    // factor), one FP64 add less per adjoint, which the compiler may not drop by itself under IEEE rules.
    std::vector<char> touched((size_t)n, 0);
    touched[(size_t)st.result] = 1;
    auto acc = [&](int tgt, const std::string& sign, const std::string& expr) {
      if (!touched[(size_t)tgt]) {
        touched[(size_t)tgt] = 1;
        o << ind << b(tgt) << " = " << (sign == "-" ? "-(" + expr + ")" : expr) << ";\n";
        return;
      }
      o << ind << b(tgt) << " = " << b(tgt) << " " << sign << " " << expr << ";\n";
    };
    for (int k = n - 1; k >= 0; k--) {
      if (!act[k]) continue;
      const Node& nd = st.nodes[k];
      const std::string bk = b(k);
      switch (nd.op) {
        case GFH_PARAM: case GFH_IPARAM: case GFH_IVAR: break;
        case GFH_INTEGRATE: {
          const Integral& in = m.integrals[nd.a];
          const std::string ks = std::to_string(k);
          for (int j = 0; j < in.n_ipars; j++) {
            int bn = m.ipar_nodes[in.ipar_off + j];
            if (act[bn]) acc(bn, "+", bk + "*g" + ks + "[" + std::to_string(j) + "]");
          }
          if (!in.upper_inf && act[in.upper]) acc(in.upper, "+", bk + "*fh" + ks);   // AD:1650-1653, 1638-1639
          if (!in.lower_inf && act[in.lower]) acc(in.lower, "-", bk + "*fl" + ks);   // AD:1645-1648, 1641-1642
          break;
        }
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
          if (fast_div) {
            std::string nb = inv(nd.b);
            if (var == 1 || var == 2) acc(nd.a, "+", bk + "*" + nb);            // AD:1521-1523, 1516-1520
            if (var == 1 || var == 3) acc(nd.b, "-", bk + "*" + v(k) + "*" + nb); // AD:1524-1533 (c/v/v = y/v)
          } else if (var == 1) {                                                // AD:1521-1527
            acc(nd.a, "+", bk + "/" + v(nd.b));
            acc(nd.b, "-", bk + "*" + v(k) + "/" + v(nd.b));
          } else if (var == 2) acc(nd.a, "+", bk + "*i" + std::to_string(k));   // AD:1516-1520, const = inv
          else acc(nd.b, "-", bk + "*" + v(nd.a) + "/" + v(nd.b) + "/" + v(nd.b));  // AD:1528-1533
          break;
        }
        case GFH_POW: {
          int var = variant(nd, k);
          // (shared reciprocals: x**(y-1) = x**y / x, and ln x came with the value -- one pow and one log less per node than
          // the reference's expressions, <= 2 ulp from them)
          const std::string pm1 = fast_div ? "(" + v(k) + "*" + (var == 3 ? std::string() : inv(nd.a)) + ")" : "pow(" + v(nd.a) + ", " + v(nd.b) + " - 1.0)";
          const std::string lg = fast_div ? "l" + std::to_string(k) : "log(" + v(nd.a) + ")";
          if (var == 1) {                                                       // AD:1534-1541
            acc(nd.a, "+", bk + "*" + v(nd.b) + "*" + pm1);
            acc(nd.b, "+", bk + "*" + lg + "*" + v(k));                       // x1**x2 recomputed = y
          } else if (var == 2)                                                  // AD:1542-1547
            acc(nd.a, "+", bk + "*" + v(nd.b) + "*" + pm1);
          else                                                                  // AD:1548-1553
            acc(nd.b, "+", bk + "*" + lg + "*" + v(k));
          break;
        }
        case GFH_POWI: {                                                        // AD:1554-1558
          std::string e = powi_expr(v(nd.a), nd.b - 1, "r" + std::to_string(k));
          acc(nd.a, "+", bk + "*" + lit((double)nd.b) + "*" + e);
          break;
        }
        case GFH_ABS:                                                           // AD:1559-1567
          touched[(size_t)nd.a] = 1;
          o << ind << b(nd.a) << " = (" << v(nd.a) << " < 0.0) ? " << b(nd.a) << " - " << bk << " : " << b(nd.a) << " + " << bk << ";\n";
          break;
        case GFH_EXP: acc(nd.a, "+", bk + "*" + v(k)); break;                  // AD:1568-1571
        case GFH_SQRT: if (fast_div) { std::string nk = inv(k); acc(nd.a, "+", bk + "*0.5*" + nk); }
                       else acc(nd.a, "+", bk + "/2.0/" + v(k));
                       break;                                                   // AD:1572-1575
        case GFH_LOG: if (fast_div) { std::string na_ = inv(nd.a); acc(nd.a, "+", bk + "*" + na_); }
                      else acc(nd.a, "+", bk + "/" + v(nd.a));
                      break;                                                    // AD:1576-1579
        case GFH_SIN: acc(nd.a, "+", bk + "*cos(" + v(nd.a) + ")"); break;     // AD:1581-1584
        case GFH_COS: acc(nd.a, "-", bk + "*sin(" + v(nd.a) + ")"); break;     // AD:1585-1588
        case GFH_TAN: o << ind << "const double rc" << k << " = cos(" << v(nd.a) << ");\n";
                      acc(nd.a, "+", bk + "/(rc" + std::to_string(k) + "*rc" + std::to_string(k) + ")"); break;   // AD:1589-1592
        case GFH_ASIN: acc(nd.a, "+", bk + "/sqrt(1.0 - " + v(nd.a) + "*" + v(nd.a) + ")"); break;  // AD:1593-1596
        case GFH_ACOS: acc(nd.a, "-", bk + "/sqrt(1.0 - " + v(nd.a) + "*" + v(nd.a) + ")"); break;  // AD:1597-1600
        case GFH_ATAN: acc(nd.a, "+", bk + "/(1.0 + " + v(nd.a) + "*" + v(nd.a) + ")"); break;      // AD:1601-1604
        case GFH_SINH: acc(nd.a, "+", bk + "*cosh(" + v(nd.a) + ")"); break;   // AD:1606-1609
        case GFH_COSH: acc(nd.a, "+", bk + "*sinh(" + v(nd.a) + ")"); break;   // AD:1610-1613
        case GFH_TANH: o << ind << "const double rc" << k << " = cosh(" << v(nd.a) << ");\n";
                       acc(nd.a, "+", bk + "/(rc" + std::to_string(k) + "*rc" + std::to_string(k) + ")"); break;  // AD:1614-1617
        case GFH_ASINH: acc(nd.a, "+", bk + "/sqrt(" + v(nd.a) + "*" + v(nd.a) + " + 1.0)"); break; // AD:1618-1621
        case GFH_ACOSH: acc(nd.a, "+", bk + "/sqrt(" + v(nd.a) + "*" + v(nd.a) + " - 1.0)"); break; // AD:1622-1625
        case GFH_ATANH: acc(nd.a, "+", bk + "/(1.0 - " + v(nd.a) + "*" + v(nd.a) + ")"); break;     // AD:1626-1629
        case GFH_ERF: acc(nd.a, "+", bk + "*" + TWO_OVER_SQRTPI + "*gfh_exp(-(" + v(nd.a) + "*" + v(nd.a) + "))"); break; // AD:1631-1635
        default: break;
      }
    }
  }

  // ---------------------------------------------------------------- forward mode (val,d,dd)
  // Active nodes carry d<k> and e<k> (= dd).  Formulas: the `else` branches of AD:454-1459.
  void emit_dd_node(int k) {
    {
      const Node& nd = st.nodes[k];
      auto D = [&](const std::string& e) { o << ind << "const double " << d(k) << " = " << e << ";\n"; };
      auto E = [&](const std::string& e) { o << ind << "const double " << dd(k) << " = " << e << ";\n"; };
      const bool leaf = nd.op == GFH_PARAM || nd.op == GFH_IPARAM || nd.op == GFH_IVAR || nd.op == GFH_INTEGRATE;
      const std::string va = nd.a >= 0 && !leaf ? v(nd.a) : "", y = v(k);
      const std::string da = nd.a >= 0 && !leaf ? d(nd.a) : "", ea = nd.a >= 0 && !leaf ? dd(nd.a) : "";
      std::string ks = std::to_string(k);
      switch (nd.op) {
        case GFH_PARAM: D("DP[" + std::to_string(nd.a) + "]"); E("0.0"); break;   // gadfit.F90:719: %d = delta1, dd = 0
        case GFH_IPARAM: D("QD[" + std::to_string(nd.a) + "]"); E("QE[" + std::to_string(nd.a) + "]"); break;
        case GFH_IVAR: D("TD"); E("TE"); break;
        case GFH_INTEGRATE: break;   // d/dd were produced together with the value (emit_integrate_call)
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
          if (fast_div) {
            std::string nb = inv(nd.b);
            if (var == 1) {
              D("(" + d(nd.a) + " - " + y + "*" + d(nd.b) + ")*" + nb);
              E("(" + dd(nd.a) + " - " + y + "*" + dd(nd.b) + " - 2.0*" + d(k) + "*" + d(nd.b) + ")*" + nb);
            } else if (var == 2) { D(d(nd.a) + "*" + nb); E(dd(nd.a) + "*" + nb); }
            else {
              D("-" + y + "*" + d(nd.b) + "*" + nb);
              E("(-" + y + "*" + dd(nd.b) + " - 2.0*" + d(k) + "*" + d(nd.b) + ")*" + nb);
            }
          } else if (var == 1) {                                                                // AD:830-831
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
            if (!fast_div) {
              o << ind << "const double l" << ks << " = log(" << x1 << ");\n";
              D(y + "*" + d(nd.b) + "*l" + ks + " + " + d(nd.a) + "*" + x2 + "*pow(" + x1 + ", " + x2 + " - 1.0)");
            } else D(y + "*" + d(nd.b) + "*l" + ks + " + " + d(nd.a) + "*" + x2 + "*(" + y + "*" + inv(nd.a) + ")");
            o << ind << "const double i" << ks << " = 1.0 / " << x1 << ";\n";
            E(d(k) + "*" + d(k) + "/" + y + " + " + y + "*(" + dd(nd.b) + "*l" + ks + " + (2.0*" + d(nd.b) + "*" + d(nd.a) +
              " + " + x2 + "*(" + dd(nd.a) + " - " + d(nd.a) + "*" + d(nd.a) + "*i" + ks + "))*i" + ks + ")");
          } else if (var == 2) {                                                                // AD:1005-1008
            if (!fast_div) D(d(nd.a) + "*" + x2 + "*pow(" + x1 + ", " + x2 + " - 1.0)");
            else D(d(nd.a) + "*" + x2 + "*(" + y + "*" + inv(nd.a) + ")");
            o << ind << "const double i" << ks << " = 1.0 / " << x1 << ";\n";
            E(d(k) + "*" + d(k) + "/" + y + " + " + y + "*" + x2 + "*(" + dd(nd.a) + " - " + d(nd.a) + "*" + d(nd.a) + "*i" + ks + ")*i" + ks);
          } else {                                                                              // AD:1076-1079
            if (!fast_div) o << ind << "const double l" << ks << " = log(" << x1 << ");\n";
            D(y + "*" + d(nd.b) + "*l" + ks);
            E(d(k) + "*" + d(k) + "/" + y + " + " + y + "*" + dd(nd.b) + "*l" + ks);
          }
          break;
        }
        case GFH_POWI: {                                                                        // AD:1051-1054
          std::string nn = lit((double)nd.b);
          if (fast_div && nd.b >= 1) {
            // x**n, n >= 1, without the reference's two divisions (d = y n dx / x, dd = d d / y + y n (ddx - dx dx / x) / x):
            // d = n x^(n-1) dx, dd = n ((n-1) x^(n-2) dx dx + x^(n-1) ddx) -- the same polynomial identities, finite at x = 0
            // where the reference's forward mode returns 0/0 (its reverse mode, AD:1554-1558, has the product form as well)
            if (nd.b == 1) { D(da); E(ea); break; }
            const std::string p1 = powi_expr(va, nd.b - 1, "f" + ks);
            D(nn + "*" + p1 + "*" + da);
            if (nd.b == 2) E(nn + "*(" + da + "*" + da + " + " + va + "*" + ea + ")");
            else {
              const std::string p2 = powi_expr(va, nd.b - 2, "g" + ks);
              E(nn + "*(" + lit((double)(nd.b - 1)) + "*" + p2 + "*" + da + "*" + da + " + " + p1 + "*" + ea + ")");
            }
            break;
          }
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
          o << ind << "const double t" << ks << " = " << TWO_OVER_SQRTPI << "*gfh_exp(-(" << va << "*" << va << "));\n";
          D(da + "*t" + ks); E("(" + ea + " - 2.0*" + da + "*" + da + "*" + va + ")*t" + ks); break;
        default: break;
      }
    }
  }
};


// ---------------------------------------------------------------------------------------
// integrate(): adaptive Gauss-Kronrod through AD on the device (numerical_integration.F90).
// Per integrand sub-tape S:   gfh_s<S>_val(T, Q)          value, everything passive (NI:238-239)
//                             gfh_s<S>_grad(T, Q, F, GQ)   value + d/dpars(:) by the unrolled reverse sweep
// Per call site I:            gfh_int<I>_val / _grad       interval bisection on values (NI:251-267), then
//                             the final pass over the intervals in storage order (NI:268-275)
// The per-lane interval workspace lives in scratch: GFH_WS1 intervals for outer, GFH_WS2 for inner integrals.  The reference's
// workspaces are user-sized, default 1000 (NI:40, 114-135); typical use is << 100, so the kernels are first compiled with
// min(100, the user's size) and a pass that exhausts that (STATUS = 1) is repeated by the host with kernels compiled at the
// user's size before the reference's error is raised (NI:282-283; context.cpp, grow_workspace).
void emit_integrand_functions(const Model& m, int S, const GenConfig& cfg, std::ostringstream& s) {
  const SubTape& st = m.sub[S];
  int nip = 0;
  for (const Node& nd : st.nodes) if (nd.op == GFH_IPARAM && nd.a + 1 > nip) nip = nd.a + 1;
  std::vector<char> none(nip > 0 ? nip : 1, 0), all(nip > 0 ? nip : 1, 1);
  s << "static __device__ double gfh_s" << S << "_val(const double T, const double* __restrict__ Q, int* STATUS) {\n";
  { Gen g(m, st, cfg.fast_div); g.in_integrand = true; g.mode = 0; g.analyse(none); g.emit_values(false); s << g.o.str() << "  return " << g.v(st.result) << ";\n}\n\n"; }
  s << "static __device__ void gfh_s" << S << "_grad(const double T, const double* __restrict__ Q, double& F, double* __restrict__ GQ, int* STATUS) {\n";
  {
    Gen g(m, st, cfg.fast_div); g.in_integrand = true; g.mode = 1; g.analyse(all); g.emit_values(false); g.emit_reverse();
    s << g.o.str() << "  F = " << g.v(st.result) << ";\n";
    for (int j = 0; j < nip; j++) {
      std::string e;
      for (int k = 0; k < (int)st.nodes.size(); k++)
        if (st.nodes[k].op == GFH_IPARAM && st.nodes[k].a == j && g.act[k]) e += (e.empty() ? "" : " + ") + g.b(k);
      s << "  GQ[" << j << "] = " << (e.empty() ? "0.0" : e) << ";\n";
    }
    s << "}\n\n";
  }
  // forward mode: (val, d, dd) with tangents QD/QE of pars(:); TA = the integration variable
  // itself carries (TD, TE) -- only needed for f(bound) with an active bound (NI:431, 435)
  for (int ta = 0; ta < 2; ta++) {
    s << "static __device__ void gfh_s" << S << (ta ? "_fwdT" : "_fwd") << "(const double T, const double TD, const double TE, "
         "const double* __restrict__ Q, const double* __restrict__ QD, const double* __restrict__ QE, double& F, double& FD, double& FE, int* STATUS) {\n";
    Gen g(m, st, cfg.fast_div); g.in_integrand = true; g.mode = 2; g.ivar_active = ta != 0; g.analyse(all); g.emit_forward_all();
    s << g.o.str() << "  F = " << g.v(st.result) << ";\n";
    if (g.act[st.result]) s << "  FD = " << g.d(st.result) << "; FE = " << g.dd(st.result) << ";\n";
    else s << "  FD = 0.0; FE = 0.0;\n";
    s << "}\n\n";
  }
}

void emit_integral_site(const Model& m, int I, const GenConfig& cfg, std::ostringstream& s) {
  const Integral& in = m.integrals[I];
  const int S = in.integrand, NQ = in.n_ipars > 0 ? in.n_ipars : 1;
  const double rel = in.rel_error >= 0 ? in.rel_error : (in.depth <= 1 ? m.rel_error_outer : m.rel_error_inner);
  const double abst = in.abs_error >= 0 ? in.abs_error : 0.0;
  // (an integrand that compares AD variables: the dispatchers gfh_sf<I>_* of emit_family stand in for gfh_s<S>_*)
  bool fam = (size_t)I < m.alts.size() && !m.alts[(size_t)I].empty();
  for (const Node& nd : m.sub[(size_t)S].nodes) if (is_guard_op(nd.op)) fam = true;
  const std::string Is = std::to_string(I), Ss = fam ? "f" + std::to_string(I) : std::to_string(S);
  const std::string WS = in.depth <= 1 ? "GFH_WS1" : "GFH_WS2";      // outer / inner workspace (NI:220-226)
  // integrand with the (a,inf) / (-inf,b) maps applied (NI:314-318, 347-351): TK 0 none, 1: f(tb-1+1/t)/t**2, 2: f(tb+1-1/t)/t**2
  s << "template <int TK> static __device__ __forceinline__ double gfh_i" << Is << "_f(const double t, const double tb, const double* __restrict__ Q, int* STATUS) {\n"
       "  if (TK == 0) return gfh_s" << Ss << "_val(t, Q, STATUS);\n"
       "  const double arg = TK == 1 ? (tb - 1.0) + 1.0 / t : (tb + 1.0) - 1.0 / t;\n"
       "  return gfh_s" << Ss << "_val(arg, Q, STATUS) * (1.0 / (t * t));\n}\n";
  s << "template <int TK> static __device__ __forceinline__ void gfh_i" << Is << "_fg(const double t, const double tb, const double* __restrict__ Q, double& F, double* __restrict__ G, int* STATUS) {\n"
       "  if (TK == 0) { gfh_s" << Ss << "_grad(t, Q, F, G, STATUS); return; }\n"
       "  const double arg = TK == 1 ? (tb - 1.0) + 1.0 / t : (tb + 1.0) - 1.0 / t;\n"
       "  gfh_s" << Ss << "_grad(arg, Q, F, G, STATUS);\n"
       "  const double i2 = 1.0 / (t * t);\n  F *= i2;\n"
       "  for (int j = 0; j < " << NQ << "; j++) G[j] *= i2;\n}\n";
  // one Gauss-Kronrod panel on values (NI:636-664)
  s << "template <int TK> static __device__ double gfh_i" << Is << "_gk(const double lo, const double hi, const double tb, const double* __restrict__ Q, double& err, int* STATUS) {\n"
       "  const double scale = (hi - lo) / 2, shift = (lo + hi) / 2;\n  double sg = 0.0, y = 0.0;\n"
       "  for (int i = 1; i <= GFH_GK_N; i++) {\n"
       "    const double f = gfh_i" << Is << "_f<TK>(scale * gfh_gk_roots[i - 1] + shift, tb, Q, STATUS);\n"
       "    if ((i & 1) == 0) sg = sg + gfh_gk_wg[i / 2 - 1] * f;\n"
       "    y = y + gfh_gk_wk[i - 1] * f;\n  }\n"
       "  y = scale * y;\n  err = fabs(y - scale * sg);\n  return y;\n}\n";
  // the same panel with the gradient w.r.t. pars(:) of its Kronrod sum (unscaled, as the final pass of NI:268-275 forms it); the value
  // and the error estimate are the operations of _gk on the same integrand values, bit for bit
  s << "template <int TK> static __device__ double gfh_i" << Is << "_gkg(const double lo, const double hi, const double tb, const double* __restrict__ Q, double& err, double* __restrict__ GS, int* STATUS) {\n"
       "  const double scale = (hi - lo) / 2, shift = (lo + hi) / 2;\n  double sg = 0.0, y = 0.0;\n"
       "  for (int j = 0; j < " << NQ << "; j++) GS[j] = 0.0;\n"
       "  for (int i = 1; i <= GFH_GK_N; i++) {\n"
       "    double f, g[" << NQ << "];\n"
       "    gfh_i" << Is << "_fg<TK>(scale * gfh_gk_roots[i - 1] + shift, tb, Q, f, g, STATUS);\n"
       "    if ((i & 1) == 0) sg = sg + gfh_gk_wg[i / 2 - 1] * f;\n"
       "    y = y + gfh_gk_wk[i - 1] * f;\n"
       "    for (int j = 0; j < " << NQ << "; j++) GS[j] += gfh_gk_wk[i - 1] * g[j];\n  }\n"
       "  y = scale * y;\n  err = fabs(y - scale * sg);\n  return y;\n}\n";
  // The mesh of one piece: bisection on values (NI:251-267) -- or, when another pass at these very parameters has left the
  // record of its bisections (MM == 2: chi2() at the trial point before the sweep of the accepted step, the sweep before
  // STEP 3), their replay: the same midpoints in the same storage order without a single integrand evaluation, so
  // everything that follows sees bitwise the mesh a fresh bisection would build.  MM == 1: this pass leaves the record
  // (MS[0] = number of bisections or 255 = none, MS[1 + k] = the interval the k-th one split).
  // carry: the bisection evaluates every panel WITH the gradient of its Kronrod sum and keeps it per interval (gs[q][:]), so the
  // final pass over the intervals (NI:268-275) has nothing left to evaluate -- (2n - 1) panels with gradient instead of (2n - 1)
  // without plus n with.  The panel sums are the same operations on the same numbers whenever they are formed, so the result is
  // bitwise the two-phase one.  Only with the small workspace the kernels carry first (scratch: 8 NQ bytes more per interval).
  const int ws_value = in.depth <= 1 ? cfg.ws_size : cfg.ws_size_inner;
  const bool can_carry = carries_gradients(NQ, ws_value, cfg.ws_global);
  // the lane's workspace in the global pool: level 1 (outer integrals) first, level 2 behind it (GFH_WSG_L2)
  const std::string ws_decl = cfg.ws_global
      ? "  double* const wl_ = gfh_wsg_lane(" + std::string(in.depth <= 1 ? "0" : "GFH_WSG_L2") + ");\n  const long long row_ = " + std::string(in.depth <= 1 ? "GFH_WSG_ROW1" : "GFH_WSG_ROW2") +
        ";\n  const gfh_wsa lo{wl_, row_}, hi{wl_ + 64, row_}, er{wl_ + 128, row_}, sm{wl_ + 192, row_};\n"
      : "  double lo[" + WS + "], hi[" + WS + "], er[" + WS + "], sm[" + WS + "];\n";
  auto mesh_build = [&](bool need_sums, bool carry = false) {
    std::ostringstream b;
    // (pool form: the gradient of a panel goes to fields 4 .. 4 + NQ - 1 of the interval's row, through a local array the panel fills)
    const bool gpool = carry && cfg.ws_global;
    auto put = [&](const char* q) { return gpool ? std::string("; for (int j = 0; j < ") + std::to_string(NQ) + "; j++) wl_[(4 + j) * 64 + (long long)(" + q + ") * row_] = gl_[j]" : std::string(); };
    if (carry && !gpool) b << "  double gs[" << WS << "][" << NQ << "];\n  bool carried = false;\n";
    if (gpool) b << "  double gl_[" << NQ << "];\n  bool carried = false;\n";
    const std::string gk0 = carry ? "_gkg<TK>(lower, upper, tb, Q, er[0], " + std::string(gpool ? "gl_" : "gs[0]") + ", STATUS)" + put("0") : "_gk<TK>(lower, upper, tb, Q, er[0], STATUS)";
    const std::string gkm = carry ? "_gkg<TK>(aa, mid, tb, Q, er[mx], " + std::string(gpool ? "gl_" : "gs[mx]") + ", STATUS)" + put("mx") : "_gk<TK>(aa, mid, tb, Q, er[mx], STATUS)";
    const std::string gkn = carry ? "_gkg<TK>(mid, bb, tb, Q, er[n], " + std::string(gpool ? "gl_" : "gs[n]") + ", STATUS)" + put("n") : "_gk<TK>(mid, bb, tb, Q, er[n], STATUS)";
    b << ws_decl <<
         "  lo[0] = lower; hi[0] = upper;\n"
         "  int n = 1;\n"
         "  if (MM == 2 && MS && MS[0] != 255) {\n"
         "    const int ns = MS[0];\n"
         "    for (int k = 0; k < ns; k++) {\n"
         "      const int mx = MS[1 + k];\n"
         "      const double aa = lo[mx], bb = hi[mx], mid = (aa + bb) / 2;\n"
         "      hi[mx] = mid; lo[n] = mid; hi[n] = bb; n++;\n"
         "    }\n";
    if (need_sums) b << "    for (int q = 0; q < n; q++) sm[q] = gfh_i" << Is << "_gk<TK>(lo[q], hi[q], tb, Q, er[q], STATUS);\n";
    b << "  } else {\n"
         "    sm[0] = gfh_i" << Is << gk0 << ";\n"
         "    bool whole = true;\n"
         "    for (;;) {\n"
         "      if (n >= " << WS << ") { if (STATUS) GFH_RAISE(STATUS, 1); whole = false; break; }            // NI:282-283\n"
         "      int mx = 0;\n      for (int q = 1; q < n; q++) if (er[q] > er[mx]) mx = q;   // maxloc: first maximum\n"
         "      const double aa = lo[mx], bb = hi[mx], mid = (aa + bb) / 2;\n"
         "      sm[mx] = gfh_i" << Is << gkm << ";\n"
         "      sm[n] = gfh_i" << Is << gkn << ";\n"
         "      hi[mx] = mid; lo[n] = mid; hi[n] = bb;\n"
         "      if (MM == 1 && MS && n < " << kMeshRecord << ") MS[n] = (unsigned char)mx;\n"
         "      n++;\n"
         "      double es = 0.0, ss = 0.0;\n      for (int q = 0; q < n; q++) { es += er[q]; ss += sm[q]; }\n"
         "      if (es < " << lit(abst) << " || es / ss < " << lit(rel) << ") break;         // NI:264-267 (no abs() on the sum)\n"
         "    }\n"
         "    if (MM == 1 && MS) MS[0] = (whole && n <= " << kMeshRecord << ") ? (unsigned char)(n - 1) : (unsigned char)255;\n"
      << (carry ? "    carried = true;\n" : "") <<
         "  }\n";
    return b.str();
  };
  // adaptive piece: the mesh, then the final pass.  WITH_GRAD adds the pars(:) gradient.
  s << "template <int TK, bool WITH_GRAD> static __device__ double gfh_i" << Is << "_piece(const double lower, const double upper, const double tb, const double* __restrict__ Q, double* __restrict__ GQ, int* STATUS, unsigned char* __restrict__ MS, const int MM) {\n"
    << "  if (!WITH_GRAD) {\n" << mesh_build(true) <<
       "  double y = 0.0;\n"
       "  for (int q = 0; q < n; q++) y = y + sm[q];       // NI:270-275\n"
       "  return y;\n  } else {\n" << mesh_build(false, can_carry) <<
       "  double y = 0.0;\n"
       "  for (int j = 0; j < " << NQ << "; j++) GQ[j] = 0.0;\n"
    << (can_carry ?
       "  if (carried) {\n"
       "    for (int q = 0; q < n; q++) {\n"
       "      const double scale = (hi[q] - lo[q]) / 2;\n"
       "      y = y + sm[q];\n"
       "      for (int j = 0; j < " + std::to_string(NQ) + "; j++) GQ[j] += scale * " + (cfg.ws_global ? std::string("wl_[(4 + j) * 64 + (long long)q * row_]") : std::string("gs[q][j]")) + ";\n"
       "    }\n"
       "    return y;\n"
       "  }\n" : std::string()) <<
       "  for (int q = 0; q < n; q++) {\n"
       "    const double scale = (hi[q] - lo[q]) / 2, shift = (lo[q] + hi[q]) / 2;\n"
       "    double yk = 0.0, gk[" << NQ << "];\n    for (int j = 0; j < " << NQ << "; j++) gk[j] = 0.0;\n"
       "    for (int i = 1; i <= GFH_GK_N; i++) {\n"
       "      double f, g[" << NQ << "];\n"
       "      gfh_i" << Is << "_fg<TK>(scale * gfh_gk_roots[i - 1] + shift, tb, Q, f, g, STATUS);\n"
       "      yk = yk + gfh_gk_wk[i - 1] * f;\n"
       "      for (int j = 0; j < " << NQ << "; j++) gk[j] += gfh_gk_wk[i - 1] * g[j];\n    }\n"
       "    const double sq = scale * yk;      // (rounded before it is added, as the panel sums of the value-only pass are: chi2() at these\n"
       "    y = y + sq;                        //  parameters then returns bitwise this pass's sum r^2 -- the look-ahead schedule builds on that)\n"
       "    for (int j = 0; j < " << NQ << "; j++) GQ[j] += scale * gk[j];\n  }\n"
       "  return y;\n  }\n}\n";
  // site: compose the pieces for the bound kinds (NI:291-369)
  auto body = [&](bool grad) {
    std::string g = grad ? "true" : "false", GQ = grad ? "GQ" : "nullptr";
    std::ostringstream b;
    if (!in.lower_inf && !in.upper_inf) b << "  double y = gfh_i" << Is << "_piece<0, " << g << ">(lower, upper, 0.0, Q, " << GQ << ", STATUS, MS, MM);\n";
    else if (!in.lower_inf && in.upper_inf > 0) b << "  double y = gfh_i" << Is << "_piece<1, " << g << ">(0.0, 1.0, lower, Q, " << GQ << ", STATUS, MS, MM);\n";
    else if (!in.lower_inf && in.upper_inf < 0) b << "  double y = 0.0 - gfh_i" << Is << "_piece<2, " << g << ">(0.0, 1.0, lower, Q, " << GQ << ", STATUS, MS, MM);\n" << (grad ? "  for (int j = 0; j < " + std::to_string(NQ) + "; j++) GQ[j] = -GQ[j];\n" : "");
    else if (in.lower_inf < 0 && !in.upper_inf) b << "  double y = gfh_i" << Is << "_piece<2, " << g << ">(0.0, 1.0, upper, Q, " << GQ << ", STATUS, MS, MM);\n";
    else if (in.lower_inf > 0 && !in.upper_inf) b << "  double y = 0.0 - gfh_i" << Is << "_piece<1, " << g << ">(0.0, 1.0, upper, Q, " << GQ << ", STATUS, MS, MM);\n" << (grad ? "  for (int j = 0; j < " + std::to_string(NQ) + "; j++) GQ[j] = -GQ[j];\n" : "");
    else {   // both infinite: integrate_inf_real(lower, 0) + integrate_real_inf(0, upper), NI:367-368
      std::string g2 = grad ? "G2" : "nullptr";
      if (grad) b << "  double G2[" << NQ << "];\n";
      b << "  double y1 = " << (in.lower_inf < 0 ? "" : "0.0 - ") << "gfh_i" << Is << "_piece<" << (in.lower_inf < 0 ? 2 : 1) << ", " << g << ">(0.0, 1.0, 0.0, Q, " << GQ << ", STATUS, nullptr, 0);\n";
      if (grad && in.lower_inf > 0) b << "  for (int j = 0; j < " << NQ << "; j++) GQ[j] = -GQ[j];\n";
      b << "  double y2 = " << (in.upper_inf > 0 ? "" : "0.0 - ") << "gfh_i" << Is << "_piece<" << (in.upper_inf > 0 ? 1 : 2) << ", " << g << ">(0.0, 1.0, 0.0, Q, " << g2 << ", STATUS, nullptr, 0);\n";
      if (grad) b << "  for (int j = 0; j < " << NQ << "; j++) GQ[j] += " << (in.upper_inf > 0 ? "" : "-") << "G2[j];\n";
      b << "  double y = y1 + y2;\n";
    }
    return b.str();
  };
  // forward-mode piece: same mesh (values), final pass carries (d, dd) linearly through the rule
  s << "template <int TK> static __device__ void gfh_i" << Is << "_piece_fwd(const double lower, const double upper, const double tb, const double* __restrict__ Q, "
       "const double* __restrict__ QD, const double* __restrict__ QE, double& Y, double& YD, double& YE, int* STATUS, unsigned char* __restrict__ MS, const int MM) {\n"
    << mesh_build(false) <<
       "  double y = 0.0, yd = 0.0, ye = 0.0;\n"
       "  for (int q = 0; q < n; q++) {\n"
       "    const double scale = (hi[q] - lo[q]) / 2, shift = (lo[q] + hi[q]) / 2;\n"
       "    double yk = 0.0, ykd = 0.0, yke = 0.0;\n"
       "    for (int i = 1; i <= GFH_GK_N; i++) {\n"
       "      const double t = scale * gfh_gk_roots[i - 1] + shift;\n"
       "      double f, fd, fe;\n"
       "      if (TK == 0) gfh_s" << Ss << "_fwd(t, 0.0, 0.0, Q, QD, QE, f, fd, fe, STATUS);\n"
       "      else {\n"
       "        const double arg = TK == 1 ? (tb - 1.0) + 1.0 / t : (tb + 1.0) - 1.0 / t;\n"
       "        gfh_s" << Ss << "_fwd(arg, 0.0, 0.0, Q, QD, QE, f, fd, fe, STATUS);\n"
       "        const double i2 = 1.0 / (t * t); f *= i2; fd *= i2; fe *= i2;\n"
       "      }\n"
       "      yk = yk + gfh_gk_wk[i - 1] * f; ykd = ykd + gfh_gk_wk[i - 1] * fd; yke = yke + gfh_gk_wk[i - 1] * fe;\n"
       "    }\n"
       "    const double sq = scale * yk;\n"
       "    y = y + sq; yd = yd + scale * ykd; ye = ye + scale * yke;\n"
       "  }\n"
       "  Y = y; YD = yd; YE = ye;\n}\n";
  {
    std::ostringstream b;
    auto call = [&](int tk, const std::string& lo_, const std::string& hi_, const std::string& tb_, const std::string& sfx) {
      b << "  double y" << sfx << ", yd" << sfx << ", ye" << sfx << ";\n"
        << "  gfh_i" << Is << "_piece_fwd<" << tk << ">(" << lo_ << ", " << hi_ << ", " << tb_ << ", Q, QD, QE, y" << sfx << ", yd" << sfx << ", ye" << sfx << ", STATUS, "
        << (sfx.empty() ? "MS, MM" : "nullptr, 0") << ");\n";
    };
    if (!in.lower_inf && !in.upper_inf) call(0, "lower", "upper", "0.0", "");
    else if (!in.lower_inf && in.upper_inf > 0) call(1, "0.0", "1.0", "lower", "");
    else if (!in.lower_inf && in.upper_inf < 0) { call(2, "0.0", "1.0", "lower", ""); b << "  y = 0.0 - y; yd = -yd; ye = -ye;\n"; }
    else if (in.lower_inf < 0 && !in.upper_inf) call(2, "0.0", "1.0", "upper", "");
    else if (in.lower_inf > 0 && !in.upper_inf) { call(1, "0.0", "1.0", "upper", ""); b << "  y = 0.0 - y; yd = -yd; ye = -ye;\n"; }
    else {
      call(in.lower_inf < 0 ? 2 : 1, "0.0", "1.0", "0.0", "1");
      if (in.lower_inf > 0) b << "  y1 = 0.0 - y1; yd1 = -yd1; ye1 = -ye1;\n";
      call(in.upper_inf > 0 ? 1 : 2, "0.0", "1.0", "0.0", "2");
      if (in.upper_inf < 0) b << "  y2 = 0.0 - y2; yd2 = -yd2; ye2 = -ye2;\n";
      b << "  double y = y1 + y2, yd = yd1 + yd2, ye = ye1 + ye2;\n";
    }
    s << "template <bool LA, bool UA> static __device__ void gfh_int" << Is << "_fwd(const double lower, const double lowerD, const double lowerE, "
         "const double upper, const double upperD, const double upperE, const double* __restrict__ Q, const double* __restrict__ QD, "
         "const double* __restrict__ QE, double& Y, double& YD, double& YE, int* STATUS, unsigned char* __restrict__ MS, const int MM) {\n" << b.str();
    // Leibniz terms of active bounds (NI:425-437 / 480-487 / 527-534): f at the bound with the
    // bound passive (dummy) and with the bound active (dir_deriv)
    if (!in.lower_inf) s << "  if (LA) {\n    double f0, f0d, f0e, f1, f1d, f1e;\n"
         "    gfh_s" << Ss << "_fwd(lower, 0.0, 0.0, Q, QD, QE, f0, f0d, f0e, STATUS);\n"
         "    gfh_s" << Ss << "_fwdT(lower, lowerD, lowerE, Q, QD, QE, f1, f1d, f1e, STATUS);\n"
         "    yd = yd - lowerD * f0;\n    ye = ye - lowerE * f0 - lowerD * (f1d + f0d);\n  }\n";
    if (!in.upper_inf) s << "  if (UA) {\n    double f0, f0d, f0e, f1, f1d, f1e;\n"
         "    gfh_s" << Ss << "_fwd(upper, 0.0, 0.0, Q, QD, QE, f0, f0d, f0e, STATUS);\n"
         "    gfh_s" << Ss << "_fwdT(upper, upperD, upperE, Q, QD, QE, f1, f1d, f1e, STATUS);\n"
         "    yd = yd + upperD * f0;\n    ye = ye + upperE * f0 + upperD * (f1d + f0d);\n  }\n";
    s << "  Y = y; YD = yd; YE = ye;\n}\n";
  }
  s << "static __device__ double gfh_int" << Is << "_val(const double lower, const double upper, const double* __restrict__ Q, int* STATUS, unsigned char* __restrict__ MS, const int MM) {\n"
    << body(false) << "  return y;\n}\n";
  // LA / UA: the bound is an AD variable with a live adjoint -- only then is f evaluated THERE for the Leibniz term (NI:413-417), as
  // in the reference.  (An iterated integral int_0^x w(t) int_0^t f du dt would otherwise evaluate its inner integral over the empty
  // range [0, 0], whose error test 0/0 never passes: "Number of iterations was insufficient" where the reference computes nothing.)
  s << "template <bool LA, bool UA> static __device__ void gfh_int" << Is << "_grad(const double lower, const double upper, const double* __restrict__ Q, double& Y, double* __restrict__ GQ, double& FL, double& FH, int* STATUS, unsigned char* __restrict__ MS, const int MM) {\n"
    << body(true)
    << "  Y = y;\n"
    << "  FL = " << (in.lower_inf ? "0.0" : "LA ? gfh_s" + Ss + "_val(lower, Q, STATUS) : 0.0") << ";\n"
    << "  FH = " << (in.upper_inf ? "0.0" : "UA ? gfh_s" + Ss + "_val(upper, Q, STATUS) : 0.0") << ";\n}\n\n";
}


// ---------------------------------------------------------------------------------------
// Branching eval() (AD:315-395; gadfit.F90:679-690 evaluates eval() afresh at every point, so a model may take another
// branch from one point to the next and from one parameter set to the next).  Every recorded path is a straight-line
// tape ("variant") whose comparisons are guard nodes with the outcome they had on that path.  The variants of one model
// share their nodes up to the first guard that came out differently, so together they form a decision tree: walk the
// common nodes, evaluate the guard AT THE CURRENT PARAMETERS on the device, descend.  gfh_select is that walk with values
// only (every parameter passive, the reference's expression shapes: comparisons look at val alone); it returns the variant
// whose body the lane then runs, or -1 where the point takes a turn no recording has taken yet -- the lane then reports
// its slot and the outcomes so far (gfh_report_unseen), the host records eval() at that point along those outcomes, adds
// the variant and repeats the pass (context.cpp, recover_unseen).
// Where two variants part ways WITHOUT a guard (a Fortran eval() that branches on the plain real x: invisible to the
// recorder), only the host can tell which points go where: a per-point column (Model::hint_aux) names the variant each
// point took when the columns were tabulated, and the walk follows it at such a fork.
struct TrieNode {
  int kind = 0;              // 0 leaf, 1 guard, 2 fork without a guard
  int rep = 0;               // a variant of this subtree: its nodes [from, to) are evaluated before the decision
  int from = 0, to = 0;
  int leaf = -1;             // kind 0: the variant
  int guard = -1;            // kind 1: node index of the guard in rep
  int child[2] = {-1, -1};   // kind 1: [outcome false, outcome true]; -1 = never recorded
  std::vector<int> kids;     // kind 2
  std::vector<int> members;  // variants below this node
};

struct Trie {
  const Model& m;
  std::vector<TrieNode> nodes;
  std::string err;
  bool forks = false;
  std::function<const SubTape&(int)> tape;       // the recordings the tree is built over: eval() of variant v, or the members of an integrand family
  explicit Trie(const Model& mm) : m(mm), tape([&mm](int v) -> const SubTape& { return mm.eval(v); }) {}
  Trie(const Model& mm, std::function<const SubTape&(int)> t) : m(mm), tape(std::move(t)) {}

  int build(const std::vector<int>& S, int pos) {
    TrieNode t; t.rep = S[0]; t.from = pos; t.members = S;
    int q = pos;
    for (;;) {
      bool any_end = false, all_end = true;
      for (int v : S) { if (q >= (int)tape(v).nodes.size()) any_end = true; else all_end = false; }
      if (all_end) {
        if (S.size() > 1) { err = "two variants record the same operations"; return -1; }
        t.kind = 0; t.leaf = S[0]; t.to = q;
        nodes.push_back(t); return (int)nodes.size() - 1;
      }
      bool same = !any_end;
      if (same) for (int v : S) if (!same_node(tape(v).nodes[(size_t)q], tape(S[0]).nodes[(size_t)q])) { same = false; break; }
      if (same && !is_guard_op(tape(S[0]).nodes[(size_t)q].op)) { q++; continue; }
      t.to = q;
      if (same) {                                   // the same comparison on every path through here
        t.kind = 1; t.guard = q;
        std::vector<int> side[2];
        for (int v : S) side[(tape(v).nodes[(size_t)q].flags & GFH_F_TAKEN) ? 1 : 0].push_back(v);
        const int me = (int)nodes.size(); nodes.push_back(t);
        for (int o = 0; o < 2; o++) if (!side[o].empty()) { const int c = build(side[o], q + 1); if (c < 0) return -1; nodes[(size_t)me].child[o] = c; }
        return me;
      }
      // different operations (or one path ends here): classes of equal next node
      t.kind = 2; forks = true;
      std::vector<std::vector<int>> cls;
      for (int v : S) {
        bool placed = false;
        for (auto& c : cls) {
          const int w = c[0];
          const bool ve = q >= (int)tape(v).nodes.size(), we = q >= (int)tape(w).nodes.size();
          if (ve || we ? (ve && we && tape(v).result == tape(w).result)
                       : same_node(tape(v).nodes[(size_t)q], tape(w).nodes[(size_t)q])) { c.push_back(v); placed = true; break; }
        }
        if (!placed) cls.push_back({v});
      }
      if (cls.size() < 2) { err = "two variants record the same operations"; return -1; }
      const int me = (int)nodes.size(); nodes.push_back(t);
      for (auto& c : cls) { const int k = build(c, q); if (k < 0) return -1; nodes[(size_t)me].kids.push_back(k); }
      return me;
    }
  }
};

// nodes of variant v whose values some guard of v needs (transitively)
std::vector<char> guard_needs_of(const Model& m, const SubTape& st);
std::vector<char> guard_needs(const Model& m, int v) { return guard_needs_of(m, m.eval(v)); }
std::vector<char> guard_needs_of(const Model& m, const SubTape& st) {
  std::vector<char> need(st.nodes.size(), 0);
  for (int k = (int)st.nodes.size() - 1; k >= 0; k--) {
    const Node& nd = st.nodes[(size_t)k];
    if (!is_guard_op(nd.op) && !need[(size_t)k]) continue;
    auto want = [&](int r) { if (r >= 0) need[(size_t)r] = 1; };
    switch (nd.op) {
      case GFH_CONST: case GFH_X: case GFH_AUX: case GFH_PARAM: case GFH_IVAR: case GFH_IPARAM: break;
      case GFH_INTEGRATE: {
        const Integral& in = m.integrals[(size_t)nd.a];
        if (!in.lower_inf) want(in.lower);
        if (!in.upper_inf) want(in.upper);
        for (int q = 0; q < in.n_ipars; q++) want(m.ipar_nodes[(size_t)in.ipar_off + q]);
        break;
      }
      case GFH_ADD: case GFH_SUB: case GFH_MUL: case GFH_DIV: case GFH_POW: case GFH_GUARD_GT: case GFH_GUARD_LT: want(nd.a); want(nd.b); break;
      default: want(nd.a); break;      // unary, LIFT, NEG, POWI
    }
  }
  return need;
}

// gfh_select (see above).  PATH / NG: the outcomes of the guards passed so far and their number, for the report of an unseen turn.
bool emit_selector(const Model& m, std::ostringstream& s, std::string* err) {
  const int V = m.n_variants();
  Trie T(m);
  std::vector<int> all(V); for (int v = 0; v < V; v++) all[(size_t)v] = v;
  const int root = T.build(all, 0);
  if (root < 0) { *err = "gfh_set_model_variants: " + T.err; return false; }
  if (T.forks && m.hint_aux < 0) {
    *err = "the recorded variants of eval() part ways without a comparison of AD variables (control flow on plain real values): "
           "the per-point variant column is needed (gfh_set_model_variants, hint_aux)";
    return false;
  }
  std::vector<std::vector<char>> need((size_t)V);
  for (int v = 0; v < V; v++) need[(size_t)v] = guard_needs(m, v);
  const std::vector<char> none((size_t)std::max(1, m.n_pars), 0);
  s << "\n// Which recorded path of eval() does this point take at these parameters?  (codegen.cpp, Trie)\n"
       "static __device__ __forceinline__ int gfh_select(const double X, const double* __restrict__ P, int* STATUS,\n"
       "                                                 const double* __restrict__ AXP, const i64 LDA, unsigned long long& PATH, int& NG) {\n"
       "  PATH = 0ull; NG = 0;\n  GFH_LANE_STASH\n";
  // (HV: the per-point variant column once a fork without a comparison has been passed.  The walk then ends on a variant the HOST
  // has seen this point take -- HV itself -- or reports the point: behind such a fork the recordings of one class need not share the
  // side of the plain real branch that set them apart (two of them may agree by accident in the node where a third differs, an
  // affine literal's slope in its last bit), so a leaf other than HV is not taken on the device's word alone; the host records the
  // point along the outcomes met, tabulates the column anew at the parameters of this pass, and the pass is repeated.)
  if (T.forks) s << "  int HV = -2;\n";
  bool ok = true;
  // `fail`: what a walk does where it cannot go on (a comparison comes out a way nobody recorded, a leaf the host has not seen this
  // point take): report the point -- or, inside a fork's kid, give up on that kid and try the next one (round 5).  Recordings may
  // part ways without a comparison for a reason that a LATER comparison settles (the same source line holding another per-point
  // column on either side of it): which kid is the point's cannot be read off a column alone then, but only one of the candidates
  // can be walked to a leaf that the host has seen the point take under the outcomes met on the way.
  int n_labels = 0;
  std::function<void(int, int, const std::string&, const std::string&)> walk = [&](int idx, int depth, const std::string& ind, const std::string& fail) {
    const TrieNode& t = T.nodes[(size_t)idx];
    auto failing = [&](int ng) { return fail.empty() ? "{ NG = " + std::to_string(ng) + "; return -1; }" : "{ " + fail + " }"; };
    {
      Gen g(m, m.eval(t.rep), false); g.mode = 0; g.ind = ind; g.analyse(none);
      for (int k = t.from; k < t.to; k++) {
        bool wanted = false;
        for (int v : t.members) if (need[(size_t)v][(size_t)k]) { wanted = true; break; }
        if (wanted) g.emit_value_node(k);
      }
      s << g.o.str();
    }
    if (t.kind == 0) {
      // (behind a fork the walk ends on a leaf the HOST has seen this point take under exactly these outcomes -- the column of the
      // leaf's own set of outcomes names a tape of this leaf -- or fails)
      if (T.forks) {
        s << ind << "if (HV != -2) { const int hl = (int)AXP[(i64)" << m.hint_col_of_variant(t.leaf) << " * LDA]; if (!(";
        const std::vector<int> tp = m.tapes_of_variant(t.leaf);
        for (size_t q = 0; q < tp.size(); q++) s << (q ? " || " : "") << "hl == " << tp[q];
        s << ")) " << failing(depth) << " }\n";
      }
      s << ind << "return " << t.leaf << ";\n";
      return;
    }
    if (t.kind == 1) {
      if (depth >= 64) { ok = false; *err = "more than 64 comparisons on one path through eval()"; return; }
      const Node& nd = m.eval(t.rep).nodes[(size_t)t.guard];
      s << ind << "const bool c" << t.guard << " = v" << nd.a << (nd.op == GFH_GUARD_GT ? " > " : " < ") << "v" << nd.b << ";      // AD:315-395: values only\n";
      s << ind << "PATH |= (c" << t.guard << " ? 1ull : 0ull) << " << depth << ";\n";
      for (int o = 1; o >= 0; o--) {
        s << ind << (o ? "if (c" + std::to_string(t.guard) + ") {\n" : "} else {\n");
        if (t.child[o] >= 0) walk(t.child[o], depth + 1, ind + "  ", fail);
        else s << ind << "  " << failing(depth + 1) << "\n";
      }
      s << ind << "}\n";
      return;
    }
    // A fork without a comparison.  Kid by kid: the column of the kid's own outcomes (its first recording's) must name a tape of the
    // kid -- the host has seen this point go there under those outcomes -- and the walk inside must reach a leaf; else the next kid.
    s << ind << "const unsigned long long pf" << idx << " = PATH;\n";
    for (size_t c = 0; c < t.kids.size(); c++) {
      const TrieNode& k = T.nodes[(size_t)t.kids[c]];
      const int lab = n_labels++;
      s << ind << "{\n" << ind << "  const int h" << idx << "_" << c << " = (int)AXP[(i64)" << m.hint_col_of_variant(k.members[0]) << " * LDA];      // the tape this point follows under that kid's outcomes\n";
      s << ind << "  if (";
      bool first = true;
      for (size_t q = 0; q < k.members.size(); q++)
        for (int tpi : m.tapes_of_variant(k.members[q])) { s << (first ? "" : " || ") << "h" << idx << "_" << c << " == " << tpi; first = false; }
      s << ") {\n" << ind << "    HV = h" << idx << "_" << c << ";\n";
      walk(t.kids[c], depth, ind + "    ", "goto gfh_next" + std::to_string(lab) + ";");
      s << ind << "  }\n" << ind << "}\n" << ind << "gfh_next" << lab << ": PATH = pf" << idx << ";\n";
    }
    s << ind << failing(depth) << "\n";
  };
  walk(root, 0, "  ", "");
  s << "}\n";
  return ok;
}

// The recordings of ONE integrand that compares AD variables (Model::alts): which of them is the path of this abscissa, and the four
// forms of the integrand (value, value + gradient, forward mode with and without a tangent on the integration variable) as
// dispatchers under the names gfh_sf<I>_* that call site I uses instead of gfh_s<S>_*.  The comparisons are decided on values, per
// evaluation, exactly as the reference's integrand decides them when the quadrature calls it (AD:315-395).  An abscissa whose
// outcomes no recording has raises status 2 (the host reports it: the recordings sampled the integration variable too coarsely).
bool emit_family(const Model& m, int I, std::ostringstream& s, std::string* err) {
  const Integral& in = m.integrals[(size_t)I];
  std::vector<int> mem{in.integrand};
  if ((size_t)I < m.alts.size()) mem.insert(mem.end(), m.alts[(size_t)I].begin(), m.alts[(size_t)I].end());
  const int M = (int)mem.size();
  Trie T(m, [&m, mem](int k) -> const SubTape& { return m.sub[(size_t)mem[(size_t)k]]; });
  std::vector<int> all((size_t)M); for (int k = 0; k < M; k++) all[(size_t)k] = k;
  const int root = T.build(all, 0);
  if (root < 0) { *err = "integrand recordings: " + T.err; return false; }
  if (T.forks) { *err = "the recordings of an integrand part ways without a comparison of AD variables (control flow on plain real values inside an integrand)"; return false; }
  std::vector<std::vector<char>> need((size_t)M);
  int nip = 0;
  for (int k = 0; k < M; k++) {
    need[(size_t)k] = guard_needs_of(m, m.sub[(size_t)mem[(size_t)k]]);
    for (const Node& nd : m.sub[(size_t)mem[(size_t)k]].nodes) if (nd.op == GFH_IPARAM && nd.a + 1 > nip) nip = nd.a + 1;
  }
  const std::vector<char> none((size_t)std::max(1, nip), 0);
  const std::string Is = std::to_string(I);
  s << "// integrand of call site " << I << ": " << M << " recorded path(s) through its comparisons\n"
       "static __device__ __forceinline__ int gfh_sf" << Is << "_sel(const double T, const double* __restrict__ Q) {\n";
  bool ok = true;
  std::function<void(int, const std::string&)> walk = [&](int idx, const std::string& ind) {
    const TrieNode& t = T.nodes[(size_t)idx];
    {
      Gen g(m, m.sub[(size_t)mem[(size_t)t.rep]], false); g.in_integrand = true; g.mode = 0; g.ind = ind; g.analyse(none);
      for (int k = t.from; k < t.to; k++) {
        bool wanted = false;
        for (int v : t.members) if (need[(size_t)v][(size_t)k]) { wanted = true; break; }
        if (wanted) g.emit_value_node(k);
      }
      s << g.o.str();
    }
    if (t.kind == 0) { s << ind << "return " << t.leaf << ";\n"; return; }
    if (t.kind != 1) { ok = false; return; }
    const Node& nd = m.sub[(size_t)mem[(size_t)t.rep]].nodes[(size_t)t.guard];
    s << ind << "if (v" << nd.a << (nd.op == GFH_GUARD_GT ? " > " : " < ") << "v" << nd.b << ") {      // AD:315-395: values only\n";
    if (t.child[1] >= 0) walk(t.child[1], ind + "  "); else s << ind << "  return -1;\n";
    s << ind << "} else {\n";
    if (t.child[0] >= 0) walk(t.child[0], ind + "  "); else s << ind << "  return -1;\n";
    s << ind << "}\n";
  };
  walk(root, "  ");
  s << "}\n";
  if (!ok) { *err = "integrand recordings: unexpected fork"; return false; }
  auto cases = [&](const std::string& call_tail, const std::string& on_miss) {
    s << "  switch (gfh_sf" << Is << "_sel(T, Q)) {\n";
    for (int k = 0; k < M; k++) s << "    case " << k << ": " << "gfh_s" << mem[(size_t)k] << call_tail << "\n";
    s << "    default: if (STATUS) GFH_RAISE(STATUS, 2); " << on_miss << "\n  }\n";
  };
  s << "static __device__ double gfh_sf" << Is << "_val(const double T, const double* __restrict__ Q, int* STATUS) {\n";
  s << "  switch (gfh_sf" << Is << "_sel(T, Q)) {\n";
  for (int k = 0; k < M; k++) s << "    case " << k << ": return gfh_s" << mem[(size_t)k] << "_val(T, Q, STATUS);\n";
  s << "    default: if (STATUS) GFH_RAISE(STATUS, 2); return 0.0;\n  }\n}\n";
  s << "static __device__ void gfh_sf" << Is << "_grad(const double T, const double* __restrict__ Q, double& F, double* __restrict__ GQ, int* STATUS) {\n"
       "  for (int j = 0; j < " << std::max(1, in.n_ipars) << "; j++) GQ[j] = 0.0;      // (a recording writes the entries of the pars(:) it reads)\n";
  cases("_grad(T, Q, F, GQ, STATUS); return;", "F = 0.0; for (int j = 0; j < " + std::to_string(std::max(1, in.n_ipars)) + "; j++) GQ[j] = 0.0; return;");
  s << "}\n";
  for (int ta = 0; ta < 2; ta++) {
    s << "static __device__ void gfh_sf" << Is << (ta ? "_fwdT" : "_fwd") << "(const double T, const double TD, const double TE, "
         "const double* __restrict__ Q, const double* __restrict__ QD, const double* __restrict__ QE, double& F, double& FD, double& FE, int* STATUS) {\n";
    cases(std::string(ta ? "_fwdT" : "_fwd") + "(T, TD, TE, Q, QD, QE, F, FD, FE, STATUS); return;", "F = 0.0; FD = 0.0; FE = 0.0; return;");
    s << "}\n";
  }
  s << "\n";
  return true;
}

// does call site I pick its integrand per evaluation (several recordings, or one that compares AD variables)?
bool site_is_family(const Model& m, int I) {
  if ((size_t)I < m.alts.size() && !m.alts[(size_t)I].empty()) return true;
  for (const Node& nd : m.sub[(size_t)m.integrals[(size_t)I].integrand].nodes) if (is_guard_op(nd.op)) return true;
  return false;
}

}  // namespace

bool Model::needs_hint() const {
  if (n_variants() < 2) return false;
  Trie T(*this);
  std::vector<int> all((size_t)n_variants());
  for (int v = 0; v < n_variants(); v++) all[(size_t)v] = v;
  return T.build(all, 0) >= 0 && T.forks;
}

// Where the workspaces live and how many intervals the translation unit carries (model.h, WsPlan).
// (scratch form: 8 NQ bytes more per interval and lane, small workspaces only; pool form: NQ more fields in the interval's row)
bool carries_gradients(int n_ipars, int ws, bool global) {
  const int nq = n_ipars > 0 ? n_ipars : 1;
  return global ? nq <= kWsgCarryMax : (nq <= 4 && ws <= 128);
}
int wsg_row_fields(const Model& m, int level) {
  int nq = 0;
  for (const Integral& in : m.integrals)
    if ((in.depth <= 1) == (level == 1)) { const int q = in.n_ipars > 0 ? in.n_ipars : 1; if (q <= kWsgCarryMax) nq = std::max(nq, q); }
  return 4 + nq;
}

static long ws_scratch_bytes(const Model& m, int ws1, int ws2) {
  int nq1 = 0, nq2 = 0; bool nested = false;
  for (const Integral& in : m.integrals) {
    const int nq = in.n_ipars > 0 ? in.n_ipars : 1;
    if (in.depth <= 1) nq1 = std::max(nq1, nq); else { nq2 = std::max(nq2, nq); nested = true; }
  }
  long b = (long)ws1 * (32 + (carries_gradients(nq1, ws1, false) ? 8 * nq1 : 0));
  if (nested) b += (long)ws2 * (32 + (carries_gradients(nq2, ws2, false) ? 8 * nq2 : 0));
  return b;
}

WsPlan plan_workspaces(const Model& m, int fast, bool grown) {
  WsPlan p{m.ws_size, m.ws_size_inner, false};
  if (!m.has_integrals()) return p;
  if (!grown && fast >= 2) {
    p.ws_size = std::min(fast, m.ws_size); p.ws_size_inner = std::min(fast, m.ws_size_inner);
    // (nested integrals with several bound parameters: both levels shrink together until the budget holds them)
    while (ws_scratch_bytes(m, p.ws_size, p.ws_size_inner) > kScratchBudget && std::max(p.ws_size, p.ws_size_inner) > 2) {
      const int top = std::max(p.ws_size, p.ws_size_inner) - 1;
      p.ws_size = std::min(p.ws_size, top); p.ws_size_inner = std::min(p.ws_size_inner, top);
    }
    return p;
  }
  p.global = ws_scratch_bytes(m, p.ws_size, p.ws_size_inner) > kScratchBudget;
  return p;
}

int mesh_sites(const Model& m) {
  if (!m.has_integrals()) return 0;
  int most = 0;
  for (int v = 0; v < m.n_variants(); v++) {
    int n = 0;
    for (const Node& nd : m.eval(v).nodes)
      if (nd.op == GFH_INTEGRATE) {
        const Integral& in = m.integrals[(size_t)nd.a];
        if (in.depth <= 1 && !(in.lower_inf && in.upper_inf) && n < kMeshSitesMax) n++;
      }
    most = std::max(most, n);
  }
  return most;
}

// The cooperative form of the fused kernel (GFH_COOP, 5 ... 8 tiles): its switch and, per wave W of the workgroup, the straight-line
// code of its share of the tile pairs -- a contiguous run of the row-major upper triangle:
//   GFH_CLOAD_W(B, U)  fragment reads of k-step U into buffer B: one per DISTINCT tile of the run + the residual row
//   GFH_CMMA_W(B)      one v_mfma_f64_16x16x4_f64 per pair; J^T r of tile t with the owner of pair (t, t)
//   GFH_CPUT_W         the wave's accumulators into the workgroup's image
static std::string coop_defines(int NA, const GenConfig& cfg) {
  std::ostringstream o;
  const bool on = fused_coop(NA, cfg);
  o << "\n#define GFH_COOP " << (on ? 1 : 0);
  if (!on) return o.str();
  const int T = (NA + 15) / 16, npair = T * (T + 1) / 2, fw = fused_waves_for(NA, cfg);
  std::vector<std::pair<int, int>> pairs;
  for (int ti = 0; ti < T; ti++) for (int tj = ti; tj < T; tj++) pairs.push_back({ti, tj});
  int cnd = 0;
  std::ostringstream body;
  for (int w = 0, p0 = 0; w < 4; w++) {
    const int nk = w < fw ? npair / fw + (w < npair % fw ? 1 : 0) : 0;
    std::vector<int> tiles;                                   // distinct tiles of this wave's run, in order of first use
    auto slot = [&](int t) { for (size_t d = 0; d < tiles.size(); d++) if (tiles[d] == t) return (int)d; tiles.push_back(t); return (int)tiles.size() - 1; };
    std::ostringstream mma0, mma, put;
    for (int k = 0; k < nk; k++) {
      const int ti = pairs[(size_t)(p0 + k)].first, tj = pairs[(size_t)(p0 + k)].second;
      const int ia = slot(ti), ib = slot(tj);
      (k == 0 ? mma0 : mma) << " GFH_MFMA(f[B_][" << ia << "], f[B_][" << ib << "], acc[" << k << "]);";
      if (ti == tj) mma << " accr[" << k << "] += f[B_][" << ia << "] * fr[B_];";
      put << " GFH_CPUT(" << k << ", " << (p0 + k) << ")";
      if (ti == tj) put << " GFH_CPUTR(" << k << ", " << ti << ")";
    }
    body << "\n#define GFH_CLOAD_" << w << "(B_, U_) fr[B_] = sb[16 * GFH_T * GFH_S + 4 * (U_) + q];";
    for (size_t d = 0; d < tiles.size(); d++) body << " f[B_][" << d << "] = sb[(16 * " << tiles[d] << " + r) * GFH_S + 4 * (U_) + q];";
    body << "\n#define GFH_CMMA0_" << w << "(B_)" << mma0.str();
    body << "\n#define GFH_CMMA_" << w << "(B_)" << mma.str();
    body << "\n#define GFH_CPUT_" << w << put.str();
    cnd = std::max(cnd, (int)tiles.size());
    p0 += nk;
  }
  o << "\n#define GFH_CK " << (npair + fw - 1) / fw << "\n#define GFH_CND " << std::max(1, cnd) << body.str();
  return o.str();
}

bool generate_source(const Model& m, const std::vector<int32_t>& active, const GenConfig& cfg,
                     std::string* src, std::string* err) {
  const SubTape& st = m.sub[0];
  for (int v = 0; v < m.n_variants(); v++)
    for (const Node& nd : m.eval(v).nodes)
      if (nd.op == GFH_IVAR || nd.op == GFH_IPARAM) { *err = "integrand node in eval() tape"; return false; }
  const int NA = (int)active.size(), NP = m.n_pars;
  std::vector<char> pa(NP, 0), none(NP, 0);
  for (int a : active) { if (a < 0 || a >= NP) { *err = "active parameter out of range"; return false; } pa[a] = 1; }
  // adjoint source per active parameter: every PARAM node of that parameter
  std::ostringstream s;
  s << "// generated by libgadfit_hip codegen -- model with " << st.nodes.size() << " tape nodes, "
    << NP << " parameters, " << NA << " active\n";
  s << "#define GFH_OMEGA_JT " << (cfg.omega_jt ? 1 : 0) << "\n#define GFH_FW " << fused_waves_for(NA, cfg) << "\n#define GFH_HALF " << (fused_half_stage(NA, cfg) ? 1 : 0) << "\n#define GFH_FUSED_WPE " << cfg.fused_wpe << "\n#define GFH_FRAG_LATE " << cfg.frag_late << "\n#define GFH_RED1 " << (fused_single_image(NA, cfg) ? 1 : 0) << "\n#define GFH_FUSED_MAX " << fused_max_active(cfg) << coop_defines(NA, cfg) << "\n#define GFH_FAST_DIV " << (cfg.fast_div ? 1 : 0)
    << "\n#define GFH_STORE_J " << (cfg.store_j ? 1 : 0) << "\n#define GFH_STORE_RES " << (cfg.store_res ? 1 : 0) << "\n#define GFH_LOSS " << cfg.loss << "\n#define GFH_BLOCK " << cfg.block << "\n#define GFH_NP " << NP
    << "\n#define GFH_NA " << (NA > 0 ? NA : 1) << "\n#define GFH_VALU_GRAM_MAX " << kValuGramMax << "\n#define GFH_VAHEAD " << valu_ahead_for(NA, cfg) << "\n#define GFH_AHEAD " << std::max(1, std::min(2, cfg.frag_ahead)) << "\n#define GFH_ABLATE " << cfg.ablate << "\n#define GFH_MATRIX_PRIO " << (cfg.store_j ? 0 : cfg.matrix_prio) << "\n";
  s << "#define GFH_PARG " << cfg.kernarg_pars << "\n";
  s << R"(
// exp(x): the operations of the device library's exp (ROCm device-libs, __ocml_exp_f64: n = rint(x log2 e), two-step
// Cody-Waite reduction, its degree-11 polynomial, ldexp), so the same bits for every x that is not a NaN.  What differs
// is how the ends of the range are handled: the library computes ldexp(p, n) and then SELECTS +inf for x > 1024 and 0 for
// x < -1075 -- two v_cmp_f64 and three v_cndmask_b32 per call, and a v_cndmask_b32 that takes its mask from VCC costs
// 16-18 cycles per wave on gfx950 against 4-5 for an FP64 multiply-add (tools/microbench/fp64_rates.hip): a third of the
// call.  Here x is clamped to [-1075, 1024] first (two full-rate instructions; ldexp then overflows to +inf and
// underflows to 0 by itself, at the same x), and the one thing a clamp loses -- a NaN argument -- is put back with a
// compare into an SGPR pair and a select on the high word that takes its mask from there (4-5 cycles each).
typedef int int2_t_ __attribute__((ext_vector_type(2)));
static __device__ __forceinline__ double gfh_exp(const double x) {
  const double xc = __builtin_fmin(__builtin_fmax(x, -1075.0), 1024.0);
  const double dn = __builtin_rint(xc * 0x1.71547652b82fep+0);
  double r = __builtin_fma(-dn, 0x1.62e42fefa39efp-1, xc);
  r = __builtin_fma(-dn, 0x1.abc9e3b39803fp-56, r);
  double p = __builtin_fma(r, 0x1.ade156a5dcb37p-26, 0x1.28af3fca7ab0cp-22);
  p = __builtin_fma(r, p, 0x1.71dee623fde64p-19);
  p = __builtin_fma(r, p, 0x1.a01997c89e6b0p-16);
  p = __builtin_fma(r, p, 0x1.a01a014761f6ep-13);
  p = __builtin_fma(r, p, 0x1.6c16c1852b7b0p-10);
  p = __builtin_fma(r, p, 0x1.1111111122322p-7);
  p = __builtin_fma(r, p, 0x1.55555555502a1p-5);
  p = __builtin_fma(r, p, 0x1.5555555555511p-3);
  p = __builtin_fma(r, p, 0x1.000000000000bp-1);
  p = __builtin_fma(r, p, 1.0);
  p = __builtin_fma(r, p, 1.0);
  int2_t_ z = __builtin_bit_cast(int2_t_, __builtin_ldexp(p, (int)dn));
  unsigned long long is_nan;
  asm("v_cmp_u_f64 %0, %1, %1" : "=s"(is_nan) : "v"(x));
  int hi = z.y;
  asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(hi) : "v"(hi), "v"(0x7ff80000), "s"(is_nan));
  z.y = hi;
  return __builtin_bit_cast(double, z);
}
)";
  {
    bool has_pow = false;
    for (const SubTape& t_ : m.sub) for (const Node& nd : t_.nodes) if (nd.op == GFH_POW && !(nd.flags & GFH_F_REAL)) has_pow = true;
    for (const SubTape& t_ : m.more_evals) for (const Node& nd : t_.nodes) if (nd.op == GFH_POW && !(nd.flags & GFH_F_REAL)) has_pow = true;
    if (has_pow && cfg.fast_div) s << R"(
// x**a and ln x from ONE extended-precision logarithm.  The device library's pow is 226 VALU instructions (28 of them selects
// on special cases), its log 98, and the derivative code of x**a wants both -- at every Kronrod node of every bisection of
// a quadrature model.  Here: x = 2^e m with m in [sqrt(1/2), sqrt(2)), s = (m - 1) / (m + 1) as a double-double (the
// quotient corrected by its own residual), ln m = 2 s + s z (2/3 + 2 z / 5 + ... + 2 z^9 / 21), z = s^2 (the truncation is
// below 2^-60 of the result for |s| <= 0.1716); ln x = e ln 2 + ln m summed as a double-double; x**a = exp(a ln x) with the
// low part of the product applied to first order.  About 90 VALU instructions for both results; measured against 60-digit
// references (tests/test_gpu_parity.py, test_device_pow_accuracy): <= 3 ulp (1.3 from the logarithm and the product, the rest gfh_exp) for |a ln x| <= 700 and 2^-1022 <= x < inf.
// Everything else -- x <= 0, subnormal, inf, NaN, overflowing exponents -- takes the library's pow and log, whose special
// cases are the reference's (IEEE pow).
static __device__ __forceinline__ double gfh_pow_ln(const double x, const double a, double& lnx) {
  if (x >= 0x1p-1022 && x < __builtin_inf()) {
    int e = __builtin_amdgcn_frexp_exp(x);
    double m = __builtin_amdgcn_frexp_mant(x);                    // [0.5, 1)
    if (m < 0x1.6a09e667f3bcdp-1) { m = m + m; e -= 1; }         // [sqrt(1/2), sqrt(2))
    const double f = m - 1.0;                                      // exact
    const double dh = 2.0 + f, dl = (2.0 - dh) + f;               // m + 1 as a double-double
    double g = __builtin_amdgcn_rcp(dh);
    g = __builtin_fma(__builtin_fma(-dh, g, 1.0), g, g);
    g = __builtin_fma(__builtin_fma(-dh, g, 1.0), g, g);
    const double sh = f * g;
    const double sl = __builtin_fma(-sh, dl, __builtin_fma(-sh, dh, f)) * g;
    const double z = sh * sh;
    double p = __builtin_fma(z, 0x1.8618618618618p-4, 0x1.af286bca1af28p-4);      // 2/21, 2/19
    p = __builtin_fma(z, p, 0x1.e1e1e1e1e1e1ep-4);                                  // 2/17
    p = __builtin_fma(z, p, 0x1.1111111111111p-3);                                  // 2/15
    p = __builtin_fma(z, p, 0x1.3b13b13b13b14p-3);                                  // 2/13
    p = __builtin_fma(z, p, 0x1.745d1745d1746p-3);                                  // 2/11
    p = __builtin_fma(z, p, 0x1.c71c71c71c71cp-3);                                  // 2/9
    p = __builtin_fma(z, p, 0x1.2492492492492p-2);                                  // 2/7
    p = __builtin_fma(z, p, 0x1.999999999999ap-2);                                  // 2/5
    p = __builtin_fma(z, p, 0x1.5555555555555p-1);                                  // 2/3
    const double mh = sh + sh;
    const double ml = __builtin_fma(sh * z, p, sl + sl);                           // ln m = mh + ml
    const double ed = (double)e;
    const double th = ed * 0x1.62e42fee00000p-1;                                    // e ln2_hi: exact (ln2_hi carries 32 trailing zero bits)
    const double lh = th + mh;
    const double bb = lh - th;
    const double le = (th - (lh - bb)) + (mh - bb);                                // two-sum: th + mh = lh + le
    const double ll = __builtin_fma(ed, 0x1.a39ef35793c76p-33, le + ml);           // + e ln2_lo
    const double nh = lh + ll, nl = ll - (nh - lh);                                // renormalised: ln x = nh + nl, |nl| <= ulp(nh) / 2
    lnx = nh;
    const double ph = a * nh;
    const double pl = __builtin_fma(a, nl, __builtin_fma(a, nh, -ph));
    if (__builtin_fabs(ph) < 700.0) {
      const double r = gfh_exp(ph);
      return __builtin_fma(r, pl, r);
    }
  }
  lnx = log(x);
  return pow(x, a);
}
)";
  }
  bool lane_stash = false;          // some integrand reads the data point's abscissa or auxiliary columns (Gen::in_integrand)
  for (size_t k = 1; k < m.sub.size(); k++) for (const Node& nd : m.sub[k].nodes) if (nd.op == GFH_X || nd.op == GFH_AUX) lane_stash = true;
  s << "\ntypedef long long i64;\n// kernels raise the status word with an agent-scope atomic: visible to whichever workgroup posts it to the host\n#define GFH_RAISE(p, v) __hip_atomic_fetch_max((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)\n"
       "#define GFH_STATUS_SLOT(st) ((st) == 0 ? 0.0 : (st) == 1 ? 1.0 : (st) == 2 ? 4096.0 : 16777216.0)\n";
  if (lane_stash)
    s << "// what an integrand takes from the enclosing eval() without passing it through pars(:): the lane's abscissa and its auxiliary\n"
         "// columns, left here by the point functions (same thread writes and reads: program order)\n"
         "__shared__ double gfh_lane_x[512];\n__shared__ const double* gfh_lane_axp[512];\n__shared__ i64 gfh_lane_lda;\n"
         "#define GFH_LANE_STASH gfh_lane_x[threadIdx.x] = X; gfh_lane_axp[threadIdx.x] = AXP; gfh_lane_lda = LDA;\n";
  else s << "#define GFH_LANE_STASH\n";
  s << R"(
// The parameter block [n_datasets][GFH_NP].  Up to 480 doubles (GFH_PARG = n_datasets * GFH_NP) it travels in the
// kernel-argument segment: no host-to-device copy is queued in front of every pass, and the
// parameters are scalar loads from the kernarg pointer.  Otherwise it is a device array.
#if GFH_PARG
struct gfh_parg { double v[GFH_PARG]; };
#define GFH_PARS_DECL const gfh_parg pars
#define GFH_DPARS_DECL const gfh_parg dpars
#if GFH_PARG == GFH_NP
#define GFH_PARS_AT(ds) pars.v
#define GFH_DPARS_AT(ds) dpars.v
#else
#define GFH_PARS_AT(ds) (pars.v + (ds) * GFH_NP)      // wave-uniform dataset index: scalar loads at a register offset
#define GFH_DPARS_AT(ds) (dpars.v + (ds) * GFH_NP)
#endif
#else
#define GFH_PARS_DECL const double* __restrict__ pars
#define GFH_PARS_AT(ds) (pars + (i64)(ds) * GFH_NP)
#define GFH_DPARS_DECL const double* __restrict__ dpars
#define GFH_DPARS_AT(ds) (dpars + (i64)(ds) * GFH_NP)
#endif
)";
  if (m.has_integrals()) {
    const double *roots = gk15_roots, *wg = gk15_wg, *wk = gk15_wk; int npts = 15;
    switch (m.gk_points) {
      case 21: roots = gk21_roots; wg = gk21_wg; wk = gk21_wk; npts = 21; break;
      case 31: roots = gk31_roots; wg = gk31_wg; wk = gk31_wk; npts = 31; break;
      case 41: roots = gk41_roots; wg = gk41_wg; wk = gk41_wk; npts = 41; break;
      case 51: roots = gk51_roots; wg = gk51_wg; wk = gk51_wk; npts = 51; break;
      case 61: roots = gk61_roots; wg = gk61_wg; wk = gk61_wk; npts = 61; break;
      default: break;
    }
    s << "// Gauss-Kronrod rule (numerical_integration.F90:139-171), reference node order: even 1-based = Gauss nodes\n";
    // (intervals an adaptive integral may use, per nesting level: ws(1) / ws(2) of the reference, NI:70, 84-98)
    s << "#define GFH_GK_N " << npts << "\n#define GFH_WS1 " << cfg.ws_size << "\n#define GFH_WS2 " << cfg.ws_size_inner << "\n";
    if (cfg.ws_global) {
      // Workspaces in the context's global pool (model.h, plan_workspaces; NI:40-51, 128-134: the reference's are heap arrays of the
      // user's size).  One slot per wave of the launch -- workgroup b's wave v owns slot b * waves-per-workgroup + v, the host caps
      // the grid at the slots there are and the kernels stride over their tiles -- laid out [level][interval][lo|hi|err|sum][lane]:
      // the wave's access to one field of one interval is one coalesced 512 B row.  The kernels post the pool's address in LDS.
      // (round 5: a level whose call sites carry their panels' gradients has one more field per integrand parameter in its rows --
      // wsg_row_fields, model.h -- so that the pool form spares the final pass its re-evaluation as the scratch form does)
      const long l2 = 64L * wsg_row_fields(m, 1) * cfg.ws_size;
      s << "#define GFH_WSG 1\n#define GFH_WSG_L2 " << l2 << "LL\n#define GFH_WSG_WAVE " << wsg_wave_doubles(m, cfg.ws_size, cfg.ws_size_inner) << "LL\n"
           "#define GFH_WSG_ROW1 " << 64L * wsg_row_fields(m, 1) << "LL\n#define GFH_WSG_ROW2 " << 64L * wsg_row_fields(m, 2) << "LL\n"
           "__shared__ double* gfh_wsg_base;\n"
           "struct gfh_wsa { double* p; long long row; __device__ __forceinline__ double& operator[](const int q) const { return p[(long long)q * row]; } };\n"
           "static __device__ __forceinline__ double* gfh_wsg_lane(const long long level) {\n"
           "  return gfh_wsg_base + ((long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * GFH_WSG_WAVE + level + (threadIdx.x & 63);\n}\n";
    }
    auto arr = [&](const char* name, const double* a, int n) {
      s << "static __device__ const double " << name << "[" << n << "] = {";
      for (int i = 0; i < n; i++) s << (i ? ", " : "") << lit(a[i]);
      s << "};\n";
    };
    arr("gfh_gk_roots", roots, npts); arr("gfh_gk_wg", wg, npts / 2); arr("gfh_gk_wk", wk, npts);
    s << "\n";
    // (call sites of different variants may share one integrand sub-tape, Model::load_variants: its functions are emitted once)
    std::vector<char> sub_done(m.sub.size(), 0);
    // call sites that some recording of eval() (or of an integrand) reaches; the pooled call sites of further recordings of an
    // integrand (Model::alts) only lent their sub-tapes
    std::vector<char> used(m.integrals.size(), 0);
    for (int v = 0; v < m.n_variants(); v++) for (const Node& nd : m.eval(v).nodes) if (nd.op == GFH_INTEGRATE) used[(size_t)nd.a] = 1;
    for (bool more = true; more;) {
      more = false;
      for (size_t I = 0; I < m.integrals.size(); I++) {
        if (!used[I]) continue;
        std::vector<int> mem{m.integrals[I].integrand};
        if (I < m.alts.size()) mem.insert(mem.end(), m.alts[I].begin(), m.alts[I].end());
        for (int sidx : mem) for (const Node& nd : m.sub[(size_t)sidx].nodes) if (nd.op == GFH_INTEGRATE && !used[(size_t)nd.a]) { used[(size_t)nd.a] = 1; more = true; }
      }
    }
    // a call site after the call sites its integrand (any recording of it) calls itself
    std::vector<char> site_done(m.integrals.size(), 0);
    bool ok_sites = true;
    std::function<void(int, int)> emit_site = [&](int I, int depth) {
      if (site_done[(size_t)I] || !ok_sites) return;
      if (depth > 4) { ok_sites = false; *err = "integrate() call sites refer to each other in a cycle"; return; }
      site_done[(size_t)I] = 1;
      std::vector<int> mem{m.integrals[(size_t)I].integrand};
      if ((size_t)I < m.alts.size()) mem.insert(mem.end(), m.alts[(size_t)I].begin(), m.alts[(size_t)I].end());
      for (int S : mem) for (const Node& nd : m.sub[(size_t)S].nodes) if (nd.op == GFH_INTEGRATE) emit_site(nd.a, depth + 1);
      if (!ok_sites) return;
      for (int S : mem) { if (!sub_done[(size_t)S]) emit_integrand_functions(m, S, cfg, s); sub_done[(size_t)S] = 1; }
      if (site_is_family(m, I) && !emit_family(m, I, s, err)) { ok_sites = false; return; }
      emit_integral_site(m, I, cfg, s);
    };
    for (int I = 0; I < (int)m.integrals.size() && ok_sites; I++) if (used[(size_t)I]) emit_site(I, 0);
    if (!ok_sites) return false;
  }
  // The model bodies.  A model with ONE recorded path and no comparison gets the four point functions below under their plain
  // names.  A branching model (Model::branching) gets them once per variant (suffix _v<k>), the selector, and dispatchers
  // under the plain names that take the lane's slot as one more argument (for the report of an unseen turn).
  // register cap of the plain kernels (GenConfig::waves_per_eu): models with integrate() are bound by VALU issue and want waves, not registers
  if (cfg.waves_per_eu > 0) s << "#define GFH_OCC __attribute__((amdgpu_waves_per_eu(" << cfg.waves_per_eu << ")))\n";
  else s << "#define GFH_OCC\n";
  const bool multi = m.branching();
  const int V = m.n_variants();
  s << (multi ? "#define GFH_SLOT_DECL , const i64 SLOT\n#define GFH_SLOT(i) , (i64)(i)\n#define GFH_SLOT_PASS , SLOT\n"
              : "#define GFH_SLOT_DECL\n#define GFH_SLOT(i)\n#define GFH_SLOT_PASS\n");
  // Mesh hand-over (emit_integral_site, mesh_build): models with integrate() carry a per-lane record pointer and a mode
  // (0 none, 1 this pass records its bisections, 2 it replays recorded ones) from the kernels down to the call sites.
  const int n_mesh = cfg.finite_diff ? 0 : mesh_sites(m);
  if (n_mesh > 0)
    s << "#define GFH_MESH_STRIDE " << kMeshRecord * n_mesh << "\n#define GFH_MESH_DECL , unsigned char* __restrict__ MESH, const int MESH_MODE\n"
         "#define GFH_MESH_PASS , MESH, MESH_MODE\n#define GFH_MESH_AT(i) , (mesh ? mesh + (i64)(i) * GFH_MESH_STRIDE : (unsigned char*)nullptr), mesh_mode\n"
         "#define GFH_MESH_NONE , (unsigned char*)nullptr, 0\n#define GFH_MESH_KPARAMS , unsigned char* __restrict__ mesh, const int mesh_mode\n";
  else
    s << "#define GFH_MESH_DECL\n#define GFH_MESH_PASS\n#define GFH_MESH_AT(i)\n#define GFH_MESH_NONE\n#define GFH_MESH_KPARAMS\n";
  // Order of dispatch (context.cpp, build_orders): the cost of a point of a model with integrate() is the number of its bisections,
  // workgroups are dispatched in index order, and x-sorted data put the expensive tiles last -- they would run alone at the end.  The
  // plain kernels of such models take the tile / block a workgroup works on from a table sorted by measured cost, expensive first;
  // the sweep measures (shader clock per tile).  Which workgroup does a tile changes no result: every sum is defined on the fixed
  // partition.
  if (n_mesh > 0)        // (the same kernels that take the mesh arguments: the host passes both groups or neither)
    s << "#define GFH_HAS_ORDER 1\n#define GFH_ORDER_KPARAMS , const int* __restrict__ order, int* __restrict__ cost\n#define GFH_ORD(b) (order ? order[b] : (int)(b))\n";
  else
    s << "#define GFH_HAS_ORDER 0\n#define GFH_ORDER_KPARAMS\n#define GFH_ORD(b) ((int)(b))\n";
  // (kernels whose quadrature workspaces are the global pool take its address as their last argument and post it in LDS)
  if (m.has_integrals() && cfg.ws_global)
    s << "#define GFH_WSG_KPARAMS , double* __restrict__ wsg\n#define GFH_WSG_INIT if (threadIdx.x == 0) gfh_wsg_base = wsg; __syncthreads();\n";
  else
    s << "#define GFH_WSG 0\n#define GFH_WSG_KPARAMS\n#define GFH_WSG_INIT\n";
  const bool mesh_on = n_mesh > 0;
  const std::string A7 = "const double* __restrict__ AXP, const i64 LDA GFH_MESH_DECL";
  auto grad_expr = [&](const Gen& g, const SubTape& t, int j) {
    std::string e;
    for (int k = 0; k < (int)t.nodes.size(); k++)
      if (t.nodes[k].op == GFH_PARAM && t.nodes[k].a == active[j] && g.act[k]) e += (e.empty() ? "" : " + ") + g.b(k);
    return e.empty() ? std::string("0.0") : e;
  };
  auto emit_value_fn = [&](const SubTape& t, const std::string& sfx, const std::string& slot) {
    s << "\n// One data point, every parameter passive (chi2 path): value only.\n"
         "static __device__ __forceinline__ double gfh_point_value" << sfx << "(const double X, const double* __restrict__ P, int* STATUS,\n"
         "                                                         " << A7 << slot << ") {\n";
    s << "  GFH_LANE_STASH\n";
    Gen g(m, t, cfg.fast_div); g.mode = 0; g.mesh_top = mesh_on; g.analyse(none); g.emit_values(false);
    s << g.o.str() << "  return " << g.v(t.result) << ";\n}\n";
  };
  auto emit_grad_fn = [&](const SubTape& t, const std::string& sfx, const std::string& slot) {
    s << "\n// One data point, reverse mode: value F and gradient G[a] = dF/dp_active(a).\n"
         "static __device__ __forceinline__ void gfh_point_grad" << sfx << "(const double X, const double* __restrict__ P,\n"
         "                                                      double& F, double (&G)[GFH_NA], int* STATUS,\n"
         "                                                      " << A7 << slot << ") {\n";
    s << "  GFH_LANE_STASH\n";
    Gen g(m, t, cfg.fast_div); g.mode = 1; g.mesh_top = mesh_on; g.analyse(pa); g.emit_values(false); g.emit_reverse();
    s << g.o.str() << "  F = " << g.v(t.result) << ";\n";
    for (int j = 0; j < NA; j++) s << "  G[" << j << "] = " << grad_expr(g, t, j) << ";\n";
    s << "}\n";
  };
  auto emit_dd_fn = [&](const SubTape& t, const std::string& sfx, const std::string& slot) {
    s << "\n// One data point, forward mode: second directional derivative along DP (per-parameter d seeds).\n"
         "static __device__ __forceinline__ double gfh_point_dd" << sfx << "(const double X, const double* __restrict__ P,\n"
         "                                                      const double* __restrict__ DP, int* STATUS,\n"
         "                                                      " << A7 << slot << ") {\n";
    s << "  GFH_LANE_STASH\n";
    Gen g(m, t, cfg.fast_div); g.mode = 2; g.mesh_top = mesh_on; g.analyse(pa); g.emit_forward_all();
    s << g.o.str();
    if (g.act[t.result]) s << "  return " << g.dd(t.result) << ";\n"; else s << "  return 0.0;\n";
    s << "}\n";
  };
  if (multi) {
    s << "\n// A lane whose point takes a turn through eval() that no recorded variant covers: status 3 and one entry {slot, outcomes of\n"
         "// the guards passed, their number} in the report area behind the status word (context.h, kStatusBytes).\n"
         "#define GFH_UNSEEN_CAP " << 120 << "\n"
         "struct gfh_unseen { i64 slot; unsigned long long path; int n_guards; int pad; };\n"
         "static __device__ __attribute__((noinline)) void gfh_report_unseen(int* STATUS, const i64 slot, const unsigned long long path, const int ng) {\n"
         "  GFH_RAISE(STATUS, 3);\n"
         "  const unsigned k = __hip_atomic_fetch_add((unsigned*)((char*)STATUS + 64), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);\n"
         "  if (k < GFH_UNSEEN_CAP) { gfh_unseen* e = (gfh_unseen*)((char*)STATUS + 128) + k; e->slot = slot; e->path = path; e->n_guards = ng; e->pad = 0; }\n"
         "}\n";
    if (!emit_selector(m, s, err)) return false;
    for (int v = 0; v < V; v++) {
      const std::string sfx = "_v" + std::to_string(v);
      emit_value_fn(m.eval(v), sfx, "");
      if (!cfg.finite_diff) { emit_grad_fn(m.eval(v), sfx, ""); emit_dd_fn(m.eval(v), sfx, ""); }
    }
    s << "\nstatic __device__ __forceinline__ double gfh_point_value(const double X, const double* __restrict__ P, int* STATUS,\n"
         "                                                         " << A7 << " GFH_SLOT_DECL) {\n"
         "  unsigned long long path; int ng;\n  switch (gfh_select(X, P, STATUS, AXP, LDA, path, ng)) {\n";
    for (int v = 0; v < V; v++) s << "    case " << v << ": return gfh_point_value_v" << v << "(X, P, STATUS, AXP, LDA GFH_MESH_PASS);\n";
    s << "    default: gfh_report_unseen(STATUS, SLOT, path, ng); return 0.0;\n  }\n}\n";
    if (!cfg.finite_diff) {
      s << "\nstatic __device__ __forceinline__ void gfh_point_grad(const double X, const double* __restrict__ P,\n"
           "                                                      double& F, double (&G)[GFH_NA], int* STATUS,\n"
           "                                                      " << A7 << " GFH_SLOT_DECL) {\n"
           "  unsigned long long path; int ng;\n  switch (gfh_select(X, P, STATUS, AXP, LDA, path, ng)) {\n";
      for (int v = 0; v < V; v++) s << "    case " << v << ": gfh_point_grad_v" << v << "(X, P, F, G, STATUS, AXP, LDA GFH_MESH_PASS); break;\n";
      s << "    default:\n      F = 0.0;\n#pragma unroll\n      for (int a = 0; a < GFH_NA; a++) G[a] = 0.0;\n      gfh_report_unseen(STATUS, SLOT, path, ng);\n  }\n}\n";
      s << "\nstatic __device__ __forceinline__ double gfh_point_dd(const double X, const double* __restrict__ P,\n"
           "                                                      const double* __restrict__ DP, int* STATUS,\n"
           "                                                      " << A7 << " GFH_SLOT_DECL) {\n"
           "  unsigned long long path; int ng;\n  switch (gfh_select(X, P, STATUS, AXP, LDA, path, ng)) {\n";
      for (int v = 0; v < V; v++) s << "    case " << v << ": return gfh_point_dd_v" << v << "(X, P, DP, STATUS, AXP, LDA GFH_MESH_PASS);\n";
      s << "    default: gfh_report_unseen(STATUS, SLOT, path, ng); return 0.0;\n  }\n}\n";
    }
  }
  if (cfg.finite_diff) {
    // use_ad = .false. (gadfit.F90:684-687, 721-726): every parameter passive, the gradient by forward differences
    // (fitfunction.F90:155-174) and the second directional derivative by a central difference (188-203); the
    // value body is inlined once per evaluation the reference makes (it re-evaluates f(p); the value is the same).
    // (A branching model takes its branch afresh in each of those evaluations, as the reference's eval() does.)
    if (!multi) emit_value_fn(st, "", " GFH_SLOT_DECL");
    s << R"(
// One data point, finite differences (grad_finite, fitfunction.F90:155-174): step = sqrt(epsilon)*p, taken as
// (p + step) - p; G[a] = (f(p + step e_a) - f(p)) / step.
static __device__ __forceinline__ void gfh_point_grad(const double X, const double* __restrict__ P,
                                                      double& F, double (&G)[GFH_NA], int* STATUS,
                                                      const double* __restrict__ AXP, const i64 LDA GFH_MESH_DECL GFH_SLOT_DECL) {
  double Q[GFH_NP];
#pragma unroll
  for (int k = 0; k < GFH_NP; k++) Q[k] = P[k];
  F = gfh_point_value(X, Q, STATUS, AXP, LDA GFH_MESH_PASS GFH_SLOT_PASS);
)";
    for (int j = 0; j < NA; j++) {
      const int pj = active[j];
      // (fd_col_sets: the per-point columns of this evaluation are those the host tabulated at p + step e_j -- set 1 + j)
      const std::string axp = cfg.fd_col_sets && m.n_aux > 0 ? "AXP + (i64)" + std::to_string((j + 1) * m.n_aux) + " * LDA" : "AXP";
      s << "  { const double saved = Q[" << pj << "]; double step = 0x1p-26 * saved; Q[" << pj << "] = saved + step; step = Q[" << pj
        << "] - saved;\n    const double fp = gfh_point_value(X, Q, STATUS, " << axp << ", LDA GFH_MESH_PASS GFH_SLOT_PASS); Q[" << pj << "] = saved; G[" << j
        << "] = (fp - F) / step; }\n";
    }
    s << R"(}

// One data point, second directional derivative along DP by finite differences (dir_deriv_2nd_finite,
// fitfunction.F90:188-203): h = epsilon**(1/4); (f(p + h d) + f(p - h d) - 2 f(p)) / sqrt(epsilon).
static __device__ __forceinline__ double gfh_point_dd(const double X, const double* __restrict__ P,
                                                      const double* __restrict__ DP, int* STATUS,
                                                      const double* __restrict__ AXP, const i64 LDA GFH_MESH_DECL GFH_SLOT_DECL) {
  double Q[GFH_NP];
#pragma unroll
  for (int k = 0; k < GFH_NP; k++) Q[k] = P[k];
)";
    for (int j = 0; j < NA; j++) s << "  Q[" << active[j] << "] = P[" << active[j] << "] + 0x1p-13 * DP[" << active[j] << "];\n";
    s << "  double y = gfh_point_value(X, Q, STATUS, AXP, LDA GFH_MESH_PASS GFH_SLOT_PASS);\n";
    for (int j = 0; j < NA; j++) s << "  Q[" << active[j] << "] = P[" << active[j] << "] - 0x1p-13 * DP[" << active[j] << "];\n";
    s << "  y = y + gfh_point_value(X, Q, STATUS, AXP, LDA GFH_MESH_PASS GFH_SLOT_PASS);\n";
    for (int j = 0; j < NA; j++) s << "  Q[" << active[j] << "] = P[" << active[j] << "];\n";
    s << "  y = y - 2.0 * gfh_point_value(X, Q, STATUS, AXP, LDA GFH_MESH_PASS GFH_SLOT_PASS);\n  return y / 0x1p-26;\n}\n";
  } else if (!multi) {
    emit_grad_fn(st, "", " GFH_SLOT_DECL");
    emit_value_fn(st, "", " GFH_SLOT_DECL");
    emit_dd_fn(st, "", " GFH_SLOT_DECL");
  }
  // ---- hand-written kernel skeletons (the model body above is the only generated part)
  s << R"(
// Device layout (DESIGN.md "Data layout"): slots are data points padded per dataset to a
// multiple of the tile so every tile is full and belongs to one dataset; pad slots carry
// w = 0.  x, y, w, res, omega: [n_slots]; J: [NA][ldj] (parameter-major: a wave's store of
// one Jacobian column is 64 consecutive doubles = one fully coalesced 512 B write).
#define GFH_TILE GFH_BLOCK

// Robust cost of the C++ solver (lm_solver.cpp:255-284, 303-317): the weighted residual and its
// Jacobian row are scaled by sqrt(rho'(res^2)); chi2() stays the plain sum (lm_solver.cpp:513-529).
#if GFH_LOSS == 1
#define GFH_ROBUST(R, Wv) { const double ls_ = sqrt(1.0 / (1.0 + (R) * (R))); R *= ls_; Wv *= ls_; }
#elif GFH_LOSS == 2
#define GFH_ROBUST(R, Wv) { const double ls_ = (R) * (R) > 1.0 ? sqrt(1.0 / fabs(R)) : 1.0; R *= ls_; Wv *= ls_; }
#else
#define GFH_ROBUST(R, Wv)
#endif

typedef double gfh_d4 __attribute__((ext_vector_type(4)));
typedef int gfh_v2i __attribute__((ext_vector_type(2)));

// One wave stores 64 consecutive doubles at a WAVE-UNIFORM base: buffer_store_dwordx2 with
// the descriptor in SGPRs (built by scalar adds) and a 32-bit lane offset -- no per-lane
// 64-bit address VALU work and half the address bytes through the vector-memory issue path.
static __device__ __forceinline__ void gfh_store64(double* base, const int lane8, const double v) {
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, 512, 0x00020000);
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(gfh_v2i, v), rs, lane8, 0, 2);   // aux 2 = nt: written once, streamed
}

extern "C" __global__ __launch_bounds__(GFH_BLOCK) GFH_OCC
void gfh_k_sweep(const double* __restrict__ x, const double* __restrict__ y, const double* __restrict__ w,
                 GFH_PARS_DECL, const int* __restrict__ tile_ds, const int n_tiles,
                 double* __restrict__ res, double* __restrict__ J, const i64 ldj, int* __restrict__ status,
                 const double* __restrict__ aux, const i64 lda GFH_MESH_KPARAMS GFH_ORDER_KPARAMS GFH_WSG_KPARAMS) {
  GFH_WSG_INIT
  for (int tb = blockIdx.x; tb < n_tiles; tb += gridDim.x) {
    const int t = GFH_ORD(tb);
#if GFH_HAS_ORDER
    const unsigned long long c0_ = __builtin_amdgcn_s_memtime();
#endif
    const double* __restrict__ P = GFH_PARS_AT(tile_ds[t]);   // wave-uniform: scalar loads
    const i64 i = (i64)t * GFH_TILE + threadIdx.x;
    const i64 iw = (i64)t * GFH_TILE + 64 * __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // this wave's first slot
    const int lane8 = (threadIdx.x & 63) * 8;
    const double X = x[i], Y = y[i];
    double W = w[i];
    double F, G[GFH_NA];
    gfh_point_grad(X, P, F, G, status, aux + i, lda GFH_MESH_AT(i) GFH_SLOT(i));
    double R = (Y - F) * W;                     // gadfit.F90:682-683
    GFH_ROBUST(R, W)
    gfh_store64(res + iw, lane8, R);
#pragma unroll
    for (int a = 0; a < GFH_NA; a++) gfh_store64(J + (i64)a * ldj + iw, lane8, G[a] * W);   // gadfit.F90:689-690
#if GFH_HAS_ORDER
    if (cost && threadIdx.x == 0) { const unsigned long long d_ = (__builtin_amdgcn_s_memtime() - c0_) >> 6; cost[t] = d_ < 0x7fffffffull ? (int)d_ : 0x7fffffff; }
#endif
  }
}

// Cross-workgroup hand-off without fences (MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement &
// inter-workgroup visibility", valid forms and the table's first row): every handed-off byte is stored `sc1` (write-through,
// past the XCD's L2), every storing wave drains (`s_waitcnt vmcnt(0)`) before a workgroup barrier, one lane then adds to
// an agent-scope counter, and the workgroup whose add came last reads the bytes with `sc1` loads -- global_ instructions,
// never flat_: the pointers are cast to the global address space so the compiler cannot fall back to flat accesses.
#define GFH_GLOBAL(p) ((__attribute__((address_space(1))) __typeof__(*(p))*)(p))
#define GFH_ST_DEV(p, v) __hip_atomic_store(GFH_GLOBAL(p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define GFH_LD_DEV(p) __hip_atomic_load(GFH_GLOBAL(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define GFH_ST_SYS(p, v) __hip_atomic_store(GFH_GLOBAL(p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)

// lane l reads lane l + N of its row of 16 (DPP row_shl:N; lanes that would read past the row get 0): the low levels of a wave tree
template <int N> static __device__ __forceinline__ double gfh_row_down(const double v) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)b, 0x100 | N, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0x100 | N, 0xf, 0xf, true);
  return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
// the wave tree t_l += t_(l+32), (l+16), (l+8), (l+4), (l+2), (l+1) -- lane 0 ends with the sum, the additions of the __shfl_down loop it
// replaces bit for bit -- with the four levels inside a row as DPP moves instead of trips through the LDS crossbar (ds_bpermute)
static __device__ __forceinline__ double gfh_wave_sum(double t) {
  t += __shfl_down(t, 32, 64);
  t += __shfl_down(t, 16, 64);
  t += gfh_row_down<8>(t); t += gfh_row_down<4>(t); t += gfh_row_down<2>(t); t += gfh_row_down<1>(t);
  return t;
}

// (the fused kernels exist for up to 128 active parameters = 8 tiles, model.h kFusedMaxActive / fused_max_active; beyond that STEP 1 and STEP 2 run as
// gfh_k_sweep + k_gram_block launches; models whose quadrature workspaces are the global pool never run them: context.cpp, fusable_model)
#if GFH_NA <= GFH_FUSED_MAX && !GFH_WSG
// Fused STEP 1 + STEP 2 (gadfit.F90:675-699): the sweep above plus J^T J / J^T r / sum r^2 of
// the same points on the FP64 matrix cores, so J is written once and never re-read.
// One wave = 64 points per pass.  After the AD body each lane holds its point's weighted
// gradient; the wave transposes it through a private LDS stage [row = parameter][col = point]
// (stride 66 doubles: the 16 rows x 2 columns a half-wave reads hit 32 distinct bank pairs)
// into v_mfma_f64_16x16x4_f64 fragments: lane (r = l&15, q = l>>4) reads stage[16t+r][4s+q]
// for k-step s; the same fragment is A operand of row tile t and B operand of column tile t.
// Workgroup partial layout is identical to k_gram's, so the reduction/assembly kernels are shared.
#define GFH_T ((GFH_NA + 15) / 16)
#define GFH_NPAIR (GFH_T * (GFH_T + 1) / 2)
// GFH_HALF: the stage holds 32 points (stride 34) and a pass runs as two half-passes -- lanes 0-31 stage their points and the
// matrix cores take k-steps 0-7, then lanes 32-63 and k-steps 8-15: the k-steps in the order of the full stage, so the same sums
// bit for bit, for half the LDS per wave (more waves per SIMD; the gradient of the upper half waits in registers meanwhile).
#if GFH_HALF
#define GFH_S 34
#define GFH_NH 2
#define GFH_KS 8
#else
#define GFH_S 66
#define GFH_NH 1
#define GFH_KS 16
#endif
#if GFH_FUSED_WPE > 0
#define GFH_FOCC __attribute__((amdgpu_waves_per_eu(GFH_FUSED_WPE)))
#else
#define GFH_FOCC
#endif
// Descriptor of the fused kernel's tail (filled by the host, context.cpp TailDesc).
struct gfh_tail {
  const int* ds_first_gb;          // [nd+1] first workgroup of each dataset
  const int* inv;                  // [nd][dim] inverse of Jacobian_indices
  double* slice;                   // [nd][32][pstride] slice sums
  double* G;                       // [nd][pstride] per-dataset Gram images
  double* packed;                  // [dim*dim + dim + 1]
  double* host_out;                // pinned result mailbox
  unsigned long long* host_flag;   // pinned sequence flag
  unsigned* counters;              // [1 + nd*32], zero between launches
  int nd, dim, n_slices, pad;
};

// GFH_FW waves per workgroup, kept in phase (__syncthreads between the AD phase and the matrix phase):
// on gfx950 FP64 VALU and FP64 MFMA share one datapath and mixing the two kinds from different waves of
// a SIMD costs throughput (tools/microbench/fp64_overlap.hip), so a SIMD runs one kind at a time.
#define GFH_FTHREADS (64 * GFH_FW)
// Without the Jacobian store (gfh_set_keep_jacobian) the kernel carries another name, so that
// profiles keep the two apart.
#if GFH_STORE_J
#define GFH_K_SWEEP_GRAM gfh_k_sweep_gram
#else
#define GFH_K_SWEEP_GRAM gfh_k_sweep_gram_nostore
#endif
extern "C" __global__ __launch_bounds__(GFH_FTHREADS) GFH_FOCC
void GFH_K_SWEEP_GRAM(const double* __restrict__ x, const double* __restrict__ y, const double* __restrict__ w,
                      GFH_PARS_DECL, const i64* __restrict__ gb_start,
                      const int* __restrict__ gb_slots, const int* __restrict__ gb_ds,
                      double* __restrict__ res, double* __restrict__ J, const i64 ldj,
                      double* __restrict__ partial, const int pstride, int* __restrict__ status, const double* __restrict__ aux, const i64 lda,
                      const gfh_tail* __restrict__ tl, const unsigned long long seq, const int tail_mode) {
#if GFH_NA <= GFH_VALU_GRAM_MAX
  // Up to 8 active parameters a 16-row matrix tile would be half empty and the whole outer product of a point is
  // NA (NA + 1) / 2 + NA + 1 <= 45 multiply-adds: it stays on the VALU, in per-lane accumulators -- no LDS stage, no
  // transposition, no matrix instructions (16 of them per pass = 1024 cycles of the FP64 pipe against 180 here) -- and the
  // kernel is left with the store stream.  Every lane sums its own points pass by pass; wave tree and the waves in order
  // at the end (for sum r^2 that is gfh_k_chi2's order, as in the matrix path).  Same partial image as the matrix path.
  constexpr int NP_ = GFH_NA * (GFH_NA + 1) / 2, NACC = NP_ + GFH_NA + 1;
  __shared__ double red[GFH_FW][NACC];
  __shared__ double tot[NACC];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const i64 s0 = gb_start[blockIdx.x];
  const i64 e = s0 + gb_slots[blockIdx.x];                   // multiple of GFH_FTHREADS slots
  const double* __restrict__ P = GFH_PARS_AT(gb_ds[blockIdx.x]);
  double av[NACC];
#pragma unroll
  for (int k = 0; k < NACC; k++) av[k] = 0.0;
  i64 iw = s0 + 64 * __builtin_amdgcn_readfirstlane(wv);
  double Xc = (x + iw)[lane], Yc = (y + iw)[lane], Wc = (w + iw)[lane];
  // GFH_VAHEAD == 2 (GADFIT_HIP_VALU_AHEAD=2, an experiment of round 6 that changed nothing: model.h, valu_ahead_for): the inputs are
  // loaded TWO passes ahead through three rotating register sets; the default loads one pass ahead.
  auto body = [&](const double XC, const double YC, const double WC) __attribute__((always_inline)) {
    double* __restrict__ Jw = J + iw;
    double F, G[GFH_NA];
    gfh_point_grad(XC, P, F, G, status, aux + iw + lane, lda GFH_MESH_NONE GFH_SLOT(iw + lane));
    double R = (YC - F) * WC;                               // gadfit.F90:682-683
    double Wl = WC;
    GFH_ROBUST(R, Wl)
    gfh_store64(res + iw, lane * 8, R);
#pragma unroll
    for (int a = 0; a < GFH_NA; a++) {
      G[a] = G[a] * Wl;                                     // gadfit.F90:689-690
#if GFH_STORE_J
      gfh_store64(Jw + (i64)a * ldj, lane * 8, G[a]);
#endif
    }
    int p = 0;
#pragma unroll
    for (int a = 0; a < GFH_NA; a++)
#pragma unroll
      for (int b = a; b < GFH_NA; b++, p++) av[p] += G[a] * G[b];      // gadfit.F90:697
#pragma unroll
    for (int a = 0; a < GFH_NA; a++) av[NP_ + a] += G[a] * R;          // gadfit.F90:698
    av[NP_ + GFH_NA] += R * R;
  };
#if GFH_VAHEAD >= 2
  const i64 i1 = iw + GFH_FTHREADS < e ? iw + GFH_FTHREADS : iw;
  double Xn = (x + i1)[lane], Yn = (y + i1)[lane], Wn = (w + i1)[lane];
  double Xf = 0.0, Yf = 0.0, Wf = 0.0;                      // (the third register set: free at entry)
  asm volatile("" :: "v"(Xc), "v"(Yc), "v"(Wc), "v"(Xn), "v"(Yn), "v"(Wn));      // (see the matrix path: keeps the per-pass wait a counted one)
  // One pass: the loads of the pass after next go out into the free register set (XL ...), then the pass on (XC ...).  Three passes per
  // trip with the roles of the sets rotating, so that no register move has to wait for a load on its way.
  auto pass = [&](const double XC, const double YC, const double WC, double& XL, double& YL, double& WL) __attribute__((always_inline)) {
    const i64 in = iw + 2 * GFH_FTHREADS < e ? iw + 2 * GFH_FTHREADS : iw;
    XL = (x + in)[lane]; YL = (y + in)[lane]; WL = (w + in)[lane];
    body(XC, YC, WC);
    iw += GFH_FTHREADS;
  };
  while (iw < e) {
    pass(Xc, Yc, Wc, Xf, Yf, Wf);
    if (iw >= e) break;
    pass(Xn, Yn, Wn, Xc, Yc, Wc);
    if (iw >= e) break;
    pass(Xf, Yf, Wf, Xn, Yn, Wn);
  }
#else
  asm volatile("" :: "v"(Xc), "v"(Yc), "v"(Wc));             // (see the matrix path: keeps the per-pass wait a counted one)
  for (; iw < e; iw += GFH_FTHREADS) {
    const i64 in = iw + GFH_FTHREADS < e ? iw + GFH_FTHREADS : iw;
    const double Xn = (x + in)[lane], Yn = (y + in)[lane], Wn = (w + in)[lane];
    body(Xc, Yc, Wc);
    Xc = Xn; Yc = Yn; Wc = Wn;
  }
#endif
  // wave tree of the NACC sums: t_l += t_(l+32), += t_(l+16) through the LDS crossbar (ds_bpermute, what __shfl_down compiles to), then
  // += t_(l+8), (l+4), (l+2), (l+1) as DPP row shifts inside the 16 lanes of row 0 -- the additions __shfl_down's tree makes, the same
  // bits, with a third of the crossbar operations: 45 sums x 6 levels x 2 halves = 540 ds_bpermute per wave, all waves of the chip
  // at once at the end of the launch, were most of this kernel's epilogue (round 6)
#pragma unroll
  for (int k = 0; k < NACC; k++) {
    const double t = gfh_wave_sum(av[k]);
    if (lane == 0) red[wv][k] = t;
  }
  __syncthreads();
  if (threadIdx.x < NACC) {
    double t = red[0][threadIdx.x];
#pragma unroll
    for (int wq = 1; wq < GFH_FW; wq++) t += red[wq][threadIdx.x];
    tot[threadIdx.x] = t;
  }
  __syncthreads();
  double* out = partial + (i64)blockIdx.x * pstride;
  __shared__ double tail_img[273];                                   // (read by the single-workgroup tail below)
  for (int idx = threadIdx.x; idx < 273; idx += GFH_FTHREADS) {      // [16][16] tile (both triangles) | JTr[16] | rTr
    double t;
    if (idx < 256) {
      int a = idx >> 4, b = idx & 15;
      if (a > b) { const int t_ = a; a = b; b = t_; }
      t = b < GFH_NA ? tot[a * GFH_NA - a * (a - 1) / 2 + (b - a)] : 0.0;
    } else if (idx < 272) t = idx - 256 < GFH_NA ? tot[NP_ + idx - 256] : 0.0;
    else t = tot[NP_ + GFH_NA];
    GFH_ST_DEV(out + idx, t);
    tail_img[idx] = t;
  }
#elif GFH_COOP
  // ---- Workgroup-cooperative Gram (round 6): 81 ... 128 active parameters, 6 ... 8 tiles (model.h, fused_coop).  Up to 4 tiles every wave keeps ALL
  // T (T + 1) / 2 accumulator tiles for its own 64 points; that grows as T^2 (15 tiles = 120 registers at T = 5, 36 = 288 at T = 8)
  // next to a gradient of 2 NA registers that waits for the half-passes.  Here the waves still differentiate and stage their own
  // points (half stages: 32 points, stride 34) but after a barrier every wave reads ALL stages of the workgroup and owns a contiguous
  // run of the row-major list of tile pairs (GFH_CK = ceil(NPAIR / FW) of them: 4 ... 9 accumulator tiles): the registers stop
  // growing as T^2, nothing spills, and no cross-wave reduction of the pair images is left -- a pair's accumulator IS the workgroup's
  // sum.  Every pair is a plain v_mfma_f64_16x16x4_f64 on two fragments (the diagonal tiles too: their 4x4x4 form saves a third of
  // a tile's cycles but needs rotated fragments per owner); J^T r of tile t rides with the owner of pair (t, t) on the VALU; sum r^2
  // stays per lane over the lane's own points (gfh_k_chi2's order).  Points enter a pair's sum in the order stage of wave 0, 1, ...,
  // half 0 then half 1, pass by pass: fixed, so deterministic.
  constexpr int ROWS = 16 * GFH_T + 1;                       // parameters (padded to 16T) + residual row
  constexpr int STAGE = ROWS * GFH_S;
  constexpr int IMG = GFH_NPAIR * 256 + 16 * GFH_T + 1;      // the workgroup's sums (partial image)
  constexpr int EPI = GFH_T * 64 + 8 + IMG;                  // epilogue: J^T r fragments per tile | wave sums of r^2 | the image, laid over the stages
  __shared__ double lds[GFH_FW * STAGE > EPI ? GFH_FW * STAGE : EPI];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int r = lane & 15, q = lane >> 4;
  double* __restrict__ st = lds + wv * STAGE;
  const i64 s0 = gb_start[blockIdx.x];
  const i64 e = s0 + gb_slots[blockIdx.x];                   // multiple of GFH_FTHREADS slots
  const double* __restrict__ P = GFH_PARS_AT(gb_ds[blockIdx.x]);
#pragma unroll
  for (int a = GFH_NA; a < 16 * GFH_T; a++) st[a * GFH_S + (lane & 31)] = 0.0;      // padding rows: zero once
  // this wave's pairs: a contiguous run of the row-major upper triangle (generator: coop_defines -- per wave W the macros
  // GFH_CLOAD_W(B, U): the fragments of the DISTINCT tiles its pairs touch, read once per k-step (a run of K pairs touches about
  // K / 2 + 2 tiles: a third of the 2 K reads a pair-by-pair form makes, and the LDS reads were this kernel's bottleneck);
  // GFH_CMMA_W(B): its matrix instructions and, for its diagonal pairs, J^T r on the VALU; GFH_CPUT_W: its part of the epilogue)
  constexpr int CK = GFH_CK;
  const int wvu = __builtin_amdgcn_readfirstlane(wv);
  gfh_d4 acc[CK];
  double accr[CK];
#pragma unroll
  for (int k = 0; k < CK; k++) { acc[k] = (gfh_d4){0.0, 0.0, 0.0, 0.0}; accr[k] = 0.0; }
  double accc = 0.0;
  i64 iw = s0 + 64 * __builtin_amdgcn_readfirstlane(wv);
  double Xc = (x + iw)[lane], Yc = (y + iw)[lane], Wc = (w + iw)[lane];
  asm volatile("" :: "v"(Xc), "v"(Yc), "v"(Wc));             // (see the matrix path below: keeps the per-pass wait a counted one)
  for (; iw < e; iw += GFH_FTHREADS) {
    const i64 in = iw + GFH_FTHREADS < e ? iw + GFH_FTHREADS : iw;
    const double Xn = (x + in)[lane], Yn = (y + in)[lane], Wn = (w + in)[lane];
    double F, G[GFH_NA];
#if GFH_ABLATE & 8
    F = Xc * 0.5;
#pragma unroll
    for (int a = 0; a < GFH_NA; a++) G[a] = Xc + (double)a;
#else
    gfh_point_grad(Xc, P, F, G, status, aux + iw + lane, lda GFH_MESH_NONE GFH_SLOT(iw + lane));
#endif
    double R = (Yc - F) * Wc;                               // gadfit.F90:682-683
    double Wl = Wc;
    GFH_ROBUST(R, Wl)
    gfh_store64(res + iw, lane * 8, R);
    accc += R * R;                                          // every lane sums its own points pass by pass: the order gfh_k_chi2 uses
#pragma unroll
    for (int a = 0; a < GFH_NA; a++) {
      G[a] = G[a] * Wl;                                     // gadfit.F90:689-690
#if GFH_STORE_J
      gfh_store64(J + iw + (i64)a * ldj, lane * 8, G[a]);
#endif
    }
#pragma unroll 1
    for (int h = 0; h < 2; h++) {
      if ((lane >> 5) == h) {                                // this half's 32 points into the wave's stage
        st[16 * GFH_T * GFH_S + (lane & 31)] = R;
#pragma unroll
        for (int a = 0; a < GFH_NA; a++) st[a * GFH_S + (lane & 31)] = G[a];
      }
      __syncthreads();
      // 8 k-steps per stage, the stages in wave order; inside a stage the fragments of step u + 1 are read before the matrix
      // instructions of step u (the stage loop itself stays a loop: unrolled over all 8 FW steps the kernel spilled hundreds of registers)
#define GFH_CSTAGES(W_)                                                                                             \
      _Pragma("unroll 1") for (int sw = 0; sw < GFH_FW; sw++) {                                                     \
        const double* __restrict__ sb = lds + sw * STAGE;                                                           \
        double f[2][GFH_CND], fr[2];                                                                                 \
        GFH_CLOAD_##W_(0, 0)                                                                                        \
        _Pragma("unroll") for (int u = 0; u < 8; u++) {                                                             \
          /* the first matrix instruction of the step, THEN the next step's fragment reads (issued while it runs: a wave issues in   \
             order, and reads in front of the step's first matrix instruction cost their whole issue time), then the rest */         \
          __builtin_amdgcn_sched_barrier(0);                                                                        \
          if (u & 1) { GFH_CMMA0_##W_(1) } else { GFH_CMMA0_##W_(0) }                                               \
          __builtin_amdgcn_sched_barrier(0);                                                                        \
          if (u + 1 < 8) { if (u & 1) { GFH_CLOAD_##W_(0, u + 1) } else { GFH_CLOAD_##W_(1, u + 1) } }              \
          __builtin_amdgcn_sched_barrier(0);                                                                        \
          if (u & 1) { GFH_CMMA_##W_(1) } else { GFH_CMMA_##W_(0) }                                                 \
          __builtin_amdgcn_sched_barrier(0);                                                                        \
        }                                                                                                           \
      }
#if GFH_ABLATE & 4
#define GFH_MFMA(A_, B_, C_) asm volatile("" :: "v"(A_), "v"(B_))
#else
#define GFH_MFMA(A_, B_, C_) C_ = __builtin_amdgcn_mfma_f64_16x16x4f64(A_, B_, C_, 0, 0, 0)
#endif
      if (wvu == 0) { GFH_CSTAGES(0) }
#if GFH_FW > 1
      else if (wvu == 1) { GFH_CSTAGES(1) }
#endif
#if GFH_FW > 2
      else if (wvu == 2) { GFH_CSTAGES(2) }
      else { GFH_CSTAGES(3) }
#endif
      __syncthreads();
    }
    Xc = Xn; Yc = Yn; Wc = Wn;
  }
  // epilogue: the owners write their pairs straight into the workgroup's image (global partial + the LDS copy the one-workgroup tail reads)
  double* vecs = lds;
  double* wsum = lds + GFH_T * 64;
  double* tail_img = wsum + 8;
  double* out = partial + (i64)blockIdx.x * pstride;
#define GFH_CPUT(K_, PP_)                                                                                            \
  _Pragma("unroll") for (int j = 0; j < 4; j++) {       /* f64 16x16 C/D map: row = (l>>4) + 4*reg, column = l & 15 */ \
    const int idx = (PP_) * 256 + (q + 4 * j) * 16 + r;                                                              \
    GFH_ST_DEV(out + idx, acc[K_][j]);                                                                               \
    tail_img[idx] = acc[K_][j]; }
#define GFH_CPUTR(K_, T_) vecs[(T_) * 64 + lane] = accr[K_];
  if (wvu == 0) { GFH_CPUT_0 }
#if GFH_FW > 1
  else if (wvu == 1) { GFH_CPUT_1 }
#endif
#if GFH_FW > 2
  else if (wvu == 2) { GFH_CPUT_2 }
  else { GFH_CPUT_3 }
#endif
  {
    const double t = gfh_wave_sum(accc);                     // wave tree, then the waves in order: gfh_k_chi2's order
    if (lane == 0) wsum[wv] = t;
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < 16 * GFH_T; idx += GFH_FTHREADS) {
    const int t = idx >> 4, rr_ = idx & 15;
    const double sacc = ((vecs[t * 64 + rr_] + vecs[t * 64 + 16 + rr_]) + vecs[t * 64 + 32 + rr_]) + vecs[t * 64 + 48 + rr_];
    GFH_ST_DEV(out + GFH_NPAIR * 256 + idx, sacc);
    tail_img[GFH_NPAIR * 256 + idx] = sacc;
  }
  if (threadIdx.x == 0) {
    double sacc = wsum[0];
#pragma unroll
    for (int wq = 1; wq < GFH_FW; wq++) sacc += wsum[wq];
    GFH_ST_DEV(out + GFH_NPAIR * 256 + 16 * GFH_T, sacc);
    tail_img[GFH_NPAIR * 256 + 16 * GFH_T] = sacc;
  }
#else
  constexpr int ROWS = 16 * GFH_T + 1;                       // parameters (padded to 16T) + residual row
  constexpr int STAGE = ROWS * GFH_S;
  constexpr int RED = GFH_NPAIR * 256 + GFH_T * 64 + 4;      // cross-wave reduction image (as k_gram)
  constexpr int IMG = GFH_NPAIR * 256 + 16 * GFH_T + 1;      // the workgroup's own sums (partial image), kept for the single-workgroup tail
#if GFH_RED1
  // 5 and 6 tiles: ONE pair image that the waves add into in order + the waves' J^T r / r^T r vectors + the workgroup's sums,
  // laid over the stages once the pass loop is done (model.h, fused_lds_bytes_for)
  constexpr int VEC = GFH_T * 64 + 4;
  constexpr int RED1 = GFH_NPAIR * 256 + GFH_FW * VEC + IMG;
  __shared__ double lds[GFH_FW * STAGE > RED1 ? GFH_FW * STAGE : RED1];
#else
  __shared__ double lds[GFH_FW * (STAGE > RED ? STAGE : RED)];
#endif
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int r = lane & 15, q = lane >> 4;
  double* __restrict__ st = lds + wv * STAGE;
  const i64 s0 = gb_start[blockIdx.x];
  const i64 e = s0 + gb_slots[blockIdx.x];                   // multiple of GFH_FTHREADS slots
  const double* __restrict__ P = GFH_PARS_AT(gb_ds[blockIdx.x]);

  // rows GFH_NA .. 16T-1 of the stage are padding: zero once
#pragma unroll
  for (int a = GFH_NA; a < 16 * GFH_T; a++) st[a * GFH_S + (lane & (64 / GFH_NH - 1))] = 0.0;

  gfh_d4 acc[GFH_NPAIR];
#pragma unroll
  for (int p = 0; p < GFH_NPAIR; p++) acc[p] = (gfh_d4){0.0, 0.0, 0.0, 0.0};
  // Diagonal tiles: only 10 of the 16 4x4 blocks of a symmetric 16x16 tile are distinct, and
  // v_mfma_f64_4x4x4_4b_f64 (four independent 4x4 blocks, 17.5 cycles against 64, tools/microbench/mfma_4x4.hip) takes
  // its A operand in exactly the fragment layout of the 16x16x4 form (lane = 16 k + 4 block + row).  With B = the same
  // fragment it yields the four diagonal blocks (b,b); with B read from rows rotated by one block, (b,b+1 mod 4) -- which is
  // (0,1) (1,2) (2,3) and (3,0) = (0,3) transposed; the remaining (0,2) (1,3) of TWO tiles share one more instruction whose
  // lanes of blocks 0,1 read the first tile and those of blocks 2,3 the second (rows rotated by two blocks for B).
  // 2.5 x 17.5 cycles per diagonal tile and k-step instead of 64.
  constexpr int NMIX = GFH_T / 2;
  double dga[GFH_T], dgb[GFH_T], dgm[NMIX + 1];
#pragma unroll
  for (int t = 0; t < GFH_T; t++) dga[t] = dgb[t] = 0.0;
#pragma unroll
  for (int m = 0; m <= NMIX; m++) dgm[m] = 0.0;
  const int r4 = (r + 4) & 15, r8 = (r + 8) & 15, hi = r >> 3;
  double accr[GFH_T];
#pragma unroll
  for (int t = 0; t < GFH_T; t++) accr[t] = 0.0;
  double accc = 0.0;

  // iw: first slot of this wave's pass, kept wave-uniform (SGPRs) so every global access is
  // "scalar base + lane*8": no per-lane 64-bit address arithmetic, 32-bit offsets to the TA
  i64 iw = s0 + 64 * __builtin_amdgcn_readfirstlane(wv);
  // every workgroup owns at least one whole pass (gb_slots is a positive multiple of GFH_FTHREADS)
  double Xc = (x + iw)[lane], Yc = (y + iw)[lane], Wc = (w + iw)[lane];
  // The first pass's inputs are consumed here, outside the loop.  vmcnt counts loads and stores in
  // issue order; if these loads were still pending at the loop header the compiler would have to
  // wait for the loop-carried inputs with vmcnt(2) -- correct for this entry path, but on the
  // back edge it means "every Jacobian store of the previous pass has completed": a full drain of
  // the store queue at the top of every pass.  With a clean entry state the wait inside the loop
  // is the counted one (the 3 prefetch loads are OLDER than the pass's stores).
  asm volatile("" :: "v"(Xc), "v"(Yc), "v"(Wc));
#if GFH_MATRIX_PRIO < 0
  __builtin_amdgcn_s_setprio(-(GFH_MATRIX_PRIO));            // (the AD phase of the first pass; GenConfig::matrix_prio)
#endif
  for (; iw < e; iw += GFH_FTHREADS) {
    // prefetch the next pass's inputs before the long compute phase (the last pass re-reads its
    // own: no branch, so the number of memory operations in flight is the same on every path)
    const i64 in = iw + GFH_FTHREADS < e ? iw + GFH_FTHREADS : iw;
    const double Xn = (x + in)[lane], Yn = (y + in)[lane], Wn = (w + in)[lane];
    double* __restrict__ Jw = J + iw;
    double F, G[GFH_NA];
#if GFH_ABLATE & 8
    F = Xc * 0.5;
#pragma unroll
    for (int a = 0; a < GFH_NA; a++) G[a] = Xc + (double)a;
#else
    gfh_point_grad(Xc, P, F, G, status, aux + iw + lane, lda GFH_MESH_NONE GFH_SLOT(iw + lane));
#endif
    double R = (Yc - F) * Wc;                               // gadfit.F90:682-683
    double Wl = Wc;
    GFH_ROBUST(R, Wl)
    gfh_store64(res + iw, lane * 8, R);
    accc += R * R;                                          // every lane sums its own points pass by pass: the order gfh_k_chi2 uses
#if !(GFH_ABLATE & 1) && !GFH_HALF
    st[16 * GFH_T * GFH_S + lane] = R;
#endif
#pragma unroll
    for (int a = 0; a < GFH_NA; a++) {
      G[a] = G[a] * Wl;                                     // gadfit.F90:689-690
#if !(GFH_ABLATE & 1) && !GFH_HALF
      st[a * GFH_S + lane] = G[a];
#elif GFH_ABLATE & 1
      asm volatile("" :: "v"(G[a]));
#endif
    }
#if GFH_STORE_J
    // phase alignment (the stage itself is wave-private): with the Jacobian stores in the matrix phase the kernel is faster when
    // the waves of a workgroup are in the same phase (0.53 against 0.58 ms); without them it is the FP64 pipe alone and any
    // barrier is idle time (0.352 against 0.334 ms)
    __syncthreads();
#endif
    // k-steps: the fragment reads of step s+1 are issued before the MFMAs of step s so the
    // LDS latency hides under the 64-cycle matrix instructions (sched_barrier pins the order)
    // GFH_AHEAD sets of fragments in flight: the reads of step s + GFH_AHEAD go out before the matrix instructions of step s
    double fn[GFH_AHEAD][GFH_T], f4n[GFH_AHEAD][GFH_T], man[GFH_AHEAD][NMIX + 1], mbn[GFH_AHEAD][NMIX + 1], rn[GFH_AHEAD];
#if GFH_ABLATE & 2
#define GFH_FRAGS(S_)                                                                                            \
    _Pragma("unroll") for (int t = 0; t < GFH_T; t++) {                                                          \
      fn[(S_) % GFH_AHEAD][t] = G[(2 * (S_) + t) % GFH_NA];                                                      \
      f4n[(S_) % GFH_AHEAD][t] = G[(2 * (S_) + t + 5) % GFH_NA];                                                 \
    }                                                                                                            \
    _Pragma("unroll") for (int m = 0; m <= NMIX; m++) {                                                          \
      man[(S_) % GFH_AHEAD][m] = G[(2 * (S_) + m + 9) % GFH_NA];                                                 \
      mbn[(S_) % GFH_AHEAD][m] = G[(2 * (S_) + m + 13) % GFH_NA];                                                \
    }                                                                                                            \
    rn[(S_) % GFH_AHEAD] = G[(S_) % GFH_NA];
#else
#define GFH_FRAGS(S_)                                                                                            \
    _Pragma("unroll") for (int t = 0; t < GFH_T; t++) {                                                          \
      fn[(S_) % GFH_AHEAD][t] = st[(16 * t + r) * GFH_S + 4 * (S_) + q];                                         \
      f4n[(S_) % GFH_AHEAD][t] = st[(16 * t + r4) * GFH_S + 4 * (S_) + q];                                       \
    }                                                                                                            \
    _Pragma("unroll") for (int m = 0; m < NMIX; m++) {                                                           \
      man[(S_) % GFH_AHEAD][m] = st[(16 * (2 * m + hi) + r) * GFH_S + 4 * (S_) + q];                             \
      mbn[(S_) % GFH_AHEAD][m] = st[(16 * (2 * m + hi) + r8) * GFH_S + 4 * (S_) + q];                            \
    }                                                                                                            \
    if (GFH_T & 1) mbn[(S_) % GFH_AHEAD][NMIX] = st[(16 * (GFH_T - 1) + r8) * GFH_S + 4 * (S_) + q];             \
    rn[(S_) % GFH_AHEAD] = st[16 * GFH_T * GFH_S + 4 * (S_) + q];
#endif
#if GFH_MATRIX_PRIO > 0
    __builtin_amdgcn_s_setprio(GFH_MATRIX_PRIO);
#elif GFH_MATRIX_PRIO < 0
    __builtin_amdgcn_s_setprio(0);
#endif
#pragma unroll
    for (int h = 0; h < GFH_NH; h++) {
#if GFH_HALF && !(GFH_ABLATE & 1)
      // this half's 32 points into the stage (the fragment reads of the half before are older LDS operations of this wave:
      // the LDS executes a wave's operations in order)
      if ((lane >> 5) == h) {
        st[16 * GFH_T * GFH_S + (lane & 31)] = R;
#pragma unroll
        for (int a = 0; a < GFH_NA; a++) st[a * GFH_S + (lane & 31)] = G[a];
      }
#endif
#pragma unroll
      for (int s = 0; s < GFH_AHEAD; s++) { GFH_FRAGS(s) }
#pragma unroll
      for (int s = 0; s < GFH_KS; s++) {
        double fa[GFH_T], f4[GFH_T], ma[NMIX + 1], mb[NMIX + 1];
#pragma unroll
        for (int t = 0; t < GFH_T; t++) { fa[t] = fn[s % GFH_AHEAD][t]; f4[t] = f4n[s % GFH_AHEAD][t]; }
#pragma unroll
        for (int m = 0; m <= NMIX; m++) { ma[m] = man[s % GFH_AHEAD][m]; mb[m] = mbn[s % GFH_AHEAD][m]; }
        const double rr = rn[s % GFH_AHEAD];
#if !GFH_FRAG_LATE
        if (s + GFH_AHEAD < GFH_KS) { GFH_FRAGS(s + GFH_AHEAD) }
#endif
        __builtin_amdgcn_sched_barrier(0);
        int p = 0;
#if GFH_ABLATE & 4
#pragma unroll
        for (int t = 0; t < GFH_T; t++) asm volatile("" :: "v"(fa[t]), "v"(f4[t]));
#pragma unroll
        for (int m = 0; m <= NMIX; m++) asm volatile("" :: "v"(ma[m]), "v"(mb[m]));
#else
#pragma unroll
        for (int ti = 0; ti < GFH_T; ti++)
#pragma unroll
          for (int tj = ti; tj < GFH_T; tj++, p++)
            if (tj > ti) acc[p] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[ti], fa[tj], acc[p], 0, 0, 0);
#if GFH_FRAG_LATE
        // the next step's fragment reads go out BEHIND this step's 64-cycle matrix instructions (issued, they run by themselves):
        // the wave's LDS instructions then cost the shared FP64 pipe no idle issue slots
        __builtin_amdgcn_sched_barrier(0);
        if (s + GFH_AHEAD < GFH_KS) { GFH_FRAGS(s + GFH_AHEAD) }
        __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
        for (int t = 0; t < GFH_T; t++) {
          dga[t] = __builtin_amdgcn_mfma_f64_4x4x4f64(fa[t], fa[t], dga[t], 0, 0, 0);
          dgb[t] = __builtin_amdgcn_mfma_f64_4x4x4f64(fa[t], f4[t], dgb[t], 0, 0, 0);
        }
#pragma unroll
        for (int m = 0; m < NMIX; m++) dgm[m] = __builtin_amdgcn_mfma_f64_4x4x4f64(ma[m], mb[m], dgm[m], 0, 0, 0);
        if (GFH_T & 1) dgm[NMIX] = __builtin_amdgcn_mfma_f64_4x4x4f64(fa[GFH_T - 1], mb[NMIX], dgm[NMIX], 0, 0, 0);
#endif
#pragma unroll
        for (int t = 0; t < GFH_T; t++) accr[t] += fa[t] * rr;
#if GFH_STORE_J
        // Jacobian columns leave for HBM a few per k-step, under the matrix instructions,
        // instead of as one burst that stalls the wave on a full store queue
#pragma unroll
        for (int a = (h * GFH_KS + s) * ((GFH_NA + 15) / 16); a < (h * GFH_KS + s + 1) * ((GFH_NA + 15) / 16) && a < GFH_NA; a++)
          gfh_store64(Jw + (i64)a * ldj, lane * 8, G[a]);
#endif
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#if GFH_MATRIX_PRIO > 0
    __builtin_amdgcn_s_setprio(0);
#elif GFH_MATRIX_PRIO < 0
    __builtin_amdgcn_s_setprio(-(GFH_MATRIX_PRIO));
#endif
#if GFH_STORE_J
    __syncthreads();
#endif
    Xc = Xn; Yc = Yn; Wc = Wn;
  }

#if GFH_MATRIX_PRIO
  __builtin_amdgcn_s_setprio(0);
#endif
#if GFH_RED1
  // cross-wave reduction, 5 and 6 tiles: the waves add their accumulators into ONE image in wave order -- ((w0 + w1) + w2) + w3,
  // the order in which the per-wave images of the smaller kernels are added -- then J^T r and r^T r from per-wave vectors as there
  __syncthreads();                                           // (every wave is done with its stage: the image lies over them)
  double* img1 = lds;
  double* vecs = lds + GFH_NPAIR * 256;
  double* tail_img = vecs + GFH_FW * VEC;
#define GFH_PUT(IDX_, V_) { if (first) img1[IDX_] = (V_); else img1[IDX_] += (V_); }
  for (int wq = 0; wq < GFH_FW; wq++) {
    if (wv == wq) {
      const bool first = wq == 0;
      int p = 0;
#pragma unroll
      for (int ti = 0; ti < GFH_T; ti++)
#pragma unroll
        for (int tj = ti; tj < GFH_T; tj++, p++) {
          if (tj > ti) {
#pragma unroll
            for (int j = 0; j < 4; j++) GFH_PUT(p * 256 + (q + 4 * j) * 16 + r, acc[p][j])
          } else {
            const int row = (r & 12) + q;
            GFH_PUT(p * 256 + row * 16 + r, dga[ti])
            GFH_PUT(p * 256 + row * 16 + r4, dgb[ti])
            GFH_PUT(p * 256 + r4 * 16 + row, dgb[ti])
          }
        }
#pragma unroll
      for (int m = 0; m < NMIX; m++) {
        const int t = 2 * m + hi, pd = t * GFH_T - t * (t - 1) / 2, row = (r & 12) + q;
        GFH_PUT(pd * 256 + row * 16 + r8, dgm[m])
        GFH_PUT(pd * 256 + r8 * 16 + row, dgm[m])
      }
      if ((GFH_T & 1) && !hi) {
        const int t = GFH_T - 1, pd = t * GFH_T - t * (t - 1) / 2, row = (r & 12) + q;
        GFH_PUT(pd * 256 + row * 16 + r8, dgm[NMIX])
        GFH_PUT(pd * 256 + r8 * 16 + row, dgm[NMIX])
      }
    }
    __syncthreads();
  }
#undef GFH_PUT
  {
    double* myvec = vecs + wv * VEC;
#pragma unroll
    for (int t = 0; t < GFH_T; t++) myvec[t * 64 + lane] = accr[t];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) accc += __shfl_down(accc, off, 64);
    if (lane == 0) myvec[GFH_T * 64] = accc;
  }
  __syncthreads();
  double* out = partial + (i64)blockIdx.x * pstride;
  for (int idx = threadIdx.x; idx < GFH_NPAIR * 256; idx += GFH_FTHREADS) {
    const double sacc = img1[idx];
    GFH_ST_DEV(out + idx, sacc);
    tail_img[idx] = sacc;
  }
  for (int idx = threadIdx.x; idx < 16 * GFH_T; idx += GFH_FTHREADS) {
    const int t = idx >> 4, rr_ = idx & 15;
    double sacc = 0.0;
#pragma unroll
    for (int wq = 0; wq < 4 * GFH_FW; wq++) sacc += vecs[(wq >> 2) * VEC + t * 64 + (wq & 3) * 16 + rr_];
    GFH_ST_DEV(out + GFH_NPAIR * 256 + idx, sacc);
    tail_img[GFH_NPAIR * 256 + idx] = sacc;
  }
  if (threadIdx.x == 0) {
    double sacc = vecs[GFH_T * 64];
#pragma unroll
    for (int wq = 1; wq < GFH_FW; wq++) sacc += vecs[wq * VEC + GFH_T * 64];
    GFH_ST_DEV(out + GFH_NPAIR * 256 + 16 * GFH_T, sacc);
    tail_img[GFH_NPAIR * 256 + 16 * GFH_T] = sacc;
  }
#else
  // cross-wave reduction in fixed order (deterministic), same image as k_gram
  __syncthreads();
  double* mine = lds + wv * RED;
  {
    int p = 0;
#pragma unroll
    for (int ti = 0; ti < GFH_T; ti++)
#pragma unroll
      for (int tj = ti; tj < GFH_T; tj++, p++) {
        if (tj > ti) {
#pragma unroll
          for (int j = 0; j < 4; j++) mine[p * 256 + (q + 4 * j) * 16 + r] = acc[p][j];   // f64 16x16 C/D map: row = (l>>4) + 4*reg
        } else {
          // 4x4x4 C/D map: lane = 16 row + 4 block + column; both triangles of the tile image are filled
          const int row = (r & 12) + q;
          mine[p * 256 + row * 16 + r] = dga[ti];
          mine[p * 256 + row * 16 + r4] = dgb[ti];
          mine[p * 256 + r4 * 16 + row] = dgb[ti];
        }
      }
#pragma unroll
    for (int m = 0; m < NMIX; m++) {
      const int t = 2 * m + hi, pd = t * GFH_T - t * (t - 1) / 2, row = (r & 12) + q;
      mine[pd * 256 + row * 16 + r8] = dgm[m];
      mine[pd * 256 + r8 * 16 + row] = dgm[m];
    }
    if ((GFH_T & 1) && !hi) {             // (blocks 2,3 of the unpaired tile hold the transposes of blocks 0,1: one writer each)
      const int t = GFH_T - 1, pd = t * GFH_T - t * (t - 1) / 2, row = (r & 12) + q;
      mine[pd * 256 + row * 16 + r8] = dgm[NMIX];
      mine[pd * 256 + r8 * 16 + row] = dgm[NMIX];
    }
  }
#pragma unroll
  for (int t = 0; t < GFH_T; t++) mine[GFH_NPAIR * 256 + t * 64 + lane] = accr[t];
  // sum r^2: wave tree, then the waves in order -- the same tree and order as gfh_k_chi2, so chi2() at the
  // parameters of a sweep returns bitwise this sweep's sum
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) accc += __shfl_down(accc, off, 64);
  if (lane == 0) mine[GFH_NPAIR * 256 + GFH_T * 64] = accc;
  __syncthreads();
  double* out = partial + (i64)blockIdx.x * pstride;
  // (the sums also stay in LDS for the single-workgroup tail below: the pair images in tail_pairs, J^T r and r^T r behind them)
  __shared__ double tail_img[GFH_NPAIR * 256 + 16 * GFH_T + 1];
  for (int idx = threadIdx.x; idx < GFH_NPAIR * 256; idx += GFH_FTHREADS) {
    double sacc = lds[idx];
#pragma unroll
    for (int wq = 1; wq < GFH_FW; wq++) sacc += lds[wq * RED + idx];
    GFH_ST_DEV(out + idx, sacc);
    tail_img[idx] = sacc;
  }
  for (int idx = threadIdx.x; idx < 16 * GFH_T; idx += GFH_FTHREADS) {
    const int t = idx >> 4, rr_ = idx & 15;
    double sacc = 0.0;
#pragma unroll
    for (int wq = 0; wq < 4 * GFH_FW; wq++) sacc += lds[(wq >> 2) * RED + GFH_NPAIR * 256 + t * 64 + (wq & 3) * 16 + rr_];
    GFH_ST_DEV(out + GFH_NPAIR * 256 + idx, sacc);
    tail_img[GFH_NPAIR * 256 + idx] = sacc;
  }
  if (threadIdx.x == 0) {
    double sacc = lds[GFH_NPAIR * 256 + GFH_T * 64];
#pragma unroll
    for (int wq = 1; wq < GFH_FW; wq++) sacc += lds[wq * RED + GFH_NPAIR * 256 + GFH_T * 64];
    GFH_ST_DEV(out + GFH_NPAIR * 256 + 16 * GFH_T, sacc);
    tail_img[GFH_NPAIR * 256 + 16 * GFH_T] = sacc;
  }
#endif  // GFH_RED1
#endif  // GFH_NA <= GFH_VALU_GRAM_MAX
  if (!tail_mode) return;

  // ---- tail (STEP 2's sum over workgroups, gadfit.F90:698-699, and the scatter through
  // Jacobian_indices): what k_reduce_partials + k_assemble + k_publish do as three more launches,
  // done here by the workgroups that finish last, in exactly their order of additions (bitwise the
  // same numbers).  Level 1: the workgroups b0+sl, b0+sl+32, ... of a dataset form slice sl; the last
  // of them to arrive adds their partials in ascending order.  Level 2: the workgroup that completes
  // the last slice adds the 32 slice sums of every dataset in slice order, assembles the packed
  // [JTJ | JTres | chi2] and (tail_mode 2) writes it, the status word and the call's sequence number
  // into the host mailbox.
  // Cross-workgroup traffic (partials, slice sums, counters) moves ONLY through device-scope atomic
  // loads/stores (sc1: written through to / read from memory, past the per-XCD L2s, which are not
  // coherent with each other), each producer waiting for its stores to be acknowledged (vmcnt(0))
  // before its arrival is counted.  A release fence would do the same job by writing back the whole
  // L2 -- which in this kernel is full of dirty Jacobian lines: measured +50 us per launch.
  constexpr int W = GFH_NPAIR * 256 + 16 * GFH_T + 1;
  __shared__ int role;
  const int d = gb_ds[blockIdx.x];
  // Assembly of the packed [JTJ | JTres | chi2] from per-dataset images (source `src`, image of dataset dd at src + dd * stride,
  // datasets [d_lo, d_hi)), the scatter through Jacobian_indices, and (tail_mode 2) the host mailbox.
  auto assemble_and_post = [&](auto at, const int d_lo, const int d_hi) {       // at(dd, k): entry k of dataset dd's image
    const int dim = tl->dim;
    const i64 nn = (i64)dim * dim, total = nn + dim + 1;
    const int* __restrict__ inv = tl->inv;
    double* packed = tl->packed;
    double* host_out = tl->host_out;
    for (i64 idx = threadIdx.x; idx < total; idx += GFH_FTHREADS) {
      double v = 0.0;
      if (idx < nn) {
        const int col = (int)(idx / dim), row = (int)(idx % dim);
        for (int dd = d_lo; dd < d_hi; dd++) {
          int a = inv[dd * dim + row], b = inv[dd * dim + col];
          if (a < 0 || b < 0) continue;
          if (a > b) { const int t_ = a; a = b; b = t_; }     // upper triangle of tile pairs is stored
          const int ti = a >> 4, tj = b >> 4;
          const int p = ti * GFH_T - ti * (ti - 1) / 2 + (tj - ti);
          v += at(dd, p * 256 + (a & 15) * 16 + (b & 15));
        }
      } else if (idx < nn + dim) {
        const int row = (int)(idx - nn);
        for (int dd = d_lo; dd < d_hi; dd++) {
          const int a = inv[dd * dim + row];
          if (a >= 0) v += at(dd, GFH_NPAIR * 256 + a);
        }
      } else {
        for (int dd = d_lo; dd < d_hi; dd++) v += at(dd, GFH_NPAIR * 256 + 16 * GFH_T);
      }
      packed[idx] = v;
      if (tail_mode == 2) GFH_ST_SYS(host_out + idx, v);       // pinned host memory is uncached: the store goes straight out
    }
    if (tail_mode != 2) {
      // (one process per GPU: element `total` of the packed buffer is the status slot of the cross-rank sum that follows --
      // 0, 1, 4096, 2^24 by code, so the sum over the ranks still tells which codes occurred: context.cpp, allreduce_sum)
      if (threadIdx.x == 0) packed[total] = GFH_STATUS_SLOT(GFH_LD_DEV(status));
      return;
    }
    // the status word travels with the data (every workgroup's status updates were acknowledged before its arrival was
    // counted, this workgroup's own before the barrier in front of this call): ONE wait for the stores to host memory, then the flag
    if (threadIdx.x == 0) GFH_ST_SYS(host_out + total, (double)GFH_LD_DEV(status));
    asm volatile("s_waitcnt vmcnt(0)\n" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(GFH_GLOBAL(tl->host_flag), seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  };
  if (gridDim.x == 1) {
    // One workgroup (the small fits most of gadfit's use consists of): its partial IS the sum over workgroups of its dataset --
    // the two levels of the hand-off below would add 0.0 to it twice and cost five round trips to memory.  The same numbers
    // (0.0 + t in the assembly, as there), bitwise.
    // The sums are still in LDS (tail_img, written next to the partial image above): no trip through memory either.
    __syncthreads();
    assemble_and_post([&](int, int k) { return tail_img[k]; }, d, d + 1);
    return;
  }
  const int b0 = tl->ds_first_gb[d], b1 = tl->ds_first_gb[d + 1];
  const int sl = ((int)blockIdx.x - b0) & 31;
  unsigned* cnt = tl->counters;
  asm volatile("s_waitcnt vmcnt(0)\n" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned members = (unsigned)((b1 - b0 - sl + 31) >> 5);
    const bool last = __hip_atomic_fetch_add(GFH_GLOBAL(cnt + 1 + d * 32 + sl), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == members - 1;
    if (last) GFH_ST_DEV(cnt + 1 + d * 32 + sl, 0u);          // ready for the next launch (stream-ordered)
    role = last;
  }
  __syncthreads();
  if (!role) return;
  {
    double* sdst = tl->slice + ((i64)d * 32 + sl) * pstride;
    for (int idx = threadIdx.x; idx < W; idx += GFH_FTHREADS) {
      double sacc = 0.0;
      for (int b = b0 + sl; b < b1; b += 32 * 16) {           // 16 loads in flight, added in ascending order
        double v[16];
#pragma unroll
        for (int u = 0; u < 16; u++) v[u] = b + 32 * u < b1 ? GFH_LD_DEV(partial + (i64)(b + 32 * u) * pstride + idx) : 0.0;
#pragma unroll
        for (int u = 0; u < 16; u++) if (b + 32 * u < b1) sacc += v[u];
      }
      GFH_ST_DEV(sdst + idx, sacc);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)\n" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const bool last = __hip_atomic_fetch_add(GFH_GLOBAL(cnt), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)tl->n_slices - 1;
    if (last) GFH_ST_DEV(cnt, 0u);
    role = last;
  }
  __syncthreads();
  if (!role) return;
  const int nd = tl->nd;
  double* G = tl->G;                                          // written and read by this workgroup only
  for (int dd = 0; dd < nd; dd++) {
    const int nb = tl->ds_first_gb[dd + 1] - tl->ds_first_gb[dd];
    const double* ssrc = tl->slice + (i64)dd * 32 * pstride;
    for (int idx = threadIdx.x; idx < W; idx += GFH_FTHREADS) {
      double v[32];
#pragma unroll
      for (int k = 0; k < 32; k++) v[k] = k < nb ? GFH_LD_DEV(ssrc + (i64)k * pstride + idx) : 0.0;
      double t = v[0];
#pragma unroll
      for (int k = 1; k < 32; k++) t += v[k];
      G[(i64)dd * pstride + idx] = t;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)\n" ::: "memory");
  __syncthreads();
  assemble_and_post([&](int dd, int k) { return G[(i64)dd * pstride + k]; }, 0, nd);
}

#endif  // GFH_NA <= GFH_FUSED_MAX && !GFH_WSG

// chi2() (gadfit.F90:1015-1034): every parameter passive, value only.  Same partition and thread-to-point
// map as the fused kernel -- one workgroup of GFH_FW waves per gram block, wave wv of pass k takes the 64 slots
// at s0 + 64 wv + k * 64 GFH_FW -- every lane sums its own points pass by pass, then the wave tree, the waves
// in order, the workgroups by slices of 32 and the datasets in order: the order of additions of the fused
// kernel's sum r^2, so chi2() is bitwise the sum a sweep at the same parameters returns (GFH_FAST_DIV = 1: the
// reference's own value-only and active division forms differ by rounding, AD:814-913).  The parameter block
// is fixed per workgroup, so parameter-only subexpressions (reciprocals of widths ...) leave the pass loop.
// The next pass's inputs are loaded before the current pass's value is computed.
// tail_mode 0: workgroup sums only; 1: the last workgroup to arrive adds them up into out[0]; 2: and posts
// {sum, status} to the host mailbox.  The hand-off is the release / acquire form (MI355X_MICROARCH.md,
// inter-workgroup visibility: valid for any number of workgroups per CU).
#define GFH_CW (GFH_NA <= GFH_FUSED_MAX ? GFH_FW : 8)      // (beyond that there is no fused kernel to agree with)
#define GFH_CTHREADS (64 * GFH_CW)
extern "C" __global__ __launch_bounds__(GFH_CTHREADS) GFH_OCC
void gfh_k_chi2(const double* __restrict__ x, const double* __restrict__ y, const double* __restrict__ w,
                GFH_PARS_DECL, const i64* __restrict__ gb_start, const int* __restrict__ gb_slots,
                const int* __restrict__ gb_ds, double* __restrict__ res, double* partial, int* __restrict__ status,
                const double* __restrict__ aux, const i64 lda, const int* __restrict__ ds_first_gb, const int nd,
                double* out, double* host_out, unsigned long long* host_flag, unsigned* counter,
                const unsigned long long seq, const int tail_mode GFH_MESH_KPARAMS GFH_ORDER_KPARAMS GFH_WSG_KPARAMS) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  __shared__ double ws[GFH_CW];
  __shared__ double sl_sum[512];
  __shared__ double ds_sum[16];
  __shared__ int role;
  GFH_WSG_INIT
#if GFH_WSG
  // (workspaces in the global pool: the grid is capped at the pool's slots and a workgroup takes gram blocks blockIdx.x, + gridDim.x, ...;
  // every sum is defined on the partition into gram blocks, so which workgroup does a block changes no bit)
  for (int bb_ = blockIdx.x, nb_ = ds_first_gb[nd]; bb_ < nb_; bb_ += gridDim.x) {
  const int B = GFH_ORD(bb_);
#else
  {
  const int B = GFH_ORD(blockIdx.x);                                       // the gram block this workgroup works on
#endif
  const i64 s0 = gb_start[B];                                              // gb_slots: a positive multiple of GFH_CTHREADS slots
  const double* __restrict__ P = GFH_PARS_AT(gb_ds[B]);
  // Two passes per trip; the inputs of a trip are loaded during the trip before it, i.e. two passes (about a
  // microsecond of arithmetic) ahead: one pass ahead is less than the latency of an HBM load under load, and the
  // waves of a workgroup run in step, so they would all wait for it together.
  const int np = gb_slots[B] / GFH_CTHREADS;                              // passes of this workgroup (wave-uniform)
  const double* __restrict__ xb = x + s0 + threadIdx.x; const double* __restrict__ yb = y + s0 + threadIdx.x;
  const double* __restrict__ wb = w + s0 + threadIdx.x; const double* __restrict__ ab = aux + s0 + threadIdx.x;
  double* __restrict__ rb = res + s0 + threadIdx.x;
  double X0 = xb[0], Y0 = yb[0], W0 = wb[0];
  const i64 o1 = np > 1 ? GFH_CTHREADS : 0;
  double X1 = xb[o1], Y1 = yb[o1], W1 = wb[o1];
  double s = 0.0;
  for (int k = 0; k < np; k += 2) {
    const i64 oa = (i64)(k + 2 < np ? k + 2 : k) * GFH_CTHREADS, ob = (i64)(k + 3 < np ? k + 3 : k) * GFH_CTHREADS;
    const double Xa = xb[oa], Ya = yb[oa], Wa = wb[oa], Xb = xb[ob], Yb = yb[ob], Wb = wb[ob];
    const i64 oc = (i64)k * GFH_CTHREADS;
    const double r0 = (Y0 - gfh_point_value(X0, P, status, ab + oc, lda GFH_MESH_AT(s0 + threadIdx.x + oc) GFH_SLOT(s0 + threadIdx.x + oc))) * W0;   // gadfit.F90:1024-1026
#if GFH_STORE_RES
    __builtin_nontemporal_store(r0, rb + oc);
#endif
    s += r0 * r0;
    if (k + 1 < np) {
      const double r1 = (Y1 - gfh_point_value(X1, P, status, ab + oc + GFH_CTHREADS, lda GFH_MESH_AT(s0 + threadIdx.x + oc + GFH_CTHREADS) GFH_SLOT(s0 + threadIdx.x + oc + GFH_CTHREADS))) * W1;
#if GFH_STORE_RES
      __builtin_nontemporal_store(r1, rb + oc + GFH_CTHREADS);
#endif
      s += r1 * r1;
    }
    X0 = Xa; Y0 = Ya; W0 = Wa; X1 = Xb; Y1 = Yb; W1 = Wb;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if (lane == 0) ws[wv] = s;
  if (tail_mode) asm volatile("s_waitcnt vmcnt(0)\n" ::: "memory");       // this wave's status raise (if any) has landed
  __syncthreads();
  if (threadIdx.x == 0) {
    double tot = ws[0];
#pragma unroll
    for (int k = 1; k < GFH_CW; k++) tot += ws[k];
    partial[B] = tot;
  }
#if GFH_WSG
  __syncthreads();                                                        // (ws[] is written again in the next round)
#endif
  }
  if (threadIdx.x == 0) {
    if (tail_mode) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)\n" ::: "memory");
      const bool last = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
      if (last) {
        __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch (stream-ordered)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)\n" ::: "memory");
      }
      role = last;
    }
  }
  if (!tail_mode) return;
  __syncthreads();
  if (!role) return;
  // level 1: slice sl of dataset d adds its workgroups b0+sl, b0+sl+32, ... in ascending order; level 2: the 32
  // slice sums in slice order; level 3: the datasets in order (k_reduce_partials + k_gather_sum, and the fused tail)
  double total = 0.0;
  constexpr int DPR = GFH_CTHREADS / 32 < 16 ? GFH_CTHREADS / 32 : 16;     // datasets per round
  for (int d0 = 0; d0 < nd; d0 += DPR) {
    const int dl = threadIdx.x >> 5, sl = threadIdx.x & 31;
    if (dl < DPR && d0 + dl < nd) {
      const int b1 = ds_first_gb[d0 + dl + 1];
      double a = 0.0;
      for (int b = ds_first_gb[d0 + dl] + sl; b < b1; b += 32) a += partial[b];
      sl_sum[dl * 32 + sl] = a;
    }
    __syncthreads();
    if (threadIdx.x < DPR && d0 + (int)threadIdx.x < nd) {
      double t = sl_sum[threadIdx.x * 32];
#pragma unroll
      for (int k = 1; k < 32; k++) t += sl_sum[threadIdx.x * 32 + k];
      ds_sum[threadIdx.x] = t;
    }
    __syncthreads();
    if (threadIdx.x == 0) for (int k = 0; k < DPR && d0 + k < nd; k++) total += ds_sum[k];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out[0] = total;
    if (tail_mode == 1) out[1] = GFH_STATUS_SLOT(GFH_LD_DEV(status));      // the status slot of the cross-rank sum that follows
    if (tail_mode == 2) {
      GFH_ST_SYS(host_out, total);
      GFH_ST_SYS(host_out + 1, (double)GFH_LD_DEV(status));
      asm volatile("s_waitcnt vmcnt(0)\n" ::: "memory");
      __hip_atomic_store(GFH_GLOBAL(host_flag), seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// The tangent block of STEP 3 (delta1 per parameter).  Parameters and tangents are wave-uniform; with more than 16 of each
// they no longer fit the scalar registers next to each other, and what the compiler then does with loop-invariant scalars
// is to park them in VGPR lanes and fetch them back with v_readlane_b32 every pass (216 of them per pass at 32 parameters: a
// quarter of the loop's VALU issue).  Re-reading the tangents through the scalar cache inside the loop (constant address
// space, pointer made opaque so the loads stay in the loop) costs four s_load_dwordx16 per pass instead.
typedef const double __attribute__((address_space(4))) * gfh_cptr;
#if GFH_PARG
// (by value with the kernel arguments: addressed through the kernarg segment itself -- x, w, pars, dpars are the first
// four parameters of both STEP 3 kernels, so dpars sits at 16 + sizeof(gfh_parg); taking the address of the parameter
// object instead would make the compiler copy it to scratch.  tests/test_cpu_generated_source.py checks the offset
// against the code object's metadata.)
#define GFH_DPARS_CONST(ds) ((gfh_cptr)((const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr() + 16 + sizeof(gfh_parg)) + (GFH_PARG == GFH_NP ? 0 : (ds) * GFH_NP))
#else
#define GFH_DPARS_CONST(ds) ((gfh_cptr)(unsigned long long)(dpars + (i64)(ds) * GFH_NP))
#endif
#if GFH_NP > 16
#define GFH_TANGENTS(DPl, ds)                                                                  \
  double DPl[GFH_NP];                                                                          \
  { gfh_cptr c_ = GFH_DPARS_CONST(ds); asm volatile("" : "+s"(c_));                            \
    _Pragma("unroll") for (int k_ = 0; k_ < GFH_NP; k_++) DPl[k_] = c_[k_]; }
#else
#define GFH_TANGENTS(DPl, ds) const double* __restrict__ DPl = GFH_DPARS_AT(ds);
#endif

// omega kernel (STEP 3, forward mode): a workgroup owns a CONTIGUOUS chunk of tiles.  When the whole chunk
// lies in one dataset (always, unless a dataset boundary falls inside it) the parameter block is
// fixed for the loop, so everything that depends on parameters only leaves the per-point code.  The host sizes
// the grid to what is resident at once (context.cpp, resident_grid), so no workgroup waits for a second round.
extern "C" __global__ __launch_bounds__(GFH_BLOCK) GFH_OCC
void gfh_k_omega(const double* __restrict__ x, const double* __restrict__ w,
                 GFH_PARS_DECL, GFH_DPARS_DECL,
                 const int* __restrict__ tile_ds, const int n_tiles, double* __restrict__ omega, int* __restrict__ status,
                 const double* __restrict__ aux, const i64 lda GFH_MESH_KPARAMS GFH_ORDER_KPARAMS GFH_WSG_KPARAMS) {
  GFH_WSG_INIT
#if GFH_WSG
  // (workspaces in the global pool: the grid is capped at the pool's slots; a workgroup takes tiles blockIdx.x, + gridDim.x, ... of
  // the table sorted by cost, like gfh_k_sweep)
  for (int tb = blockIdx.x; tb < n_tiles; tb += gridDim.x) {
    const int t = GFH_ORD(tb);
    const double* __restrict__ P = GFH_PARS_AT(tile_ds[t]);
    const double* __restrict__ DP = GFH_DPARS_AT(tile_ds[t]);
    for (i64 i = (i64)t * GFH_TILE + threadIdx.x; i < (i64)(t + 1) * GFH_TILE; i += GFH_BLOCK)
      omega[i] = -gfh_point_dd(x[i], P, DP, status, aux + i, lda GFH_MESH_AT(i) GFH_SLOT(i)) * w[i];
  }
  return;
#endif
  // tiles split as evenly as integers allow: workgroup b takes [b n / G, (b + 1) n / G)
  const int bi = gridDim.x == (unsigned)n_tiles ? GFH_ORD(blockIdx.x) : (int)blockIdx.x;      // (one tile per workgroup: in the order of cost)
  const int t0 = (int)((i64)bi * n_tiles / gridDim.x), t1 = (int)((i64)(bi + 1) * n_tiles / gridDim.x);
  if (t0 >= t1) return;
  if (tile_ds[t0] == tile_ds[t1 - 1]) {
    const double* __restrict__ P = GFH_PARS_AT(tile_ds[t0]);
    const int ds0 = tile_ds[t0];                                 // delta1 scattered per dataset
    const i64 e = (i64)t1 * GFH_TILE;
    i64 i = (i64)t0 * GFH_TILE + threadIdx.x;
    double Xc = x[i], Wc = w[i];
    for (; i < e; i += GFH_BLOCK) {
      const i64 in = i + GFH_BLOCK < e ? i + GFH_BLOCK : i;       // next pass's inputs (the last pass re-reads its own)
      const double Xn = x[in], Wn = w[in];
      GFH_TANGENTS(DPl, ds0)
      omega[i] = -gfh_point_dd(Xc, P, DPl, status, aux + i, lda GFH_MESH_AT(i) GFH_SLOT(i)) * Wc;                  // gadfit.F90:722-723
      Xc = Xn; Wc = Wn;
    }
  } else {
    for (int t = t0; t < t1; t++) {
      const double* __restrict__ P = GFH_PARS_AT(tile_ds[t]);
      const double* __restrict__ DP = GFH_DPARS_AT(tile_ds[t]);
      for (i64 i = (i64)t * GFH_TILE + threadIdx.x; i < (i64)(t + 1) * GFH_TILE; i += GFH_BLOCK)
        omega[i] = -gfh_point_dd(x[i], P, DP, status, aux + i, lda GFH_MESH_AT(i) GFH_SLOT(i)) * w[i];
    }
  }
}
)";
  if (cfg.omega_jt && !cfg.finite_diff && !m.has_integrals() && cfg.loss == 0 && NA <= 64) {
    // One data point, forward mode AND reverse mode over ONE evaluation of the forward values: the second directional
    // derivative along DP and the gradient.  The forward values are the expressions of gfh_point_grad / gfh_point_dd (the same
    // emit_value_node), the reverse sweep is gfh_point_grad's, so G is bitwise the Jacobian row the sweep kernel stores.
    auto emit_dd_grad_fn = [&](const SubTape& t, const std::string& sfx, const std::string& slot) {
      s << "\nstatic __device__ __forceinline__ double gfh_point_dd_grad" << sfx << "(const double X, const double* __restrict__ P,\n"
           "                                                           const double* __restrict__ DP, double (&G)[GFH_NA], int* STATUS,\n"
           "                                                           " << A7 << slot << ") {\n";
      s << "  GFH_LANE_STASH\n";
      Gen g(m, t, cfg.fast_div); g.mode = 2; g.analyse(pa); g.emit_forward_all(); g.emit_reverse();
      s << g.o.str();
      for (int j = 0; j < NA; j++) s << "  G[" << j << "] = " << grad_expr(g, t, j) << ";\n";
      if (g.act[t.result]) s << "  return " << g.dd(t.result) << ";\n}\n";
      else s << "  return 0.0;\n}\n";
    };
    if (!multi) emit_dd_grad_fn(st, "", " GFH_SLOT_DECL");
    else {
      for (int v = 0; v < V; v++) emit_dd_grad_fn(m.eval(v), "_v" + std::to_string(v), "");
      s << "\nstatic __device__ __forceinline__ double gfh_point_dd_grad(const double X, const double* __restrict__ P,\n"
           "                                                           const double* __restrict__ DP, double (&G)[GFH_NA], int* STATUS,\n"
           "                                                           " << A7 << " GFH_SLOT_DECL) {\n"
           "  unsigned long long path; int ng;\n  switch (gfh_select(X, P, STATUS, AXP, LDA, path, ng)) {\n";
      for (int v = 0; v < V; v++) s << "    case " << v << ": return gfh_point_dd_grad_v" << v << "(X, P, DP, G, STATUS, AXP, LDA GFH_MESH_PASS);\n";
      s << "    default:\n#pragma unroll\n      for (int a = 0; a < GFH_NA; a++) G[a] = 0.0;\n      gfh_report_unseen(STATUS, SLOT, path, ng); return 0.0;\n  }\n}\n";
    }
  }
  if (cfg.omega_jt && !cfg.finite_diff && !m.has_integrals() && cfg.loss == 0 && NA <= 64) s << R"(
// STEP 3 in one pass (gadfit.F90:715-735): omega_i = -f''_delta1(x_i) w_i in forward mode AND
// J^T omega, with the Jacobian row of the point recomputed in registers (the reverse sweep of
// gfh_k_sweep over the forward values the forward-mode pass has just formed: the same expressions,
// so the same J_i) instead of re-read from HBM -- 8*p B/point
// of traffic less than J^T omega from the stored Jacobian, and STEP 3 no longer needs J in HBM
// at all.  One workgroup per gram block; the thread-to-point map, the order of additions, the wave
// and workgroup reductions are those of k_jtv (kernels.hip), so partial[b][a] is bitwise what
// k_jtv returns from the stored J.
extern "C" __global__ __launch_bounds__(256)
void gfh_k_omega_jt(const double* __restrict__ x, const double* __restrict__ w,
                    GFH_PARS_DECL, GFH_DPARS_DECL,
                    const i64* __restrict__ gb_start, const int* __restrict__ gb_slots, const int* __restrict__ gb_ds,
                    double* __restrict__ omega, double* __restrict__ partial, const int pstride, int* __restrict__ status,
                    const double* __restrict__ aux, const i64 lda) {
  const i64 s0 = gb_start[blockIdx.x], e = s0 + gb_slots[blockIdx.x];
  const double* __restrict__ P = GFH_PARS_AT(gb_ds[blockIdx.x]);
  const int ds0 = gb_ds[blockIdx.x];
  double acc[GFH_NA];
#pragma unroll
  for (int a = 0; a < GFH_NA; a++) acc[a] = 0.0;
  for (i64 i = s0 + threadIdx.x; i < e; i += 256) {
    const double X = x[i], W = w[i];                   // (no prefetch of the next pass here: at 32 parameters it would not fit 256 VGPRs)
    double G[GFH_NA];
    GFH_TANGENTS(DPl, ds0)
    const double om = -gfh_point_dd_grad(X, P, DPl, G, status, aux + i, lda GFH_MESH_NONE GFH_SLOT(i)) * W;    // gadfit.F90:722-723
    omega[i] = om;
#pragma unroll
    for (int a = 0; a < GFH_NA; a++) {
      const double j = G[a] * W;                                                      // the stored J entry, gadfit.F90:689-690
      acc[a] += j * om;                                                               // gadfit.F90:734
    }
  }
  __shared__ double ws[GFH_NA][4];
#pragma unroll
  for (int a = 0; a < GFH_NA; a++) {
    const double v = gfh_wave_sum(acc[a]);
    if ((threadIdx.x & 63) == 0) ws[a][threadIdx.x >> 6] = v;
  }
  __syncthreads();
  if (threadIdx.x < GFH_NA) partial[(i64)blockIdx.x * pstride + threadIdx.x] =
      ((ws[threadIdx.x][0] + ws[threadIdx.x][1]) + ws[threadIdx.x][2]) + ws[threadIdx.x][3];
}
)";
  *src = s.str();
  return true;
}

}  // namespace gfh

// The Gauss-Kronrod rule the kernels are generated with, for the Fortran layer's host-side integrate() (gadf_print and calls of
// eval() outside gadf_fit: numerical_integration.F90, host_integral): reference node order, even 1-based positions = Gauss nodes.
extern "C" __attribute__((visibility("default"))) int gfh_gk_rule(int points, double* roots, double* wg, double* wk) {
  const double *r = nullptr, *g = nullptr, *k = nullptr;
  if (!roots || !wg || !wk) return 1;
  switch (points) {
    case 15: r = gk15_roots; g = gk15_wg; k = gk15_wk; break;
    case 21: r = gk21_roots; g = gk21_wg; k = gk21_wk; break;
    case 31: r = gk31_roots; g = gk31_wg; k = gk31_wk; break;
    case 41: r = gk41_roots; g = gk41_wg; k = gk41_wk; break;
    case 51: r = gk51_roots; g = gk51_wg; k = gk51_wk; break;
    case 61: r = gk61_roots; g = gk61_wg; k = gk61_wk; break;
    default: return 1;
  }
  for (int i = 0; i < points; i++) { roots[i] = r[i]; wk[i] = k[i]; }
  for (int i = 0; i < points / 2; i++) wg[i] = g[i];
  return 0;
}

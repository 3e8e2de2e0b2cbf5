"""Random fitting functions written TWICE from one seed: as a Python callable over gadfit_amd.ad (traced into a tape for the CPU
oracle) and as Fortran source -- a module extending `fitfunc` plus a main program that fits through gadf_init / gadf_add_dataset /
gadf_set / gadf_fit.  The Fortran text goes through everything the Python tracer bypasses: the recorder of module ad, the
classification of literals (constants, reals affine in x, per-point columns, reals formed from a parameter's %val), the capture
over the data, the tape the layer builds.  tests/test_gpu_fortran_fuzz.py runs both and compares the fits.

Only straight-line bodies; every operation keeps its argument inside the function's domain (x in [0.3, 1.6], parameters in
[0.6, 1.8])."""
import numpy as np

from gadfit_amd import ad

NP_ = 5


def _lit(c):
    s = '%r_kp' % float(c)
    return '(%s)' % s if c < 0 else s


class E:
    """an expression: fn(p, x) over gadfit_amd.ad, its Fortran text, the parameters it reads; stmts: Fortran statements that must
    have run before the text is valid (the if-blocks of a branching body, which assign the temporaries the text names)"""
    def __init__(self, fn, f90, used, stmts=()):
        self.fn, self.f90, self.used, self.stmts = fn, f90, frozenset(used), list(stmts)


class Scope:
    """where an expression lives: eval() (parameters this%pars(:), the plain real abscissa x) or an integrand (its pars(:), the
    advar integration variable t -- whose .val an integrand must not read, nor a parameter's)"""
    def __init__(self, par='this%%pars(%d)', x='x', n_pars=NP_, val=True, pvx=False):
        self.par, self.x, self.n_pars, self.val, self.pvx = par, x, n_pars, val, pvx


EVAL = Scope()
# eval() of the kind 'pvx' (round 5): leaves that form reals from a parameter's %val TOGETHER with the abscissa as well
EVAL_PVX = Scope(pvx=True)
INTEGRAND = Scope(par='pars(%d)', x='t', n_pars=3, val=False)


def _leaf_pvx(rng, sc):
    """reals that eval() forms in plain arithmetic from the VALUE of a parameter and the abscissa -- another value at every point and
    (where the parameter is fitted) every pass; no derivative flows through them (AD:%val)"""
    j = int(rng.integers(0, sc.n_pars)); i = int(rng.integers(0, sc.n_pars))
    c = float(rng.uniform(0.5, 1.5))
    P, Pi, X = sc.par % (j + 1), sc.par % (i + 1), sc.x
    k = int(rng.integers(0, 5))
    if k == 0:
        return E(lambda p, x: ad.cos(ad.value(p[j]) * x * c) * p[i], '(cos(%s%%val*%s*%s)*%s)' % (P, X, _lit(c), Pi), [i])
    if k == 1:          # (affine in x at any one set of parameters; zero at x = 0 whatever the parameter)
        return E(lambda p, x: (ad.value(p[j]) * x) * p[i], '((%s%%val*%s)*%s)' % (P, X, Pi), [i])
    if k == 2:
        return E(lambda p, x: ad.exp(-(ad.value(p[j]) * x * c)) + p[i], '(exp(-(%s%%val*%s*%s)) + %s)' % (P, X, _lit(c), Pi), [i])
    if k == 3:          # (of the parameter alone: a pseudo-parameter under AD, a column under finite differences)
        return E(lambda p, x: ad.exp(-(ad.value(p[j]) * c)) * p[i], '(exp(-(%s%%val*%s))*%s)' % (P, _lit(c), Pi), [i])
    return E(lambda p, x: ad.sqrt(1.0 + ad.value(p[j]) * x) - p[i], '(sqrt(1.0_kp + %s%%val*%s) - %s)' % (P, X, Pi), [i])


def _leaf(rng, sc):
    if sc.pvx and rng.random() < 0.35:
        return _leaf_pvx(rng, sc)
    k = int(rng.integers(0, 7 if sc.val else 6))
    j = int(rng.integers(0, sc.n_pars))
    c = float(rng.uniform(0.5, 1.5))
    P = sc.par % (j + 1)
    X = sc.x
    if k == 0:
        return E(lambda p, x: p[j], P, [j])
    if k == 1:                                        # real arithmetic on x: affine (a literal that follows x)
        return E(lambda p, x: x * c, '(%s*%s)' % (X, _lit(c)), [])
    if k == 2:
        return E(lambda p, x: p[j] * x, '(%s*%s)' % (P, X), [j])
    if k == 3:
        c2 = float(rng.uniform(-2.0, 2.0))
        return E(lambda p, x: c2 + p[j], '(%s + %s)' % (_lit(c2), P), [j])
    if k == 4:                                        # a real function of x outside the affine ones: a per-point column in Fortran
        which = int(rng.integers(0, 4))
        py = [lambda p, x: ad.sin(x * c), lambda p, x: ad.exp(-(x * c)), lambda p, x: x ** 2 * c, lambda p, x: ad.sqrt(1.0 + x * c)][which]
        f = ['sin(%s*%s)', 'exp(-(%s*%s))', '(%s**2*%s)', 'sqrt(1.0_kp + %s*%s)'][which] % (X, _lit(c))
        return E(py, f, [])
    if k == 5:                                        # affine with an offset
        c2 = float(rng.uniform(-1.0, 1.0))
        return E(lambda p, x: x * c + c2, '(%s*%s + %s)' % (X, _lit(c), _lit(c2)), [])
    # a real formed from the VALUE of a parameter (no derivative through it, AD:%val), meeting an advar at once
    i = int(rng.integers(0, sc.n_pars))
    return E(lambda p, x: ad.sin(ad.value(p[j])) * p[i], '(sin(%s%%val)*%s)' % (P, sc.par % (i + 1)), [i])      # (no derivative reaches parameter j through its value)


def rand_expr(rng, depth, sc=EVAL):
    if depth <= 0 or rng.random() < 0.15:
        return _leaf(rng, sc)
    a = rand_expr(rng, depth - 1, sc)
    op = int(rng.integers(0, 27))
    if op < 8:
        b = rand_expr(rng, depth - 1, sc)
        c = float(rng.uniform(0.3, 2.5))
        A, B, C = a.f90, b.f90, _lit(c)
        return [E(lambda p, x: a.fn(p, x) + b.fn(p, x), '(%s + %s)' % (A, B), a.used | b.used),
                E(lambda p, x: a.fn(p, x) - b.fn(p, x), '(%s - %s)' % (A, B), a.used | b.used),
                E(lambda p, x: a.fn(p, x) * b.fn(p, x), '(%s*%s)' % (A, B), a.used | b.used),
                E(lambda p, x: a.fn(p, x) / (1.5 + abs(b.fn(p, x))), '(%s/(1.5_kp + abs(%s)))' % (A, B), a.used | b.used),
                E(lambda p, x: c - a.fn(p, x), '(%s - %s)' % (C, A), a.used),
                E(lambda p, x: a.fn(p, x) / c, '(%s/%s)' % (A, C), a.used),
                E(lambda p, x: c / (1.3 + abs(a.fn(p, x))), '(%s/(1.3_kp + abs(%s)))' % (C, A), a.used),
                E(lambda p, x: c * a.fn(p, x), '(%s*%s)' % (C, A), a.used)][op]
    if op == 8:
        b = rand_expr(rng, depth - 1, sc)
        return E(lambda p, x: (1.2 + abs(a.fn(p, x))) ** ad.tanh(b.fn(p, x)), '((1.2_kp + abs(%s))**tanh(%s))' % (a.f90, b.f90), a.used | b.used)
    if op == 9:
        c = float(rng.uniform(-1.5, 2.5))
        return E(lambda p, x: (1.2 + abs(a.fn(p, x))) ** c, '((1.2_kp + abs(%s))**%s)' % (a.f90, _lit(c)), a.used)
    if op == 10:
        c = float(rng.uniform(1.1, 3.0))
        return E(lambda p, x: c ** ad.tanh(a.fn(p, x)), '(%s**tanh(%s))' % (_lit(c), a.f90), a.used)
    if op == 11:
        n = int(rng.integers(-2, 4))
        return E(lambda p, x: (0.7 + abs(a.fn(p, x))) ** n, '((0.7_kp + abs(%s))**(%d))' % (a.f90, n), a.used)
    A = a.f90
    table = [(lambda p, x: ad.exp(ad.tanh(a.fn(p, x))), 'exp(tanh(%s))' % A),
             (lambda p, x: ad.sqrt(0.5 + abs(a.fn(p, x))), 'sqrt(0.5_kp + abs(%s))' % A),
             (lambda p, x: ad.log(1.1 + abs(a.fn(p, x))), 'log(1.1_kp + abs(%s))' % A),
             (lambda p, x: ad.sin(a.fn(p, x)), 'sin(%s)' % A),
             (lambda p, x: ad.cos(a.fn(p, x)), 'cos(%s)' % A),
             (lambda p, x: ad.tan(0.5 * ad.tanh(a.fn(p, x))), 'tan(0.5_kp*tanh(%s))' % A),
             (lambda p, x: ad.asin(0.9 * ad.tanh(a.fn(p, x))), 'asin(0.9_kp*tanh(%s))' % A),
             (lambda p, x: ad.acos(0.9 * ad.tanh(a.fn(p, x))), 'acos(0.9_kp*tanh(%s))' % A),
             (lambda p, x: ad.atan(a.fn(p, x)), 'atan(%s)' % A),
             (lambda p, x: ad.sinh(ad.tanh(a.fn(p, x))), 'sinh(tanh(%s))' % A),
             (lambda p, x: ad.cosh(ad.tanh(a.fn(p, x))), 'cosh(tanh(%s))' % A),
             (lambda p, x: ad.tanh(a.fn(p, x)), 'tanh(%s)' % A),
             (lambda p, x: ad.asinh(a.fn(p, x)), 'asinh(%s)' % A),
             (lambda p, x: ad.acosh(1.5 + abs(a.fn(p, x))), 'acosh(1.5_kp + abs(%s))' % A),
             (lambda p, x: ad.atanh(0.9 * ad.tanh(a.fn(p, x))), 'atanh(0.9_kp*tanh(%s))' % A),
             (lambda p, x: ad.erf(a.fn(p, x)), 'erf(%s)' % A),
             (lambda p, x: abs(a.fn(p, x)), 'abs(%s)' % A),
             (lambda p, x: -a.fn(p, x), '(-%s)' % A)]
    py, f = table[(op - 12) % len(table)]
    return E(py, f, a.used)


def make_case(seed, depth=4, pvx=False):
    """-> (E root, active parameter indices, start values, truth values).  The root always reads parameter 0 (so that eval()
    returns an advar and at least one parameter can be fitted).  pvx: the kind whose leaves also form reals from %val and x."""
    rng = np.random.default_rng((51000 if pvx else 31000) + seed)
    body = rand_expr(rng, depth, EVAL_PVX if pvx else EVAL)
    if pvx:             # (at least one such real in every case of the kind)
        lf = _leaf_pvx(rng, EVAL_PVX)
        body = E(lambda p, x, a=body, b=lf: a.fn(p, x) + b.fn(p, x), '(%s + %s)' % (body.f90, lf.f90), body.used | lf.used)
    c = float(rng.uniform(0.5, 1.5))
    root = E(lambda p, x: body.fn(p, x) + c * p[0], '(%s + %s*this%%pars(1))' % (body.f90, _lit(c)), body.used | {0})
    truth = rng.uniform(0.6, 1.8, size=NP_)
    used = sorted(root.used)
    mask = rng.random(len(used)) < 0.7
    active = [u for u, m in zip(used, mask) if m] or [0]
    start = truth.copy()
    for k in active:
        start[k] = truth[k] * (1.0 + 0.04 * rng.uniform(-1, 1))
    return root, active, start, truth


def _rand_cond(rng):
    """a random comparison (the reference's 14 specifics come down to advar-advar, advar-real, real-advar; and the plain real one
    that operator overloading never sees): -> (fn(p, x) -> bool, Fortran text, parameters read)"""
    k = int(rng.integers(0, 6))
    c = float(rng.uniform(0.6, 1.3))
    i, j = int(rng.integers(0, NP_)), int(rng.integers(0, NP_))
    if k == 0:
        return (lambda p, x: x < p[j] * c), '(x < this%%pars(%d)*%s)' % (j + 1, _lit(c)), {j}
    if k == 1:
        return (lambda p, x: p[i] * x > p[j]), '(this%%pars(%d)*x > this%%pars(%d))' % (i + 1, j + 1), {i, j}
    if k == 2:
        a = rand_expr(rng, 1)
        return (lambda p, x: a.fn(p, x) > c), '(%s > %s)' % (a.f90, _lit(c)), set(a.used)
    if k == 3:
        a = rand_expr(rng, 1)
        c9 = c * 0.9
        return (lambda p, x: c9 < a.fn(p, x)), '(%s < %s)' % (_lit(c9), a.f90), set(a.used)
    if k == 4:
        a = rand_expr(rng, 1); b = rand_expr(rng, 1)
        return (lambda p, x: a.fn(p, x) < b.fn(p, x)), '(%s < %s)' % (a.f90, b.f90), set(a.used | b.used)
    return (lambda p, x: x < c), '(x < %s)' % _lit(c), set()


def rand_branching(rng, depth, counter, sc=EVAL):
    """a random body that BRANCHES: if-blocks nested `depth` deep, each side its own random expression"""
    if depth <= 0:
        return rand_expr(rng, 2, sc)
    cond, cond_f, cond_used = _rand_cond(rng)
    a = rand_branching(rng, depth - 1, counter, sc)
    extra = rand_expr(rng, 1, sc) if rng.random() < 0.5 else None
    b = rand_branching(rng, depth - 1, counter, sc)
    c = float(rng.uniform(0.5, 1.5))
    counter[0] += 1
    t = 't%d' % counter[0]
    then_f = a.f90 if extra is None else '(%s + %s)' % (a.f90, extra.f90)
    stmts = ['if %s then' % cond_f] + ['  ' + ln for ln in a.stmts] + ['  %s = %s' % (t, then_f), 'else'] + \
            ['  ' + ln for ln in b.stmts] + ['  %s = (%s*%s)' % (t, b.f90, _lit(c)), 'end if']

    def fn(p, x):
        if cond(p, x):
            return a.fn(p, x) if extra is None else a.fn(p, x) + extra.fn(p, x)
        return b.fn(p, x) * c
    return E(fn, t, a.used | b.used | (extra.used if extra is not None else frozenset()), stmts)


def make_branching_case(seed, depth=2):
    """as make_case, the body branching `depth` deep.  (The conditions are not counted as uses: a parameter that only decides a
    branch has no derivative.)"""
    rng = np.random.default_rng(41000 + seed)
    counter = [0]
    body = rand_branching(rng, depth, counter)
    c = float(rng.uniform(0.5, 1.5))
    root = E(lambda p, x: body.fn(p, x) + c * p[0], '(%s + %s*this%%pars(1))' % (body.f90, _lit(c)), body.used | {0}, body.stmts)
    root.n_temps = counter[0]
    truth = rng.uniform(0.6, 1.8, size=NP_)
    # (only parameter 1 is certain to carry a derivative on EVERY path: the others are fitted where the data reach them)
    used = sorted(root.used)
    mask = rng.random(len(used)) < 0.5
    active = sorted(set([u for u, m in zip(used, mask) if m] + [0]))
    start = truth.copy()
    for k in active:
        start[k] = truth[k] * (1.0 + 0.02 * rng.uniform(-1, 1))
    return root, active, start, truth


RULES = [15, 21, 31, 41, 51, 61]


def make_integral_case(seed, branching=False, nested=False, ipv=False):
    """eval() = an integral of a random integrand (branching = True: one that takes one of two random expressions, by a comparison of
    its integration variable with a parameter, decided anew at every abscissa of the quadrature): envelope exp(-q1 t**2) (integrable on every range) times 1 + 0.3 tanh(random
    expression in (t, q)); one of six kinds of bounds (finite with the upper one following x; ACTIVE bounds; (a, inf); (-inf, b);
    (-inf, inf); an active lower bound with +inf), a random Gauss-Kronrod rule.  -> (root, active, start, truth, integrand E, rule)"""
    rng = np.random.default_rng((53000 if nested else 52000 if branching else 54000 if ipv else 51000) + seed)
    kind = int(rng.integers(0, 6))
    rule = RULES[int(rng.integers(0, 6))]
    body = rand_expr(rng, 2, INTEGRAND)
    if nested:
        kind = kind % 2          # (finite outer ranges: the inner range follows the outer variable)
        # the integrand holds an integral of its own (the reference's two workspaces, NI:70): int_0^{c t} exp(-q2 s) (1 + 0.3 tanh(e(s, q))) ds
        inner_body = rand_expr(rng, 1, INTEGRAND)
        cn = float(rng.uniform(0.5, 1.2))
        first = body

        def inner(s_, q):
            return ad.exp(-(q[1] * s_)) * (1.0 + 0.3 * ad.tanh(inner_body.fn(q, s_)))

        def integrand(t, q):
            return ad.exp(-(q[0] * t * t)) * (1.0 + 0.3 * ad.tanh(first.fn(q, t))) * (1.0 + 0.2 * ad.integrate(inner, q, 0.0, t * cn))
        integrand_f90 = ['y = (exp(-(pars(1)*t*t))*(1.0_kp + 0.3_kp*tanh(%s))*(1.0_kp + 0.2_kp*integrate(fuzz_inner, pars, 0.0_kp, t*%s)))' % (first.f90, _lit(cn))]
        inner_f90 = '(exp(-(pars(2)*t))*(1.0_kp + 0.3_kp*tanh(%s)))' % inner_body.f90
        body = E(None, '', set(first.used) | set(inner_body.used) | {1})
        body.inner_f90 = inner_f90
    elif branching:
        other = rand_expr(rng, 2, INTEGRAND)
        cb = float(rng.uniform(0.6, 1.4))
        first = body

        def integrand(t, q):
            b = first.fn(q, t) if t * cb < q[1] else other.fn(q, t)
            return ad.exp(-(q[0] * t * t)) * (1.0 + 0.3 * ad.tanh(b))
        integrand_f90 = ['if (t*%s < pars(2)) then' % _lit(cb), '  b = %s' % first.f90, 'else', '  b = %s' % other.f90, 'end if',
                         'y = (exp(-(pars(1)*t*t))*(1.0_kp + 0.3_kp*tanh(b)))']
        body = E(None, '', set(first.used) | set(other.used))       # (the comparison itself carries no derivative: AD:315-395)
    elif ipv:
        # (round 5) the integrand forms a real from the %val of one of ITS parameters in plain arithmetic: one more, passive entry of
        # its pars(:) on the device, bound at the call site to a pseudo-parameter refreshed before every pass
        jv = int(rng.integers(0, 3)); cv = float(rng.uniform(0.5, 1.5))

        def integrand(t, q):
            return ad.exp(-(q[0] * t * t)) * (1.0 + 0.3 * ad.tanh(body.fn(q, t))) * (1.0 + 0.1 * ad.cos(ad.value(q[jv]) * cv))
        integrand_f90 = '(exp(-(pars(1)*t*t))*(1.0_kp + 0.3_kp*tanh(%s))*(1.0_kp + 0.1_kp*cos(pars(%d)%%val*%s)))' % (body.f90, jv + 1, _lit(cv))
    else:
        def integrand(t, q):
            return ad.exp(-(q[0] * t * t)) * (1.0 + 0.3 * ad.tanh(body.fn(q, t)))
        integrand_f90 = '(exp(-(pars(1)*t*t))*(1.0_kp + 0.3_kp*tanh(%s)))' % body.f90

    def q_of(p):
        return [p[0], p[1], p[2]]
    I = 'integrate(fuzz_integrand, q, %s, %s)'
    forms = [(lambda p, x: ad.integrate(integrand, q_of(p), 0.1, x) + p[3], '(' + I % ('0.1_kp', 'x') + ' + this%pars(4))', {3}),
             (lambda p, x: ad.integrate(integrand, q_of(p), p[3] * 0.2, x * p[4]), I % ('this%pars(4)*0.2_kp', 'x*this%pars(5)'), {3, 4}),
             (lambda p, x: ad.integrate(integrand, q_of(p), x * 0.5, ad.INFINITY), I % ('x*0.5_kp', 'INFINITY'), set()),
             (lambda p, x: ad.integrate(integrand, q_of(p), -ad.INFINITY, x - p[3]), I % ('-INFINITY', 'x - this%pars(4)'), {3}),
             (lambda p, x: ad.integrate(integrand, q_of(p), -ad.INFINITY, ad.INFINITY) * x, '(' + I % ('-INFINITY', 'INFINITY') + '*x)', set()),
             (lambda p, x: p[4] * ad.integrate(integrand, q_of(p), p[3] * 0.1, ad.INFINITY), '(this%pars(5)*' + I % ('this%pars(4)*0.1_kp', 'INFINITY') + ')', {3, 4})]
    fn, f90, more = forms[kind]
    root = E(fn, f90, set(body.used) | {0} | more, ['q = this%pars(1:3)'])
    root.decls = ['type(advar) :: q(3)']
    root.inner_f90 = getattr(body, 'inner_f90', None)
    truth = rng.uniform(0.7, 1.6, size=NP_)
    used = sorted(root.used)
    mask = rng.random(len(used)) < 0.7
    active = [u for u, m in zip(used, mask) if m] or [0]
    start = truth.copy()
    for k in active:
        start[k] = truth[k] * (1.0 + 0.03 * rng.uniform(-1, 1))
    return root, active, start, truth, integrand_f90, rule


def wrap(text, width=120):
    """Fortran free-form continuation lines for a long expression"""
    out = []
    while len(text) > width:
        cut = max(text.rfind('(', 12, width), text.rfind(' ', 12, width))      # (before a parenthesis or at a blank: never inside a token)
        assert cut > 12, text
        out.append(text[:cut] + ' &')
        text = '         & ' + text[cut:]
    out.append(text)
    return '\n'.join(out)


def fortran_source(root, active, start, lam, max_iter, integrand=None, init_args='', use_ad=True):
    nt = getattr(root, 'n_temps', 0)
    decls = ('    type(advar) :: ' + ', '.join('t%d' % (k + 1) for k in range(nt))) if nt else ''
    decls = '\n'.join([decls] + ['    ' + d for d in getattr(root, 'decls', [])])
    extra = '' if integrand is None else '''  type(advar) function fuzz_integrand(t, pars) result(y)
    type(advar), intent(in) :: t
    type(advar), intent(in out) :: pars(:)
    type(advar) :: b
%s
  end function fuzz_integrand
''' % (wrap('    y = ' + integrand) if isinstance(integrand, str) else '\n'.join(wrap('    ' + ln) for ln in integrand))
    if getattr(root, 'inner_f90', None):
        extra += '''  type(advar) function fuzz_inner(t, pars) result(y)
    type(advar), intent(in) :: t
    type(advar), intent(in out) :: pars(:)
%s
  end function fuzz_inner
''' % wrap('    y = ' + root.inner_f90)
    body = '\n'.join(wrap('    ' + ln) for ln in root.stmts + ['y = ' + root.f90])
    sets = '\n'.join("  call gadf_set(%d, %s, %s)" % (k + 1, '%r_kp' % float(start[k]), '.true.' if k in active else '.false.') for k in range(NP_))
    return '''! generated by tests/fortran_fuzz.py
module fuzz_model
  use ad
  use fitfunction
  use gadf_constants
  use numerical_integration
  implicit none
  type, extends(fitfunc) :: fuzz_t
   contains
     procedure :: init => fuzz_init
     procedure :: eval => fuzz_eval
  end type fuzz_t
contains
  subroutine fuzz_init(this)
    class(fuzz_t), intent(out) :: this
    allocate(this%%pars(%d))
  end subroutine fuzz_init
  type(advar) function fuzz_eval(this, x) result(y)
    class(fuzz_t), intent(in) :: this
    real(kp), intent(in) :: x
%s
%s
  end function fuzz_eval
%send module fuzz_model

program fuzz
  use fuzz_model
  use gadfit
  implicit none
  type(fuzz_t) :: f
  character(len=512) :: path
  integer :: k
  call get_command_argument(1, path)
  call gadf_init(f%s)
  call gadf_add_dataset(trim(path))
%s
  call gadf_set_errors(NONE)
  call gadf_set_verbosity(output='/dev/null')
  call gadf_fit(%s, max_iter=%d%s)
  do k = 1, %d
     write(*, '(a, i0, 1x, es25.17)') 'par ', k, fitfuncs(1)%%pars(k)%%val
  end do
  write(*, '(a, es25.17)') 'chi2 ', gadf_chi2
  write(*, '(a, i0)') 'iterations ', gadf_iterations
  call gadf_close()
  print '(a)', 'DONE'
end program fuzz
''' % (NP_, decls, body, extra, init_args, sets, repr(float(lam)), max_iter, '' if use_ad else ', use_ad=.false.', NP_)


ERROR_MODES = ['NONE', 'SQRT_Y', 'PROPTO_Y', 'INVERSE_Y', 'USER']


def make_layout_case(seed, branching=False, pvx=False):
    """a straight-line body (branching = True: one that branches two deep) fitted to 1-3 datasets at once: every parameter global or local (a local one with its own start value
    per dataset), one of the five kinds of data errors, geodesic acceleration on or off, a random lambda.
    -> dict(root, active, is_global, start [nd][NP], truth [nd][NP], nd, mode, accth, lam, max_iter)"""
    rng = np.random.default_rng((71000 if branching else 61000) + (20000 if pvx else 0) + seed)
    counter = [0]
    sc = EVAL_PVX if pvx else EVAL          # (pvx: leaves that form reals from %val and x as well -- with LOCAL parameters one column per dataset's values)
    body = rand_branching(rng, 2, counter, sc) if branching else rand_expr(rng, 3, sc)
    if pvx:
        lf = _leaf_pvx(rng, sc)
        body = E(lambda p, x, a=body, b=lf: a.fn(p, x) + b.fn(p, x), '(%s + %s)' % (body.f90, lf.f90), body.used | lf.used, body.stmts)
    c = float(rng.uniform(0.5, 1.5))
    root = E(lambda p, x: body.fn(p, x) + c * p[0], '(%s + %s*this%%pars(1))' % (body.f90, _lit(c)), body.used | {0}, body.stmts)
    root.n_temps = counter[0]
    nd = int(rng.integers(1, 4))
    is_global = [int(v) for v in rng.integers(0, 2, size=NP_)]
    base = rng.uniform(0.6, 1.8, size=NP_)
    truth = np.array([[base[k] if is_global[k] else base[k] * (1.0 + 0.1 * rng.uniform(-1, 1)) for k in range(NP_)] for _ in range(nd)])
    used = sorted(root.used)
    mask = rng.random(len(used)) < 0.7
    active = [u for u, m in zip(used, mask) if m] or [0]
    if branching:
        active = sorted(set(active + [0]))         # (the one parameter that carries a derivative on every path)
    start = truth.copy()
    for k in active:
        if is_global[k]:
            start[:, k] = truth[0, k] * (1.0 + 0.03 * rng.uniform(-1, 1))
        else:
            start[:, k] = truth[:, k] * (1.0 + 0.03 * rng.uniform(-1, 1, size=nd))
    case = dict(root=root, active=active, is_global=is_global, start=start, truth=truth, nd=nd, mode=ERROR_MODES[int(rng.integers(0, 5))],
                accth=(0.9 if rng.random() < 0.5 else None), lam=float(rng.choice([0.1, 1.0, 10.0])), max_iter=int(rng.integers(2, 4)))
    # (drawn after everything else, so that the cases of the earlier seeds stay what they were)
    menu = [dict(), dict(lam_incs=4, nielsen=True), dict(umnigh=True, uphill=1), dict(rel_error=1e-5, cos_phi=1e-3, grad_chi2=1e-3, max_iter=8),
            dict(damp_max=False, chi2_rel=1e-9), dict(lam_up=5.0, lam_down=3.0), dict(use_ad=False), dict(chi2_abs=1e-3, max_iter=6)]
    case['more'] = menu[int(rng.integers(0, len(menu)))]
    if '%val' in root.f90 + ' '.join(root.stmts) and not case['more'].get('use_ad', True):
        if not pvx:
            case['more'] = dict()                 # (refused up to round 4; the cases of the earlier seeds stay what they were)
        elif case['accth'] is not None:
            case['accth'] = None                  # (use_ad=.false. over such reals with geodesic acceleration: refused, refused_literals.F90 'fdacc')
    # a second gadf_fit after the program has changed its mind about one parameter: fitted <-> fixed, the value moved by 1 %
    case['refit'] = None
    if rng.random() < 0.5:
        k = int(rng.choice(used))
        case['refit'] = dict(par=k, active=(k not in active) or len(active) == 1, scale=1.0 + 0.01 * float(rng.uniform(-1, 1)))
    case['images'] = int(rng.choice([1, 1, 2, 3]))           # (members of a single-process device group, sharing the card)
    return case


def fortran_source_layout(case):
    root, nd, start, active, is_global = case['root'], case['nd'], case['start'], case['active'], case['is_global']
    sets = []
    for k in range(NP_):
        act = '.true.' if k in active else '.false.'
        if is_global[k]:
            sets.append('  call gadf_set(%d, %s, %s)' % (k + 1, '%r_kp' % float(start[0, k]), act))
        else:
            for d in range(nd):
                sets.append('  call gadf_set(%d, %d, %s, %s)' % (d + 1, k + 1, '%r_kp' % float(start[d, k]), act))
    nt = getattr(root, 'n_temps', 0)
    decls = ('    type(advar) :: ' + ', '.join('t%d' % (k + 1) for k in range(nt))) if nt else ''
    body = '\n'.join(wrap('    ' + ln) for ln in root.stmts + ['y = ' + root.f90])
    kw = dict(max_iter=case['max_iter'])
    if case['accth'] is not None:
        kw['accth'] = case['accth']
    kw.update(case.get('more', {}))

    def f90(v):
        return ('.true.' if v else '.false.') if isinstance(v, bool) else ('%d' % v if isinstance(v, int) else repr(float(v)))
    fit_args = repr(float(case['lam'])) + ''.join(', %s=%s' % (k, f90(v)) for k, v in kw.items())
    refit = ''
    if case.get('refit'):
        rf = case['refit']
        k = rf['par']
        act = '.true.' if rf['active'] else '.false.'
        if is_global[k]:
            refit = '  call gadf_set(%d, fitfuncs(1)%%pars(%d)%%val*%s, %s)\n' % (k + 1, k + 1, _lit(rf['scale']), act)
        else:
            refit = ''.join('  call gadf_set(%d, %d, fitfuncs(%d)%%pars(%d)%%val*%s, %s)\n' % (d + 1, k + 1, d + 1, k + 1, _lit(rf['scale']), act) for d in range(nd))
        refit += '  call gadf_fit(%s, max_iter=2)\n' % repr(float(case['lam']))
    return '''! generated by tests/fortran_fuzz.py (layout case)
module fuzz_model
  use ad
  use fitfunction
  use gadf_constants
  implicit none
  type, extends(fitfunc) :: fuzz_t
   contains
     procedure :: init => fuzz_init
     procedure :: eval => fuzz_eval
  end type fuzz_t
contains
  subroutine fuzz_init(this)
    class(fuzz_t), intent(out) :: this
    allocate(this%%pars(%d))
  end subroutine fuzz_init
  type(advar) function fuzz_eval(this, x) result(y)
    class(fuzz_t), intent(in) :: this
    real(kp), intent(in) :: x
%s
%s
  end function fuzz_eval
end module fuzz_model

program fuzz
  use fuzz_model
  use gadfit
  implicit none
  type(fuzz_t) :: f
  character(len=512) :: path
  integer :: k, d
  call gadf_init(f, %d)
  do d = 1, %d
     call get_command_argument(d, path)
     call gadf_add_dataset(trim(path))
  end do
%s
  call gadf_set_errors(%s)
  if (command_argument_count() <= %d) call gadf_set_verbosity(output='/dev/null')      ! (one argument more: the iteration log)
  call gadf_fit(%s)
  write(*, '(a, i0)') 'iterations1 ', gadf_iterations
%s  do d = 1, %d
     do k = 1, %d
        write(*, '(a, i0, 1x, i0, 1x, es25.17)') 'par ', d, k, fitfuncs(d)%%pars(k)%%val
     end do
  end do
  write(*, '(a, es25.17)') 'chi2 ', gadf_chi2
  write(*, '(a, i0)') 'iterations ', gadf_iterations
  call gadf_close()
  print '(a)', 'DONE'
end program fuzz
''' % (NP_, decls, body, nd, nd, '\n'.join(sets), case['mode'], nd, fit_args, refit, nd, NP_)


def fortran_source_two_sessions(case_a, case_b):
    """ONE program that runs two layout cases one after the other -- gadf_init ... gadf_close, then again with another model, other
    datasets, other options: whatever the layer keeps between sessions (paths, classes, reserved slots, flags) must not leak.
    The command line holds the files of the first case, then those of the second."""
    def parts(src, tag, offset, nd):
        mod, prog = src.split('\nprogram fuzz\n', 1)
        mod += '\n'
        mod = mod.replace('fuzz_model', 'fuzz_model_' + tag)
        body = prog.split('end program fuzz')[0]
        body = body.replace('use fuzz_model', 'use fuzz_model_' + tag)
        body = body.replace('call get_command_argument(d, path)', 'call get_command_argument(d + %d, path)' % offset)
        body = body.replace('if (command_argument_count() <= %d) call gadf_set_verbosity' % nd, 'call gadf_set_verbosity')
        body = body.replace("  print '(a)', 'DONE'\n", "  print '(a)', 'SESSION %s DONE'\n" % tag)
        return mod, '  subroutine run_%s()\n' % tag + body + '  end subroutine run_%s\n' % tag
    ma, ra = parts(fortran_source_layout(case_a), 'a', 0, case_a['nd'])
    mb, rb = parts(fortran_source_layout(case_b), 'b', case_a['nd'], case_b['nd'])
    main = "program fuzz\n  implicit none\n  call run_a()\n  call run_b()\n  print '(a)', 'DONE'\ncontains\n"
    return ma + mb + main + ra + rb + 'end program fuzz\n'

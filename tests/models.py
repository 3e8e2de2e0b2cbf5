"""Synthetic workloads of SURVEY.md §8(d) / BASELINE.json configs, shared by tests and bench.py.
Deterministic: SplitMix64 -> Box-Muller, seed 20240601, identical wherever it runs."""
import numpy as np

from gadfit_amd.ad import exp

SEED = 20240601


def splitmix64(n, seed):
    """n uint64 from counter-based SplitMix64 (vectorised)."""
    with np.errstate(over='ignore'):
        z = (np.arange(1, n + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)) + np.uint64(seed)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def normal(n, seed):
    u1 = (splitmix64(n, seed) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    u2 = (splitmix64(n, seed + 7919) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    return np.sqrt(-2.0 * np.log(np.maximum(u1, 1e-300))) * np.cos(2.0 * np.pi * u2)


# ---- cfg 2: 4-exponential decay, 8 active parameters ------------------------------------
def model_exp4(p, x):
    y = p[0] * exp(-(x / p[1]))
    for k in range(1, 4):
        y = y + p[2 * k] * exp(-(x / p[2 * k + 1]))
    return y


EXP4_TRUTH = np.array([5.0, 0.5, 3.0, 2.0, 2.0, 8.0, 1.0, 30.0])


def exp4_numpy(p, x):
    return sum(p[2 * k] * np.exp(-x / p[2 * k + 1]) for k in range(4))


# ---- cfg 5 / headline: 8 Gaussians with a linear skew, 32 active parameters ---------------
def model_gauss8(p, x):
    y = None
    for k in range(8):
        A, mu, w, s = p[4 * k], p[4 * k + 1], p[4 * k + 2], p[4 * k + 3]
        d = x - mu
        t = A * exp(-((d / w) ** 2)) * (1 + s * d)
        y = t if y is None else y + t
    return y


def gauss8_truth():
    p = np.zeros(32)
    for k in range(8):
        p[4 * k] = 1.0 + 4.0 * k / 7.0          # A = 1..5
        p[4 * k + 1] = 6.0 + 12.0 * k           # mu_k = 6 + 12k
        p[4 * k + 2] = 2.0 + 2.0 * k / 7.0      # w = 2..4
        p[4 * k + 3] = 0.01 * (1 + (k % 3))     # small skew
    return p


def gauss8_numpy(p, x):
    y = np.zeros_like(x)
    for k in range(8):
        d = x - p[4 * k + 1]
        y += p[4 * k] * np.exp(-(d / p[4 * k + 2]) ** 2) * (1 + p[4 * k + 3] * d)
    return y


# ---- cfg 3: global fit, 4 local (A, B, C, bgr) + 3 global (tau1, tau2, tau3) ---------------
def model_global7(p, x):
    A, B, Cc, bgr, t1, t2, t3 = p
    return A * exp(-(x / t1)) + B * exp(-(x / t2)) + Cc * x * exp(-(x / t3)) + bgr


GLOBAL7_TAUS = np.array([1.5, 6.0, 20.0])


def global7_numpy(p, x):
    return p[0] * np.exp(-x / p[4]) + p[1] * np.exp(-x / p[5]) + p[2] * x * np.exp(-x / p[6]) + p[3]


def start_values(truth):
    """truth x (1 +- 5 %) alternating sign."""
    s = np.where(np.arange(truth.size) % 2 == 0, 1.05, 0.95)
    return truth * s


def make_single(fn_numpy, truth, n, x_lo, x_hi, seed=SEED):
    """x ascending in (x_lo, x_hi); y = f + sigma*N(0,1); sigma = 0.01(1+|f|).  Returns x, y, sigma."""
    x = x_lo + (x_hi - x_lo) * (np.arange(n, dtype=np.float64) + 0.5) / n
    f = fn_numpy(truth, x)
    sigma = 0.01 * (1.0 + np.abs(f))
    y = f + sigma * normal(n, seed)
    return x, y, sigma


def make_global7(n_datasets, n_per, seed=SEED):
    xs, ys, ss, truths = [], [], [], []
    for d in range(n_datasets):
        u = (splitmix64(4, seed + 1000 * (d + 1)) >> np.uint64(11)).astype(np.float64) / 9007199254740992.0
        loc = np.array([2.0 + 3.0 * u[0], 1.0 + 2.0 * u[1], 0.1 + 0.4 * u[2], 0.5 * u[3]])
        truth = np.concatenate([loc, GLOBAL7_TAUS])
        n = n_per if np.isscalar(n_per) else n_per[d]
        x, y, s = make_single(global7_numpy, truth, n, 0.0, 60.0, seed + d)
        xs.append(x); ys.append(y); ss.append(s); truths.append(truth)
    return xs, ys, ss, np.array(truths)


def make_single_slice(fn_numpy, truth, n_total, begin, count, x_lo, x_hi, seed=SEED):
    """The [begin, begin+count) slice of make_single(fn, truth, n_total, ...) without
    materialising the whole array (counter-based RNG): identical values on every rank."""
    i = np.arange(begin, begin + count, dtype=np.float64)
    x = x_lo + (x_hi - x_lo) * (i + 0.5) / n_total
    f = fn_numpy(truth, x)
    sigma = 0.01 * (1.0 + np.abs(f))
    with np.errstate(over='ignore'):
        ctr = np.arange(begin + 1, begin + count + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)

        def mix(z):
            z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            return z ^ (z >> np.uint64(31))
        u1 = (mix(ctr + np.uint64(seed)) >> np.uint64(11)).astype(np.float64) / 9007199254740992.0
        u2 = (mix(ctr + np.uint64(seed + 7919)) >> np.uint64(11)).astype(np.float64) / 9007199254740992.0
    z = np.sqrt(-2.0 * np.log(np.maximum(u1, 1e-300))) * np.cos(2.0 * np.pi * u2)
    return x, f + sigma * z, sigma


# ---- cfg 1: 2-exponential decay, 200 points, 4 active parameters --------------------
def model_exp2(p, x):
    return p[0] * exp(-(x / p[1])) + p[2] * exp(-(x / p[3]))


EXP2_TRUTH = np.array([5.0, 2.0, 2.0, 30.0])


def exp2_numpy(p, x):
    return p[0] * np.exp(-x / p[1]) + p[2] * np.exp(-x / p[3])


# ---- K skewed Gaussians (4K parameters): exercises 3 and 4 row tiles of the Gram kernels -------
def make_model_gaussK(K):
    def model(p, x):
        y = None
        for k in range(K):
            d = x - p[4 * k + 1]
            t = p[4 * k] * exp(-((d / p[4 * k + 2]) ** 2)) * (1 + p[4 * k + 3] * d)
            y = t if y is None else y + t
        return y
    return model


def gaussK_truth(K):
    p = np.zeros(4 * K)
    for k in range(K):
        p[4 * k] = 1.0 + 3.0 * ((k * 7) % K) / K
        p[4 * k + 1] = 100.0 * (k + 0.5) / K
        p[4 * k + 2] = 1.5 + 1.0 * ((k * 3) % K) / K
        p[4 * k + 3] = 0.01 * (1 + (k % 3))
    return p


def gaussK_numpy(K):
    def f(p, x):
        y = np.zeros_like(x)
        for k in range(K):
            d = x - p[4 * k + 1]
            y += p[4 * k] * np.exp(-(d / p[4 * k + 2]) ** 2) * (1 + p[4 * k + 3] * d)
        return y
    return f

"""The list epilogue's threshold (nn_mfma.h, coarse_unit MODE 2) is the resolve's bound tau_s(d) (nn_mfma.h: tau_from_a /
split_tau) evaluated in fp32 with every input rounded up.  It must never come out BELOW the fp64 bound -- a slot holding a
target within the row's distance bound would go unlisted.  Restated here in numpy (fp32 operations one by one, no
contraction, as capi.hip is built -ffp-contract=off) and checked over the ranges the kernels see: bounds from 0 to 1e30,
split frame terms from 1e-3 to 1e6, the row's fp32 images made exactly as RowBounds / k_knn_prebound make them."""
import numpy as np

U = 5.9604644775390625e-08            # 2^-24
REPR_EPS = 1.52587890625e-05 + U      # kReprEps
ARITH = 52.0                          # kArithBound


def tau64(a, d):
    """tau_from_a (nn_mfma.h): the fp64 bound on the coarse value of a target at exact distance d, frame term a."""
    eps = REPR_EPS * a * (1.0 + 1e-6)
    tau = d + eps * (2.0 * np.sqrt(d) + eps) + ARITH * U * a * a
    return tau * (1.0 + 5e-6) + 1e-300


def up32(x64):
    """(float)x rounded up, as RowBounds does for ub (x >= 0)."""
    with np.errstate(over="ignore"):
        f = x64.astype(np.float32)       # (beyond fp32's range: +Inf, as on the device)
    low = f.astype(np.float64) < x64
    return np.where(low, np.nextafter(f, np.float32(np.inf)), f).astype(np.float32)


def thr32(p2_f32, rho64, ub64):
    """coarse_unit<MODE 2>: fp32, operation by operation."""
    f = np.float32
    ubf = up32(ub64)
    sqf = np.sqrt(ubf, dtype=np.float32)
    sqf = np.nextafter(np.nextafter(sqf, f(np.inf)), f(np.inf))          # + 2 ulps
    a = (np.sqrt(p2_f32, dtype=np.float32) + rho64.astype(np.float32)) * f(1.0001)
    eps = f(1.5260e-05) * a
    t = (ubf + eps * (f(2.0) * sqf + eps)) + f(ARITH * U) * (a * a)
    return t * f(1.00002)


def test_fp32_threshold_dominates_the_fp64_bound():
    rng = np.random.default_rng(11)
    n = 400_000
    dist = 10.0 ** rng.uniform(-3, 6, n)                  # |p - c_s|
    rho = 10.0 ** rng.uniform(-3, 5, n)                   # split radius
    ub = np.concatenate([np.zeros(1000), 10.0 ** rng.uniform(-12, 30, n - 1000)])
    # what the kernel holds: |P~|^2 of the represented point (within 2^-16 of the true centred point, five fp32 roundings)
    wobble = 1.0 + rng.uniform(-1, 1, n) * (2.0 ** -16 + 2.0 ** -21)
    p2 = ((dist * wobble) ** 2).astype(np.float32)
    a_true = dist * (1.0 + 1e-6) + rho                    # split_tau's frame term (its fp32 sqrt is inside the 1e-6)
    t64 = tau64(a_true, ub)
    t32 = thr32(p2, rho, ub).astype(np.float64)
    ok = (t32 >= t64) | ~np.isfinite(t32)                 # (+Inf lists everything)
    assert ok.all(), (int((~ok).sum()), float((t64[~ok] / t32[~ok]).max()))
    # and it is not uselessly loose: within 0.1 % wherever the bound itself is not dominated by the 1e-4 margins on a
    tight = t32[np.isfinite(t32)] / t64[np.isfinite(t32)]
    assert np.median(tight) < 1.001 and tight.max() < 1.01


def test_special_rows():
    f = np.float32
    one = np.ones(1)
    # a row with a non-finite coordinate: NaN bound -> NaN threshold (nothing is <= NaN: nothing listed)
    assert np.isnan(thr32(one.astype(f), one, np.array([np.nan])))
    # no previous match: +Inf bound -> +Inf threshold (everything listed, the resolve's exhaustive path takes the row)
    assert np.isposinf(thr32(one.astype(f), one, np.array([np.inf])))
    # a coordinate beyond fp32's range makes |P|^2 overflow: +Inf again, never a finite threshold that is too small
    assert np.isposinf(thr32(np.array([np.inf], dtype=f), one, np.array([1e80])))

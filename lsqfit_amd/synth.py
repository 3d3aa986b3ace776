"""Synthetic correlated-Gaussian fit problems (benchmark generator).

Restates the recipe of ``lsqfit.fake_fitargs`` (src/lsqfit/_extras.py:2560-2589)
for the bench model of SURVEY.md 8(d): y-sdev = fsig * |f(x, p_true)| (:2569-2570),
correlation = normalize(U U^T) with U ~ Uniform(0.1, 0.9) (:2576-2581), data mean
= f(x, p_true) + a draw from that distribution (:2582-2583).  numpy only; the
reference draws from gvar's RNG without a fixed seed, here seeds are explicit
(PCG64, 20261 + config index).
"""
import numpy as np

from .models import cosmix


def cosmix_f(x, p):
    K = p.size // 2
    return np.cos(np.outer(x, p[K:])) @ p[:K]


def make_cosmix(N, P, seed, block=0, prior_corr=False, fsig=1e-3, dtype=np.float64):
    """-> dict(model, x, ymean, yerr, prior=(mean, err), p_true, p0).

    block = 0: uncorrelated data (C2); block = B: block-diagonal data covariance with
    B x B blocks corr = normalize(U U^T), U ~ Uniform(0.1, 0.9)^{B x 2B} (C4); block = N:
    one dense block of the same kind (C3)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    K = P // 2
    x = np.arange(N) * (2 * np.pi / N)
    a = rng.uniform(0.5, 1.5, K)
    w = np.arange(1, K + 1) + rng.uniform(-0.05, 0.05, K)
    p_true = np.concatenate([a, w])
    ybar = cosmix_f(x, p_true)
    sig = fsig * np.abs(ybar)
    sig[sig <= 0] = sig[sig > 0].min()
    z = rng.standard_normal(N)
    if block and block > 1:
        blocks = []
        noise = np.empty(N)
        for r0 in range(0, N, block):
            B = min(block, N - r0)
            # rectangular U (B x 2B): same structure as the reference's square recipe
            # (positive correlations, unit diagonal) but cond(corr) ~ 55 B instead of the
            # 1e9..1e12 the square one gives at B = 256 (measured), which would put ~10 %
            # of the blocks on the svdcut = 1e-12 threshold (SURVEY.md 8d makes the same
            # substitution for C3)
            U = rng.uniform(0.1, 0.9, (B, 2 * B))
            corr = U @ U.T
            d = 1.0 / np.sqrt(np.diag(corr))
            corr *= np.outer(d, d)
            s = sig[r0:r0 + B]
            cov = corr * np.outer(s, s)
            L = np.linalg.cholesky(cov)
            noise[r0:r0 + B] = L @ z[r0:r0 + B]
            if B > 1:
                blocks.append((r0, cov))
        yerr = dict(sdev=sig, blocks=blocks)
    else:
        noise = sig * z
        yerr = sig
    ymean = ybar + noise
    pm = np.concatenate([np.ones(K), np.arange(1, K + 1.0)])
    ps = np.concatenate([np.full(K, 0.5), np.full(K, 0.1)])
    if prior_corr:
        U = rng.uniform(0.1, 0.9, (P, 2 * P))
        corr = U @ U.T
        d = 1.0 / np.sqrt(np.diag(corr))
        corr *= np.outer(d, d)
        perr = corr * np.outer(ps, ps)
    else:
        perr = ps
    return dict(model=cosmix(K), x=x, ymean=ymean, yerr=yerr, prior=(pm, perr), p_true=p_true,
                p0=pm.copy())

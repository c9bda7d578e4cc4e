"""CPU checks of oracle/depth_filter.py (DepthFilter::Update, server_pose_graph/src/dense_mapping/depth_filter.cpp:177-259)
against the reference's lines evaluated by hand in scalar Python (IEEE double like the reference's `double`s)."""
import math

import numpy as np

from oracle.depth_filter import DepthFilter


def scalar_update(a, b, mu, cov, new_mu, new_cov, inv_range=100 - 0.01):
    """depth_filter.cpp:186-251 for one pixel, statement by statement; -> (a, b, mu, cov)"""
    old_sigma = math.sqrt(cov)
    old_sq = old_sigma * old_sigma
    new_sigma = math.sqrt(new_cov)
    if new_mu < 0.01 or new_mu > 100:
        return a, b + 1, mu, cov
    new_sq = new_sigma * new_sigma
    m = (new_sq * mu + old_sq * new_mu) / (old_sq + new_sq)
    s = (new_sq * old_sq) / (new_sq + old_sq)
    ssum = new_sq + old_sq
    pdf = math.exp(-(new_mu - mu) * (new_mu - mu) / (2.0 * ssum)) * math.sqrt(2.0 * 3.14159 * ssum)
    c1 = (a / (a + b)) * pdf
    c2 = (b / (a + b)) * 1.0 / inv_range
    norm = c1 + c2
    c1 /= norm
    c2 /= norm
    f = c1 * ((a + 1.0) / (a + b + 1.0)) + c2 * (a / (a + b + 1.0))
    e = c1 * ((a + 1.0) * (a + 2.0)) / ((a + b + 1.0) * (a + b + 2.0)) + c2 * (a * (a + 1.0)) / ((a + b + 1.0) * (a + b + 2.0))
    if math.isnan(c1 * m):
        return a, b, mu, cov
    fused_mu = c1 * m + c2 * mu
    fused_sigma = c1 * (s + m * m) + c2 * (old_sq + mu * mu) - fused_mu * fused_mu
    fused_a = (e - f) / (f - e / f)
    fused_b = fused_a * (1.0 - f) / f
    return fused_a, fused_b, fused_mu, fused_sigma * fused_sigma


def test_constructor_and_readout():
    f = DepthFilter(3, 4)
    assert (f.a == 15.0).all() and (f.b == 15.0).all() and (f.mu == 0.5).all() and (f.cov == 100.0).all()
    assert (f.ratio() == 0.5).all() and (f.inv_depth() == 0.5).all()   # ratio 0.5 is not below 0.5
    f.b[0, 0] = 16.0
    assert f.inv_depth()[0, 0] == 0.00001 and f.inv_depth()[0, 1] == 0.5


def test_update_matches_the_scalar_statements():
    rng = np.random.default_rng(5)
    H, W = 6, 7
    f = DepthFilter(H, W)
    state = [[(15.0, 15.0, 0.5, 100.0) for _ in range(W)] for _ in range(H)]
    for it in range(6):
        mu = rng.uniform(0.2, 1.5, (H, W))
        mu[0, 0] = 0.001      # outlier: below the range
        mu[0, 1] = 250.0      # outlier: above
        mu[1, 0] = np.nan     # NaN reading: the pixel is skipped (isnan(c1 * m))
        cov = rng.uniform(1e-4, 1e-2, (H, W)) if it % 2 else np.full((H, W), 4e-3)
        f.update(mu, cov)
        for v in range(H):
            for u in range(W):
                state[v][u] = scalar_update(*state[v][u], float(mu[v, u]), float(cov[v, u]))
        for v in range(H):
            for u in range(W):
                got = (f.a[v, u], f.b[v, u], f.mu[v, u], f.cov[v, u])
                for g, w in zip(got, state[v][u]):
                    assert g == w or (math.isnan(g) and math.isnan(w)) or abs(g - w) <= 4e-15 * abs(w), (it, v, u, got, state[v][u])
    assert f.b[0, 0] == 15.0 + 6 and f.b[0, 1] == 15.0 + 6 and f.a[0, 0] == 15.0
    assert (f.a[1, 0], f.b[1, 0], f.mu[1, 0], f.cov[1, 0]) == (15.0, 15.0, 0.5, 100.0)


def test_filter_converges_on_a_steady_reading():
    f = DepthFilter(2, 2)
    for _ in range(12):
        f.update(np.full((2, 2), 0.8), 4e-3)
    assert np.allclose(f.mu, 0.8, atol=1e-3) and (f.ratio() > 0.5).all()
    assert np.allclose(f.inv_depth(), f.mu)

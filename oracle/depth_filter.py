"""CPU restatement (numpy, float64) of the Bayesian inverse-depth filter that feeds the TSDF -- TEST INFRASTRUCTURE ONLY.

Follows server_pose_graph/src/dense_mapping/depth_filter.cpp: the constructor (:130-142: a = b = 15, mu = 0.5, cov = 100,
inverse-depth range 100 - 0.01), NormPdf (:10-16: exp(-(x - mu)^2 / (2 s2)) * sqrt(2 PI s2) with PI = 3.14159 -- a product, as
written), Update(mu, cov) (:177-259: per pixel, outlier branch b += 1 for a reading outside [0.01, 100], Gaussian x uniform
mixture update, `continue` when c1 * m is NaN, the new covariance stored as the SQUARE of the fused variance, as written) and
the read-out of DepthEstimator (depth_estimator.cpp:387-398: inverse depth 1e-5 where a / (a + b) < 0.5).
PARITY UNPINNED: the file includes OpenCV and Sophus and cannot be compiled here; exp() is libm's on the reference, numpy's
here and OCML's on the GPU, so the GPU test states a tolerance.  Only tests/ may import this module."""
import numpy as np

PI = 3.14159


class DepthFilter:
    def __init__(self, height, width):
        shape = (height, width)
        self.a = np.full(shape, 15.0)
        self.b = np.full(shape, 15.0)
        self.mu = np.full(shape, 0.5)
        self.cov = np.full(shape, 100.0)
        self.inv_depth_range = 100 - 0.01

    def update(self, new_mu, new_cov):
        new_mu = np.asarray(new_mu, np.float64)
        new_cov = np.broadcast_to(np.asarray(new_cov, np.float64), new_mu.shape)
        with np.errstate(all="ignore"):
            a, b, old_mu = self.a, self.b, self.mu
            old_sigma = np.sqrt(self.cov)
            old_sq = old_sigma * old_sigma
            new_sigma = np.sqrt(new_cov)
            outlier = (new_mu < 0.01) | (new_mu > 100)
            new_sq = new_sigma * new_sigma
            m = (new_sq * old_mu + old_sq * new_mu) / (old_sq + new_sq)
            s = (new_sq * old_sq) / (new_sq + old_sq)
            ssum = new_sq + old_sq
            pdf = np.exp(-(new_mu - old_mu) * (new_mu - old_mu) / (2.0 * ssum)) * np.sqrt(2.0 * PI * ssum)
            c1 = (a / (a + b)) * pdf
            c2 = (b / (a + b)) * 1.0 / self.inv_depth_range
            norm = c1 + c2
            c1 = c1 / norm
            c2 = c2 / norm
            f = c1 * ((a + 1.0) / (a + b + 1.0)) + c2 * (a / (a + b + 1.0))
            e = (c1 * ((a + 1.0) * (a + 2.0)) / ((a + b + 1.0) * (a + b + 2.0)) +
                 c2 * (a * (a + 1.0)) / ((a + b + 1.0) * (a + b + 2.0)))
            skip = np.isnan(c1 * m)
            fused_mu = c1 * m + c2 * old_mu
            fused_sigma = c1 * (s + m * m) + c2 * (old_sq + old_mu * old_mu) - fused_mu * fused_mu
            fused_a = (e - f) / (f - e / f)
            fused_b = fused_a * (1.0 - f) / f
            keep = outlier | skip
            self.b = np.where(outlier, b + 1, np.where(keep, b, fused_b))
            self.a = np.where(keep, a, fused_a)
            self.mu = np.where(keep, old_mu, fused_mu)
            self.cov = np.where(keep, self.cov, fused_sigma * fused_sigma)

    def ratio(self):
        with np.errstate(all="ignore"):
            return self.a / (self.a + self.b)

    def inv_depth(self):
        out = self.mu.copy()
        out[self.ratio() < 0.5] = 0.00001
        return out

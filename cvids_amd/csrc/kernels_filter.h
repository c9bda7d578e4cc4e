// kernels_filter.h -- the Bayesian inverse-depth filter in front of the TSDF path.
// Reference: server_pose_graph/src/dense_mapping/depth_filter.cpp:10-16 (NormPdf), :130-142 (constructor), :177-259
// (DepthFilter::Update(mu, cov)); read-out: depth_estimator.cpp:387-398.  One thread per pixel, all arithmetic in double, in the
// reference's order; exp() is OCML's (libm's on the reference: the last bit may differ, see tests/test_gpu_filter.py).
#pragma once
#include <hip/hip_runtime.h>

namespace chisel_hip {

struct FilterView {
    double *a, *b, *mu, *cov;  // m_mA, m_mB, m_mInvDepthMu, m_mInvDepthCov (depth_filter.h)
    int n;                     // height * width
    double inv_depth_range;    // m_nMaxInvDepth - m_nMinInvDepth
};

__global__ void filter_init_kernel(FilterView F) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= F.n) return;
    F.a[i] = 15.0;
    F.b[i] = 15.0;
    F.mu[i] = 0.5;
    F.cov[i] = 100.0;
}

__device__ inline double norm_pdf(double x, double mu, double sigma_sq) {  // depth_filter.cpp:10-16, PI = 3.14159
    return (exp(-(x - mu) * (x - mu) / (2.0 * sigma_sq))) * sqrt(2.0 * 3.14159 * sigma_sq);
}

// upd_cov == nullptr: the same covariance for every pixel (depth_estimator.cpp:293); reciprocal: the update is 1 / upd_mu[i]
// (the "mResultMap = 1.0 / mResultMap" of depth_estimator.cpp:286 fused in)
__global__ void filter_update_kernel(FilterView F, const double *__restrict__ upd_mu, const double *__restrict__ upd_cov, double cov_all,
                                     int reciprocal) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= F.n) return;
    const double nA = F.a[i];
    double nB = F.b[i];
    const double nOldMu = F.mu[i];
    const double nOldSigma = sqrt(F.cov[i]);
    const double nOldSigmaSq = nOldSigma * nOldSigma;
    double nNewMu = upd_mu[i];
    if (reciprocal) nNewMu = 1.0 / nNewMu;
    const double nNewSigma = sqrt(upd_cov ? upd_cov[i] : cov_all);
    if (nNewMu < 0.01 || nNewMu > 100) {  // outlier
        nB += 1;
        F.b[i] = nB;
        return;
    }
    const double nNewSigmaSq = nNewSigma * nNewSigma;
    const double nM = (nNewSigmaSq * nOldMu + nOldSigmaSq * nNewMu) / (nOldSigmaSq + nNewSigmaSq);
    const double nS = (nNewSigmaSq * nOldSigmaSq) / (nNewSigmaSq + nOldSigmaSq);
    double nC1 = (nA / (nA + nB)) * norm_pdf(nNewMu, nOldMu, nNewSigmaSq + nOldSigmaSq);
    double nC2 = (nB / (nA + nB)) * 1.0 / F.inv_depth_range;
    const double nNorm = nC1 + nC2;
    nC1 /= nNorm;
    nC2 /= nNorm;
    const double nF = nC1 * ((nA + 1.0) / (nA + nB + 1.0)) + nC2 * (nA / (nA + nB + 1.0));
    const double nE = nC1 * ((nA + 1.0) * (nA + 2.0)) / ((nA + nB + 1.0) * (nA + nB + 2.0)) +
                      nC2 * ((nA) * (nA + 1.0)) / ((nA + nB + 1.0) * (nA + nB + 2.0));
    if (isnan(nC1 * nM)) return;
    const double nFusedMu = nC1 * nM + nC2 * nOldMu;
    const double nFusedSigma = nC1 * (nS + nM * nM) + nC2 * (nOldSigmaSq + nOldMu * nOldMu) - nFusedMu * nFusedMu;
    const double nFusedA = (nE - nF) / (nF - nE / nF);
    const double nFusedB = nFusedA * (1.0 - nF) / nF;
    F.a[i] = nFusedA;
    F.b[i] = nFusedB;
    F.mu[i] = nFusedMu;
    F.cov[i] = nFusedSigma * nFusedSigma;
}

// which: 0 a, 1 b, 2 mu, 3 cov, 4 ratio a / (a + b) (DepthFilter::GetRatio), 5 inverse depth with 1e-5 where the ratio is
// below 0.5 (depth_estimator.cpp:387-398), 6 the depth map 1.0 / that (server_keyframe.cpp:1117: what PublishDenseInfo conditions)
__global__ void filter_read_kernel(FilterView F, int which, double *out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= F.n) return;
    double v;
    if (which == 0) v = F.a[i];
    else if (which == 1) v = F.b[i];
    else if (which == 2) v = F.mu[i];
    else if (which == 3) v = F.cov[i];
    else {
        const double ratio = F.a[i] / (F.a[i] + F.b[i]);
        if (which == 4) v = ratio;
        else v = ratio < 0.5 ? 0.00001 : F.mu[i];
    }
    out[i] = which == 6 ? 1.0 / v : v;
}

}  // namespace chisel_hip

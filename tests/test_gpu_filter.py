"""GPU parity of the inverse-depth filter (chisel_hip_depth_filter_*) against oracle/depth_filter.py.

All arithmetic is IEEE double in the reference's order except exp(): libm on the reference, numpy here, OCML on the GPU.  One
ulp of exp() reaches the state through divisions, so the tolerance is relative 1e-9 on a, b and mu after ten updates -- far
below the 1e-4 m the TSDF parity bar allows downstream.  The stored covariance is the SQUARE of a variance obtained by
cancellation (c1 (s + m^2) + c2 (s_old + mu_old^2) - mu^2: terms of order 1, result down to 1e-12, depth_filter.cpp:244), which
is ill-conditioned in the reference itself; it is compared as a variance (square root) with an absolute tolerance of 1e-12."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
RTOL = 1e-9


def _close(got, want, what, atol=0.0):
    assert (np.isnan(got) == np.isnan(want)).all(), what
    ok = np.isfinite(want)
    err = np.abs(got[ok] - want[ok]) - atol
    rel = err / np.maximum(np.abs(want[ok]), 1e-300)
    assert rel.max() <= RTOL, "%s: relative error %g" % (what, rel.max())


def test_filter_matches_the_oracle(hip_lib):
    from cvids_amd.chisel import DepthFilter
    from oracle.depth_filter import DepthFilter as Oracle
    H, W = 120, 160
    rng = np.random.default_rng(9)
    gf, of = DepthFilter(H, W), Oracle(H, W)
    for name, which in (("a", gf.A), ("b", gf.B), ("mu", gf.INV_DEPTH), ("cov", gf.COV)):
        assert np.array_equal(gf.read(which), getattr(of, name))       # constructor values, exactly
    truth = rng.uniform(0.3, 1.2, (H, W))
    for it in range(10):
        mu = truth + rng.normal(0.0, 0.02, (H, W))
        mu[rng.random((H, W)) < 0.05] = 300.0           # outliers above the range
        mu[rng.random((H, W)) < 0.02] = 0.0             # below
        mu[rng.random((H, W)) < 0.01] = np.nan
        mu[:12, :12] = 300.0                            # a corner that never gets a valid reading: its ratio falls below 0.5
        if it % 3 == 2:
            cov = rng.uniform(1e-4, 1e-2, (H, W))
            gf.Update(mu, cov)
            of.update(mu, cov)
        else:
            gf.Update(mu, 4.05e-3)                      # (3 * DEP_SAMPLE)^2-like constant, depth_estimator.cpp:293
            of.update(mu, 4.05e-3)
        for name, which in (("a", gf.A), ("b", gf.B), ("mu", gf.INV_DEPTH)):
            _close(gf.read(which), getattr(of, name), "update %d %s" % (it, name))
        _close(np.sqrt(gf.read(gf.COV)), np.sqrt(of.cov), "update %d variance" % it, atol=1e-12)
    _close(gf.GetRatio(), of.ratio(), "ratio")
    # the masked read-out can only differ where the ratio sits within rounding of 0.5
    inv_g, inv_o = gf.read(gf.INV_DEPTH_MASKED), of.inv_depth()
    decided = np.abs(of.ratio() - 0.5) > 1e-9
    _close(inv_g[decided], inv_o[decided], "masked inverse depth")
    assert (inv_o == 0.00001).sum() > 50 and (inv_o != 0.00001).sum() > H * W // 2
    _close(gf.read(gf.DEPTH)[decided], (1.0 / inv_o)[decided], "depth")


def test_filter_reciprocal_input_and_device_chain(hip_lib):
    """depth_estimator.cpp:286 (mResultMap = 1.0 / mResultMap) fused into the update; filter -> depth map -> PublishDenseInfo
    conditioning -> TSDF with every intermediate in HBM."""
    import ctypes as C

    import torch
    from cvids_amd import chisel as ch
    from cvids_amd import synth
    from oracle.depth_filter import DepthFilter as Oracle
    H, W = 480, 640
    intr = synth.intrinsics(W, H)
    pose = synth.trajectory_pose(0)
    depth = synth.render_depth("sphere_room", pose, intr, W, H).astype(np.float64)
    gf, of = ch.DepthFilter(H, W), Oracle(H, W)
    d_depth = torch.from_numpy(depth).cuda()
    for _ in range(8):
        gf.Update(d_depth, 4.05e-3, reciprocal=True)
        with np.errstate(all="ignore"):
            of.update(1.0 / depth, 4.05e-3)
    _close(gf.GetInvDepth(), of.mu, "mu after reciprocal updates")
    d_map = torch.empty((H, W), dtype=torch.float64, device="cuda")
    gf.read(gf.DEPTH, out=d_map)
    d_f32 = torch.empty((H, W), dtype=torch.float32, device="cuda")
    K = (C.c_double * 4)(*intr)
    assert hip_lib.chisel_hip_condition_depth(d_map.data_ptr(), W, H, 1, d_f32.data_ptr(), W, H, 1, K, None) == 0
    torch.cuda.synchronize()
    cond = d_f32.cpu().numpy()
    want = (1.0 / of.inv_depth()).astype(np.float32)
    ok = np.isfinite(cond)
    assert ok.mean() > 0.9 and np.abs(cond[ok] - want[ok]).max() < 1e-5
    gm = ch.Chisel((16, 16, 16), 0.02, False)
    integ = ch.ProjectionIntegrator(ch.InverseTruncator(2.0), ch.ConstantWeighter(1.0), 0.05, True)
    cam = ch.PinholeCamera(*intr, W, H, 0.05, 5.0)
    gm.IntegrateDepthScan(integ, d_f32, pose, cam)
    assert gm.NumChunks() > 100
    ok_q, dist = gm.GetSDF((0.0, 0.0, 2.49))
    assert ok_q and abs(dist) < 0.05

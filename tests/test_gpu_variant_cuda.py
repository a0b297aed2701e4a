"""the reference's CUDA-branch DualTVL1 semantics (cv2.cuda.OpticalFlowDual_TVL1,
calculate_optical_flow.py:572-575, 633-639) as TF_VARIANT_CUDA, against oracle variant 1 -- kernel by kernel (warp) and
end to end (identical executed-iteration counts, bit-identical flow).  Like the CPU variant, parity with real OpenCV is
unpinned (the oracle restates cudaoptflow from memory; see oracle/tvl1_oracle.c)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng_cuda():
    import tee_optical_flow_amd as T
    e = T.cuda_OpticalFlowDual_TVL1_create(device_id=0)
    yield e
    e.close()


def test_warp_kernel_matches_oracle(eng_cuda, oracle):
    from tee_optical_flow_amd import _lib
    from tee_optical_flow_amd.synth import speckle_pair
    rng = np.random.default_rng(4)
    for H, W in ((64, 80), (97, 131), (40, 40)):
        I0, I1, _ = speckle_pair(11, H, W)
        I0f, I1f = I0.astype(np.float32), I1.astype(np.float32)
        u1 = (rng.normal(0, 2.0, (H, W))).astype(np.float32)
        u2 = (rng.normal(0, 2.0, (H, W))).astype(np.float32)
        u1[::7, ::5] = np.round(u1[::7, ::5])                      # integer coordinates: five taps, the outer ones weigh 0
        u2[::3, ::4] = 0.0
        u1[0, :] = -6.0; u2[:, -1] = 9.0                           # far outside: every tap clamps to the border
        I1x, I1y = oracle.centered_gradient(I1f)
        ref = [np.empty((H, W), np.float32) for _ in range(4)]
        oracle.lib().orc_warp_cuda(I0f, I1f, I1x, I1y, u1, u2, W, H, *ref)
        got = [np.empty((H, W), np.float32) for _ in range(3)]
        L = _lib.load()
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        _lib.check(L.tf_dbg_warp(eng_cuda._h, p(I0f), p(I1f), p(u1), p(u2), W, H, p(got[0]), p(got[1]), p(got[2])), eng_cuda._h, "tf_dbg_warp")
        assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1]), "I1wx / I1wy differ"
        assert np.array_equal(got[2], ref[3]), "rho_c differs"


@pytest.mark.parametrize("seed,H,W", [(0, 128, 128), (3, 97, 131), (2, 256, 256)])
def test_pair_matches_oracle_variant(eng_cuda, oracle, seed, H, W):
    from tee_optical_flow_amd.synth import speckle_pair
    I0, I1, truth = speckle_pair(seed, H, W)
    p = oracle.default_params(variant=1)
    ref, ref_it, nl = oracle.tvl1_calc(I0, I1, p, return_iters=True)
    out = eng_cuda.calc(I0, I1, None)
    it = eng_cuda.last_iters()[0]
    assert np.array_equal(it, ref_it[:nl]), f"iteration counts differ:\n{it[..., 0]}\n{ref_it[:nl, :, 0]}"
    assert (it[..., 0] % 2 == 0).all() and (it[..., 1] == 0).all()          # stops only after odd iterations; no outer loop
    assert np.array_equal(out, ref)
    assert np.sqrt(((out - truth) ** 2).sum(-1))[8:-8, 8:-8].mean() < 0.1    # and it solves the problem


def test_batch_and_strip_kernels_variant(oracle):
    """A batch large enough for the row-strip kernel (the single-pair tests above take the tile kernel)."""
    import tee_optical_flow_amd as T
    from tee_optical_flow_amd.synth import speckle_pairs
    I0s, I1s = speckle_pairs(range(40, 80), 224, 224)
    I1s[5] = I0s[5]
    eng = T.DenseFlow(device_id=0, max_batch=40, variant="cuda")
    try:
        flows = eng.calc_pairs(I0s, I1s)
        its = eng.last_iters()
        p = oracle.default_params(variant=1)
        for b in (0, 5, 17, 39):
            ref, ref_it, nl = oracle.tvl1_calc(I0s[b], I1s[b], p, return_iters=True)
            assert np.array_equal(its[b], ref_it[:nl]) and np.array_equal(flows[b], ref)
        assert np.all(flows[5] == 0)
    finally:
        eng.close()


def test_variant_differs_from_cpu_variant_and_rejects_odd_totals(engine):
    import tee_optical_flow_amd as T
    from tee_optical_flow_amd.synth import speckle_pair
    I0, I1, _ = speckle_pair(9, 96, 96)
    e = T.cuda_OpticalFlowDual_TVL1_create()
    try:
        assert not np.array_equal(e.calc(I0, I1, None), engine.calc(I0, I1, None))
        with pytest.raises(T.OpticalFlowCalculationError):
            e.setInnerIterations(3)                                       # 3 x 10 = 30 is even: fine ...
            e.setOuterIterations(5)                                       # ... 3 x 5 = 15 is not
    finally:
        e.close()

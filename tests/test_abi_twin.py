"""the CPU oracle exports the same C ABI as the product.  tests/abi_driver.py is one script of raw
ABI calls; here it runs against the checker library alone (CPU) and, on the GPU box, against both libraries with every
observable compared: error codes, parameter handling, stats, iteration-count layout and the flows bit for bit."""
import numpy as np
import pytest


def _inputs():
    from tee_optical_flow_amd.synth import speckle_pairs, speckle_sequence
    return speckle_sequence(3, 4, 64, 80), speckle_pairs(range(60, 63), 72, 96)


@pytest.fixture(scope="module")
def cpu_run(oracle):
    from tests import abi_driver as D
    frames, pairs = _inputs()
    return D.drive(D.bind(D.CPU_LIB), frames, pairs), frames, pairs


def test_checker_library_speaks_the_abi(cpu_run, oracle):
    out, frames, (I0s, I1s) = cpu_run
    assert out["abi"] == 2
    assert out["codes"] == [0, 2, 2, 1, 2, 1, 1] and out["median_after_bad_set"] == 5.0   # TF_OK, UNSUPPORTED x2, INVALID_ARG, UNSUPPORTED (scaleStep 0.5), INVALID_ARG x2
    assert out["seq_too_short"] == 1 and out["bad_variant"] == 1 and out["deepflow_set_param"] == 2
    ref, it, nl = oracle.tvl1_calc(I0s[0], I1s[0], return_iters=True)
    assert np.array_equal(out["pair_flow"], ref) and out["pair_stats"][:3] == (1, nl, 5)
    assert out["pair_stats"][3] == int(it[:nl, :, 0].sum()) and out["pair_stats"][4] == int(it[:nl, :, 1].sum())
    for b in range(3):
        r, i2, n2 = oracle.tvl1_calc(I0s[b], I1s[b], return_iters=True)
        assert np.array_equal(out["pairs_flow"][b], r) and np.array_equal(out["pairs_iters"][b], i2[:n2])
    p = oracle.default_params(warps=3, epsilon=0.03)
    for i in range(3):
        assert np.array_equal(out["seq_flow"][i], oracle.tvl1_calc(frames[i], frames[i + 1], p) * np.float32(2.0))
    assert np.array_equal(out["queued_flow"], out["pairs_flow"]) and np.array_equal(out["queued_iters"].reshape(out["pairs_iters"].shape), out["pairs_iters"])
    assert np.array_equal(out["async_pairs_flow"], out["pairs_flow"]) and np.array_equal(out["async_pairs_iters"], out["queued_iters"])
    for i in range(3):
        assert np.array_equal(out["async_seq_flow"][i], oracle.tvl1_calc(frames[i], frames[i + 1]) * np.float32(0.5))
    assert out["async_seq_pairs"] == 3 and out["wait_twice"] == 1 and out["wait_all_when_none"] == 0
    assert np.array_equal(out["variant_flow"], oracle.tvl1_calc(I0s[1], I1s[1], oracle.default_params(variant=1))) and out["variant_outer"] == 0
    assert np.array_equal(out["deepflow_flow"], oracle.deepflow_calc(I0s[0], I1s[0]))


@pytest.mark.gpu
def test_product_and_checker_agree_call_for_call(cpu_run):
    from tee_optical_flow_amd import _lib
    from tests import abi_driver as D
    ref, frames, pairs = cpu_run
    got = D.drive(D.bind(_lib.LIB_PATH), frames, pairs)
    assert set(got) == set(ref)
    for k in ref:
        if isinstance(ref[k], np.ndarray):
            assert np.array_equal(got[k], ref[k]), f"{k} differs between libteeflow_hip.so and libteeflow_cpu.so"
        else:
            assert got[k] == ref[k], (k, got[k], ref[k])

"""Large and uneven batches through the lock-step driver: pairs that stop at very different iterations (identical frames,
unrelated frames, ordinary pairs) in one call, one lane against two, sequence mode with sub-batching, and the BASELINE.json
configs[2] per-GPU shard (128 pairs x 512^2 in one call) -- identical executed-iteration counts and bit-identical flows
against the CPU oracle on a spread sample."""
import numpy as np
import pytest


pytestmark = pytest.mark.gpu


def _engine(B, **params):
    import tee_optical_flow_amd as T
    return T.DenseFlow(device_id=0, max_batch=B, **params)


def _mixed_pairs(n, H, W, seed0=100):
    """Pairs that converge at very different speeds: speckle pairs, identical frames (stop at once), unrelated frames."""
    from tee_optical_flow_amd.synth import speckle_pair
    I0s, I1s = [], []
    for i in range(n):
        a, b, _ = speckle_pair(seed0 + i, H, W)
        if i % 7 == 3:
            b = a.copy()                                    # zero flow: every stage stops after its first iteration
        elif i % 11 == 5:
            b = speckle_pair(seed0 + 1000 + i, H, W)[0]      # nothing to match: many iterations
        I0s.append(a); I1s.append(b)
    return np.stack(I0s), np.stack(I1s)


def _run(eng, I0s, I1s, lanes=1):
    eng.set_tuning("lanes", lanes)
    f = np.array(eng.calc_pairs(I0s, I1s))
    return f, eng.last_iters().copy()


@pytest.mark.parametrize("n,H,W,params", [
    (40, 256, 256, {}),
    (96, 97, 131, {}),                                                     # ragged size, idle lanes, odd strip lengths
    (36, 240, 320, {"median_filtering": 3, "inner_iterations": 4, "outer_iterations": 3, "warps": 2, "nscales": 3}),
    (36, 240, 320, {"median_filtering": 1, "inner_iterations": 6, "outer_iterations": 2, "epsilon": 0.05}),
    (48, 200, 200, {"lambda_": 0.05, "theta": 0.2, "tau": 0.2, "scale_step": 0.7, "nscales": 4}),
])
def test_mixed_batch_one_lane_equals_two_and_the_oracle(oracle, n, H, W, params):
    I0s, I1s = _mixed_pairs(n, H, W)
    eng = _engine(n, **params)
    try:
        f0, it0 = _run(eng, I0s, I1s, lanes=1)
        spread = it0[..., 0].sum(axis=(1, 2))
        assert spread.max() > 1.5 * spread.min(), "the batch should mix fast and slow pairs"
        f2, it2 = _run(eng, I0s, I1s, lanes=2)                               # the batch cut over two independent lanes
        assert np.array_equal(it0, it2) and np.array_equal(f0, f2)
        op = oracle.default_params(**params)
        for b in (3, 5, n - 1):                                              # a zero-flow pair, an unrelated pair, the last one
            ref, ref_it, nl = oracle.tvl1_calc(I0s[b], I1s[b], op, return_iters=True)
            assert np.array_equal(it0[b], ref_it[:nl]) and np.array_equal(f0[b], ref), f"pair {b}"
    finally:
        eng.close()


def test_mixed_batch_matches_oracle_on_sample(oracle):
    I0s, I1s = _mixed_pairs(32, 256, 256, seed0=300)
    eng = _engine(32)
    try:
        f, it = _run(eng, I0s, I1s)
        for b in (0, 3, 5, 17, 31):
            ref, ref_it, nl = oracle.tvl1_calc(I0s[b], I1s[b], return_iters=True)
            assert np.array_equal(it[b], ref_it[:nl]), f"pair {b}: iteration counts differ from the oracle"
            assert np.array_equal(f[b], ref), f"pair {b}: not bit-identical to the oracle"
    finally:
        eng.close()


def test_sequence_mode_and_sub_batches(oracle):
    """tf_calc_seq (pairs share frames) with a call larger than the engine's capacity."""
    from tee_optical_flow_amd.synth import speckle_sequence
    frames = speckle_sequence(5, 70, 224, 224)
    eng = _engine(40)
    try:
        flows = np.array(eng.calc_batch(frames, scale=2.5))                  # 69 pairs > capacity 40: two sub-batches
        it = eng.last_iters()
        for i in (0, 39, 40, 68):
            ref, ref_it, nl = oracle.tvl1_calc(frames[i], frames[i + 1], return_iters=True)
            assert np.array_equal(it[i], ref_it[:nl])
            assert np.array_equal(flows[i], ref * np.float32(2.5))
    finally:
        eng.close()


def test_config3_shard_128_pairs_512(oracle):
    """BASELINE.json configs[2]: the per-GPU shard, 128 pairs x 512^2 in ONE call through the C ABI (two lanes of 64).
    Parity on a sample spread over both lanes; the executed-iteration histogram must not be degenerate."""
    from bench import make_inputs
    B, H, W = 128, 512, 512
    I0s, I1s = make_inputs(range(B), H, W, allow_pool=False)      # no child processes from a GPU-initialised process
    eng = _engine(B)
    try:
        eng.set_tuning("lanes", 2)
        got = eng.calc_pairs(I0s, I1s)
        it = eng.last_iters()
        per_pair = it[..., 0].sum(axis=(1, 2))
        assert per_pair.min() >= 25 and per_pair.max() <= 7500
        assert len(np.unique(per_pair)) > 20 and per_pair.max() > 1.3 * per_pair.min(), "degenerate iteration histogram"
        for b in (0, 1, 37, 63, 64, 101, 127):                               # both lanes, first / last of each
            ref, ref_it, nl = oracle.tvl1_calc(I0s[b], I1s[b], return_iters=True)
            assert np.array_equal(it[b], ref_it[:nl]), f"pair {b}: iteration counts differ from the oracle"
            assert np.array_equal(got[b], ref), f"pair {b}: not bit-identical to the oracle"
    finally:
        eng.close()


def test_config4_batch_128_pairs_512_deepflow_every_sor_form_agrees(oracle):
    """BASELINE.json configs[3] (DeepFlow, all defaults) at the bench's batch: 128 pairs x 512^2 in one call, two lanes of 64.  The oracle needs
    ~0.35 s per pair, so at this size the property checked over ALL 128 flows is form-independence -- the co-resident launches (the default),
    the tiled register kernel and the one-colour-per-launch plain form must give the same bits -- and a spread sample is held against the oracle."""
    from bench import make_inputs
    import tee_optical_flow_amd as T
    B, H, W = 128, 512, 512
    I0s, I1s = make_inputs(range(B), H, W, allow_pool=False)
    eng = T.DenseFlow(device_id=0, max_batch=B, algo="deepflow")
    try:
        flows = np.asarray(eng.calc_pairs(I0s, I1s)).copy()
        assert eng.counter("coop_launches") > 1000 and eng.counter("coop_aborts") == 0
        n = eng.counter("coop_launches")
        eng.set_tuning("sor_coop", 0)
        tiled = np.asarray(eng.calc_pairs(I0s, I1s))
        assert eng.counter("coop_launches") == n
        assert np.array_equal(flows, tiled), "co-resident and tiled SOR disagree"
        eng.set_tuning("sor_rt", 0)
        plain = np.asarray(eng.calc_pairs(I0s[:16], I1s[:16]))
        assert np.array_equal(flows[:16], plain), "register-tile and plain SOR disagree"
        for b in (0, 63, 64, 127):
            assert np.array_equal(flows[b], oracle.deepflow_calc(I0s[b], I1s[b])), f"pair {b}: not bit-identical to the oracle"
    finally:
        eng.close()

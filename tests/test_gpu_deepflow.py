"""DeepFlow path (BASELINE config 4) on the GPU against oracle/deepflow_oracle.c: bit-exact.
(The oracle itself is a from-memory restatement of OpenCV's OpticalFlowDeepFlow / VariationalRefinement: parity with real
OpenCV is UNPINNED -- see the oracle's header.)"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


@pytest.fixture(scope="module")
def deep():
    import tee_optical_flow_amd as T
    e = T.createOptFlow_DeepFlow()
    yield e
    e.close()


@pytest.mark.parametrize("shape", [(64, 64), (97, 131), (30, 27), (1, 40), (3, 2)])
def test_blur_bit_exact(deep, oracle, shape):
    from tee_optical_flow_amd import _lib
    L = _lib.load()
    h, w = shape
    src = np.random.default_rng(0).uniform(0, 255, (h, w)).astype(np.float32)
    ref = oracle.deepflow_gauss_blur3(src, 0.6)
    out = np.empty_like(src)
    _lib.check(L.tf_dbg_df_blur(deep._h, _ptr(src), w, h, _ptr(out)), deep._h)
    assert np.array_equal(out, ref)


@pytest.mark.parametrize("fuse,shape_knob", [(0, 1), (1, 1), (2, 1), (3, 1), (4, 1), (5, 1), (6, 1), (7, 1), (8, 1),
                                             (1, 2), (3, 2), (5, 2), (6, 2), (8, 2), (5, 3), (3, 3)])
@pytest.mark.parametrize("shape,amp", [((64, 64), 1.0), ((97, 131), 4.0), ((40, 52), 40.0), ((26, 26), 0.5), ((70, 200), 2.0), ((96, 96), 3.0), ((150, 301), 2.0), ((333, 141), 1.0)])
def test_variational_refinement_bit_exact(deep, oracle, shape, amp, fuse, shape_knob):
    """One cv::VariationalRefinement::calcUV (warp, 8 derivative planes, 5 x [data term, smoothness, 25 red-black SOR sweeps]) in
    every SOR form: one colour per launch (fuse 0), n sweeps per launch of the register-tile kernel on 128 x 64 regions held by
    16 bands x 4 rows (shape 1), on 128 x 32 regions held by 8 bands x 4 rows (shape 2), or the
    launcher's own choice between 1 and 2 (shape 3, the default); a level that fits one region runs all 25 sweeps in one launch."""
    from scipy import ndimage
    from tee_optical_flow_amd import _lib
    L = _lib.load()
    h, w = shape
    rng = np.random.default_rng(1)
    I0 = ndimage.gaussian_filter(rng.uniform(0, 255, (h, w)), 1.5).astype(np.float32)
    I1 = ndimage.shift(I0, (0.7, -1.2), order=1, mode="nearest").astype(np.float32)
    u = rng.uniform(-amp, amp, (h, w)).astype(np.float32)
    v = rng.uniform(-amp, amp, (h, w)).astype(np.float32)
    ru, rv = oracle.deepflow_variational_refine(I0, I1, u, v)
    gu, gv = u.copy(), v.copy()
    deep.set_tuning("sor_fuse", fuse)
    deep.set_tuning("sor_rt_shape", shape_knob)
    deep.set_tuning("sor_coop", 0)                      # the tiled forms are what this test is about
    try:
        _lib.check(L.tf_dbg_df_refine(deep._h, _ptr(I0), _ptr(I1), w, h, _ptr(gu), _ptr(gv)), deep._h)
    finally:
        deep.set_tuning("sor_fuse", 5)
        deep.set_tuning("sor_rt_shape", 3)
        deep.set_tuning("sor_coop", 1)
    assert np.array_equal(gu, ru), f"u: {np.sum(gu != ru)} differ, max {np.abs(gu - ru).max()}"
    assert np.array_equal(gv, rv)


@pytest.mark.parametrize("shape_knob", [1, 2])
def test_sor_plain_division_path_bit_exact(deep, oracle, shape_knob):
    """A block whose diagonals leave the range the pre-scaled division is exact for takes the plain IEEE division; real data never
    gets there, so the path is forced (sor_plain_div) -- both divisions are correctly rounded, the results must not move."""
    from scipy import ndimage
    from tee_optical_flow_amd import _lib
    L = _lib.load()
    h, w = 150, 301
    rng = np.random.default_rng(13)
    I0 = ndimage.gaussian_filter(rng.uniform(0, 255, (h, w)), 1.5).astype(np.float32)
    I1 = ndimage.shift(I0, (0.3, -0.8), order=1, mode="nearest").astype(np.float32)
    u = rng.uniform(-2, 2, (h, w)).astype(np.float32)
    v = rng.uniform(-2, 2, (h, w)).astype(np.float32)
    ru, rv = oracle.deepflow_variational_refine(I0, I1, u, v)
    gu, gv = u.copy(), v.copy()
    deep.set_tuning("sor_plain_div", 1)
    deep.set_tuning("sor_rt_shape", shape_knob)
    try:
        _lib.check(L.tf_dbg_df_refine(deep._h, _ptr(I0), _ptr(I1), w, h, _ptr(gu), _ptr(gv)), deep._h)
    finally:
        deep.set_tuning("sor_plain_div", 0)
        deep.set_tuning("sor_rt_shape", 3)
    assert np.array_equal(gu, ru) and np.array_equal(gv, rv)


@pytest.mark.parametrize("S,plain", [(1, 0), (2, 0), (3, 0), (4, 0), (5, 0), (5, 1), (6, 0), (7, 0), (3, 1)])
@pytest.mark.parametrize("shape", [(150, 301), (333, 141), (97, 131), (70, 200), (65, 129), (512, 512), (200, 520), (301, 75), (139, 255)])
def test_sor_coresident_regions_bit_exact(deep, oracle, shape, S, plain):
    """k_df_sor_rt_coop: the 25 sweeps of a fixed-point iteration in ONE launch -- the regions of the level are resident together, keep the
    linear system in registers and trade (du, dv) with the regions they overlap every S sweeps through memory (flags, no grid barrier).
    Same regions, halo and arithmetic as the tiled form; checked against the oracle for every S (phases of unequal length included)."""
    from scipy import ndimage
    from tee_optical_flow_amd import _lib
    L = _lib.load()
    h, w = shape
    rng = np.random.default_rng(31)
    I0 = ndimage.gaussian_filter(rng.uniform(0, 255, (h, w)), 1.4).astype(np.float32)
    I1 = ndimage.shift(I0, (-0.5, 1.1), order=1, mode="nearest").astype(np.float32)
    u = rng.uniform(-2, 2, (h, w)).astype(np.float32)
    v = rng.uniform(-2, 2, (h, w)).astype(np.float32)
    ru, rv = oracle.deepflow_variational_refine(I0, I1, u, v)
    gu, gv = u.copy(), v.copy()
    before = deep.counter("coop_launches")
    deep.set_tuning("sor_coop", 2)                     # 2: also for a single pair (1 leaves that to the tiled form's 128 x 32 regions)
    deep.set_tuning("sor_coop_s", S)
    deep.set_tuning("sor_plain_div", plain)
    try:
        _lib.check(L.tf_dbg_df_refine(deep._h, _ptr(I0), _ptr(I1), w, h, _ptr(gu), _ptr(gv)), deep._h)
    finally:
        deep.set_tuning("sor_coop", 1)
        deep.set_tuning("sor_coop_s", 5)
        deep.set_tuning("sor_plain_div", 0)
    assert deep.counter("coop_launches") == before + 5          # one launch per fixed-point iteration: the form under test did run
    assert deep.counter("coop_aborts") == 0
    assert np.array_equal(gu, ru), f"u: {np.sum(gu != ru)} differ, max {np.abs(gu - ru).max()}"
    assert np.array_equal(gv, rv)


@pytest.mark.parametrize("plain", [0, 1])
@pytest.mark.parametrize("shape", [(64, 62), (65, 62), (128, 61), (100, 33), (127, 60), (30, 27), (61, 2), (5, 45), (129, 62), (64, 63)])
def test_sor_two_bands_per_wave_bit_exact(deep, oracle, shape, plain):
    """Levels at most 62 px wide and 128 rows high run with two bands per wave (k_df_sor_rt<.., HALF>), lanes 31 / 32 being neighbours in
    the wave but not in the image; the shapes either side of those limits take the ordinary forms."""
    from scipy import ndimage
    from tee_optical_flow_amd import _lib
    L = _lib.load()
    h, w = shape
    rng = np.random.default_rng(21)
    I0 = ndimage.gaussian_filter(rng.uniform(0, 255, (h, w)), 1.2).astype(np.float32)
    I1 = ndimage.shift(I0, (0.6, 0.9), order=1, mode="nearest").astype(np.float32)
    u = rng.uniform(-3, 3, (h, w)).astype(np.float32)
    v = rng.uniform(-3, 3, (h, w)).astype(np.float32)
    ru, rv = oracle.deepflow_variational_refine(I0, I1, u, v)
    gu, gv = u.copy(), v.copy()
    deep.set_tuning("sor_plain_div", plain)
    try:
        _lib.check(L.tf_dbg_df_refine(deep._h, _ptr(I0), _ptr(I1), w, h, _ptr(gu), _ptr(gv)), deep._h)
    finally:
        deep.set_tuning("sor_plain_div", 0)
    assert np.array_equal(gu, ru) and np.array_equal(gv, rv)


@pytest.mark.parametrize("ds", [0, 2])
@pytest.mark.parametrize("shape,amp", [((97, 131), 4.0), ((40, 52), 40.0), ((26, 26), 0.5), ((150, 301), 2.0), ((333, 141), 1.0), ((65, 258), 3.0)])
def test_data_and_smoothness_term_forms_bit_exact(deep, oracle, shape, amp, ds):
    """The linear system of a fixed-point iteration in its three kernel forms: k_df_data + k_df_smooth (0), both in one pass with one
    pixel per thread (1), four pixels per thread with 16-byte loads (2, the default) -- through a full calcUV against the oracle."""
    from scipy import ndimage
    from tee_optical_flow_amd import _lib
    L = _lib.load()
    h, w = shape
    rng = np.random.default_rng(5)
    I0 = ndimage.gaussian_filter(rng.uniform(0, 255, (h, w)), 1.5).astype(np.float32)
    I1 = ndimage.shift(I0, (-0.4, 0.9), order=1, mode="nearest").astype(np.float32)
    u = rng.uniform(-amp, amp, (h, w)).astype(np.float32)
    v = rng.uniform(-amp, amp, (h, w)).astype(np.float32)
    ru, rv = oracle.deepflow_variational_refine(I0, I1, u, v)
    gu, gv = u.copy(), v.copy()
    deep.set_tuning("df_fuse_ds", ds)
    try:
        _lib.check(L.tf_dbg_df_refine(deep._h, _ptr(I0), _ptr(I1), w, h, _ptr(gu), _ptr(gv)), deep._h)
    finally:
        deep.set_tuning("df_fuse_ds", 2)
    assert np.array_equal(gu, ru) and np.array_equal(gv, rv)


@pytest.mark.parametrize("seed,H,W", [(0, 96, 96), (1, 120, 160), (2, 64, 200), (3, 100, 75), (4, 55, 123), (5, 75, 139)])     # odd widths: the last lane's second column is row padding
def test_deepflow_pair_matches_oracle(deep, oracle, seed, H, W):
    from tee_optical_flow_amd.synth import speckle_pair
    I0, I1, truth = speckle_pair(seed, H, W)
    ref, nl = oracle.deepflow_calc(I0, I1, return_levels=True)
    out = deep.calc(I0, I1, None)
    assert out.shape == (H, W, 2) and out.dtype == np.float32
    assert deep.last_stats["nscales_used"] == nl
    e = np.sqrt(((out - ref) ** 2).sum(-1))
    assert e.mean() <= 1e-3 and e.max() <= 1e-2        # north_star tolerance ...
    assert np.array_equal(out, ref)                     # ... and in fact bit-exact
    assert np.sqrt(((out - truth) ** 2).sum(-1))[12:-12, 12:-12].mean() < 0.1


def test_deepflow_512_config4(deep, oracle):
    """BASELINE.json configs[3]: OF_algo='deepflow', one 512x512 pair, all defaults (60 pyramid levels)."""
    from tee_optical_flow_amd.synth import speckle_pair
    I0, I1, truth = speckle_pair(0, 512, 512)
    ref, nl = oracle.deepflow_calc(I0, I1, return_levels=True)
    out = deep.calc(I0, I1, None)
    assert nl == 60 and deep.last_stats["nscales_used"] == 60
    assert np.array_equal(out, ref)


def test_deepflow_batch_and_sequence(deep, oracle):
    from tee_optical_flow_amd.synth import speckle_pairs, speckle_sequence
    I0s, I1s = speckle_pairs(range(10, 14), 72, 80)
    flows = deep.calc_pairs(I0s, I1s)
    for b in range(4):
        assert np.array_equal(flows[b], oracle.deepflow_calc(I0s[b], I1s[b])), f"pair {b}"
    fr = speckle_sequence(5, 4, 64, 72)
    fs = deep.calc_batch(fr, scale=1.5)
    for i in range(3):
        assert np.array_equal(fs[i], oracle.deepflow_calc(fr[i], fr[i + 1]) * np.float32(1.5))


@pytest.mark.parametrize("B,lanes,S", [(5, 1, 5), (7, 1, 3), (36, 2, 5), (33, 2, 4)])
def test_deepflow_batches_through_coresident_regions(oracle, B, lanes, S):
    """Full solves whose larger levels run k_df_sor_rt_coop: several pairs per launch, several launches per level (the regions of a
    level x the pairs exceed the CUs a lane may fill), one lane or two lanes that split the CUs -- first, middle and last pair against the oracle."""
    import tee_optical_flow_amd as T
    from tee_optical_flow_amd.synth import speckle_pairs
    H, W = 150, 420                                      # 4 x 4 regions at the first level
    I0s, I1s = speckle_pairs(range(300, 300 + B), H, W)
    eng = T.DenseFlow(algo="deepflow", max_batch=B)
    try:
        eng.set_tuning("lanes", lanes)
        eng.set_tuning("sor_coop", 2)
        eng.set_tuning("sor_coop_s", S)
        flows = eng.calc_pairs(I0s, I1s)
        assert eng.counter("coop_launches") > 0 and eng.counter("coop_aborts") == 0
        for b in sorted({0, B // 2, B - 1}):
            assert np.array_equal(flows[b], oracle.deepflow_calc(I0s[b], I1s[b])), f"pair {b}"
    finally:
        eng.close()


@pytest.mark.parametrize("S,plain", [(1, 0), (3, 0), (4, 0), (5, 0), (5, 1), (6, 0)])      # S = 6: the halo would reach past the neighbours' 8-row cores -> 128 x 64 regions
@pytest.mark.parametrize("shape", [(150, 301), (333, 141), (97, 131), (512, 512), (200, 520), (40, 300)])
def test_sor_coresident_small_regions_bit_exact(deep, oracle, shape, S, plain):
    """Few pairs (here one): the co-resident form takes 128 x 32 regions -- 512-thread blocks, two per CU -- so that a single pair still
    spreads over the chip; same exchange protocol, checked against the oracle."""
    from scipy import ndimage
    from tee_optical_flow_amd import _lib
    L = _lib.load()
    h, w = shape
    rng = np.random.default_rng(37)
    I0 = ndimage.gaussian_filter(rng.uniform(0, 255, (h, w)), 1.4).astype(np.float32)
    I1 = ndimage.shift(I0, (0.8, -0.7), order=1, mode="nearest").astype(np.float32)
    u = rng.uniform(-2, 2, (h, w)).astype(np.float32)
    v = rng.uniform(-2, 2, (h, w)).astype(np.float32)
    ru, rv = oracle.deepflow_variational_refine(I0, I1, u, v)
    gu, gv = u.copy(), v.copy()
    before = deep.counter("coop_launches")
    deep.set_tuning("sor_coop", 3)                     # 3: the small-region form whenever the batch is small (1 lets the size rules decide)
    deep.set_tuning("sor_coop_s", S)
    deep.set_tuning("sor_plain_div", plain)
    try:
        _lib.check(L.tf_dbg_df_refine(deep._h, _ptr(I0), _ptr(I1), w, h, _ptr(gu), _ptr(gv)), deep._h)
    finally:
        deep.set_tuning("sor_coop", 1)
        deep.set_tuning("sor_coop_s", 5)
        deep.set_tuning("sor_plain_div", 0)
    assert deep.counter("coop_launches") == before + 5 and deep.counter("coop_aborts") == 0
    assert np.array_equal(gu, ru), f"u: {np.sum(gu != ru)} differ, max {np.abs(gu - ru).max()}"
    assert np.array_equal(gv, rv)


def test_coresident_form_is_chosen_per_level_by_how_well_it_fills_the_cus(oracle):
    """With the default knobs a level runs co-resident only if whole pairs fill >= 85 % of the CUs the handle may use; the others run
    tiled.  Same flows whatever the rule decides (0 = always co-resident, 101 = never)."""
    import tee_optical_flow_amd as T
    from tee_optical_flow_amd.synth import speckle_pairs
    I0s, I1s = speckle_pairs(range(60, 66), 300, 400)
    ref = [oracle.deepflow_calc(I0s[b], I1s[b]) for b in (0, 5)]
    launches = {}
    for util in (0, 85, 101):
        eng = T.DenseFlow(algo="deepflow", max_batch=6)
        try:
            eng.set_tuning("sor_coop_min_util", util)
            eng.set_tuning("sor_coop_small", 0)            # the small-batch form (128 x 32 regions) is not subject to this rule
            flows = eng.calc_pairs(I0s, I1s)
            launches[util] = eng.counter("coop_launches")
            assert np.array_equal(flows[0], ref[0]) and np.array_equal(flows[5], ref[1]), f"min_util {util}"
        finally:
            eng.close()
    assert launches[101] == 0 and launches[0] >= launches[85] > 0, launches


@pytest.mark.parametrize("form", [2, 3])          # 128 x 64 regions / the small-batch form (128 x 32 regions, two blocks per CU)
def test_coresident_launch_that_cannot_meet_gives_up_backs_off_and_re_arms(oracle, form):
    """Every wait in k_df_sor_rt_coop is bounded.  With block 0 muted (it never raises its flag) its neighbours poll ~0.1 s, raise the
    launch's abort word and leave, every other block follows, the kernel ends; the host sees the word after the solve and solves the
    batch again with the tiled form -- same flows as ever.  The form is not dropped for good: the handle sits out a number of tiled solves
    (16, doubled by every further abort; 2 here), then tries again; tf_last_error says what happened."""
    import tee_optical_flow_amd as T
    from tee_optical_flow_amd.synth import speckle_pairs
    I0s, I1s = speckle_pairs(range(40, 43), 150, 300)
    eng = T.DenseFlow(algo="deepflow", max_batch=3)
    try:
        eng.set_tuning("sor_coop", form)
        eng.set_tuning("coop_backoff", 1)                  # the next abort sits out 2 solves instead of 16
        eng.set_tuning("coop_test_mute", 1)
        flows = eng.calc_pairs(I0s, I1s)
        assert eng.counter("coop_aborts") == 1 and eng.counter("coop_disabled") == 1 and eng.counter("coop_cooldown") == 2
        n = eng.counter("coop_launches")
        for b in range(3):
            assert np.array_equal(flows[b], oracle.deepflow_calc(I0s[b], I1s[b])), f"pair {b}"
        eng.set_tuning("coop_test_mute", 0)                # the neighbour on the GPU has gone
        flows2 = eng.calc_pairs(I0s, I1s)                  # sits out (tiled), 1 left
        assert eng.counter("coop_launches") == n and eng.counter("coop_disabled") == 1 and eng.counter("coop_cooldown") == 1
        flows3 = eng.calc_pairs(I0s, I1s)                  # sits out (tiled), re-armed at its end
        assert eng.counter("coop_launches") == n and eng.counter("coop_disabled") == 0 and eng.counter("coop_rearms") == 1
        flows4 = eng.calc_pairs(I0s, I1s)                  # co-resident again
        assert eng.counter("coop_launches") > n and eng.counter("coop_aborts") == 1
        for f in (flows2, flows3, flows4):
            assert np.array_equal(np.asarray(f), np.asarray(flows))
        # a second abort doubles the wait; setting the knob re-arms at once
        eng.set_tuning("coop_test_mute", 1)
        eng.calc_pairs(I0s, I1s)
        assert eng.counter("coop_aborts") == 2 and eng.counter("coop_cooldown") == 4
        eng.set_tuning("coop_test_mute", 0)
        eng.set_tuning("sor_coop", form)
        n2 = eng.counter("coop_launches")
        flows5 = eng.calc_pairs(I0s, I1s)
        assert eng.counter("coop_launches") > n2 and eng.counter("coop_disabled") == 0
        assert np.array_equal(np.asarray(flows5), np.asarray(flows))
    finally:
        eng.close()


def test_coresident_form_is_sized_from_the_occupancy_query(oracle):
    """tf_create_deepflow's buffers come with a hipOccupancyMaxActiveBlocksPerMultiprocessor query for both co-resident instantiations: one
    1024-thread block per CU (128 x 64 regions) and two 512-thread blocks (128 x 32) are what the form counts on.  A runtime that promises
    less gets the tiled form (same bits)."""
    import tee_optical_flow_amd as T
    from tee_optical_flow_amd.synth import speckle_pairs
    I0s, I1s = speckle_pairs(range(50, 53), 150, 300)
    eng = T.DenseFlow(algo="deepflow", max_batch=3)
    try:
        eng.set_tuning("sor_coop", 2)
        ref = eng.calc_pairs(I0s, I1s)
        assert eng.counter("coop_occ16") >= 1 and eng.counter("coop_occ8") >= 2, "this build no longer fits the co-resident form's block shapes"
        n = eng.counter("coop_launches")
        assert n > 0
        eng.set_tuning("coop_test_occ16", 0)               # as if a 1024-thread block did not fit a CU
        out = eng.calc_pairs(I0s, I1s)
        assert eng.counter("coop_launches") == n and eng.counter("coop_aborts") == 0
        assert np.array_equal(np.asarray(out), np.asarray(ref))
        eng.set_tuning("coop_test_occ16", -1)
        eng.set_tuning("sor_coop", 3)                      # the small-batch form needs two blocks per CU
        eng.set_tuning("coop_test_occ8", 1)
        out = eng.calc_pairs(I0s, I1s)
        assert eng.counter("coop_launches") == n
        eng.set_tuning("coop_test_occ8", -1)
        out2 = eng.calc_pairs(I0s, I1s)
        assert eng.counter("coop_launches") > n
        assert np.array_equal(np.asarray(out), np.asarray(ref)) and np.array_equal(np.asarray(out2), np.asarray(ref))
    finally:
        eng.close()

"""Host-side logic of bench.py that needs no GPU: the kernel-source fingerprint that decides whether a stored PMC / ISA
record may be used, the profiler detection that keeps child processes away from a GPU-initialised process, the input
cache, and the FETCH_SIZE factors."""
import json
import os
import sys

import pytest

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def test_fingerprint_gates_stored_records(tmp_path, monkeypatch):
    import bench
    fp = bench.kernel_source_fingerprint()
    assert len(fp) == 16 and fp == bench.kernel_source_fingerprint()
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "kernel_source_fingerprint", lambda: fp)
    rec, why = bench.stored_record("hbm_traffic.json", "k_iter2_rows")
    assert rec is None and "missing" in why
    (prof / "hbm_traffic.json").write_text(json.dumps({"k_iter2_rows": {"bytes_per_launch": 1.0, "source_fingerprint": "0" * 16, "round": "r01"}}))
    rec, why = bench.stored_record("hbm_traffic.json", "k_iter2_rows")
    assert rec is None and "not used" in why                     # another build's counters are never replayed
    (prof / "hbm_traffic.json").write_text(json.dumps({"k_iter2_rows": {"bytes_per_launch": 5.0, "source_fingerprint": fp, "round": "r02"}}))
    rec, why = bench.stored_record("hbm_traffic.json", "k_iter2_rows")
    assert why is None and rec["bytes_per_launch"] == 5.0
    rec, why = bench.stored_record("hbm_traffic.json", "k_df_sor_rt")
    assert rec is None and "no record" in why


def test_profiler_detection_and_input_cache(tmp_path, monkeypatch):
    import bench
    for k in list(os.environ):
        if k.startswith(("ROCPROF", "ROCP_")):
            monkeypatch.delenv(k)
    monkeypatch.setenv("LD_PRELOAD", "")
    assert not bench.under_profiler()
    monkeypatch.setenv("LD_PRELOAD", "/opt/rocm/lib/librocprofiler-sdk-tool.so")
    assert bench.under_profiler()
    monkeypatch.setenv("TMPDIR", str(tmp_path))
    a0, a1 = bench.make_inputs([3, 4], 40, 48)                   # under a "profiler": generated in-process, no pool
    assert a0.shape == (2, 40, 48) and a0.dtype == np.uint8 and any(f.startswith("teeflow_bench_inputs_") for f in os.listdir(tmp_path))
    b0, b1 = bench.make_inputs([3, 4], 40, 48)                   # second call: from the cache, identical
    assert np.array_equal(a0, b0) and np.array_equal(a1, b1)
    from tee_optical_flow_amd.synth import speckle_pair
    assert np.array_equal(a0[1], speckle_pair(4, 40, 48)[0])


def test_fetch_factors_and_committed_records_are_well_formed():
    import bench
    for algo in ("TVL1", "deepflow"):                            # calibrated on the GPU (profiles/r03_fetch_calibration.json): 2.0 at every width
        f, src = bench.fetch_factor(algo)
        assert f == 2.0 and "fetch_calibration" in src
    cal = json.load(open(os.path.join(ROOT, "profiles", "r03_fetch_calibration.json")))["kernels"]
    assert set(cal) >= {"4", "8", "16", "8_tile_rows"} and all(abs(k["fetch_factor"] - 2.0) < 0.01 and abs(k["write_factor"] - 1.0) < 0.01 for k in cal.values())
    for name in ("hbm_traffic.json", "isa_stats.json"):
        p = os.path.join(ROOT, "profiles", name)
        if os.path.exists(p):
            d = json.load(open(p))
            for kern, rec in d.items():
                assert "source_fingerprint" in rec and len(rec["source_fingerprint"]) == 16, (name, kern)


def test_spawn_ranks_sets_the_rendezvous_and_reports_failures(tmp_path):
    """`python bench.py --gpus N` from a plain shell starts its own ranks.  The launcher itself, on a child
    that needs no GPU: every rank sees RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*; rank 0's stdout comes back; a failing rank
    makes the whole run fail."""
    import bench
    child = tmp_path / "child.py"
    child.write_text("import json, os, sys\n"
                     "r = int(os.environ['RANK'])\n"
                     "open(sys.argv[1] + f'/rank{r}.json', 'w').write(json.dumps({k: os.environ.get(k) for k in "
                     "('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT', 'HSA_ENABLE_IPC_MODE_LEGACY')}))\n"
                     "print(json.dumps({'n_gpus': int(os.environ['WORLD_SIZE']), 'rank': r}))\n"
                     "sys.exit(int(sys.argv[2]) if r == 1 else 0)\n")
    rc, out = bench.spawn_ranks(3, [str(tmp_path), "0"], script=str(child), timeout=60)
    assert rc == 0 and json.loads(out) == {"n_gpus": 3, "rank": 0}
    envs = [json.loads((tmp_path / f"rank{r}.json").read_text()) for r in range(3)]
    assert [e["RANK"] for e in envs] == ["0", "1", "2"] and [e["LOCAL_RANK"] for e in envs] == ["0", "1", "2"]
    assert {e["WORLD_SIZE"] for e in envs} == {"3"} and {e["MASTER_ADDR"] for e in envs} == {"127.0.0.1"}
    assert len({e["MASTER_PORT"] for e in envs}) == 1 and {e["HSA_ENABLE_IPC_MODE_LEGACY"] for e in envs} == {"0"}
    rc, out = bench.spawn_ranks(2, [str(tmp_path), "7"], script=str(child), timeout=60)
    assert rc == 7 and json.loads(out)["rank"] == 0


def test_limiter_is_derived_from_counter_records(tmp_path, monkeypatch):
    """roofline.limiter comes from the stored SQ counters of the build, not from a literal."""
    import bench
    fp = "f" * 16
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "kernel_source_fingerprint", lambda: fp)
    (tmp_path / "profiles").mkdir()
    assert bench.classify_limiter("k_x", 0.5)[0].startswith("unclassified")
    rec = {"k_x": {"source_fingerprint": fp, "derived": {"valu_pipe_busy_per_simd_all_launches": 0.91, "wave_time_share_parked_waitcnt_or_barrier": 0.2,
                                                          "wave_time_share_issue_stalled": 0.3, "mean_waves_per_simd_resident": 3.9}},
           "k_y": {"source_fingerprint": fp, "derived": {"valu_pipe_busy_per_simd_all_launches": 0.35, "wave_time_share_parked_waitcnt_or_barrier": 0.55,
                                                          "wave_time_share_issue_stalled": 0.1, "mean_waves_per_simd_resident": 3.0}},
           "k_z": {"source_fingerprint": "0" * 16, "derived": {"valu_pipe_busy_per_simd_all_launches": 0.99}}}
    (tmp_path / "profiles" / "r09_sq_counters.json").write_text(json.dumps(rec))
    lim, ev = bench.classify_limiter("k_x", 0.4)
    assert lim.startswith("valu") and ev["valu_pipe_busy_per_simd"] == 0.91 and ev["source"] == "profiles/r09_sq_counters.json"
    assert bench.classify_limiter("k_y", 0.85)[0].startswith("hbm")
    assert bench.classify_limiter("k_y", 0.45)[0].startswith("memory latency")
    assert bench.classify_limiter("k_z", 0.45)[0].startswith("unclassified")          # another build's counters are not used


def _fake_run(E, first, n, collectives, fail_at=None, nbuf=None):
    """run_steps_pipelined with fake submits / collects and recorded collectives; returns (stats, log, pending buffers)."""
    import bench
    nbuf = nbuf or E + 1
    log, pending, running = [], set(), set()

    def submit(k, buf):
        assert buf == k % nbuf
        assert buf not in pending, f"step {k} is solved into buffer {buf} while its all-gather is still pending"
        assert buf not in running, f"step {k} is solved into buffer {buf} while an earlier step is still writing it"
        assert len(running) < E, "more steps in flight than asked for"
        running.add(buf)
        log.append(("submit", k, buf))
        if E == 1:                                         # the synchronous form: submit is the solve
            if k == fail_at:
                running.discard(buf)
                raise RuntimeError(f"injected failure in step {k}")
            running.discard(buf)
            return {"k": k}
        return ("ticket", k, buf)

    def collect(hd):
        if E == 1:
            return hd
        _, k, buf = hd
        running.discard(buf)
        log.append(("collect", k, buf))
        if k == fail_at:
            raise RuntimeError(f"injected failure in step {k}")
        return {"k": k}

    def issue(buf):
        assert buf not in pending and buf not in running
        pending.add(buf)
        log.append(("issue", None, buf))

    def retire(buf):
        pending.discard(buf)
        log.append(("retire", None, buf))
    stats = bench.run_steps_pipelined(first, n, E, nbuf, submit, collect, issue if collectives else None, retire if collectives else None)
    return stats, log, pending, running


@pytest.mark.parametrize("E,first,n", [(1, 0, 5), (2, 1, 9), (3, 5, 20), (3, 0, 2), (4, 3, 13)])
@pytest.mark.parametrize("collectives", [False, True], ids=["one-rank", "with-collectives"])
def test_pipelined_steps_order_and_buffer_safety(E, first, n, collectives):
    """The N > 1 form of the timed loop cannot run on the hardware here (one GPU): its logic is pinned with fakes.  Every step is submitted
    once, in order, with at most E in flight; collectives are issued in step order, each right after its step has been collected; no step
    is solved into a buffer that is still being gathered or written (asserted inside the fakes); the last min(n, E + 1) all-gathers are
    left for the caller's drain."""
    stats, log, pending, running = _fake_run(E, first, n, collectives)
    assert [s["k"] for s in stats] == list(range(first, first + n)) and not running
    assert [e[1] for e in log if e[0] == "submit"] == list(range(first, first + n))
    if E > 1:
        assert [e[1] for e in log if e[0] == "collect"] == list(range(first, first + n))
        # a step is collected only when E are in flight (or at the end): the overlap is real
        pos = {("submit", e[1]): i for i, e in enumerate(log) if e[0] == "submit"}
        pos.update({("collect", e[1]): i for i, e in enumerate(log) if e[0] == "collect"})
        for k in range(first, first + n - E):
            assert pos[("submit", k + E - 1)] < pos[("collect", k)] < pos[("submit", k + E)]
    if collectives:
        assert [e[2] for e in log if e[0] == "issue"] == [k % (E + 1) for k in range(first, first + n)]      # step order
        assert len(pending) == min(n, E + 1)
    else:
        assert not [e for e in log if e[0] in ("issue", "retire")]


@pytest.mark.parametrize("E", [1, 3])
@pytest.mark.parametrize("collectives", [False, True])
def test_pipelined_steps_failure_leaves_nothing_in_flight(E, collectives):
    """A step that fails must not leave later steps submitted and uncollected."""
    import bench  # noqa: F401
    with pytest.raises(RuntimeError, match="injected failure in step 9"):
        _fake_run(E, 2, 15, collectives, fail_at=9)

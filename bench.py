#!/usr/bin/env python3
"""bench.py -- frame-pairs/s of the MI355X DualTVL1 path (BASELINE.json metric), plus a DeepFlow leg (BASELINE configs[3]).

One "step" = one pass of the hot path over one batch = ONE synchronous call of the C-ABI boundary (tf_calc_pairs_device): every rank
solves `--batch` (default 384) independent 512x512 uint8 frame pairs (synthetic "speckle-warp v1"; pair shape of
BASELINE configs[1]; three sub-batches of configs[2]'s per-GPU shard size, 128) with all-default DualTVL1 (lambda 0.15), inputs already
resident in HBM, and -- for N > 1 -- the (u,v) fields are all-gathered over RCCL (the one exchange step north_star names), overlapped
with the next step's compute.  value = pairs all ranks solved / max-over-ranks wall time.

Inside the call the LIBRARY cuts the batch into 128-pair sub-batches that its three lanes (engines of their own: stream, buffers, host
thread) take from a queue, so one sub-batch's tail runs under the others' full launches (include/teeflow.h "Sub-batches and lanes").
Round 4 reached that overlap with three engines driven by Python threads in this file; rounds 1-3 timed 128-pair calls that were split
in two contiguous halves and joined -- the default N=1 run still times that form right after and reports it as "steps_joined", and
the same 128-pair batches submitted without waiting (tf_submit_pairs_device, three in flight) as "jobs_in_flight".  `--in-flight E`
(E > 1) makes the timed region itself use tf_submit_* with E steps in flight.  Every pair's flow is bit-identical in all forms
(tests/test_gpu_queue.py, tools/queue_forms.py).

The default N=1 run then measures BASELINE configs[3] (OF_algo='deepflow', the algorithm the reference's own CLI
hard-codes, calculate_optical_flow.py:735-739) on 128 pairs per call and reports it under "deepflow" in the same JSON line.

Everything that is not GPU work (synthetic inputs, the optional `--pmc` counter passes, which run this script as a
child under rocprofv3) happens BEFORE the first GPU call, so no process is ever started from a GPU-initialised one.

  python bench.py [--gpus N] [--steps K] [--warmup W]          (N > 1 from a plain shell: this process starts the N ranks itself)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
"""
import argparse
import glob
import hashlib
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s measured streaming ceiling)
HBM_STREAM_GBS = 6290.0
SIMDS = 256 * 4         # 256 CUs x 4 SIMDs
TVL1_KERNEL = "k_iter2_rows"
DF_KERNEL = "k_df_sor_rt"
# FETCH_SIZE correction.  MI355X_MICROARCH.md (HBM): gfx950 tallies the 128-B requests of 16-B-per-lane loads at 64 B, "other
# widths: calibrate".  Calibrated (tools/calibrate_fetch.sh -> profiles/r03_fetch_calibration.json: 1-GiB streaming copies under
# --pmc): the counter reports half the bytes at EVERY width tried -- 4, 8, 16 B per lane and the SOR kernel's 8-B tile-row
# shape -- and WRITE_SIZE is exact.  Round 2 took the 8-B loads of the DeepFlow kernel raw (factor 1): its read traffic was
# under-reported by 2x.
LOAD_BYTES_PER_LANE = {"TVL1": "16", "deepflow": "8_tile_rows"}


def fetch_factor(algo):
    """(factor, source) for FETCH_SIZE of the dominant kernel's load shape, from the newest calibration under profiles/."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_fetch_calibration.json")))
    if files:
        try:
            with open(files[-1]) as f:
                k = json.load(f)["kernels"][LOAD_BYTES_PER_LANE[algo]]
            return float(round(k["fetch_factor"], 3)), f"profiles/{os.path.basename(files[-1])} [{LOAD_BYTES_PER_LANE[algo]} B per lane]"
        except (OSError, ValueError, KeyError):
            pass
    return 2.0, "MI355X_MICROARCH.md (16-B-per-lane streaming reads); no calibration file under profiles/"


def classify_limiter(kernel, frac_of_streaming_ceiling):
    """What the stored SQ / GRBM counter passes of THIS build (profiles/*_sq_counters.json, source fingerprint checked) say limits
    the dominant kernel, next to the measured share of the streaming ceiling.  Returns (limiter string, evidence dict)."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_sq_counters.json")))
    for f in reversed(files):
        try:
            with open(f) as fh:
                rec = json.load(fh).get(kernel)
        except (OSError, ValueError):
            continue
        if not rec or "derived" not in rec:
            continue
        if rec.get("source_fingerprint") not in (None, kernel_source_fingerprint()):
            continue
        d = rec["derived"]
        busy, parked, stalled = d.get("valu_pipe_busy_per_simd_all_launches"), d.get("wave_time_share_parked_waitcnt_or_barrier"), d.get("wave_time_share_issue_stalled")
        ev = {"source": f"profiles/{os.path.basename(f)}", "source_fingerprint": rec.get("source_fingerprint"), "valu_pipe_busy_per_simd": busy,
              "wave_time_parked_on_waitcnt_or_barrier": parked, "wave_time_issue_stalled": stalled,
              "mean_waves_per_simd_resident": d.get("mean_waves_per_simd_resident"), "frac_of_streaming_ceiling": frac_of_streaming_ceiling}
        if rec.get("source_fingerprint") is None:
            ev["note"] = "counter record carries no source fingerprint (taken before round 3): it may describe an older build of this kernel"
        if busy is not None and busy >= 0.8:
            lim = "valu (vector ALU busy %.2f of the time)" % busy
        elif frac_of_streaming_ceiling is not None and frac_of_streaming_ceiling >= 0.8:
            lim = "hbm (%.2f of the streaming ceiling)" % frac_of_streaming_ceiling
        elif parked is not None and parked >= 0.4 and (busy or 0) < 0.6:
            lim = "memory latency / synchronisation (waves parked on s_waitcnt or barriers %.2f of their time, vector ALU busy %.2f)" % (parked, busy or 0)
        else:
            lim = "mixed: hbm %.2f of the streaming ceiling, vector ALU busy %s" % (frac_of_streaming_ceiling or 0, "%.2f" % busy if busy is not None else "n/a")
        return lim, ev
    return "unclassified (no SQ counter record of this build under profiles/)", None


# ---------------------------------------------------------------------------------------------------------------
# N > 1 without a launcher: the CPU-only parent starts one child per GPU (it never imports torch or touches HIP)
# ---------------------------------------------------------------------------------------------------------------
def spawn_ranks(n, argv, script=None, timeout=None):
    """Run `script argv` as n rank processes of one node (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set the
    way torch.distributed.run sets them), relay rank 0's stdout, and return (rc, rank-0 stdout).  rc is the first
    non-zero child code (a killed child counts).  No exec: the children are ordinary subprocesses of a parent that has
    not initialised the GPU."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    script = script or os.path.abspath(__file__)
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC only on this pool: RCCL needs it in every rank
        procs.append(subprocess.Popen([sys.executable, script] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=sys.stderr))
    rc, out0 = 0, ""
    t_end = None if timeout is None else time.time() + timeout
    try:
        out0 = procs[0].communicate(timeout=timeout)[0].decode(errors="replace")
        for p_ in procs:
            left = None if t_end is None else max(1.0, t_end - time.time())
            p_.wait(timeout=left)
    except subprocess.TimeoutExpired:
        rc = 124
    finally:
        for p_ in procs:                      # exactly the processes started here, never a pattern
            if p_.poll() is None:
                p_.kill()
                p_.wait()
    for p_ in procs:
        if p_.returncode and not rc:
            rc = p_.returncode if p_.returncode > 0 else 128 - p_.returncode
    return rc, out0


# ---------------------------------------------------------------------------------------------------------------
# inputs (CPU only; runs before any GPU call)
# ---------------------------------------------------------------------------------------------------------------
def _gen_pair(args):
    from tee_optical_flow_amd.synth import speckle_pair
    seed, H, W = args
    I0, I1, _ = speckle_pair(seed, H, W)
    return I0, I1


def under_profiler():
    """rocprofv3 preloads its tool library, which initialises the GPU before main(): no child processes then."""
    pre = os.environ.get("LD_PRELOAD", "").lower()
    return "rocprof" in pre or any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ)


def make_inputs(seeds, H, W, allow_pool=True):
    """[I0s, I1s] uint8 [B,H,W].  Cached under $TMPDIR (the generator is deterministic); a process pool only when
    nothing in this process has touched the GPU and no profiler is attached."""
    seeds = list(seeds)
    key = hashlib.sha1(repr((seeds, H, W, "speckle-warp v1")).encode()).hexdigest()[:16]
    cache = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"teeflow_bench_inputs_{key}.npz")
    if os.path.exists(cache):
        try:
            z = np.load(cache)
            if z["I0s"].shape == (len(seeds), H, W):
                return z["I0s"], z["I1s"]
        except (OSError, ValueError, KeyError):
            pass
    n = min(8, len(seeds), os.cpu_count() or 1)
    if allow_pool and n > 1 and not under_profiler():
        import multiprocessing as mp
        with mp.get_context("spawn").Pool(n) as pool:
            res = pool.map(_gen_pair, [(s, H, W) for s in seeds])
    else:
        res = [_gen_pair((s, H, W)) for s in seeds]
    I0s, I1s = np.stack([r[0] for r in res]), np.stack([r[1] for r in res])
    try:
        tmp = cache + f".{os.getpid()}.npz"     # np.savez appends nothing: the name already ends in .npz
        np.savez(tmp, I0s=I0s, I1s=I1s)
        os.replace(tmp, cache)
    except OSError:
        pass
    return I0s, I1s


# ---------------------------------------------------------------------------------------------------------------
# evidence kept under profiles/: PMC traffic and ISA statistics, each tied to the kernel source it was taken from
# ---------------------------------------------------------------------------------------------------------------
def kernel_source_fingerprint():
    """sha256 over the HIP sources + the build recipe: a stored PMC / ISA record is used only for the build it came from."""
    h = hashlib.sha256()
    for p in sorted(glob.glob(os.path.join(ROOT, "tee_optical_flow_amd", "csrc", "*"))):
        if p.endswith((".hip", ".h", "Makefile")):
            with open(p, "rb") as f:
                h.update(os.path.basename(p).encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def stored_record(fname, kernel):
    """(record, why_not) from profiles/<fname> for `kernel`, accepted only when its source fingerprint is this build's."""
    path = os.path.join(ROOT, "profiles", fname)
    try:
        with open(path) as f:
            rec = json.load(f).get(kernel)
    except (OSError, ValueError):
        return None, f"profiles/{fname} missing or unreadable"
    if not rec:
        return None, f"profiles/{fname} has no record for {kernel}"
    fp = kernel_source_fingerprint()
    if rec.get("source_fingerprint") != fp:
        return None, (f"profiles/{fname} was taken from kernel sources {rec.get('source_fingerprint')} (round {rec.get('round')}), "
                      f"this build is {fp}: not used")
    return rec, None


def pmc_child_passes(algo, out_dir, tag="live"):
    """`--pmc`: run this script as a CHILD under rocprofv3 (one pass per counter, as MI355X_MICROARCH.md prescribes) and
    return {'fetch_kb','write_kb','launches'} averaged per launch of the dominant kernel.  Must be called before this
    process touches the GPU."""
    import csv
    kern = TVL1_KERNEL if algo == "TVL1" else DF_KERNEL
    res = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = os.path.join(out_dir, f"pmc_{tag}_{algo}_{ctr}")
        cmd = ["rocprofv3", "--pmc", ctr, "--kernel-include-regex", kern, "--output-format", "csv", "-d", d, "-o", "pmc", "--",
               sys.executable, os.path.abspath(__file__), "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-profile",
               "--in-flight", "1", "--lanes", "1", "--no-deepflow", "--steps-only", "--algo", algo, "--batch", "128"]     # one sub-batch, one lane: the launches the event pairs time
        env = dict(os.environ, TMPDIR=os.environ.get("TMPDIR", "/tmp"))
        r = subprocess.run(cmd, cwd=os.environ.get("TMPDIR", "/tmp"), env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
        if r.returncode != 0:
            return {"error": f"rocprofv3 --pmc {ctr} failed rc={r.returncode}: {r.stderr.decode(errors='replace')[-300:]}"}
        per = {}
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    if row["Counter_Name"] == ctr and kern in row["Kernel_Name"]:
                        per[row["Dispatch_Id"]] = per.get(row["Dispatch_Id"], 0.0) + float(row["Counter_Value"])
        if not per:
            return {"error": f"no {ctr} rows for {kern}"}
        v = list(per.values())
        res["fetch_kb" if ctr == "FETCH_SIZE" else "write_kb"] = sum(v) / len(v)
        res["launches"] = len(v)
        res["command"] = (f"rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE (one pass each) --kernel-include-regex {kern} -- python3 bench.py "
                          + " ".join(cmd[cmd.index("--steps"):]))
    return res


# ---------------------------------------------------------------------------------------------------------------
# CPU baseline (the oracle; test infrastructure used here only as the reported baseline and the parity checker)
# ---------------------------------------------------------------------------------------------------------------
def cv2_baseline(I0s, I1s, n_sample, algo):
    """BASELINE.md tier B2: real OpenCV on this box's host cores -- only when cv2 happens to be importable (probed, never
    required, never installed).  Returns (record, flows) or (None, None)."""
    import importlib.util
    if importlib.util.find_spec("cv2") is None:
        return None, None
    try:
        import cv2
        if algo == "TVL1":
            m = cv2.optflow.createOptFlow_DualTVL1()       # reference calculate_optical_flow.py:577-578
            m.setLambda(0.15)
        else:
            m = cv2.optflow.createOptFlow_DeepFlow()       # reference :568
        threads = cv2.getNumThreads()
        m.calc(I0s[0], I1s[0], None)
        t0 = time.perf_counter()
        flows = [m.calc(I0s[i], I1s[i], None) for i in range(n_sample)]
        dt = time.perf_counter() - t0
        return {"value": n_sample / dt, "unit": "frame-pairs/s", "cores": threads, "kind": "opencv", "cv2_version": cv2.__version__,
                "sample": f"{n_sample} of the benchmark's pairs, 1 warm-up, cv2.optflow {'DualTVL1 (lambda 0.15)' if algo == 'TVL1' else 'DeepFlow'}"}, flows
    except Exception as e:   # an OpenCV build without the contrib optflow module, ...
        return {"error": f"cv2 present but unusable: {e!r}"}, None


def cpu_baseline(I0s, I1s, n_sample, algo="TVL1"):
    """Oracle (CPU restatement, NOT OpenCV) timed on this box's host cores on a bounded sample of the same workload:
    all granted cores, and one thread as the median over >= 3 pairs.  What is timed is the
    -O3 -march=native build of the oracle sources (`make -C oracle o3`, compiled on this machine by main() before the GPU is
    touched) once it has reproduced the -O2 checker bit for bit on the sample's first pair; the returned flows -- the parity
    reference -- always come from the checker build."""
    from oracle import oracle as O
    threads = O.effective_cpus()
    O.set_num_threads(threads)
    calc = O.tvl1_calc if algo == "TVL1" else O.deepflow_calc
    ref0 = calc(I0s[0], I1s[0])  # warm-up
    fast, build = O.o3_calc(algo), "-O2 checker build (oracle/Makefile)"
    if fast is not None:
        if np.array_equal(fast(I0s[0], I1s[0]), ref0):
            build = "-O3 -march=native -fopenmp -ffp-contract=off build of the same sources (bit-identical to the -O2 checker on the sample's first pair)"
        else:
            fast, build = None, "-O2 checker build (the -O3 -march=native build did NOT reproduce it bit for bit and was discarded)"
    timed = fast if fast is not None else calc
    t0 = time.perf_counter()
    for i in range(n_sample):
        timed(I0s[i], I1s[i])
    dt = time.perf_counter() - t0
    flows = [ref0] + [calc(I0s[i], I1s[i]) for i in range(1, n_sample)]
    O.set_num_threads(1)
    one = []
    for i in range(min(3, len(I0s))):
        t1 = time.perf_counter()
        timed(I0s[i], I1s[i])
        one.append(time.perf_counter() - t1)
    O.set_num_threads(threads)
    rec = {"value": n_sample / dt, "unit": "frame-pairs/s", "cores": threads, "kind": "port",
           "sample": f"{n_sample} of the benchmark's 512x512 pairs (seeds 0..{n_sample - 1}), 1 warm-up, "
                     f"oracle/{'tvl1' if algo == 'TVL1' else 'deepflow'}_oracle.c with {threads} OpenMP threads; restatement, not OpenCV",
           "build": build, "one_thread_pairs_per_s_median_of_3": 1.0 / float(np.median(one))}
    return rec, flows


# ---------------------------------------------------------------------------------------------------------------
# one measured leg
# ---------------------------------------------------------------------------------------------------------------
def launch_profile(eng):
    """Per-launch records of the last profiled single-lane solve: level, warp, first iteration, ms."""
    import ctypes as C
    from tee_optical_flow_amd import _lib
    L = _lib.load()
    n = L.tf_dbg_launch_profile(eng._h, None, None, None, None, 0)
    lv, wp, it = (np.zeros(max(n, 1), np.int32) for _ in range(3))
    ms = np.zeros(max(n, 1), np.float32)
    ptr = lambda x: x.ctypes.data_as(C.c_void_p)
    L.tf_dbg_launch_profile(eng._h, ptr(lv), ptr(wp), ptr(it), ptr(ms), n)
    return lv[:n], wp[:n], it[:n], ms[:n]


def run_steps_pipelined(first, n, E, nbuf, submit, collect, issue=None, retire=None):
    """Steps first .. first+n-1 on ONE host thread with at most E of them submitted and not yet collected.  Step k's flows go to
    buffer k % nbuf.  `submit(k, buf)` starts the step and returns what `collect(...)` turns into its statistics (E = 1: submit IS
    the synchronous solve and collect is the identity; E > 1: tf_submit_* / tf_wait -- the overlap happens on the library's lanes,
    not here).  Collectives (None on a single rank): `issue(buf)` right after the step that filled the buffer has been collected --
    so in step order, as RCCL needs on every rank -- and `retire(buf)` (host-blocking) before a buffer is solved into again; with
    nbuf > E the all-gather of a step has the whole next step to finish.  An exception leaves nothing submitted and uncollected.
    Returns the n statistics in step order; the last min(n, nbuf) all-gathers are left for the caller's drain."""
    from collections import deque
    assert E >= 1 and nbuf >= E
    stats, inflight = [], deque()

    def collect_oldest():
        k0, h0 = inflight.popleft()
        stats.append(collect(h0))
        if issue:
            issue(k0 % nbuf)
    try:
        for k in range(first, first + n):
            if len(inflight) == E:
                collect_oldest()
            if retire:
                retire(k % nbuf)
            inflight.append((k, submit(k, k % nbuf)))
        while inflight:
            collect_oldest()
    except BaseException:
        while inflight:                                   # whatever failed, nothing of this loop stays in flight
            try:
                collect(inflight.popleft()[1])
            except Exception:
                pass
        raise
    return stats


def run_leg(a, algo, B, steps, warmup, I0s, I1s, torch, dist, dev, rank, world, local_rank, live_pmc=None, cpu_sample=8):
    import tee_optical_flow_amd as T
    H = W = a.size
    frames = torch.from_numpy(np.concatenate([I0s, I1s])).to(dev)       # [2B,H,W] u8, resident in HBM
    SUB = min(a.sub_batch, B)                             # pairs per sub-batch (the engine's max_batch): what one lane solves at a time
    # ONE engine.  A step is one synchronous tf_calc_pairs_device call of B pairs; when B > SUB the library cuts it into sub-batches that
    # its lanes take from a queue (three lanes for DualTVL1, one for DeepFlow, whose co-resident SOR launches take every CU; a call of at
    # most SUB pairs is split in `--lanes` contiguous parts and joined).  `--in-flight E` > 1: steps are submitted without waiting
    # (tf_submit_pairs_device), E in flight -- the same lanes, the overlap reaching across steps.
    E = max(1, a.in_flight)
    NBUF = E + 1
    flows = [torch.empty((B, H, W, 2), dtype=torch.float32, device=dev) for _ in range(NBUF)]
    gdev = dev if a.backend == "nccl" else torch.device("cpu")
    gathered = [torch.empty((world * B, H, W, 2), dtype=torch.float32, device=gdev) for _ in range(NBUF)] if world > 1 else None
    tuning = [kv.split("=") for kv in a.tuning.split(",") if kv]
    eng = T.DenseFlow(device_id=local_rank, max_batch=SUB, algo=algo)
    eng.set_tuning("lanes", a.lanes)
    for k, v in tuning:
        eng.set_tuning(k, int(v))
    # Every tf_calc_* is host-synchronous (its lanes have drained when it returns) and tf_wait likewise: a step's flows are complete
    # when its all-gather is issued, on whatever stream they were produced.  The opposite direction needs an explicit host wait: before
    # a buffer is solved into again the host waits (tf_comm_wait) until the all-gather that read it has really finished.
    p0, p1 = frames.data_ptr(), frames.data_ptr() + B * H * W
    pending = {}                                      # flow buffer -> the all-gather that is reading it
    # The exchange is the library's own: tf_allgather_flows = ncclAllGather on librccl, issued on the engine's communication
    # stream (include/teeflow.h); torch.distributed only carries the 128-byte communicator id.  If RCCL cannot be set up
    # that way (or on the gloo rehearsal backend) the step falls back to torch's all_gather_into_tensor and says so.
    collective = "none (single GPU)"
    lib_comm = False
    if world > 1:
        collective = "torch.distributed.all_gather_into_tensor"
        if a.backend == "nccl" and not a.torch_collective:
            try:
                from tee_optical_flow_amd.distributed import init_engine_comm, torch_id_exchange
                init_engine_comm(eng, rank, world, torch_id_exchange())
                lib_comm = True
                collective = "tf_allgather_flows: ncclAllGather issued by the library on its own stream (librccl over xGMI)"
            except Exception as e:                       # keep the N>1 run alive; the line records which path ran
                collective += f" (library communicator unavailable: {type(e).__name__}: {e})"
        ok = torch.tensor([1 if lib_comm else 0], device=gdev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)       # every rank must take the same path
        if lib_comm and int(ok.item()) == 0:
            lib_comm = False
            collective = "torch.distributed.all_gather_into_tensor (another rank could not join the library communicator)"

    def retire(buf):
        tk = pending.pop(buf, None)
        if tk is not None:
            if lib_comm:
                eng.comm_wait(tk)                        # host-blocking: the buffer is free for the next solve
            else:
                tk.wait()
                if dev.type == "cuda":
                    torch.cuda.current_stream(dev).synchronize()

    def issue_gather(buf):
        if lib_comm:
            pending[buf] = eng.allgather(flows[buf].data_ptr(), flows[buf].numel(), gathered[buf].data_ptr())
        else:
            src = flows[buf] if a.backend == "nccl" else flows[buf].cpu()
            pending[buf] = dist.all_gather_into_tensor(gathered[buf], src, async_op=True)

    def submit(k, buf):
        if E == 1:
            return eng.calc_pairs_device(p0, p1, B, H, W, flows[buf].data_ptr())      # one call of the boundary = one step
        return eng.submit_pairs_device(p0, p1, B, H, W, flows[buf].data_ptr())

    def collect(hd):
        return hd if E == 1 else eng.wait(hd)

    def run_steps(first, n):
        return run_steps_pipelined(first, n, E, NBUF, submit, collect, issue_gather if world > 1 else None, retire if world > 1 else None)

    def drain():
        for buf in list(pending):
            retire(buf)

    def fence():
        drain()
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    eng.calc_pairs_device(p0, p1, B, H, W, flows[0].data_ptr())          # set-up, not a step: every lane allocates its buffers
    run_steps(0, warmup)
    fence()
    units0 = eng.counter("queue_units_done")
    t0 = time.perf_counter()
    acc = {"iter_ms": 0.0, "iter_bytes": 0.0, "iter_launches": 0, "total_bytes": 0.0, "inner": 0, "outer": 0, "ms_device": 0.0,
           "timed_iter_bytes": 0.0, "sor_px": 0.0}
    for st in run_steps(warmup, steps):
        acc["total_bytes"] += st["total_bytes"]; acc["inner"] += st["inner_iters_total"]; acc["outer"] += st["outer_iters_total"]
        acc["ms_device"] += st["ms_device"]; acc["timed_iter_bytes"] += st["iter_bytes"]
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=gdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    last_buf = (warmup + steps - 1) % NBUF
    last_flow = flows[last_buf]
    queue_lanes = eng.counter("queue_lanes")
    units_per_step = (eng.counter("queue_units_done") - units0) / max(steps, 1) if (B > SUB or E > 1) else 1     # what the library cut a step into
    lane_streams = {"GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES", "unset (HIP's default: 4)"), "streams_tried_and_dropped": eng.counter("stream_retries"),
                    "lanes_serialised_on_a_shared_hardware_queue": bool(eng.counter("streams_serialised") & 1),
                    "copy_stream_shares_a_hardware_queue_with_a_solve_stream": bool(eng.counter("streams_serialised") & 2)}
    gather_ok = None
    if world > 1:
        # every rank's shard must have arrived intact everywhere: compare checksums of the gathered segments with the
        # checksums the owning ranks computed locally (last timed step)
        bits = lambda t: t.contiguous().view(torch.int32).to(torch.int64)        # exact, order-independent: sum of the bit patterns
        mine = bits(last_flow).sum().reshape(1).to(gdev)
        sums = torch.empty(world, dtype=torch.int64, device=gdev)
        dist.all_gather_into_tensor(sums, mine)
        seg = bits(gathered[last_buf]).view(world, -1).sum(1)
        gather_ok = bool(torch.equal(seg.cpu(), sums.cpu()))
    # The same pairs in the two other forms, for the record beside `value` (N = 1): (i) SUB-pair calls, each split over `--lanes`
    # contiguous parts and joined at its end -- what rounds 1-3 timed; (ii) SUB-pair jobs submitted without waiting, three in flight.
    npx = H * W
    subs = [(c, min(SUB, B - c)) for c in range(0, B, SUB)]
    joined = inflight = None
    if world == 1 and B > SUB and not a.steps_only:
        def sub_call(c, nb, buf, fn):
            return fn(p0 + c * npx, p1 + c * npx, nb, H, W, flows[buf].data_ptr() + c * npx * 8)
        for c, nb in subs:
            sub_call(c, nb, 0, eng.calc_pairs_device)
        torch.cuda.synchronize(dev)
        tj = time.perf_counter()
        for k in range(steps):
            for c, nb in subs:
                sub_call(c, nb, k & 1, eng.calc_pairs_device)
        torch.cuda.synchronize(dev)
        dj = time.perf_counter() - tj
        joined = {"value": B * steps / dj, "unit": "frame-pairs/s", "ms_per_call": dj / (steps * len(subs)) * 1e3, "pairs_per_call": SUB, "lanes": a.lanes,
                  "note": f"one synchronous call per {SUB} pairs, each split over {a.lanes} contiguous parts and joined at its end (the form rounds 1-3 timed as a step)"}
        tk = []
        tj = time.perf_counter()
        for k in range(steps):
            for c, nb in subs:
                if len(tk) == 3:
                    eng.wait(tk.pop(0))
                tk.append(sub_call(c, nb, k & 1, eng.submit_pairs_device))
        while tk:
            eng.wait(tk.pop(0))
        torch.cuda.synchronize(dev)
        dj = time.perf_counter() - tj
        inflight = {"value": B * steps / dj, "unit": "frame-pairs/s", "pairs_per_job": SUB, "jobs_in_flight": 3,
                    "note": f"tf_submit_pairs_device of {SUB}-pair jobs, three in flight, one host thread (the library's lanes take whole jobs)"}
    # Roofline leg: the SAME pairs again as SUB-pair calls on ONE lane (no queue, no split), with every launch of the dominant kernel
    # bracketed by a HIP event pair on the engine's stream.  Kept out of the timed region above (the event records cost a few % of a
    # step); roofline.timed_regime describes the kernel in the regime the timed region runs in.
    prof = None
    if not a.no_profile:
        eng.set_tuning("lanes", 1)      # one lane: a launch's event-pair time must not include another lane's kernels
        eng.set_profile(1)
        for k in range(steps):
            last_flow = flows[k & 1]
            for c, nb in subs:
                st = eng.calc_pairs_device(p0 + c * npx, p1 + c * npx, nb, H, W, last_flow.data_ptr() + c * npx * 8)
                acc["iter_ms"] += st["iter_ms"]; acc["iter_bytes"] += st["iter_bytes"]; acc["iter_launches"] += st["iter_launches"]
                acc["sor_px"] += st["iter_pair_steps"]
                for kk in ("ms_warp", "ms_median", "ms_misc", "ms_sched", "ms_device"):
                    acc["p_" + kk] = acc.get("p_" + kk, 0.0) + st[kk]
        drain()
        torch.cuda.synchronize(dev)
        if algo == "TVL1":
            prof = launch_profile(eng)
        eng.set_profile(0)
        eng.set_tuning("lanes", a.lanes)

    out = None
    if rank == 0:
        pairs = world * B * steps
        unit_bytes = 60.0 if algo == "TVL1" else 40.0          # bytes tf_stats.iter_bytes charges per px-iteration / px-sweep
        launches = max(acc["iter_launches"], 1)
        avg_launch_ms = acc["iter_ms"] / launches if acc["iter_ms"] > 0 else None
        units_per_launch = acc["iter_bytes"] / unit_bytes / launches           # px-iterations (px-sweeps) per launch, averaged
        kern = TVL1_KERNEL if algo == "TVL1" else DF_KERNEL
        # ---- HBM traffic of the dominant kernel: live PMC passes (--pmc) or the stored passes of THIS build --------
        traffic = None
        traffic_source = None
        ffac, ffac_src = fetch_factor(algo)
        if live_pmc and "fetch_kb" in live_pmc and "write_kb" in live_pmc:
            traffic = (ffac * live_pmc["fetch_kb"] + live_pmc["write_kb"]) * 1024.0
            traffic_source = {"kind": "live", "command": live_pmc.get("command"), "launches_profiled": live_pmc.get("launches"),
                              "fetch_size_kb_mean": live_pmc["fetch_kb"], "write_size_kb_mean": live_pmc["write_kb"],
                              "fetch_size_factor": ffac, "fetch_size_factor_source": ffac_src}
        else:
            rec, why = stored_record("hbm_traffic.json", kern)
            if rec:
                traffic = rec["bytes_per_launch"]
                traffic_source = {"kind": "stored", "file": "profiles/hbm_traffic.json", "round": rec.get("round"),
                                  "source_fingerprint": rec.get("source_fingerprint"), "command": rec.get("command"),
                                  "launches_profiled": rec.get("launches_profiled"), "note": "PMC passes of this same kernel build, not of this run"}
            else:
                traffic_source = {"kind": "none", "why": (live_pmc or {}).get("error") or why}
        # compulsory bytes of the kernel that is launched: tvl1_iter 9 plane reads + 6 writes once per TWO iterations = 30 B per
        # px-iteration; SOR 8 plane reads + 2 writes per launch of `sor_fuse` sweeps = 40 B per pixel and launch (tf_stats.iter_bytes
        # charges 40 B per px-SWEEP, so the per-launch figure is iter_bytes / sweeps per launch)
        comp_per_launch = units_per_launch * 30.0 if algo == "TVL1" else acc["sor_px"] * 40.0 / launches
        secs = avg_launch_ms / 1e3 if avg_launch_ms else None
        if traffic is not None and secs:
            achieved, basis = traffic / 1e9 / secs, (f"measured HBM bytes per launch (PMC: {ffac:g} x FETCH_SIZE + WRITE_SIZE; factor from {ffac_src}) / mean launch time")
        elif secs:
            achieved, basis = comp_per_launch / 1e9 / secs, ("no PMC pass of this build: compulsory bytes of the launched kernel ("
                                                             + ("30 B per px-iteration" if algo == "TVL1" else "40 B per pixel and launch") + ") / mean launch time")
        else:
            achieved, basis = None, "profiling off"
        limiter, limiter_evidence = classify_limiter(kern, achieved / HBM_STREAM_GBS if achieved else None)
        roof = {"bound": "hbm", "limiter": limiter, "limiter_evidence": limiter_evidence,
                "kernel": kern + (" (tvl1_iter: two inner iterations per launch, full-width row strips)" if algo == "TVL1" else
                                  " and k_df_sor_rt_coop (red-black SOR on 128x64 regions held in registers: all 25 sweeps of a fixed-point iteration per launch -- "
                                  "co-resident regions that trade du, dv every 5 sweeps, or one region per pair on the small levels)"),
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS if achieved else None,
                "frac_of_streaming_ceiling": achieved / HBM_STREAM_GBS if achieved else None,
                "traffic": traffic, "traffic_source": traffic_source, "achieved_basis": basis,
                "avg_launch_ms": avg_launch_ms, "launches": acc["iter_launches"], "launches_per_step": acc["iter_launches"] / max(steps, 1),
                ("px_iterations_per_launch" if algo == "TVL1" else "px_sweeps_per_launch"): units_per_launch,
                "measured_on": f"{steps} instrumented repeats of the timed steps' pairs as single-lane {SUB}-pair calls (one HIP event pair per launch, engine stream; the GPU to itself)"}
        rate = units_per_launch / secs if secs else None
        roof["algorithmic_bytes_per_launch"] = comp_per_launch
        roof["algorithmic_GBps"] = comp_per_launch / 1e9 / secs if secs else None
        roof["algorithmic_frac"] = comp_per_launch / 1e9 / secs / HBM_PEAK_GBS if secs else None
        roof["traffic_over_algorithmic"] = traffic / comp_per_launch if traffic and comp_per_launch else None
        if algo != "TVL1":
            # With all sweeps of a fixed-point iteration in one launch the kernel reads its 8 planes once per 25 sweeps: ~25 x 56 flop per
            # 40 B = 35 flop/B, above the ~20 flop/B ridge of the (non-matrix) fp32 vector pipes -- the HBM object above is what the bench
            # contract asks for, the vector ALU is what bounds the launch.  Executed instructions from the stored counter passes of this build.
            valu = {"px_sweeps_per_s": rate, "flop_per_byte_per_launch_approx": 35.0, "fp32_vector_ridge_flop_per_byte_approx": 19.7}
            sqf = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_sq_counters.json")))
            for f in reversed(sqf):
                try:
                    with open(f) as fh:
                        rec = json.load(fh).get(kern)
                except (OSError, ValueError):
                    rec = None
                if rec and rec.get("source_fingerprint") == kernel_source_fingerprint() and "SQ_INSTS_VALU" in rec.get("totals_all_dispatches", {}):
                    # the counter pass profiles whole steps of this same workload (--steps-only): all its dispatches against the
                    # pixel-sweeps of as many steps as its dispatch count says (accepted only if that is a whole number)
                    lps = acc["iter_launches"] / max(steps, 1)
                    nst = rec["dispatches"] / lps if lps else 0.0
                    whole = abs(nst - round(nst)) < 0.02 and round(nst) >= 1
                    wi = rec["totals_all_dispatches"]["SQ_INSTS_VALU"]
                    # the vector-ALU roof: lane-instructions per second against CUs x 4 SIMDs x 16 lanes x clock
                    peak = SIMDS * 16 * a.clock_ghz * 1e9
                    t_all = round(nst) * lps * secs if whole and secs else None
                    valu.update({"peak_lane_instructions_per_s": peak, "clock_ghz_assumed": a.clock_ghz,
                                 "frac_executed": (wi * 64.0 / t_all / peak) if t_all else None,
                                 "frac_useful": (rate * 40.0 / peak) if rate else None,
                                 "frac_note": "executed = SQ_INSTS_VALU x 64 lanes / launch time; useful = 40 instructions per pixel-sweep x pixel-sweeps / launch time"})
                    valu.update({"source": f"profiles/{os.path.basename(f)}", "valu_wave_instructions_all_dispatches": wi,
                                 "dispatches_in_the_counter_pass": rec["dispatches"], "steps_in_the_counter_pass": nst,
                                 "valu_lane_instructions_per_px_sweep": (wi * 64.0 / (round(nst) * lps * units_per_launch)) if whole and units_per_launch else None,
                                 "valu_lane_instructions_per_update_in_the_loop": 40,
                                 "valu_pipe_busy_per_simd": rec.get("derived", {}).get("valu_pipe_busy_per_simd_all_launches"),
                                 "simd_cycles_per_valu_inst_all_launches": rec.get("derived", {}).get("simd_cycles_per_valu_inst_all_launches")})
                    break
            roof["valu"] = valu
            # the classifier's word: HBM is this launch's bound only if it runs near the streaming ceiling
            if limiter and not limiter.startswith("hbm"):
                roof["bound"] = "valu"
                roof["bound_note"] = ("set from the counter-derived limiter: the launch keeps its system in registers for 25 sweeps (~35 flop per compulsory byte), "
                                      "HBM traffic is far from the roof; the hbm figures above stay for the record, roofline.valu holds the fractions of the vector-ALU roof")
        if algo == "TVL1":
            # The regime `value` is measured in (three lanes' kernels sharing the GPU), from a stored kernel trace of THIS build's timed steps
            # (tools/timed_regime.py): the kernel's mean duration while it shares the GPU, the kernels in flight beside it, and its
            # share-adjusted duration -- the figures above are from launches that had the GPU to themselves.
            tr_files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_timed_regime.json")))
            tr = None
            for f in reversed(tr_files):
                try:
                    with open(f) as fh:
                        rec = json.load(fh).get(kern)
                except (OSError, ValueError):
                    rec = None
                if rec and rec.get("source_fingerprint") == kernel_source_fingerprint():
                    tr = dict(rec, source=f"profiles/{os.path.basename(f)}")
                    break
            if tr is not None:
                sad = tr["mean_share_adjusted_duration_us"] * 1e-6
                bytes_launch = traffic if traffic is not None else comp_per_launch
                tr["bytes_per_launch_used"] = bytes_launch
                tr["bytes_basis"] = "measured HBM bytes per launch (PMC)" if traffic is not None else "compulsory bytes per launch (30 B per px-iteration)"
                tr["GBps_at_mean_duration"] = bytes_launch / (tr["mean_duration_us"] * 1e-6) / 1e9
                tr["GBps_share_adjusted"] = bytes_launch / sad / 1e9
                tr["frac_of_peak_share_adjusted"] = bytes_launch / sad / 1e9 / HBM_PEAK_GBS
                tr["compulsory_frac_of_peak_share_adjusted"] = comp_per_launch / sad / 1e9 / HBM_PEAK_GBS
                roof["timed_regime"] = tr
            else:
                roof["timed_regime"] = {"note": "no kernel trace of this build's timed steps under profiles/ (tools/timed_regime.py); the figures above describe launches that had the GPU to themselves"}
            # algorithmic (compulsory) bytes of the kernel that is actually launched: 9 plane reads + 6 writes once per TWO iterations
            roof["algorithmic_GBps_30B"] = rate * 30.0 / 1e9 if rate else None
            roof["algorithmic_bytes_per_launch_30B"] = units_per_launch * 30.0
            roof["notional_GBps_60B"] = rate * 60.0 / 1e9 if rate else None     # one-iteration-per-pass kernel's compulsory traffic
            roof["notional_GBps_88B"] = rate * 88.0 / 1e9 if rate else None     # two-kernel K_A/K_B formulation
            roof["notional_note"] = ("what kernels that make one HBM pass per inner iteration would have had to move at this rate; the shipped kernel "
                                     "makes one pass per TWO iterations, so these exceed its real traffic and are not roofline fractions")
            full = None
            if prof is not None and len(prof[3]):
                lv, wp, it, ms = prof
                m = (lv == 0) & (it == 0) & (ms > 0)             # first launch of a level-0 stage: all B pairs iterate, 2 iterations
                if m.any():
                    full = float(np.median(SUB * H * W * 2.0 / (ms[m] * 1e-3)))      # (the instrumented calls hold SUB pairs each)
            isa, why = stored_record("isa_stats.json", kern)
            valu = {"px_iterations_per_s_all_launches": rate, "px_iterations_per_s_full_launches": full}
            if isa:
                ipp = isa["valu_insts_per_wave_step"] / isa["px_iterations_per_wave_step"]
                valu.update({"valu_insts_per_px_iteration": ipp, "vgprs": isa.get("vgprs"), "waves_per_simd": isa.get("waves_per_simd"),
                             "isa_source": "profiles/isa_stats.json (disassembly of this build's main loop)",
                             "simd_cycles_per_valu_inst_full_launches": (a.clock_ghz * 1e9 * SIMDS / (full * ipp) if full else None),
                             "clock_ghz_assumed": a.clock_ghz})
            else:
                valu["isa_source"] = why
            roof["valu"] = valu
        out = {
            "metric": "frame-pairs/sec @512x512 " + ("DualTVL1" if algo == "TVL1" else "DeepFlow"), "value": pairs / dt, "unit": "frame-pairs/s",
            "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": dt / steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{B} independent {H}x{W} u8 frame pairs per GPU per step (BASELINE configs[1] pair, "
                                   f"{units_per_step:g} sub-batch(es) of at most configs[2]'s per-GPU shard size {SUB}) in ONE call of the boundary, speckle-warp v1 seeds rank*B..; "
                                   + ("DualTVL1 all defaults, lambda 0.15, 5 scales x0.8, 5 warps, eps 0.01, 30x10 iterations, 5x5 median; "
                                      if algo == "TVL1" else
                                      "DeepFlow all defaults (BASELINE configs[3]): sigma 0.6, x0.95 pyramid (60 levels), 5 fixed-point x 25 SOR, omega 1.6; ")
                                   + "inputs resident in HBM; "
                                   + ("RCCL all-gather of (u,v) overlapped with the next step" if world > 1 else "single GPU, no collective"),
                       "pairs_per_gpu_per_step": B, "height": H, "width": W, "parallelism": f"pair-sharded x{world}",
                       "calls_per_step": 1, "boundary_call": ("tf_calc_pairs_device (synchronous)" if E == 1 else f"tf_submit_pairs_device / tf_wait, {E} steps in flight"),
                       "sub_batch_capacity_pairs": SUB, "sub_batches_per_step": units_per_step, "pairs_per_sub_batch": B / units_per_step,
                       # what overlaps inside the library (include/teeflow.h "Sub-batches and lanes"): lanes that take whole sub-batches from a queue
                       "steps_in_flight": E, "library_queue_lanes": queue_lanes if B > SUB or E > 1 else 0, "lane_streams": lane_streams,
                       "lanes_per_sub_batch_call": a.lanes if B <= SUB and E == 1 else (a.lanes if algo != "TVL1" else 1)},
            # data-independent rate of the whole job: executed pixel-iterations (pixel-sweeps) per second of the timed region
            ("px_iterations_per_s" if algo == "TVL1" else "px_sweeps_per_s"): world * acc["timed_iter_bytes"] / unit_bytes / dt,
            "roofline": roof,
            # NOT a bandwidth: the bytes a one-HBM-pass-per-iteration (per-sweep) formulation would move, over the device time.  The shipped
            # kernels make one pass per two iterations / per 25 sweeps, so this exceeds their traffic (and may exceed the HBM peak).
            "notional_whole_solve_GBps_one_pass_per_iteration": acc["total_bytes"] / 1e9 / (acc["ms_device"] / 1e3) if acc["ms_device"] else None,
        }
        if not a.no_profile and algo == "TVL1":
            # where a step's device time goes, from the library's own event pairs (instrumented single-lane repeats)
            out["stage_ms_per_step"] = {"tvl1_iter": acc["iter_ms"] / steps, "warp": acc.get("p_ms_warp", 0.0) / steps,
                                        "median": acc.get("p_ms_median", 0.0) / steps, "device_total": acc.get("p_ms_device", 0.0) / steps,
                                        "note": f"summed over the {len(subs)} single-lane {SUB}-pair calls that make up a step's {B} pairs (exclusive use of the GPU, not the timed regime)"}
        if algo == "TVL1":
            out["executed_inner_iterations_per_pair"] = acc["inner"] / (B * steps)
            out["executed_outer_iterations_per_pair"] = acc["outer"] / (B * steps)
            out["data_dependence_note"] = ("pairs/s depends on how early the synthetic pairs converge (executed inner iterations per pair above, of a "
                                           "possible 7500); px_iterations_per_s does not")
        if algo != "TVL1":
            # the co-resident SOR form: launches since the engine was made, and calls that had to be repeated tiled because a launch gave up waiting
            out["sor_coresident"] = {"launches": eng.counter("coop_launches"), "calls_repeated_tiled": eng.counter("coop_aborts")}
        if joined is not None:
            out["steps_joined"] = joined
            out["jobs_in_flight"] = inflight
        out["collective"] = collective
        if gather_ok is not None:
            out["allgather_checksums_match"] = gather_ok
    if rank == 0 and not a.steps_only:
        # single-pair latency (BASELINE configs[1] as a latency number), outside the timed region
        lat_eng = T.DenseFlow(device_id=local_rank, max_batch=1, algo=algo)
        for k, v in tuning:
            lat_eng.set_tuning(k, int(v))
        f1 = torch.empty((1, H, W, 2), dtype=torch.float32, device=dev)
        for _ in range(2):
            lat_eng.calc_pairs_device(p0, p1, 1, H, W, f1.data_ptr())
        torch.cuda.synchronize(dev)
        nlat = 10 if algo == "TVL1" else 4
        tl = time.perf_counter()
        for _ in range(nlat):
            lat_eng.calc_pairs_device(p0, p1, 1, H, W, f1.data_ptr())
        torch.cuda.synchronize(dev)
        out["latency_ms_single_pair"] = (time.perf_counter() - tl) / nlat * 1e3
        lat_eng.close()
        if algo == "TVL1":
            # the reference's default preprocessing branch, no_saliency=False (calculate_optical_flow.py:559-560, :586): fine-grained saliency maps
            # of a study's worth of frames, CV_32F out.  Wall time includes PCIe both ways (host frames in, host maps out); the device figure is
            # the eight kernels' HIP-event time inside the library (tools/saliency_bench.py, profiles/r05_saliency_kernel_stats.csv)
            from tee_optical_flow_amd.synth import speckle_sequence
            seq = speckle_sequence(3, 65, H, W)
            rgb = np.ascontiguousarray(np.repeat(seq[..., None], 3, axis=3))
            sal_eng = T.DenseFlow(device_id=local_rank, max_batch=1)
            sal_eng.saliency_frames(rgb)
            wall, devt = [], []
            for _ in range(5):
                ts = time.perf_counter()
                sal_eng.saliency_frames(rgb)
                wall.append(time.perf_counter() - ts)
                devt.append(sal_eng.counter("saliency_kernel_us") * 1e-6)
            sal_eng.close()
            bpp = 42.0             # compulsory bytes per pixel of the eight passes (tools/saliency_bench.py)
            dsec = float(np.median(devt))
            out["saliency"] = {"frames": 65, "height": H, "width": W, "map": "float32 in [0,1] (computeSaliency() of opencv-contrib 4.x)",
                               "saliency_ms_per_frame_512" if H == 512 else "saliency_ms_per_frame": float(np.median(wall)) / 65 * 1e3,
                               "frames_per_s_end_to_end": 65 / float(np.median(wall)), "device_us_per_frame": dsec / 65 * 1e6,
                               "compulsory_bytes_per_pixel": bpp, "device_GBps_of_compulsory_bytes": 65 * H * W * bpp / dsec / 1e9 if dsec else None,
                               "frac_of_hbm_peak": 65 * H * W * bpp / dsec / 1e9 / HBM_PEAK_GBS if dsec else None,
                               "note": "end to end = host RGB frames in, host maps out (PCIe both ways, pageable buffers); device = the eight kernels of the study, HIP events"}
        # the same step through the host-pointer entry point (PCIe in and out included) -- reported beside, never as `value`
        eng.calc_pairs(I0s, I1s)                   # result arrays come from the engine's pinned pool: this call fills it
        tp = time.perf_counter()
        eng.calc_pairs(I0s, I1s)
        out["pcie_inclusive_pairs_per_s"] = B / (time.perf_counter() - tp)
        if algo == "TVL1":
            # the same with two calls in flight (tf_submit_pairs): one call's copy-out runs under the next call's solve
            for _ in range(2):                          # the pinned pool then holds two result buffers
                tk = [eng.submit_pairs(I0s, I1s), eng.submit_pairs(I0s, I1s)]
                res = [eng.wait(t) for t in tk]
                del res
            tp = time.perf_counter()
            tk, n_calls = [], 6
            for _ in range(n_calls):
                if len(tk) == 2:
                    eng.wait(tk.pop(0))
                tk.append(eng.submit_pairs(I0s, I1s))
            while tk:
                eng.wait(tk.pop(0))
            out["pcie_inclusive_pairs_per_s_two_calls_in_flight"] = n_calls * B / (time.perf_counter() - tp)
        if world == 1 and not a.no_cpu_baseline:
            n = min(cpu_sample, B)
            cb, ref = cpu_baseline(I0s, I1s, n, algo)
            out["cpu_baseline"] = cb
            got = last_flow[:n].cpu().numpy()
            out["parity_vs_oracle_max_abs_diff_on_cpu_sample"] = max(float(np.abs(got[i] - ref[i]).max()) for i in range(n))
            cvb, cvf = cv2_baseline(I0s, I1s, min(n, 4), algo)
            if cvb is None:
                out["opencv_baseline"] = "cv2 not importable on this box (importlib.util.find_spec('cv2') is None): BASELINE.md tier B2 skipped; parity vs OpenCV stays unpinned"
            else:
                out["opencv_baseline"] = cvb
                if cvf is not None:
                    epe = [float(np.sqrt(((got[i] - cvf[i]) ** 2).sum(-1)).mean()) for i in range(len(cvf))]
                    out["mean_epe_vs_opencv"] = float(np.mean(epe))
                    out["cpu_baseline"] = cvb       # the real thing replaces the restatement as THE baseline
                    out["cpu_baseline_port"] = cb
    eng.close()
    del frames, flows, gathered
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=None, help="frame pairs per GPU per step = per call of the boundary (default 384 DualTVL1, 128 DeepFlow)")
    ap.add_argument("--sub-batch", type=int, default=128, help="pairs per sub-batch = the engine's max_batch (BASELINE configs[2]'s per-GPU shard size)")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=8)
    ap.add_argument("--no-profile", action="store_true", help="do not bracket tvl1_iter launches with HIP events")
    ap.add_argument("--algo", default="TVL1", choices=["TVL1", "deepflow"], help="BASELINE configs[1..2] (TVL1, default) or configs[3] (deepflow) as the main leg")
    ap.add_argument("--no-deepflow", action="store_true", help="skip the DeepFlow leg the default N=1 TVL1 run appends")
    ap.add_argument("--deepflow-batch", type=int, default=128, help="pairs per step of the DeepFlow leg (64 pairs: 577, 128: 601-607, 256: 608 pairs/s on one MI355X)")
    ap.add_argument("--deepflow-steps", type=int, default=4)
    ap.add_argument("--pmc", action="store_true", help="N=1 only: first run the FETCH_SIZE / WRITE_SIZE counter passes of this command as child "
                                                       "processes under rocprofv3 (adds ~1-2 min), so roofline.traffic is measured in this run")
    ap.add_argument("--pmc-dir", default=os.path.join(ROOT, "gpurun_out", "pmc_live"))
    ap.add_argument("--round-tag", default="r05")
    ap.add_argument("--steps-only", action="store_true", help="only the timed steps: no single-pair latency, no PCIe step (what the counter passes profile)")
    ap.add_argument("--in-flight", type=int, default=1, help="1 = every step is one synchronous call (default); E > 1 = steps are submitted with tf_submit_pairs_device, "
                                                             "E in flight (the library's lanes overlap them; one host thread)")
    ap.add_argument("--lanes", type=int, default=2, help="contiguous parts a call of at most one sub-batch is split over (joined at its end); 1 for clean per-kernel profiles")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend; 'gloo' + --share-device rehearses N>1 on one GPU")
    ap.add_argument("--share-device", action="store_true", help="rehearsal only: every rank uses cuda:0")
    ap.add_argument("--torch-collective", action="store_true", help="N>1: all-gather through torch.distributed instead of the library's tf_allgather_flows")
    ap.add_argument("--tuning", default="", help="experiments only: engine knobs as name=value,... (tf_set_tuning); empty = shipped defaults")
    ap.add_argument("--clock-ghz", type=float, default=2.4, help="shader clock assumed for the cycles-per-instruction figure (max clock; DESIGN.md notes 2.32 held under load)")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        # plain `python bench.py --gpus N`: be the launcher.  Nothing in this process has touched the GPU (torch is not even
        # imported yet); the ranks are children, their rank 0 prints the line, we pass it on.
        rc, line = spawn_ranks(a.gpus, sys.argv[1:])
        sys.stdout.write(line)
        sys.stdout.flush()
        raise SystemExit(rc)
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if world > 1:
        # N > 1: the all-gather of step k runs on the engine's communication stream beside step k+1's solve.  With HIP's four hardware
        # queues every queue a created stream can get is already one of the three lanes', and a collective kernel that shares a lane's
        # queue would hold that lane up while it waits for its peers; eight queues give the communication stream (and RCCL's own) room.
        # (Single-GPU runs leave the default: eight queues slow the latency-bound forms)  Before torch touches HIP.
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

    # ---- CPU-only preparation: nothing below this block may start a process ------------------------------------
    if a.batch is None:
        a.batch = 384 if a.algo == "TVL1" else 128
    B, H, W = a.batch, a.size, a.size
    I0s, I1s = make_inputs(range(rank * B, (rank + 1) * B), H, W)          # rank r owns pairs [rB, (r+1)B): no data-path exchange
    if rank == 0 and world == 1 and not a.no_cpu_baseline and not under_profiler():
        from oracle import oracle as O
        err = O.build_o3()                      # the timed baseline build is machine-specific: compile it here, before any GPU call
        if err:
            print(f"bench.py: `make -C oracle o3` failed, the -O2 checker will be timed instead: {err}", file=sys.stderr)
    want_df = a.algo == "TVL1" and world == 1 and not a.no_deepflow
    DB = min(a.deepflow_batch, B)
    live = {}
    if a.pmc and world == 1 and not under_profiler():
        live[a.algo] = pmc_child_passes(a.algo, a.pmc_dir)
        if want_df:
            live["deepflow"] = pmc_child_passes("deepflow", a.pmc_dir)

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback for the hot path")
    if a.share_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(a.backend)

    out = run_leg(a, a.algo, B, a.steps, a.warmup, I0s, I1s, torch, dist, dev, rank, world, local_rank,
                  live_pmc=live.get(a.algo), cpu_sample=a.cpu_sample)
    if want_df:
        df = run_leg(a, "deepflow", DB, a.deepflow_steps, 1, I0s[:DB], I1s[:DB], torch, dist, dev, rank, world, local_rank,
                     live_pmc=live.get("deepflow"), cpu_sample=2)
        if rank == 0:
            df["why"] = "OF_algo='deepflow' is what the reference's CLI runs (calculate_optical_flow.py:735-739); BASELINE configs[3]"
            out["deepflow"] = df
    if rank == 0:
        out["kernel_source_fingerprint"] = kernel_source_fingerprint()
        if live:
            # the counter passes of this run, in the form profiles/hbm_traffic.json keeps (copy it there to have later
            # runs of the SAME build report the traffic without --pmc)
            rec = {}
            for algo, lp in live.items():
                if "fetch_kb" in lp and "write_kb" in lp:
                    rec[TVL1_KERNEL if algo == "TVL1" else DF_KERNEL] = {
                        "bytes_per_launch": (fetch_factor(algo)[0] * lp["fetch_kb"] + lp["write_kb"]) * 1024.0, "fetch_size_kb_mean": lp["fetch_kb"],
                        "fetch_size_factor": fetch_factor(algo)[0], "fetch_size_factor_source": fetch_factor(algo)[1],
                        "write_size_kb_mean": lp["write_kb"], "launches_profiled": lp.get("launches"), "command": lp.get("command"),
                        "formula": "(fetch_size_factor*FETCH_SIZE + WRITE_SIZE)*1024 per launch, averaged over every launch of the kernel in a 1-step "
                                   "single-lane run; the factor is calibrated per load width (tools/calibrate_fetch.sh); one --pmc pass per counter",
                        "source_fingerprint": out["kernel_source_fingerprint"], "round": a.round_tag}
            if rec:
                os.makedirs(a.pmc_dir, exist_ok=True)
                with open(os.path.join(a.pmc_dir, "hbm_traffic.json"), "w") as f:
                    json.dump(rec, f, indent=1)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()

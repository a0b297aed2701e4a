"""`DenseFlow`: the object that takes the place of cv2's DenseOpticalFlow on the reference's hot path.

Reference call sites it is a drop-in for (/root/reference/optical_flow/calculate_optical_flow.py):
  :577  OF_model = cv2.optflow.createOptFlow_DualTVL1()        -> createOptFlow_DualTVL1()
  :575  OF_model = cv2.cuda.OpticalFlowDual_TVL1.create()      -> createOptFlow_DualTVL1()
  :578  OF_model.setLambda(config.lambda_value)                -> DenseFlow.setLambda
  :631/:638/:642  flow = OF_model.calc(I0, I1, None)           -> DenseFlow.calc

Same protocol: `calc(I0: uint8[H,W], I1: uint8[H,W], None) -> float32[H,W,2]`, the 12 cv2 getters/setters,
not re-entrant, result owned by the caller.  Everything is computed by hand-written HIP kernels through the
C ABI in include/teeflow.h; there is no CPU path.
"""
import ctypes as C
import weakref

import numpy as np

from . import _lib
from .exceptions import OpticalFlowCalculationError


def _u8_image_stack(a, name, ndim, allow_f32=False):
    a = np.asarray(a)
    if a.dtype != np.uint8 and not (allow_f32 and a.dtype == np.float32):
        raise OpticalFlowCalculationError(f"{name} must be uint8 (CV_8UC1){' or float32 (CV_32FC1)' if allow_f32 else ''}, got {a.dtype}")
    if a.ndim != ndim:
        raise OpticalFlowCalculationError(f"{name} must have {ndim} dimensions, got shape {a.shape}")
    return np.ascontiguousarray(a)


class _PinnedPool:
    """Result arrays backed by pinned host memory (tf_host_alloc): the library then copies flows out at PCIe speed while
    the next sub-batch is still being solved.  A buffer returns to the pool when the numpy array that owns it is garbage
    collected (the caller still gets a fresh array per call, like cv2), so a steady stream of equally sized calls
    allocates nothing."""

    def __init__(self, L, keep_bytes=2 << 30):
        self._L, self._free, self._kept, self._keep = L, {}, 0, keep_bytes
        self.closed = False

    def empty(self, shape, dtype):
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        if n == 0:
            return np.empty(shape, dtype)
        lst = self._free.get(n)
        if lst:
            ptr = lst.pop()
            self._kept -= n
        else:
            ptr = self._L.tf_host_alloc(n)
            if not ptr:
                return np.empty(shape, dtype)            # pinning refused (limits): pageable memory still works
        raw = (C.c_char * n).from_address(ptr)
        weakref.finalize(raw, self._give_back, ptr, n).atexit = False    # at interpreter exit the process frees it
        return np.frombuffer(raw, dtype=dtype).reshape(shape)

    def _give_back(self, ptr, n):
        if self.closed or self._kept + n > self._keep:
            self._L.tf_host_free(ptr)
        else:
            self._free.setdefault(n, []).append(ptr)
            self._kept += n

    def close(self):
        self.closed = True
        for lst in self._free.values():
            for ptr in lst:
                self._L.tf_host_free(ptr)
        self._free, self._kept = {}, 0


class DenseFlow:
    """MI355X DualTVL1 solver with the cv2.DenseOpticalFlow calling convention."""

    device_unit_scale = True        # calc_study(scale=, pad_last=) applies the unit scale in the output kernel (pipeline.flow_for_study)

    _SETTERS = {"Tau": "tau", "Lambda": "lambda", "Theta": "theta", "ScalesNumber": "nscales",
                "WarpingsNumber": "warps", "Epsilon": "epsilon", "InnerIterations": "inner_iterations",
                "OuterIterations": "outer_iterations", "ScaleStep": "scale_step", "Gamma": "gamma",
                "MedianFiltering": "median_filtering", "UseInitialFlow": "use_initial_flow"}

    def __init__(self, device_id=0, max_batch=128, algo="TVL1", **params):
        self._L = _lib.load()
        self._pool = _PinnedPool(self._L)
        self._jobs = {}                                    # ticket -> what a submitted job reads and writes (kept alive until wait)
        self.algo = algo
        if algo == "deepflow":
            dp = _lib.TfDeepflowParams()
            _lib.check(self._L.tf_default_deepflow_params(C.byref(dp)), None, "tf_default_deepflow_params")
            dp.max_batch = int(max_batch)
            for k, v in params.items():
                if not hasattr(dp, k):
                    raise OpticalFlowCalculationError(f"unknown DeepFlow parameter {k!r}")
                setattr(dp, k, v)
            h = C.c_void_p()
            _lib.check(self._L.tf_create_deepflow(C.byref(dp), int(device_id), C.byref(h)), None, "tf_create_deepflow")
            self._h = h
            self.device_id = int(device_id)
            self.last_stats = None
            return
        if algo != "TVL1":
            raise OpticalFlowCalculationError("OF_algo only supports deepflow or TVL1")
        p = _lib.TfParams()
        _lib.check(self._L.tf_default_params(C.byref(p)), None, "tf_default_params")
        p.max_batch = int(max_batch)
        variant = params.pop("variant", "cpu")
        if variant not in ("cpu", "cuda", 0, 1):
            raise OpticalFlowCalculationError(f"variant must be 'cpu' or 'cuda', got {variant!r}")
        p.variant = 1 if variant in ("cuda", 1) else 0          # TF_VARIANT_CUDA: cv2.cuda.OpticalFlowDual_TVL1 semantics (row a5)
        for k, v in params.items():
            key = "lambda_" if k in ("lambda", "lambda_") else k
            if not hasattr(p, key):
                raise OpticalFlowCalculationError(f"unknown DualTVL1 parameter {k!r}")
            setattr(p, key, v)
        h = C.c_void_p()
        _lib.check(self._L.tf_create(C.byref(p), int(device_id), C.byref(h)), None, "tf_create")
        self._h = h
        self.device_id = int(device_id)
        self.last_stats = None

    # ---- lifetime ------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_pool", None):
            self._pool.close()
        if getattr(self, "_h", None):
            self._L.tf_destroy(self._h)                    # jobs still queued are finished first
            self._h = None
        self._jobs = {}

    def _out(self, shape):
        """A fresh float32 result array (pinned host memory from the pool)."""
        return self._pool.empty(shape, np.float32)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- the 12 cv2 setters/getters (setLambda is the one the reference calls, :578) -------------
    def _set(self, key, value):
        _lib.check(self._L.tf_set_param(self._h, _lib.PARAM_KEYS[key], float(value)), self._h, f"set {key}")

    def _get(self, key):
        v = C.c_double()
        _lib.check(self._L.tf_get_param(self._h, _lib.PARAM_KEYS[key], C.byref(v)), self._h, f"get {key}")
        return v.value

    def __getattr__(self, name):
        if name.startswith(("set", "get")) and name[3:] in DenseFlow._SETTERS:
            key = DenseFlow._SETTERS[name[3:]]
            is_int = key in ("nscales", "warps", "inner_iterations", "outer_iterations", "median_filtering")
            if name.startswith("set"):
                return lambda v: self._set(key, v)
            if key == "use_initial_flow":
                return lambda: bool(self._get(key))
            return (lambda: int(self._get(key))) if is_int else (lambda: self._get(key))
        raise AttributeError(name)

    # ---- engine controls -----------------------------------------------------------------------
    def set_stream(self, hip_stream_ptr, external=True):
        """Run on a caller's hipStream_t (e.g. torch.cuda.current_stream().cuda_stream, 0 = legacy default stream);
        external=False returns to the handle's own non-blocking stream."""
        _lib.check(self._L.tf_set_stream(self._h, C.c_void_p(hip_stream_ptr or 0), 1 if external else 0), self._h, "tf_set_stream")

    def set_profile(self, level):
        _lib.check(self._L.tf_set_profile(self._h, int(level)), self._h, "tf_set_profile")

    def set_tuning(self, name, value):
        """Implementation knobs (never change results): iter_variant, strip_blocks, probe_cadence."""
        _lib.check(self._L.tf_set_tuning(self._h, name.encode(), int(value)), self._h, "tf_set_tuning")

    def counter(self, name):
        """Debug counters of the engine (tf_dbg_counter): coop_launches, coop_aborts, coop_disabled, coop_rearms, coop_cooldown, coop_occ16, coop_occ8,
        queue_jobs, queue_units_done, queue_units_skipped, queue_outstanding, queue_lanes, experimental."""
        return int(self._L.tf_dbg_counter(self._h, name.encode()))

    def _finish(self, st):
        self.last_stats = st.as_dict()

    def last_iters(self):
        """Executed (inner, outer) iteration counts of the last call: int32 [pairs, levels, warps, 2]."""
        s = self.last_stats
        n = s["n_pairs"] * s["nscales_used"] * s["warps"] * 2
        out = np.zeros(n, np.int32)
        w = C.c_size_t()
        _lib.check(self._L.tf_get_iters(self._h, out.ctypes.data_as(C.c_void_p), n, C.byref(w)), self._h, "tf_get_iters")
        return out.reshape(s["n_pairs"], s["nscales_used"], s["warps"], 2)

    # ---- cv2 protocol --------------------------------------------------------------------------
    def calc(self, I0, I1, flow=None):
        """flow = OF_model.calc(I0, I1, None): uint8 [H,W] x2 -> float32 [H,W,2] (x, y displacement).  float32 frames
        (CV_32FC1) are accepted like cv2 does: DualTVL1 scales values in [0,1] by 255, DeepFlow takes them as they are."""
        I0 = _u8_image_stack(I0, "I0", 2, allow_f32=True)
        I1 = _u8_image_stack(I1, "I1", 2, allow_f32=True)
        if I0.shape != I1.shape or I0.dtype != I1.dtype:
            raise OpticalFlowCalculationError(f"I0 and I1 differ: {I0.shape} {I0.dtype} vs {I1.shape} {I1.dtype}")
        H, W = I0.shape
        out = self._out((H, W, 2))
        st = _lib.TfStats()
        fn = self._L.tf_calc_pair_f32 if I0.dtype == np.float32 else self._L.tf_calc_pair
        _lib.check(fn(self._h, I0.ctypes.data, I1.ctypes.data, H, W, out.ctypes.data, C.byref(st)), self._h, "tf_calc_pair")
        self._finish(st)
        return out

    # ---- batched forms (the data-parallel path the reference lacks) ------------------------------
    def calc_batch(self, frames, scale=1.0):
        """frames uint8 [N,H,W] -> float32 [N-1,H,W,2]: flow(frame i -> i+1) * scale (reference loop :584-597)."""
        frames = _u8_image_stack(frames, "frames", 3)
        N, H, W = frames.shape
        if N < 2:
            raise OpticalFlowCalculationError("need at least 2 frames")
        out = self._out((N - 1, H, W, 2))
        st = _lib.TfStats()
        _lib.check(self._L.tf_calc_seq(self._h, frames.ctypes.data, N, H, W, float(scale), out.ctypes.data, C.byref(st)),
                   self._h, "tf_calc_seq")
        self._finish(st)
        return out

    def condition_frames(self, nparr):
        """Device version of frames.condition_frames: uint8 [N,H,W,3] -> uint8 [N,H,W] = img2uint8(rgb2gray(frame)) per frame."""
        nparr = _u8_image_stack(nparr, "nparr", 4)
        if nparr.shape[3] != 3:
            raise OpticalFlowCalculationError(f"nparr must be [N,H,W,3], got {nparr.shape}")
        N, H, W, _ = nparr.shape
        out = np.empty((N, H, W), np.uint8)
        _lib.check(self._L.tf_condition_frames(self._h, nparr.ctypes.data, N, H, W, out.ctypes.data), self._h, "tf_condition_frames")
        return out

    def calc_study(self, nparr, scale=1.0, pad_last=False):
        """RGB study uint8 [N,H,W,3] -> float32 [N-1,H,W,2]: conditioning and all pair solves stay on the device.
        `pad_last`: the result is [N,H,W,2] with the last flow repeated (reference :599) -- written into one pinned buffer, no
        concatenate on the host; with `scale` = pixel_spacing * frame_rate this is the study's whole flow array (:600)."""
        nparr = _u8_image_stack(nparr, "nparr", 4)
        if nparr.shape[3] != 3 or nparr.shape[0] < 2:
            raise OpticalFlowCalculationError(f"nparr must be [N>=2,H,W,3], got {nparr.shape}")
        N, H, W, _ = nparr.shape
        out = self._out((N if pad_last else N - 1, H, W, 2))
        st = _lib.TfStats()
        _lib.check(self._L.tf_calc_seq_rgb(self._h, nparr.ctypes.data, N, H, W, float(scale), out.ctypes.data, C.byref(st)),
                   self._h, "tf_calc_seq_rgb")
        self._finish(st)
        if pad_last:
            out[N - 1] = out[N - 2]
        return out

    def saliency_frames(self, nparr, dtype=np.float32):
        """cv2.saliency.StaticSaliencyFineGrained_create().computeSaliency(frame)[1] for every frame, on the device (reference
        calculate_optical_flow.py:559-560, :586): uint8 [N,H,W,3] or [N,H,W] -> [N,H,W].  dtype float32 (default): the CV_32F map in
        [0,1] that opencv-contrib 4.x returns; uint8: the algorithm's 8-bit map (opencv-contrib 3.x).  Three-channel frames meet
        OpenCV's BGR2GRAY in the order given, as the reference's RGB frames do."""
        nparr = np.ascontiguousarray(nparr)
        if nparr.ndim == 3:
            nparr = _u8_image_stack(nparr, "nparr", 3)
            ch = 1
        else:
            nparr = _u8_image_stack(nparr, "nparr", 4)
            ch = nparr.shape[3]
            if ch not in (1, 3):
                raise OpticalFlowCalculationError(f"nparr must be [N,H,W], [N,H,W,1] or [N,H,W,3], got {nparr.shape}")
        N, H, W = nparr.shape[:3]
        f32 = self._map_is_f32(dtype)
        out = np.empty((N, H, W), np.float32 if f32 else np.uint8)
        fn = self._L.tf_saliency_frames_f32 if f32 else self._L.tf_saliency_frames
        _lib.check(fn(self._h, nparr.ctypes.data, N, H, W, ch, out.ctypes.data), self._h, "tf_saliency_frames")
        return out

    @staticmethod
    def _map_is_f32(dtype):
        if dtype in ("f32", "float32") or (not isinstance(dtype, str) and np.dtype(dtype) == np.float32):
            return True
        if dtype in ("u8", "uint8") or (not isinstance(dtype, str) and np.dtype(dtype) == np.uint8):
            return False
        raise OpticalFlowCalculationError(f"saliency map dtype must be float32 ('f32') or uint8 ('u8'), got {dtype!r}")

    def calc_study_saliency(self, nparr, scale=1.0, pad_last=False, map_dtype="f32"):
        """RGB study uint8 [N,H,W,3] -> float32 [N-1,H,W,2] with the saliency maps as the solver's frames (no_saliency=False).
        map_dtype "f32" (default): the solver receives computeSaliency()'s CV_32F maps in [0,1], as under opencv-contrib >= 4.5 -- DualTVL1
        multiplies them by 255 in float, DeepFlow takes them as they are; "u8": the 8-bit maps.  `scale` / `pad_last` as in calc_study."""
        nparr = _u8_image_stack(nparr, "nparr", 4)
        if nparr.shape[3] not in (1, 3) or nparr.shape[0] < 2:
            raise OpticalFlowCalculationError(f"nparr must be [N>=2,H,W,3], got {nparr.shape}")
        N, H, W, ch = nparr.shape
        out = self._out((N if pad_last else N - 1, H, W, 2))
        st = _lib.TfStats()
        fn = self._L.tf_calc_seq_saliency_f32 if self._map_is_f32(map_dtype) else self._L.tf_calc_seq_saliency
        _lib.check(fn(self._h, nparr.ctypes.data, N, H, W, ch, float(scale), out.ctypes.data, C.byref(st)), self._h, "tf_calc_seq_saliency")
        self._finish(st)
        if pad_last:
            out[N - 1] = out[N - 2]
        return out

    def wase_compensate(self, flows, bkgd_mask, scale=1.0):
        """Reference :647-652, 659 for every flow of a study at once, on the device: returns (flows - background[p]) * scale
        and the float32 backgrounds.  flows float32 [P,H,W,2]; bkgd_mask bool [N,H,W,2] (mask_dict['bkgd'])."""
        flows = np.array(flows, dtype=np.float32, order="C", copy=True)
        mask = np.ascontiguousarray(bkgd_mask)
        if flows.ndim != 4 or flows.shape[3] != 2:
            raise OpticalFlowCalculationError(f"flows must be [P,H,W,2], got {flows.shape}")
        if mask.dtype != np.bool_ or mask.ndim != 4 or mask.shape[1:] != flows.shape[1:]:
            raise OpticalFlowCalculationError(f"bkgd mask must be bool [N,{flows.shape[1]},{flows.shape[2]},2], got {mask.dtype} {mask.shape}")
        P, H, W, _ = flows.shape
        bg = np.empty(P, np.float32)
        _lib.check(self._L.tf_wase_compensate(self._h, flows.ctypes.data, P, mask.view(np.uint8).ctypes.data, mask.shape[0], H, W,
                                              float(scale), bg.ctypes.data), self._h, "tf_wase_compensate")
        return flows, bg

    def calc_pairs(self, I0s, I1s):
        """B independent pairs: uint8 or float32 [B,H,W] x2 -> float32 [B,H,W,2] (float frames: DualTVL1 scales them by 255 like cv2, DeepFlow
        takes them as they are like cv2)."""
        I0s = _u8_image_stack(I0s, "I0s", 3, allow_f32=True)
        I1s = _u8_image_stack(I1s, "I1s", 3, allow_f32=True)
        if I0s.shape != I1s.shape or I0s.dtype != I1s.dtype:
            raise OpticalFlowCalculationError(f"I0s and I1s differ: {I0s.shape} {I0s.dtype} vs {I1s.shape} {I1s.dtype}")
        B, H, W = I0s.shape
        out = self._out((B, H, W, 2))
        st = _lib.TfStats()
        fn = self._L.tf_calc_pairs_f32 if I0s.dtype == np.float32 else self._L.tf_calc_pairs
        _lib.check(fn(self._h, I0s.ctypes.data, I1s.ctypes.data, B, H, W, out.ctypes.data, C.byref(st)), self._h, "tf_calc_pairs")
        self._finish(st)
        return out

    def calc_pairs_device(self, dI0s_ptr, dI1s_ptr, B, H, W, dflow_ptr, scale=1.0):
        """Device-resident form: raw device pointers (e.g. tensor.data_ptr()); result written to dflow_ptr."""
        st = _lib.TfStats()
        _lib.check(self._L.tf_calc_pairs_device(self._h, C.c_void_p(dI0s_ptr), C.c_void_p(dI1s_ptr), int(B), int(H), int(W),
                                                float(scale), C.c_void_p(dflow_ptr), C.byref(st)), self._h, "tf_calc_pairs_device")
        self._finish(st)
        return self.last_stats

    # ---- jobs in flight: the same calls, queued on the engine's lanes inside the library and collected later ------------------
    def submit_pairs_device(self, dI0s_ptr, dI1s_ptr, B, H, W, dflow_ptr, scale=1.0):
        """tf_submit_pairs_device: queue the batch on the engine's lanes and return a ticket at once; `wait(ticket)` returns its
        statistics.  The device buffers must stay untouched until then.  Consecutive jobs overlap on the GPU (one batch's tail
        under the next one's full launches), which a stream of synchronous calls cannot do."""
        t = C.c_int(-1)
        _lib.check(self._L.tf_submit_pairs_device(self._h, C.c_void_p(dI0s_ptr), C.c_void_p(dI1s_ptr), int(B), int(H), int(W), float(scale),
                                                  C.c_void_p(dflow_ptr), C.byref(t)), self._h, "tf_submit_pairs_device")
        self._jobs[t.value] = None
        return t.value

    def submit_batch(self, frames, scale=1.0):
        """calc_batch without waiting: frames uint8 [N,H,W] -> ticket; `wait(ticket)` returns float32 [N-1,H,W,2]."""
        frames = _u8_image_stack(frames, "frames", 3)
        N, H, W = frames.shape
        if N < 2:
            raise OpticalFlowCalculationError("need at least 2 frames")
        out = self._out((N - 1, H, W, 2))
        t = C.c_int(-1)
        _lib.check(self._L.tf_submit_seq(self._h, frames.ctypes.data, N, H, W, float(scale), out.ctypes.data, C.byref(t)), self._h, "tf_submit_seq")
        self._jobs[t.value] = (out, frames)                 # the library reads `frames` and writes `out` until the wait
        return t.value

    def submit_study(self, nparr, scale=1.0, pad_last=False):
        """calc_study without waiting: RGB study uint8 [N,H,W,3] -> ticket; `wait(ticket)` returns float32 [N-1,H,W,2] (or [N,H,W,2] with
        the last flow repeated, `pad_last`).  The frames are conditioned on the device before this returns (nparr may be reused at once)."""
        nparr = _u8_image_stack(nparr, "nparr", 4)
        if nparr.shape[3] != 3 or nparr.shape[0] < 2:
            raise OpticalFlowCalculationError(f"nparr must be [N>=2,H,W,3], got {nparr.shape}")
        N, H, W, _ = nparr.shape
        out = self._out((N if pad_last else N - 1, H, W, 2))
        t = C.c_int(-1)
        _lib.check(self._L.tf_submit_seq_rgb(self._h, nparr.ctypes.data, N, H, W, float(scale), out.ctypes.data, C.byref(t)), self._h, "tf_submit_seq_rgb")
        self._jobs[t.value] = (out, ("pad_last", N) if pad_last else None)
        return t.value

    def submit_pairs(self, I0s, I1s):
        """calc_pairs without waiting: uint8 [B,H,W] x2 -> ticket; `wait(ticket)` returns float32 [B,H,W,2]."""
        I0s = _u8_image_stack(I0s, "I0s", 3)
        I1s = _u8_image_stack(I1s, "I1s", 3)
        if I0s.shape != I1s.shape:
            raise OpticalFlowCalculationError(f"I0s and I1s differ: {I0s.shape} vs {I1s.shape}")
        B, H, W = I0s.shape
        out = self._out((B, H, W, 2))
        t = C.c_int(-1)
        _lib.check(self._L.tf_submit_pairs(self._h, I0s.ctypes.data, I1s.ctypes.data, B, H, W, out.ctypes.data, C.byref(t)), self._h, "tf_submit_pairs")
        self._jobs[t.value] = (out, I0s, I1s)
        return t.value

    def wait(self, ticket):
        """Collect a submitted job: its flows (host forms) or its statistics (device form).  `last_stats` / `last_iters()` then
        describe that job.  A failed job raises here."""
        if ticket not in self._jobs:
            raise OpticalFlowCalculationError(f"unknown ticket {ticket!r} (already waited for?)")
        keep = self._jobs.pop(ticket)
        st = _lib.TfStats()
        _lib.check(self._L.tf_wait(self._h, int(ticket), C.byref(st)), self._h, "tf_wait")
        self._finish(st)
        if keep is None:
            return self.last_stats
        if len(keep) == 2 and isinstance(keep[1], tuple) and keep[1][0] == "pad_last":
            keep[0][keep[1][1] - 1] = keep[0][keep[1][1] - 2]          # the last flow repeated (reference :599), inside the pinned buffer
        return keep[0]

    def calc_seq_device(self, dframes_ptr, N, H, W, dflow_ptr, scale=1.0):
        st = _lib.TfStats()
        _lib.check(self._L.tf_calc_seq_device(self._h, C.c_void_p(dframes_ptr), int(N), int(H), int(W), float(scale),
                                              C.c_void_p(dflow_ptr), C.byref(st)), self._h, "tf_calc_seq_device")
        self._finish(st)
        return self.last_stats


    # ---- multi-GPU exchange: RCCL all-gather issued by the library itself ------------------
    @staticmethod
    def comm_unique_id():
        """128 opaque bytes from ncclGetUniqueId: rank 0 makes them, the launcher hands them to every rank."""
        L = _lib.load()
        buf = (C.c_ubyte * _lib.COMM_ID_BYTES)()
        _lib.check(L.tf_comm_unique_id(buf), None, "tf_comm_unique_id")
        return bytes(buf)

    def comm_init_rank(self, nranks, rank, id_bytes):
        """Join an `nranks`-rank RCCL communicator as `rank` (one process per GPU)."""
        if len(id_bytes) != _lib.COMM_ID_BYTES:
            raise OpticalFlowCalculationError(f"communicator id must be {_lib.COMM_ID_BYTES} bytes")
        buf = (C.c_ubyte * _lib.COMM_ID_BYTES).from_buffer_copy(id_bytes)
        _lib.check(self._L.tf_comm_init_rank(self._h, int(nranks), int(rank), buf), self._h, "tf_comm_init_rank")
        self.has_comm = True

    def allgather(self, d_send_ptr, count_floats, d_recv_ptr):
        """Enqueue this rank's part of the all-gather of the (u,v) fields (device pointers, `count_floats` per rank);
        returns a ticket for comm_wait.  Returns at once: the exchange runs beside whatever is solved next."""
        t = C.c_int(-1)
        _lib.check(self._L.tf_allgather_flows(self._h, C.c_void_p(d_send_ptr), C.c_size_t(int(count_floats)), C.c_void_p(d_recv_ptr),
                                              C.byref(t)), self._h, "tf_allgather_flows")
        return t.value

    def comm_wait(self, ticket=-1):
        """Block until the all-gather `ticket` (default: every one issued so far) has finished."""
        _lib.check(self._L.tf_comm_wait(self._h, int(ticket)), self._h, "tf_comm_wait")


def createOptFlow_DeepFlow(device_id=0, **kw):
    """Name-compatible factory for cv2.optflow.createOptFlow_DeepFlow() (reference :568)."""
    return DenseFlow(device_id=device_id, algo="deepflow", **kw)


def createOptFlow_DualTVL1(device_id=0, **kw):
    """Name-compatible factory for cv2.optflow.createOptFlow_DualTVL1() (reference :577)."""
    return DenseFlow(device_id=device_id, **kw)


def cuda_OpticalFlowDual_TVL1_create(device_id=0, **kw):
    """Counterpart of cv2.cuda.OpticalFlowDual_TVL1.create() (reference :575): the CUDA branch's semantics -- 300 iterations
    in one loop per warp, no median filtering, error looked at on odd iterations only, weight-normalised Catmull-Rom warp
    with clamp addressing (TF_VARIANT_CUDA).  As on the reference's CUDA branch, nothing is set on it."""
    return DenseFlow(device_id=device_id, variant="cuda", **kw)

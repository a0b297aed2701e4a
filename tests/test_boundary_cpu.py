"""The C-ABI boundary without a GPU: the library loads, exports every symbol include/teeflow.h declares, the ctypes
struct layouts equal the C ones, and the product path fails LOUDLY (never falls back) when no device is present."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "include", "teeflow.h")


def _declared_functions():
    src = open(HDR).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(tf_[a-z_0-9]+)\s*\(", src)))


def test_header_cites_reference_interfaces():
    txt = open(HDR).read()
    for needle in ("calculate_optical_flow.py:577", "calculate_optical_flow.py:578", "calculate_optical_flow.py:642",
                   "calculate_optical_flow.py:584-600"):
        assert needle in txt


def test_library_exports_every_declared_symbol():
    from tee_optical_flow_amd import _lib
    L = _lib.load()
    names = _declared_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), f"{n} declared in teeflow.h but not exported"
    assert sorted(_lib.EXPORTED_SYMBOLS) == names
    assert L.tf_abi_version() == 2


def test_ctypes_struct_layout_equals_c(tmp_path):
    from tee_optical_flow_amd import _lib
    src = tmp_path / "lay.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "teeflow.h"\nint main(){'
                   'printf("%zu %zu %zu %zu %zu %zu\\n", sizeof(tf_params), offsetof(tf_params, nscales), offsetof(tf_params, max_batch),'
                   ' sizeof(tf_stats), offsetof(tf_stats, iter_ms), offsetof(tf_stats, ms_sched)); return 0; }\n')
    exe = tmp_path / "lay"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    vals = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    P, S = _lib.TfParams, _lib.TfStats
    assert vals == [C.sizeof(P), P.nscales.offset, P.max_batch.offset, C.sizeof(S), S.iter_ms.offset, S.ms_sched.offset]


def test_default_params_are_cv2_defaults():
    from tee_optical_flow_amd import _lib
    L = _lib.load()
    p = _lib.TfParams()
    assert L.tf_default_params(C.byref(p)) == 0
    assert (p.tau, p.lambda_, p.theta, p.nscales, p.warps, p.epsilon, p.inner_iterations, p.outer_iterations, p.scale_step,
            p.gamma, p.median_filtering, p.use_initial_flow) == (0.25, 0.15, 0.3, 5, 5, 0.01, 30, 10, 0.8, 0.0, 5, 0)


def test_no_silent_cpu_fallback():
    """Without a GPU the product path must raise, not compute on the CPU."""
    import torch
    import tee_optical_flow_amd as T
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    with pytest.raises(T.OpticalFlowCalculationError, match="no HIP device|no CPU fallback"):
        T.DenseFlow()
    with pytest.raises(T.OpticalFlowCalculationError):
        T.createOptFlow_DualTVL1()


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "tee_optical_flow_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "import oracle" not in txt and "from oracle" not in txt and "libtvl1_oracle" not in txt, f


def test_median_networks_proved_by_zero_one_principle(tmp_path):
    exe = tmp_path / "vmn"
    subprocess.check_call(["g++", "-O2", "-I", os.path.join(ROOT, "tee_optical_flow_amd", "csrc"),
                           os.path.join(ROOT, "tests", "csrc", "verify_median_net.cpp"), "-o", str(exe)])
    assert subprocess.check_output([str(exe)]).decode().strip() == "OK"


def test_hip_sources_have_no_compat_layers():
    for f in ("teeflow.hip", "teeflow_kernels.hip.h"):
        txt = open(os.path.join(ROOT, "tee_optical_flow_amd", "csrc", f)).read()
        for bad in ("__HIP_PLATFORM_AMD__", "cuda_runtime", "__CUDACC__", "triton"):
            assert bad not in txt


def test_input_validation_happens_before_any_gpu_work():
    from tee_optical_flow_amd.dense_flow import _u8_image_stack
    from tee_optical_flow_amd import OpticalFlowCalculationError
    with pytest.raises(OpticalFlowCalculationError):
        _u8_image_stack(np.zeros((4, 4), np.float32), "I0", 2)
    with pytest.raises(OpticalFlowCalculationError):
        _u8_image_stack(np.zeros((4, 4, 3), np.uint8), "I0", 2)
    a = _u8_image_stack(np.zeros((8, 8), np.uint8)[::2, ::2], "I0", 2)
    assert a.flags["C_CONTIGUOUS"]


def test_strip_rule_covers_the_image_and_fills_one_round():
    """Host arithmetic behind the device-side strip sizing of tvl1_iter: strips cover every row and the work items of a
    launch fit the resident-block budget in whole rounds."""
    from tee_optical_flow_amd import _lib
    L = _lib.load()
    R, S = C.c_int(), C.c_int()
    for H, RY in [(512, 2), (410, 2), (328, 3), (262, 3), (210, 4), (17, 1), (5, 8), (1024, 1)]:
        for slots in (256, 768, 2048):
            for n in list(range(1, 130)) + [255, 768, 769, 1024]:
                L.tf_dbg_strip_rule(n, H, RY, slots, C.byref(R), C.byref(S))
                r, s = R.value, S.value
                assert r >= 1 and s >= 1
                assert (s - 1) * r < H <= s * r, (H, RY, slots, n, r, s)          # strips tile the rows, none is empty
                rounds = -(-n // slots)
                assert n * s <= rounds * slots or s == 1, (H, RY, slots, n, r, s)  # never more items than the rounds hold
    L.tf_dbg_strip_rule(64, 512, 2, 768, C.byref(R), C.byref(S))
    assert (R.value, S.value) == (43, 12)
    L.tf_dbg_strip_rule(1, 512, 2, 768, C.byref(R), C.byref(S))
    assert R.value == 8 and S.value == 64                                       # capped: at least 4 steps per strip


def test_comm_entry_points_fail_cleanly_without_a_handle():
    """The RCCL entry points are part of the ABI; without a handle / communicator they return error
    codes instead of touching librccl (no GPU here)."""
    import ctypes as C
    from tee_optical_flow_amd import _lib
    L = _lib.load()
    assert L.tf_comm_wait(None, -1) == 1 and L.tf_comm_destroy(None) == 1           # TF_ERR_INVALID_ARG
    assert L.tf_comm_init_rank(None, 2, 0, None) == 1
    assert L.tf_allgather_flows(None, None, 0, None, None) == 1
    assert L.tf_comm_init_all(None, 0) == 1 and L.tf_comm_unique_id(None) == 1
    assert _lib.COMM_ID_BYTES == 128


def test_binding_refuses_the_cpu_checker(oracle, monkeypatch, tmp_path):
    """the checker exports the product's ABI, so the binding itself must refuse it -- by location (anything
    under an oracle/ directory) and by content (a copy elsewhere still carries orc_* symbols)."""
    import shutil
    import tee_optical_flow_amd as T
    from tee_optical_flow_amd import _lib
    from tests import abi_driver as D
    assert os.path.exists(D.CPU_LIB)
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", D.CPU_LIB)
    with pytest.raises(T.OpticalFlowCalculationError, match="oracle/ directory"):
        _lib.load()
    moved = tmp_path / "libteeflow_hip.so"
    shutil.copy(D.CPU_LIB, moved)
    monkeypatch.setattr(_lib, "LIB_PATH", str(moved))
    with pytest.raises(T.OpticalFlowCalculationError, match="CPU checker"):
        _lib.load()
    link = tmp_path / "link.so"
    os.symlink(D.CPU_LIB, link)
    monkeypatch.setattr(_lib, "LIB_PATH", str(link))
    with pytest.raises(T.OpticalFlowCalculationError, match="oracle/ directory"):
        _lib.load()

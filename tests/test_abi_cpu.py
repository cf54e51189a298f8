"""CPU: the C-ABI library loads here (no GPU) and exports every symbol include/frenet_gpu.h declares;
without a device every entry point fails loudly instead of falling back to a CPU path."""
import ctypes as C
import os
import re
import subprocess

import pytest

from conftest import ROOT
from fiss_plus_planner_amd import _abi


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_abi.LIB_PATH):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "fiss_plus_planner_amd", "csrc"), "-s"])
    return _abi.load()


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "frenet_gpu.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(fp_[a-z_0-9]+)\s*\(", src)))


def test_header_and_binding_agree():
    assert _declared_symbols() == sorted(_abi.EXPORTED_SYMBOLS)


def test_library_exports_every_declared_symbol(lib):
    for sym in _declared_symbols():
        assert hasattr(lib, sym), sym
    assert lib.fp_abi_version() == _abi.FP_ABI_VERSION


def test_struct_layouts_match_header():
    # 4 int32 + 10 double ; 6 int32 + 14 pointers ; 5 pointers
    assert C.sizeof(_abi.FpParams) == 4 * 4 + 10 * 8
    assert C.sizeof(_abi.FpBatch) == 6 * 4 + 15 * 8
    assert C.sizeof(_abi.FpResult) == 7 * 8


def test_no_device_means_loud_failure(lib):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    ctx = C.c_void_p()
    rc = lib.fp_ctx_create(0, C.byref(ctx))
    assert rc == -5 and not ctx
    assert b"no CPU fallback" in lib.fp_last_error()
    from fiss_plus_planner_amd.engine import FrenetEngine

    with pytest.raises(_abi.FrenetGpuError):
        FrenetEngine(0)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "fiss_plus_planner_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in txt.replace("no oracle", ""), f"{f} mentions the oracle"


def test_host_batch_struct_is_cached_per_batch_and_safe_to_edit():
    """engine._host_batch: the ctypes view of a batch is rebuilt only when one of its arrays was replaced, and callers get their
    own copy (editing it must not leak into the next call)."""
    import numpy as np

    from fiss_plus_planner_amd import synth
    from fiss_plus_planner_amd.engine import _host_batch

    b = synth.make_batch(2, 3, 3, 2, 4, 20, False, 5)
    fb1 = _host_batch(b)
    ego_ptr = fb1.ego
    assert ego_ptr == b.ego.ctypes.data and fb1.B == 2 and fb1.n_obs == 4
    fb1.ego = None                      # a caller scribbles on its struct ...
    fb2 = _host_batch(b)
    assert fb2.ego == ego_ptr           # ... the next call still sees the real pointer
    b.ego[0, 0] += 1.0                  # in-place update: same array, same pointer, no rebuild needed
    assert _host_batch(b).ego == ego_ptr
    b.ego = np.ascontiguousarray(b.ego.copy())   # replaced array: new pointer
    fb3 = _host_batch(b)
    assert fb3.ego == b.ego.ctypes.data and fb3.ego != ego_ptr

"""CPU: the C-ABI library loads here (no GPU) and exports every symbol include/frenet_gpu.h declares;
without a device every entry point fails loudly instead of falling back to a CPU path."""
import ctypes as C
import os
import re
import subprocess
import sys

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


def test_struct_layouts_match_header(tmp_path):
    """Every struct of include/frenet_gpu.h against its ctypes mirror: size and the offset of every field, as gcc lays them out
    (a C program built from the header prints them)."""
    pairs = {"fp_params": _abi.FpParams, "fp_batch": _abi.FpBatch, "fp_result": _abi.FpResult, "fp_fiss_opts": _abi.FpFissOpts,
             "fp_fiss_io": _abi.FpFissIo, "fp_loop_io": _abi.FpLoopIo, "fp_copy": _abi.FpCopy, "fp_shard_call": _abi.FpShardCall}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "frenet_gpu.h"', 'int main(void) {']
    for cname, cls in pairs.items():
        lines.append(f'  printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'  printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ['  printf("version %d\\n", FP_ABI_VERSION);', '  return 0;', '}']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)])
    got = dict(l.split() for l in subprocess.check_output([str(exe)], text=True).splitlines())
    assert int(got["version"]) == _abi.FP_ABI_VERSION
    for cname, cls in pairs.items():
        assert int(got[cname]) == C.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert int(got[f"{cname}.{fname}"]) == getattr(cls, fname).offset, f"{cname}.{fname}"
    # the flag bits too
    hdr = open(os.path.join(ROOT, "include", "frenet_gpu.h")).read()
    for name, val in (("SPEED", _abi.FLAG_SPEED), ("ACCEL", _abi.FLAG_ACCEL), ("COLLISION", _abi.FLAG_COLLISION), ("TRUNCATED", _abi.FLAG_TRUNCATED),
                      ("CURVATURE", _abi.FLAG_CURVATURE), ("KAPPA_D", _abi.FLAG_KAPPA_D), ("KAPPA_DD", _abi.FLAG_KAPPA_DD)):
        assert int(re.search(rf"#define FP_FLAG_{name} (\d+)u", hdr).group(1)) == val, name


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


def test_integration_md_stub_matches_the_binding():
    """The ctypes stub INTEGRATION.md shows a maintainer of the reference: its struct declarations are executed and compared with
    the product binding field by field, and its version constant with the header's (round 1 shipped a 5-field FpResult there)."""
    txt = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = txt[txt.index("class FpParams(C.Structure):"):txt.index("_lib.fp_abi_version.restype")]
    ns = {"C": C}
    exec(block, ns)
    assert ns["FP_ABI_VERSION"] == _abi.FP_ABI_VERSION
    for name in ("FpParams", "FpBatch", "FpResult"):
        doc, ours = ns[name], getattr(_abi, name)
        assert [(f[0], C.sizeof(f[1])) for f in doc._fields_] == [(f[0], C.sizeof(f[1])) for f in ours._fields_], name
        assert C.sizeof(doc) == C.sizeof(ours)
    assert "best_traj.ctypes.data, None, None, 0, 0)" in txt  # the call site passes all eleven fields


def test_no_kernel_spills_a_vgpr():
    """Every gfx950 kernel must fit its register budget without spilling a VGPR to scratch: this compiler build places the spill store
    of a value that lives across a divergent loop in the loop's exit block BEFORE the exec mask is restored (nothing is stored; the
    reload returns the scratch slot's old content - found on the run-time-shape three-per-CU lattice instances, whose workgroups of a
    launch's second round then planned with another ego's numbers).  tools/resource_usage.py's table, asserted."""
    out = subprocess.check_output(["python", os.path.join(ROOT, "tools", "resource_usage.py")], text=True)
    rows = [l for l in out.splitlines()[1:] if l.strip()]
    assert len(rows) >= 30
    for l in rows:
        cols = l.split()
        spill_v, scratch = int(cols[-5]), int(cols[-3])   # ... VGPRs Spill, SGPRs Spill, ScratchSize, Occupancy, LDS Size
        assert spill_v == 0, l
        # (scratch WITHOUT a spill is a stack object: fiss_refine_kernel<2> reserves 68 bytes and audit_kernel 20 that no instruction
        # addresses - the optimiser emptied them but did not delete them.  fiss_refine_kernel<4>, the four-points-per-lane instance for
        # trajectories beyond 128 points, keeps its per-lane difference-chain arrays there: ordinary stack accesses under their own exec
        # mask, not spill code - the rare-path instance pays for it in speed, its results are pinned by G13 and the oracle.)
        assert scratch <= 68 or "fiss_refine_kernel<4>" in l, l


def test_production_library_reports_no_diagnostic_macro_and_its_compiler(lib):
    lib.fp_build_flags.restype = C.c_char_p
    lib.fp_build_compiler.restype = C.c_char_p
    assert lib.fp_build_flags() == b""
    assert b"roc-7.2.0" in lib.fp_build_compiler() and lib.fp_build_compiler().startswith(b"22.")


def test_a_diagnostic_build_names_its_macros_and_the_binding_refuses_it(tmp_path):
    """The timing ablations (FP_ABL_*: wrong results by design), the instance switches and the stamp / counter builds compile into the
    same file name under the same ABI version: fp_build_flags() is how a caller tells, and _abi.load() refuses unless told otherwise.
    (Host-only check: one translation unit with the macro on its command line - the whole library takes a minute to build.)"""
    csrc = os.path.join(ROOT, "fiss_plus_planner_amd", "csrc")
    src = open(os.path.join(csrc, "frenet_abi.hip")).read()
    a = src.index("const char* fp_build_flags(void)")
    b = src.index("const char* fp_build_compiler(void)")
    unit = tmp_path / "flags.cpp"
    unit.write_text('#include <string>\nextern "C" {\n' + src[a:b] + "}\n")
    for extra, want in (([], ""), (["-DFP_ABL_NO_N", "-DFP_NO_OCC8"], "-DFP_ABL_NO_N -DFP_NO_OCC8"),
                        (["-DFP_BUILD_EXTRA=\"-DFP_PHASE_STAMPS -DFOO=1\"", "-DFP_PHASE_STAMPS", "-DFP_COUNTERS"], "-DFP_PHASE_STAMPS -DFOO=1 -DFP_COUNTERS")):
        so = tmp_path / f"flags{len(extra)}.so"
        subprocess.check_call(["g++", "-shared", "-fPIC", "-std=c++17", "-o", str(so), str(unit)] + extra)
        L = C.CDLL(str(so))
        L.fp_build_flags.restype = C.c_char_p
        assert L.fp_build_flags().decode() == want
    # the Makefile hands EXTRA to the library, and refuses a toolchain it was not validated on
    mk = open(os.path.join(csrc, "Makefile")).read()
    assert "FP_BUILD_EXTRA" in mk and "VALIDATED_ROCM = 7.2.0" in mk and "$(error" in mk
    assert "__clang_major__ != 22" in src
    # the binding's refusal (no GPU needed: it happens right after dlopen)
    code = ("import ctypes, sys; sys.path.insert(0, %r); from fiss_plus_planner_amd import _abi\n"
            "class Fake:\n"
            "    def __getattr__(self, n):\n"
            "        f = lambda *a: b'-DFP_ABL_NO_N' if n == 'fp_build_flags' else 0\n"
            "        return type('F', (), {'__call__': staticmethod(f), 'restype': None, 'argtypes': None})()\n"
            "real = _abi.C.CDLL\n"
            "_abi.C.CDLL = lambda p, *a, **k: Fake() if p == _abi.LIB_PATH else real(p, *a, **k)\n"
            "try:\n    _abi.load()\nexcept ImportError as e:\n    print('REFUSED', e)\n") % ROOT
    env = {k: v for k, v in os.environ.items() if k != "FP_ALLOW_DIAGNOSTIC_BUILD"}
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert "REFUSED" in out.stdout and "FP_ABL_NO_N" in out.stdout and "DIAGNOSTIC" in out.stdout, (out.stdout, out.stderr[-2000:])

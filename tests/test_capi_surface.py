"""CPU-side checks of the drop-in boundary: libea_hip.so loads, exports every symbol that
include/ea_hip.h declares, and refuses to compute without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols(header=os.path.join("include", "ea_hip.h")):
    text = open(os.path.join(ROOT, header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ea_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    from edge_alignment_amd import build_library, capi
    build_library()
    return capi.load()


def test_header_and_stub_agree():
    from edge_alignment_amd import capi
    assert _declared_symbols() == sorted(capi.EXPORTED)


def test_every_declared_symbol_is_exported(lib):
    for name in _declared_symbols():
        assert hasattr(lib, name), name


def test_measurement_hooks_live_in_their_own_header(lib):
    """the public header declares no ea_*bench* entry point; the internal one declares exactly the hooks the stub binds"""
    from edge_alignment_amd import capi
    assert not [n for n in _declared_symbols() if "bench" in n]
    dev = _declared_symbols(os.path.join("edge_alignment_amd", "csrc", "ea_hip_dev.h"))
    assert dev == sorted(capi.EXPORTED_DEV)
    for name in dev:
        assert hasattr(lib, name), name


def test_struct_layouts_match_the_header(lib):
    from edge_alignment_amd import capi
    # sizes as laid out by the C compiler for include/ea_hip.h
    assert C.sizeof(capi.Camera) == 32
    assert C.sizeof(capi.Options) == 4 + 4 + 9 * 8 + 4 * 4 + 2 * 4 + 8  # + solve_timeout_ms
    assert C.sizeof(capi.Summary) == 5 * 4 + 4 + 2 * 8 + 8 + 8 + 6 * 8 * capi.MAX_TRACE + 4 * capi.MAX_TRACE
    o = capi.default_options()
    assert (o.max_num_iterations, o.function_tolerance, o.gradient_tolerance, o.parameter_tolerance) == (50, 1e-6, 1e-10, 1e-8)
    assert (o.initial_trust_region_radius, o.min_relative_decrease, o.jacobi_scaling) == (1e4, 1e-3, 1)


def test_no_cpu_fallback_without_a_device(lib):
    from edge_alignment_amd import capi
    if capi.device_count() > 0:
        pytest.skip("a gfx950 device is visible; the no-device path is exercised on the CPU box")
    with pytest.raises(capi.EAError) as ei:
        capi.Problem(525.0, 525.0, 319.5, 239.5)
    assert ei.value.code == -3  # EA_ERR_NO_DEVICE
    assert b"fallback" in lib.ea_last_error() or b"device" in lib.ea_last_error()


def test_argument_validation_needs_no_device(lib):
    h = C.c_void_p()
    assert lib.ea_problem_create(C.byref(h), None, 0, 0) == -1
    assert lib.ea_batch_create(C.byref(h), None, 0) == -1
    assert lib.ea_problem_set_loss(None, 1, 1.0) == -1
    assert lib.ea_problem_num_points(None) == 0
    # the drivers on top of the solve and the frame producers reject missing arguments before they touch a device
    q = (C.c_double * 4)(1, 0, 0, 0)
    t = (C.c_double * 3)()
    assert lib.ea_solve_pyramid(None, 3, None, q, t, None) == -1
    from edge_alignment_amd import capi
    assert lib.ea_solve_sharded(None, None, C.cast(None, capi.ALLREDUCE_FN), None, q, t, None) == -1
    assert lib.ea_tracker_create(None, None, 0, 0, 0) == -1
    assert lib.ea_tracker_create(C.byref(h), None, 0, 0, 7) == -1          # unknown pre-processing flavour
    assert lib.ea_tracker_push_frame(None, None, None, 480, 640, 5000.0, None, q, t, None, None) == -1
    assert lib.ea_tracker_problem(None) is None
    lib.ea_tracker_destroy(None)                                            # like free(NULL)
    assert lib.ea_problem_set_ref_frame(None, None, None, 480, 640, 5000.0, 35) == -1
    assert lib.ea_problem_set_ref_frame_canny(None, None, None, 480, 640, 5000.0, 30.0, 90.0) == -1
    assert lib.ea_problem_set_ref_frame_masked(None, None, None, None, 480, 640, 5000.0, 35) == -1
    assert lib.ea_problem_set_ref_frame_ros(None, None, None, 240, 320, 150.0, 100.0) == -1
    assert lib.ea_problem_set_now_frame(None, None, 480, 640, 35, 1, 1) == -1
    assert lib.ea_problem_set_now_frame_canny(None, None, None, 480, 640, 30.0, 90.0, 1, 0.0, 1.0) == -1
    assert lib.ea_problem_set_now_frame_ros(None, None, 240, 320, 150.0, 100.0) == -1
    assert lib.ea_problem_set_point_order(None, 16) == -1
    v = C.c_int()
    assert lib.ea_problem_get_point_order(None, C.byref(v)) == -1
    ms = C.c_double()
    assert lib.ea_batch_bench_fold(None, 1, 1, C.byref(ms)) == -1
    assert lib.ea_batch_set_tuning(None, b"solve_streams", 2) == -1
    assert b"NULL" in lib.ea_last_error() or b"argument" in lib.ea_last_error()


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under edge_alignment_amd/ (or include/) may import, include,
    link or dlopen it.  (Comments may mention it.)"""
    pat = re.compile(r"(^\s*(from|import)\s+oracle\b)|(#\s*include\s*[\"<][^\">]*oracle)|(ea_oracle)|(libea_oracle)|(oracle[./]ea_)", re.M)
    for top in ("edge_alignment_amd", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, top)):
            for f in files:
                if f.endswith((".py", ".h", ".hip", ".cpp", ".hpp")):
                    src = open(os.path.join(dirpath, f)).read()
                    assert not pat.search(src), (dirpath, f, pat.search(src).group(0))


def test_header_is_plain_c99(tmp_path):
    """the boundary is a C ABI: include/ea_hip.h compiles as C99 with -pedantic -Werror, and the plain-C example builds
    against it and the shared library (no C++, no HIP headers)"""
    import subprocess
    src = tmp_path / "hdr.c"
    src.write_text('#include "ea_hip.h"\nint main(void) { ea_options o; ea_default_options(&o); return o.max_num_iterations > 0 ? 0 : 1; }\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"),
                           "-c", str(src), "-o", str(tmp_path / "hdr.o")])
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "examples"), "c_abi_demo"])
    assert os.path.exists(os.path.join(ROOT, "examples", "c_abi_demo"))

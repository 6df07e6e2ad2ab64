"""CPU tests of the drop-in boundary: the C-ABI library builds for gfx950, loads without a GPU, exports every symbol
include/dmvio_hip.h declares, and the product path refuses to run without a device (no CPU fallback)."""
import ctypes
import os
import re

import pytest


def test_library_exports_every_declared_symbol(pkg):
    lib = pkg.load_library()
    syms = pkg.declared_symbols()
    assert len(syms) >= 20
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing


def test_no_test_module_shadows_a_definition():
    """No test / helper is defined twice in one module (the later definition would silently replace the earlier test); conftest.py
    enforces the same at collection time."""
    import glob
    from conftest import shadowed_definitions
    here = os.path.dirname(os.path.abspath(__file__))
    files = sorted(glob.glob(os.path.join(here, "*.py")) + glob.glob(os.path.join(here, "dropin", "*.py")))
    assert len(files) > 30
    dup = {os.path.basename(f): shadowed_definitions(f) for f in files}
    assert not any(dup.values()), {k: v for k, v in dup.items() if v}


def test_header_cites_reference_lines(pkg):
    txt = open(pkg.INCLUDE_PATH).read()
    assert len(re.findall(r"\.(?:cpp|h):\d+", txt)) >= 15, "every entry point names the reference interface it replaces"


def test_code_object_is_gfx950_only(pkg):
    blob = open(pkg.LIB_PATH, "rb").read()
    assert b"gfx950" in blob
    for other in (b"gfx90a", b"gfx942", b"sm_80", b"sm_90"):
        assert other not in blob


def test_no_cpu_fallback_without_device(pkg):
    lib = pkg.load_library()
    if lib.dmvio_hip_device_count() > 0:
        pytest.skip("a GPU is visible; the no-device refusal is exercised on the CPU box")
    with pytest.raises(pkg.HipLibraryError):
        pkg.Context(64, 64, 2)
    lib.dmvio_hip_create.restype = ctypes.c_void_p
    assert not lib.dmvio_hip_create(0, 64, 64, 2)
    assert lib.dmvio_hip_last_error()


def test_product_does_not_import_oracle():
    """oracle/ is test infrastructure: nothing under dm-vio_amd/ may reference it."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for dp, _, files in os.walk(os.path.join(root, "dm-vio_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "oracle_py" not in txt and "liboracle" not in txt and "oracle/" not in txt.replace("see oracle/", ""), os.path.join(dp, f)


def test_library_reads_no_environment_variable_that_changes_the_computation():
    """Launch shapes, the evaluation path and the accumulation order are chosen through explicit entry points (dmvio_hip_tracker_set_launch_shape, _set_eval_server,
    _set_single_frame_mode, dmvio_hip_set_pyramid_tile_log2, dmvio_hip_ba_set_accumulators): a stray variable in the environment of a production process must not change what the
    library computes.  The one getenv left is DMVIO_HIP_BA_TIMING (a timing printout on stderr)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    found = []
    for dp, _, files in os.walk(os.path.join(root, "dm-vio_amd", "csrc")):
        for f in files:
            for n, line in enumerate(open(os.path.join(dp, f), errors="ignore"), 1):
                code = line.split("//")[0]
                if "getenv" in code:
                    found.append((f, n, re.findall(r'getenv\("([A-Z0-9_]+)"\)', code)))
    assert [x[2] for x in found] == [["DMVIO_HIP_BA_TIMING"]], found


def test_header_is_plain_c_and_cxx(pkg, tmp_path):
    """include/dmvio_hip.h is the drop-in boundary: it must compile as C99 and as C++11 on its own (no torch / HIP / Eigen types), and a
    C++ translation unit that takes the address of every declared entry point must link against the shared library."""
    import subprocess
    hdr = pkg.INCLUDE_PATH
    subprocess.check_call(["gcc", "-fsyntax-only", "-x", "c", "-std=c99", "-Wall", "-Werror", hdr])
    subprocess.check_call(["g++", "-fsyntax-only", "-x", "c++", "-std=c++11", "-Wall", "-Werror", hdr])
    src = tmp_path / "link_all.cpp"
    syms = pkg.declared_symbols()
    src.write_text('#include "%s"\n#include <cstdio>\nint main() {\n  const void* p[] = {%s};\n  std::printf("%%d\\n", (int)(sizeof(p) / sizeof(p[0])));\n  return p[0] ? 0 : 1;\n}\n'
                   % (hdr, ", ".join("(const void*)&%s" % s for s in syms)))
    exe = tmp_path / "link_all"
    libdir = os.path.dirname(pkg.LIB_PATH)
    subprocess.check_call(["g++", "-std=c++11", str(src), "-o", str(exe), "-L" + libdir, "-ldmvio_hip", "-Wl,-rpath," + libdir, "-Wl,--allow-shlib-undefined"])


def _build_c_demo(pkg, out):
    import subprocess
    root = os.path.dirname(os.path.dirname(pkg.INCLUDE_PATH))
    libdir = os.path.dirname(pkg.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c99", "-O2", "-Wall", "-Wextra", "-Werror", os.path.join(root, "examples", "c_abi_demo.c"), "-I" + os.path.dirname(pkg.INCLUDE_PATH),
                           "-L" + libdir, "-ldmvio_hip", "-lm", "-Wl,-rpath," + libdir, "-Wl,--allow-shlib-undefined", "-o", str(out)])
    return out


def _build_cpp_demo(pkg, out):
    import subprocess
    root = os.path.dirname(os.path.dirname(pkg.INCLUDE_PATH))
    libdir = os.path.dirname(pkg.LIB_PATH)
    subprocess.check_call(["g++", "-std=c++11", "-O2", "-Wall", "-Wextra", "-Werror", os.path.join(root, "examples", "cpp_adapter_demo.cpp"),
                           "-I" + os.path.dirname(pkg.INCLUDE_PATH), "-L" + libdir, "-ldmvio_hip", "-Wl,-rpath," + libdir, "-Wl,--allow-shlib-undefined", "-o", str(out)])
    return out


def test_cpp_mirror_of_the_reference_surface_compiles(pkg, tmp_path):
    """include/dmvio_hip.hpp (CoarseTracker / FrameStore with the reference's member names over the C ABI) and its demo build as C++11 without
    warnings against the header and the library."""
    exe = _build_cpp_demo(pkg, tmp_path / "cpp_adapter_demo")
    assert os.path.getsize(exe) > 0


def test_plain_c_demo_compiles_and_links(pkg, tmp_path):
    """examples/c_abi_demo.c — one tracked frame through the C ABI from plain C99 (what a maintainer's adapter calls) — builds against the
    header and the shared library without warnings; tests/test_edge_gpu.py runs it on the device."""
    exe = _build_c_demo(pkg, tmp_path / "c_abi_demo")
    assert os.path.getsize(exe) > 0

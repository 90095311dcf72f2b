"""The C-ABI shared library loads on a machine without a GPU and exports every symbol that
include/pt.h declares.  No compute call is made here."""
import ctypes as C
import os
import re
import subprocess

from path_tracer_amd import _lib, abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "pt.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"^\s*(?:const\s+char\s*\*|int|void\s*\*|void|uint32_t|uint64_t)\s+(\w+)\s*\(", text, flags=re.M)
    return sorted(set(names))


def test_header_and_binding_list_the_same_symbols():
    assert _declared_symbols() == sorted(_lib.SYMBOLS)


def test_library_exports_every_declared_symbol():
    L = _lib.lib()
    for name in _declared_symbols():
        assert getattr(L, name) is not None
    assert L.pt_abi_version() == 2


def test_struct_sizes_match_the_header(tmp_path):
    """sizeof / offsetof as gcc sees include/pt.h == what the ctypes mirror declares."""
    src = tmp_path / "sizes.c"
    src.write_text(
        '#include <stdio.h>\n#include <stddef.h>\n#include "pt.h"\n'
        'int main(void){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(PtSceneSoA), sizeof(PtCamera),'
        ' sizeof(PtRenderParams), sizeof(PtCounters), offsetof(PtSceneSoA, mat_type), offsetof(PtCamera, width),'
        ' offsetof(PtCounters, diffuse_updates), sizeof(PtDim3), offsetof(PtDim3, z)); return 0;}\n')
    exe = tmp_path / "sizes"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    got = [int(v) for v in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    want = [C.sizeof(abi.PtSceneSoA), C.sizeof(abi.PtCamera), C.sizeof(abi.PtRenderParams), C.sizeof(abi.PtCounters),
            abi.PtSceneSoA.mat_type.offset, abi.PtCamera.width.offset, abi.PtCounters.diffuse_updates.offset,
            C.sizeof(abi.PtDim3), abi.PtDim3.z.offset]
    assert got == want


def test_host_only_entry_points():
    L = _lib.lib()
    s, c = abi.PtSceneSoA(), abi.PtCamera()
    assert L.pt_builtin_scene(b"no_such_scene", 64, 64, C.byref(s), C.byref(c)) == abi.PT_ERR_INVALID
    assert L.pt_builtin_scene(b"cuda_default", 0, 64, C.byref(s), C.byref(c)) == abi.PT_ERR_INVALID
    assert L.pt_builtin_scene(b"cuda_default", 64, 36, C.byref(s), C.byref(c)) == 0
    assert s.num_spheres == 9 and c.width == 64
    assert L.pt_destroy(None) == 0  # destroying nothing is not an error


def test_last_error_is_the_last_calls():
    """pt_last_error() holds the message of the thread's last call: set by a failure, empty again after a call
    that succeeds (launchRenderKernel returns void like the reference's, so this is its only error channel)."""
    L = _lib.lib()
    s, c = abi.PtSceneSoA(), abi.PtCamera()
    assert L.pt_builtin_scene(b"no_such_scene", 64, 64, C.byref(s), C.byref(c)) == abi.PT_ERR_INVALID
    assert b"no_such_scene" in L.pt_last_error()
    assert L.pt_builtin_scene(b"cuda_default", 64, 36, C.byref(s), C.byref(c)) == 0
    assert L.pt_last_error() == b""
    # the reference's call shape, dim3 by value (raytracer_kernel.cuh:51): no context is current on this thread
    assert L.pt_make_current(None) == 0
    L.launchRenderKernel(C.c_void_p(16), 64, 36, 0, abi.PtDim3(4, 3, 1), abi.PtDim3(16, 16, 1))
    assert b"no current context" in L.pt_last_error()
    assert L.pt_frames_in_flight(None) == 0 and L.pt_last_error() == b""
    assert L.pt_fly_camera_step(None, None, None, 0.0, 0.0, 0) == abi.PT_ERR_INVALID and L.pt_last_error()
    assert L.pt_destroy(None) == 0 and L.pt_last_error() == b""


def test_product_does_not_reference_the_oracle():
    """The product path must not import, link or load anything under oracle/."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "path_tracer_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "pt_oracle" not in text and "oracle_binding" not in text and "libpt_oracle" not in text, f
    out = os.popen(f"ldd {_lib.LIB_PATH}").read()
    assert "oracle" not in out

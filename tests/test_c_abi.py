"""The C-ABI library loads on a machine without a GPU and exports every symbol include/*.h declares.
No compute call is made here (there is no GPU in the CPU test tier)."""
import ctypes
import glob
import os
import re
import subprocess

from conftest import ROOT


def declared_symbols():
    names = set()
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        txt = open(h).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        names |= set(re.findall(r"FQL_API\s+[\w\s\*]+?\b(fql_\w+)\s*\(", txt))
    return names


def test_header_declares_the_boundary():
    names = declared_symbols()
    assert {"fql_linear_fwd_f32", "fql_moe_fwd_f32", "fql_linear_workspace_bytes", "fql_moe_workspace_bytes",
            "fql_version", "fql_error_string"} <= names


def test_library_exports_every_declared_symbol():
    from fused_int4_amd import _native
    lib = ctypes.CDLL(_native.SO_PATH)
    for name in declared_symbols():
        assert hasattr(lib, name), name
    assert set(_native.exported_symbols()) <= declared_symbols()


def test_header_compiles_as_plain_c():
    """The boundary header must be consumable by a C compiler (cgo / JNI / FFI binders)."""
    src = "#include \"fql_int4.h\"\nint main(void){return FQL_OK + (int)sizeof(size_t)*0;}\n"
    out = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "include"),
                          "-x", "c", "-"], input=src.encode(), capture_output=True)
    assert out.returncode == 0, out.stderr.decode()


def test_argument_validation_without_a_gpu():
    """Error returns that are decided before any HIP call."""
    from fused_int4_amd import _native
    L = _native.lib()
    assert L.fql_version() >= 100
    assert L.fql_error_string(0) == b"ok"
    assert b"even" in L.fql_error_string(-3)
    assert L.fql_linear_fwd_f32(None, None, None, None, None, 1, 3, 1, 0, None, 0, None) == -3          # odd K
    assert L.fql_linear_fwd_f32(None, None, None, None, None, 1, 4, 1, 9, None, 0, None) == -6          # bad precision
    assert L.fql_linear_fwd_f32(None, None, None, None, None, -1, 4, 1, 0, None, 0, None) == -2         # bad shape
    assert L.fql_linear_fwd_f32(None, None, None, None, None, 1, 4, 1, 0, None, 0, None) == -1          # NULL pointers
    assert L.fql_linear_fwd_f32(None, None, None, None, None, 0, 4, 1, 0, None, 0, None) == 0           # empty batch
    assert L.fql_moe_fwd_f32(None, None, None, None, None, None, None, 2, 0, 4, 4, 0, None, 0, None) == 0
    assert L.fql_moe_fwd_f32(None, None, None, None, None, None, None, 2, 4, 5, 4, 0, None, 0, None) == -3
    # workspace sizing: B <= 4 needs none; MoE config 3 of BASELINE.json
    assert L.fql_linear_workspace_bytes(1, 4096, 11008, 0) == 0
    assert L.fql_linear_workspace_bytes(512, 4096, 11008, 0) > 3 * 512 * 4096
    ws3 = L.fql_moe_workspace_bytes(8, 1024, 4096, 11008, 0)
    ws2 = L.fql_moe_workspace_bytes(8, 1024, 4096, 11008, 2)
    assert ws3 > ws2 > 2 * 1024 * 4096
    assert L.fql_act_padded_k(4096) == 4096 and L.fql_act_padded_k(100) == 256

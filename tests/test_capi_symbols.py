"""CPU-side checks of the drop-in boundary: the HIP library builds for gfx950, loads without a GPU and
exports exactly the symbols include/sdrainer_hip.h declares (no compute calls here)."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "sdrainer_hip.h")


@pytest.fixture(scope="module")
def lib_path():
    from sdrainer_amd.csrc import build
    return build.build()


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sdr_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_what_the_binding_binds():
    from sdrainer_amd import capi
    assert declared_symbols() == sorted(capi.SYMBOLS)


def test_library_exports_every_declared_symbol(lib_path):
    out = subprocess.check_output(["nm", "-D", "--defined-only", lib_path], text=True)
    exported = {ln.split()[-1] for ln in out.splitlines() if " T " in ln}
    missing = [s for s in declared_symbols() if s not in exported]
    assert not missing, missing
    # nothing but the C ABI leaks out of the library
    leaked = [s for s in exported if not s.startswith("sdr_")]
    assert not leaked, leaked


def test_library_loads_and_reports_abi(lib_path):
    from sdrainer_amd import capi
    L = capi.load()
    assert L.sdr_abi_version() == 2
    assert [L.sdr_kernel_name(i).decode() for i in range(8)] == list(capi.KERNELS)
    assert ctypes.sizeof(capi.Config) == 56 and ctypes.sizeof(capi.Peak) == 40
    assert capi.FRAME_REC_DTYPE.itemsize == 40 and capi.EDGE_DTYPE.itemsize == 8
    # bulk delivery records (sdr_results and what it points to)
    assert ctypes.sizeof(capi.Results) == 128 and capi.PEAK_DTYPE.itemsize == 40
    assert capi.CHUNK_RESULT_DTYPE.itemsize == 24 and capi.LISTENER_RESULT_DTYPE.itemsize == 24


def test_code_object_is_gfx950_only(lib_path):
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("llvm-objdump not available")
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:  # --offloading extracts the code objects into the cwd
        out = subprocess.run([objdump, "--offloading", lib_path], capture_output=True, text=True, cwd=tmp).stdout
    archs = set(re.findall(r"gfx[0-9a-f]+", out))
    assert archs == {"gfx950"}, archs


def test_product_does_not_reference_the_oracle():
    """The product path must never import, link or execute anything under oracle/."""
    pkg = os.path.join(ROOT, "sdrainer_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                assert "liborc" not in txt and "from oracle" not in txt and "import oracle" not in txt, f

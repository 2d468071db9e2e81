"""CPU-side checks of the drop-in boundary: libgangfit.so builds for gfx950, loads, and exports every symbol that
include/gangfit.h declares.  No compute calls here (no GPU in this container)."""
import ctypes
import os
import re

import pytest

import gangfit
from gangfit import _native, build

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib_path():
    return build.build_native()


def _declared_symbols():
    text = open(os.path.join(REPO, "include", "gangfit.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gf_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_all_exported(lib_path):
    lib = ctypes.CDLL(lib_path)
    declared = _declared_symbols()
    assert len(declared) >= 15
    missing = [s for s in declared if not hasattr(lib, s)]
    assert not missing, f"declared in gangfit.h but not exported: {missing}"
    assert sorted(_native.EXPORTED_SYMBOLS) == declared


def test_version_and_struct_layout(lib_path):
    lib = _native.load()
    assert lib.gf_version() == 300
    # layout promised by the header: 64-byte app records, 16-byte results
    assert _native.APP_DTYPE.itemsize == 64 and _native.APP_DTYPE.fields["exec_off"][1] == 56
    assert _native.RESULT_DTYPE.itemsize == 16


def test_code_object_is_gfx950(lib_path):
    """Every device code object bundled into the library targets gfx950 (bundle ids `...amdhsa--<arch>`); no sorting or
    scan library is linked in (the priority sort of gangfit_snapshot.hip is hand-written), so nothing else names an arch."""
    import re

    blob = open(lib_path, "rb").read()
    targets = set(re.findall(rb"amdgcn-amd-amdhsa--(gfx[0-9a-f]+)", blob))
    assert targets == {b"gfx950"}, targets
    assert b"sm_90" not in blob and b"nvptx" not in blob


def test_init_without_device_fails_loudly(lib_path):
    if os.path.exists("/dev/kfd"):
        pytest.skip("a GPU is present")
    with pytest.raises(gangfit.GangfitError) as e:
        gangfit.Context(0)
    assert e.value.code in (_native.GF_ERR_NO_DEVICE, _native.GF_ERR_HIP)


def test_product_path_never_touches_the_oracle():
    """Nothing under k8s-spark-scheduler_amd/ or include/ may reference oracle/ (the oracle is test infrastructure)."""
    bad = []
    for root in ("k8s-spark-scheduler_amd", "include"):
        for dirpath, _, files in os.walk(os.path.join(REPO, root)):
            for f in files:
                if f.endswith((".so", ".o", ".pyc")):
                    continue
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                if re.search(r"gangfit_oracle|pyoracle|from oracle|import oracle|oracle/", text):
                    bad.append(os.path.join(dirpath, f))
    assert not bad, bad


def test_header_is_plain_c99(tmp_path):
    """The drop-in boundary is a C ABI: include/gangfit.h must compile as C99 (what cgo feeds it to) with the promised
    record layouts."""
    import subprocess

    src = tmp_path / "hdr.c"
    src.write_text('#include "gangfit.h"\n'
                   "int main(void) { return sizeof(gf_app) == 64 && sizeof(gf_result) == 16 && sizeof(gf_shard_partial) == 16 &&\n"
                   "                        sizeof(gf_shard_driver) == 16 && sizeof(gf_avg_efficiency) == 32 ? 0 : 1; }\n")
    exe = tmp_path / "hdr"
    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", inc, str(src), "-o", str(exe)])
    assert subprocess.call([str(exe)]) == 0

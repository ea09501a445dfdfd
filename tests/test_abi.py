"""CPU: the C-ABI library builds, loads and exports every symbol include/dfusion.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built_lib():
    import __graft_entry__ as g
    g.build()
    return g.LIB


def _declared():
    src = open(os.path.join(ROOT, "include", "dfusion.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    # every df_* function plus `knn_device`, the reference's own native symbol (lib/knn/src/knn_cuda_kernel.h:14-16)
    return sorted(set(re.findall(r"\b(df_[a-z0-9_]+|knn_device)\s*\(", src)))


def test_header_symbols_exported(built_lib):
    L = ctypes.CDLL(built_lib)
    names = _declared()
    assert "df_knn_device" in names and "df_knn" in names and "knn_device" in names
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/dfusion.h but not exported"


def test_python_binding_covers_header(built_lib):
    from densefusion_amd import _lib
    assert sorted(_lib.SIGNATURES) == _declared()


def test_no_torch_types_in_abi():
    src = open(os.path.join(ROOT, "include", "dfusion.h")).read()
    assert "torch" not in src.lower().replace("pytorch.c", "").replace("pytorch.h", "")
    assert "at::" not in src and "Tensor *" not in src


def test_version_and_error_string(built_lib):
    from densefusion_amd import _lib
    L = _lib.lib()
    assert L.df_version() >= 1
    # argument errors are detected on the host before any launch -> safe without a GPU
    assert L.df_knn(None, None, None, 1, 3, 10, 10, 1, None) == -1
    assert b"null" in L.df_last_error()
    with pytest.raises(RuntimeError):
        _lib.check(-1, "knn")


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "densefusion_amd")
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, fn)).read()
                assert "oracle" not in txt.replace("# oracle", "").replace("oracle is", "").replace(
                    "the oracle", "").replace("oracle/", ""), f"{fn} references the oracle package"


def test_the_shipped_library_reads_no_environment(built_lib):
    """Development switches live in the -DDF_DEV build only (libdfusion_hip_dev.so, loaded by the variant-comparison tests): the product
    library does not even import getenv."""
    import subprocess
    und = subprocess.run(["nm", "-D", "--undefined-only", built_lib], capture_output=True, text=True, check=True).stdout
    assert "getenv" not in und
    dev = built_lib.replace("libdfusion_hip.so", "libdfusion_hip_dev.so")
    assert os.path.exists(dev) and "getenv" in subprocess.run(["nm", "-D", "--undefined-only", dev], capture_output=True, text=True, check=True).stdout

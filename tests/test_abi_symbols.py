"""CPU tests: the built shared libraries load and export every symbol the public headers declare.
No compute call is made (there is no GPU in the build container)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared(header, prefix):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(" + prefix + r"[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def built():
    from pointcloudtraj_amd import build
    build.build_all()
    return build


def test_engine_exports_every_declared_symbol(built):
    names = declared("pct_engine.h", "pct_")
    assert len(names) >= 30
    L = C.CDLL(built.ENGINE_SO)
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing


def test_engine_fails_loudly_without_gpu(built):
    from pointcloudtraj_amd import engine as E
    if E.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(E.EngineError) as ei:
        E.init(0)
    assert ei.value.code == 1 and "no host fallback" in str(ei.value)
    with pytest.raises(E.EngineError):
        E.Cloud(16)

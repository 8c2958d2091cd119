"""CPU-side checks of the product boundary: the HIP library builds, loads, exports every symbol the header declares,
and refuses to run without a GPU (no silent fallback).  No compute calls here."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "flashfry_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(ffh_[a-z0-9_]+)\s*\(", txt)))


def test_library_builds_and_exports_the_declared_abi():
    from flashfry_amd import capi
    L = capi.load_library()
    declared = header_symbols()
    assert declared == sorted(capi.SYMBOLS)
    out = subprocess.check_output(["nm", "-D", "--defined-only", capi.library_path()]).decode()
    exported = set(re.findall(r"\bT (ffh_[a-z0-9_]+)", out))
    assert set(declared) <= exported, set(declared) - exported
    for s in declared:
        assert hasattr(L, s)
    assert L.ffh_version() == 1


def test_no_cpu_fallback():
    import torch
    from flashfry_amd import capi
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    L = capi.load_library()
    assert L.ffh_device_count() == 0
    with pytest.raises(capi.FlashFryHipError) as e:
        capi.Context(3)
    assert "no HIP device" in str(e.value) or "no CPU fallback" in str(e.value)


def test_product_never_touches_the_oracle():
    """nothing under flashfry_amd/ may import, link or execute anything under oracle/"""
    bad = []
    for dp, _, files in os.walk(os.path.join(ROOT, "flashfry_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".hpp", ".h", ".inc")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                if re.search(r"ff_oracle|oracle_lib|libff_oracle|from\s+oracle|import\s+oracle|oracle/", txt):
                    bad.append(os.path.join(dp, f))
    assert not bad, bad

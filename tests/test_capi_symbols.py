"""CPU suite: the C-ABI shared library loads and exports every symbol that
include/*.h declares (no compute calls -- there is no GPU here), and the product
code does not reach into oracle/."""
import ctypes
import glob
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    syms = set()
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        text = open(h).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        syms |= set(re.findall(r"\b(lsa_[a-z0-9_]+)\s*\(", text))
    return sorted(syms)


def test_header_declares_entry_points():
    syms = declared_symbols()
    for must in ("lsa_init", "lsa_g1_msm", "lsa_g2_msm", "lsa_msm_run", "lsa_g1_bases_create"):
        assert must in syms


def test_library_exports_every_declared_symbol():
    import legosnark_amd
    if not os.path.exists(legosnark_amd.LIB_PATH):
        legosnark_amd.build()
    lib = legosnark_amd.lib()          # (loads torch first when it is installed: one RCCL build per process, see lib())
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, missing


def test_library_exports_nothing_else():
    """-fvisibility=hidden + csrc/exports.map: the dynamic symbol table is the header, nothing more (the library is
    linked beside libsnark / libff builds: no stray C++ or kernel-handle symbols may interpose)."""
    import subprocess
    import legosnark_amd
    if not os.path.exists(legosnark_amd.LIB_PATH):
        legosnark_amd.build()
    libs = [legosnark_amd.LIB_PATH]
    lb = legosnark_amd.LIB_PATH.replace(".so", "_loopback.so")
    if os.path.exists(lb):
        libs.append(lb)
    for path in libs:
        out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
        names = [ln.split()[-1] for ln in out.splitlines() if ln.strip()]
        assert sorted(names) == declared_symbols(), (path, sorted(set(names) ^ set(declared_symbols()))[:10])


def test_no_cpu_fallback_without_device():
    """Without a GPU every compute entry point must fail loudly (never fall back)."""
    import legosnark_amd
    if not os.path.exists(legosnark_amd.LIB_PATH):
        legosnark_amd.build()
    if legosnark_amd.device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(legosnark_amd.LsaError):
        legosnark_amd.init(0)
    import numpy as np
    with pytest.raises(legosnark_amd.LsaError):
        legosnark_amd.msm("g1", np.zeros((1, 12), dtype=np.uint64), np.zeros((1, 4), dtype=np.uint64))


def test_product_does_not_touch_oracle():
    bad = []
    for base in ("legosnark_amd", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".so", ".o", ".pyc")):
                    continue
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                if re.search(r"oracle[/_.]|liboracle|oracle_lib", text):
                    bad.append(os.path.join(dirpath, f))
    assert not bad, bad

"""CPU suite: host-compiled unit tests of the device arithmetic headers (the same code the
kernels compile): 29-bit-limb field + bound-aware XYZZ formulas against the canonical
32-bit-limb code, and the GLV decomposition (k1 + k2*lambda == k mod r, |k_i| < 2^127)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("name", ["test_fp29", "test_glv", "test_fp29x2", "test_tower29", "test_smul", "test_w12", "test_tmiller", "test_inv29"])
def test_host_cpp(name, tmp_path):
    src = os.path.join(ROOT, "tests", "cpp", name + ".cc")
    if not os.path.exists(src):
        pytest.skip(name + " not present")
    exe = str(tmp_path / name)
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-I", os.path.join(ROOT, "legosnark_amd", "csrc"), src, "-o", exe])
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0 and "PASS" in r.stdout, r.stdout[-2000:]


@pytest.mark.parametrize("name", ["test_fp29", "test_glv", "test_tower29"])
def test_host_cpp_with_the_32_bit_limb_product(name, tmp_path):
    """On the host Fp's Montgomery product runs on four 64-bit limbs; the device compiles the
    32-bit-limb CIOS.  LSA_FP_HOST32 forces the device variant on the host: the same tests must
    pass with either, i.e. both agree with the independent 29-bit-limb code and the integer
    identities."""
    src = os.path.join(ROOT, "tests", "cpp", name + ".cc")
    exe = str(tmp_path / (name + "_32"))
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-DLSA_FP_HOST32", "-I", os.path.join(ROOT, "legosnark_amd", "csrc"), src, "-o", exe])
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0 and "PASS" in r.stdout, r.stdout[-2000:]


def test_host_64_bit_limbs_agree_with_the_device_limbs(tmp_path):
    """fp.h on the host: four 64-bit limbs, carry intrinsics, binary-Euclid inverse (what the reference's own host
    loops run on through the shim).  The device variant (-DLSA_FP_HOST32: eight 32-bit limbs, Fermat inverse) must
    print the same digests over the same operand sequence, edge values included."""
    src = os.path.join(ROOT, "tests", "cpp", "test_fp_host.cc")
    outs = []
    for tag, flags in (("64", []), ("32", ["-DLSA_FP_HOST32"])):
        exe = str(tmp_path / ("test_fp_host_" + tag))
        subprocess.check_call(["g++", "-std=c++17", "-O2", *flags, "-I", os.path.join(ROOT, "legosnark_amd", "csrc"), src, "-o", exe])
        r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
        assert r.returncode == 0 and "PASS" in r.stdout, r.stdout[-2000:]
        outs.append(r.stdout)
    assert outs[0] == outs[1] and outs[0].count("inv ") == 2

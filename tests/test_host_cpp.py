"""CPU suite: host-compiled unit tests of the device arithmetic headers (the same code the
kernels compile): 29-bit-limb field + bound-aware XYZZ formulas against the canonical
32-bit-limb code, and the GLV decomposition (k1 + k2*lambda == k mod r, |k_i| < 2^127)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("name", ["test_fp29", "test_glv", "test_fp29x2", "test_tower29", "test_smul", "test_w12", "test_tmiller", "test_inv29", "test_ntt_core"])
def test_host_cpp(name, tmp_path):
    src = os.path.join(ROOT, "tests", "cpp", name + ".cc")
    if not os.path.exists(src):
        pytest.skip(name + " not present")
    exe = str(tmp_path / name)
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-I", os.path.join(ROOT, "legosnark_amd", "csrc"), src, "-o", exe])
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0 and "PASS" in r.stdout, r.stdout[-2000:]


@pytest.mark.parametrize("name", ["test_fp29", "test_fp29x2", "test_tmiller", "test_w12", "test_ntt_core"])
def test_host_cpp_with_the_column_wise_products(name, tmp_path):
    """-DLSA_FP29_COLS / -DLSA_F29_COLS / -DLSA_FR29_COLS select the column-wise forms of the 29-bit-limb products (17
    independent column accumulators instead of one serial multiply-add chain: fs29.h f29_dot_cols, fp29.h redc_cols,
    fr29.h mul).  Round 6 measured them equal to the serial forms on the GPU and left them behind these switches; they must
    stay the same functions: every host test of the field, tower, Miller and NTT code passes with them too."""
    src = os.path.join(ROOT, "tests", "cpp", name + ".cc")
    exe = str(tmp_path / (name + "_cols"))
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-DLSA_FP29_COLS", "-DLSA_F29_COLS", "-DLSA_FR29_COLS", "-I", os.path.join(ROOT, "legosnark_amd", "csrc"), src, "-o", exe])
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


def test_frobenius_row_constants_are_the_generators_output():
    """csrc/frob_rows.h is generated (tools/gen_frob_rows.py: the Frobenius factors of bn254_constants.h multiplied out,
    conjugation and signs folded in, 29-bit-limb Montgomery form); the committed table must be what the script prints,
    and must satisfy gamma_(P,k) = FROB6_C{k>>1}[P] * (k odd ? FROB12_C1[P] : 1) against the oracle's Fq2 arithmetic."""
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "gen_frob_rows.py")], cwd=root, capture_output=True, text=True, check=True).stdout
    header = open(os.path.join(root, "legosnark_amd", "csrc", "frob_rows.h")).read()
    want = re.findall(r"0x[0-9a-f]{8}u", out)
    got = re.findall(r"0x[0-9a-f]{8}u", header)
    assert len(want) == 3 * 6 * 2 * 2 * 9 and got == want
    # spot value: power 2, k = 0 is the identity map: rows (1, -0) / (0, 1) in Montgomery form R = 2^261
    p = 21888242871839275222246405745257275088696311157297823662689037894645226208583
    one = (1 << 261) % p
    limbs = [int(x[2:-1], 16) for x in want]
    base = ((1 * 6 + 0) * 2 + 0) * 2 * 9                      # [power - 1 = 1][k = 0][part 0][h 0]
    assert sum(l << (29 * i) for i, l in enumerate(limbs[base:base + 9])) == one
    assert all(l == 0 for l in limbs[base + 9:base + 18])       # -g1 = 0


def test_final_exponentiation_scalar_exponent_is_the_generators_output():
    """csrc/w12.h: W12_FE_SCALAR_EXP is generated (tools/gen_fe_scalar_exponent.py walks libff's final-exponentiation chain
    symbolically, with conjugations NOT changing the sign of a stray Fq factor, and checks on the big-int model that the
    chain without its inversion, times n1^e, is f^(libff's exponent)); the committed words must be what the script prints."""
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "gen_fe_scalar_exponent.py")], cwd=root, capture_output=True, text=True, check=True).stdout
    header = open(os.path.join(root, "legosnark_amd", "csrc", "w12.h")).read()
    want = re.search(r"W12_FE_SCALAR_EXP\[8\] = \{([^}]*)\}", out).group(1)
    got = re.search(r"W12_FE_SCALAR_EXP\[8\] = \{([^}]*)\}", header).group(1)
    assert got == want
    bits = re.search(r"W12_FE_SCALAR_EXP_BITS = (\d+)", out).group(1)
    assert re.search(r"W12_FE_SCALAR_EXP_BITS = (\d+)", header).group(1) == bits == "254"
    # e < q - 1, and the chain has the links to run it: 264 row products follow the norm (w12.h: w12_final_exponentiation_h)
    words = [int(x.strip()[:-1], 16) for x in want.split(",")]
    e = sum(w << (32 * i) for i, w in enumerate(words))
    q = 21888242871839275222246405745257275088696311157297823662689037894645226208583
    assert 0 < e < q - 1 and e.bit_length() == 254

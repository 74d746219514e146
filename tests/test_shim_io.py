"""CPU suite: the libff-compatible shim's text wire format (/root/reference/src/utils/util.h:56-96
streams Fr / G1 / G2 with operator<< / operator>>) against strings derived from the big-int
model, under each of libff's serialisation macros; plus point decompression, is_well_formed and
the CSPRNG-backed random_element."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle", "pymodel"))
import bn254_model as m  # noqa: E402

KS = [1, 2, 3, 12345, 1000003]


def build_and_run(tmp_path, flags):
    import legosnark_amd
    if not os.path.exists(legosnark_amd.LIB_PATH):
        legosnark_amd.build()
    exe = str(tmp_path / ("shim_io" + "".join(f.replace("-D", "_") for f in flags)))
    libdir = os.path.join(ROOT, "legosnark_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-w", *flags, "-I", os.path.join(ROOT, "legosnark_amd", "shim"),
                           os.path.join(ROOT, "tests", "cpp", "test_shim_io.cc"), "-o", exe,
                           "-L" + libdir, "-llegosnark_amd", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0 and "PASS" in r.stdout, r.stdout[-3000:]
    lines = {}
    for ln in r.stdout.splitlines():
        if ln.startswith("LINE "):
            key, val = ln[5:].split(" | ", 1)
            lines[key] = val
    return lines


def fq_txt(x, mont):
    return str(x * m.MONT_R % m.P if mont else x)


@pytest.mark.parametrize("flags", [[], ["-DNO_PT_COMPRESSION"], ["-DMONTGOMERY_OUTPUT"], ["-DMONTGOMERY_OUTPUT", "-DNO_PT_COMPRESSION"]])
def test_text_format_matches_libffs(tmp_path, flags):
    lines = build_and_run(tmp_path, flags)
    mont = "-DMONTGOMERY_OUTPUT" in flags
    full = "-DNO_PT_COMPRESSION" in flags
    for k in KS:
        x, y = m.g1_mul(m.G1_GEN, k)
        want = "0 %s %s" % (fq_txt(x, mont), fq_txt(y, mont) if full else str(y & 1))
        assert lines["G1 %d" % k] == want
        (x0, x1), (y0, y1) = m.g2_mul(m.G2_GEN, k)
        xs = "%s %s" % (fq_txt(x0, mont), fq_txt(x1, mont))
        want = "0 %s %s" % (xs, ("%s %s" % (fq_txt(y0, mont), fq_txt(y1, mont))) if full else str(y0 & 1))
        assert lines["G2 %d" % k] == want
        v = (k * k - 7) % m.R
        assert lines["FR %d" % k] == str(v * m.MONT_R % m.R if mont else v)
    # libff prints the affine form of zero() = (0, 1, 0) with the flag set
    assert lines["G1 inf"].startswith("1 0 ")
    want = " ".join("%s %s" % (fq_txt(10 + i, mont), fq_txt(100 + i, mont)) for i in range(6))
    assert lines["GT x"] == want


def test_binary_output_round_trips(tmp_path):
    build_and_run(tmp_path, ["-DBINARY_OUTPUT"])
    build_and_run(tmp_path, ["-DBINARY_OUTPUT", "-DMONTGOMERY_OUTPUT"])


def test_reproducible_randomness_only_under_the_test_macro(tmp_path):
    build_and_run(tmp_path, ["-DLSA_SHIM_TEST_SEED"])


def test_evaluation_domains_select_and_behave_as_libfqfft(tmp_path):
    """tests/cpp/test_shim_domains.cc: which domain get_evaluation_domain returns for a size (basic radix-2 for powers of
    two, the step domain for 2^b + 2^s, the rounded size otherwise) and the step domain's host helpers -- points,
    vanishing polynomial, Lagrange coefficients, add_poly_Z, divide_by_Z_on_coset -- against their definitions."""
    import legosnark_amd
    if not os.path.exists(legosnark_amd.LIB_PATH):
        legosnark_amd.build()
    exe = str(tmp_path / "shim_domains")
    libdir = os.path.join(ROOT, "legosnark_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-w", "-I", os.path.join(ROOT, "legosnark_amd", "shim"),
                           os.path.join(ROOT, "tests", "cpp", "test_shim_domains.cc"), "-o", exe,
                           "-L" + libdir, "-llegosnark_amd", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0 and "PASS" in r.stdout, r.stdout[-3000:]

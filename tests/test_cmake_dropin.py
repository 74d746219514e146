"""CPU suite: LegoSNARK's own, untouched CMakeLists.txt configures and builds against the
drop-in packages of legosnark_amd/shim/cmake (CMake targets `snark`, `ff`,
`fmt::fmt-header-only`; /root/reference/src/CMakeLists.txt:22, src/examples/CMakeLists.txt:1-11).
Needs the reference checkout, which only exists in the dev container."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src")), reason="reference checkout not present")
@pytest.mark.skipif(shutil.which("cmake") is None or shutil.which("ninja") is None, reason="cmake/ninja not present")
def test_reference_cmake_builds_cplink_against_the_dropin_packages(tmp_path):
    import legosnark_amd
    if not os.path.exists(legosnark_amd.LIB_PATH):
        legosnark_amd.build()
    tree, bld = str(tmp_path / "tree"), str(tmp_path / "build")
    subprocess.check_call([os.path.join(ROOT, "legosnark_amd", "shim", "cmake", "stage_tree.sh"), REF, tree])
    # every file of the reference is a symlink into the read-only checkout
    assert os.path.realpath(os.path.join(tree, "src")) == os.path.join(REF, "src")
    assert os.path.realpath(os.path.join(tree, "CMakeLists.txt")) == os.path.join(REF, "CMakeLists.txt")
    subprocess.check_call(["cmake", "-S", tree, "-B", bld, "-G", "Ninja", "-DWITH_PROCPS=OFF", "-DOPT_FLAGS=-O1",
                           "-DCMAKE_POLICY_VERSION_MINIMUM=3.5", "-Wno-dev"], stdout=subprocess.DEVNULL)
    r = subprocess.run(["cmake", "--build", bld, "--target", "cplink"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-3000:]
    exe = os.path.join(bld, "src", "examples", "cplink")
    assert os.path.exists(exe)
    needed = subprocess.check_output(["readelf", "-d", exe], text=True)
    assert "liblegosnark_amd.so" in needed


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src")), reason="reference checkout not present")
@pytest.mark.parametrize("src", ["gadgets/lipmaa.cc", "gadgets/subspace.cc", "utils/sparsemexp.cc", "examples/hadamard.cc"])
def test_reference_sources_compile_in_the_multicore_configuration(src, tmp_path):
    """-DMULTICORE=ON (/root/reference/CMakeLists.txt:35-39,57-59,78-80) = -fopenmp -DMULTICORE=1 under the reference's
    own warning flags: src/utils/globl.h:52,68 and src/utils/sparsemexp.cc:6,17 call omp_get_max_threads() without
    including <omp.h> -- libff's headers do under that macro, so the shim's must."""
    R = os.path.join(REF, "src")
    cmd = ["g++", "-std=c++17", "-O0", "-Wall", "-Wextra", "-Wfatal-errors", "-pthread", "-fopenmp", "-DMULTICORE=1",
           "-DBN_SUPPORT_SNARK=1", "-DCURVE_BN128", "-DNO_PROCPS", "-I", os.path.join(ROOT, "legosnark_amd", "shim")]
    for d in ("", "gadgets", "prototools", "examples", "utils"):
        cmd += ["-I", os.path.join(R, d)]
    r = subprocess.run(cmd + ["-c", os.path.join(R, src), "-o", str(tmp_path / "o.o")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-3000:]
    syms = subprocess.check_output(["nm", "-u", str(tmp_path / "o.o")], text=True)
    if src != "examples/hadamard.cc":
        assert "omp_get_max_threads" in syms or "GOMP_parallel" in syms


def test_shim_header_has_no_openmp_dependency_without_multicore(tmp_path):
    """The default configuration must not need libgomp: <omp.h> is only pulled in under MULTICORE."""
    src = tmp_path / "t.cc"
    src.write_text('#include "libff/lsa_libff.hpp"\n#ifdef _OMP_H\n#error omp.h included without MULTICORE\n#endif\nint main() { return 0; }\n')
    subprocess.check_call(["g++", "-std=c++17", "-fsyntax-only", "-I", os.path.join(ROOT, "legosnark_amd", "shim"), str(src)])

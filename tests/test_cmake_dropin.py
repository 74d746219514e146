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

"""GPU: LegoSNARK's UNCHANGED example programs, compiled against the libff-compatible shim
(legosnark_amd/shim, built by __graft_entry__.build() in the dev container where
/root/reference exists) and linked with the MI355X library, run end to end.
cplink asserts MYREQUIRE(ss.verify(...)) itself (/root/reference/src/examples/cplink.cc:114):
prover MSM <-> keygen sparse-MSM / batch_exp <-> 3 pairings must be mutually consistent."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "build", "reference")


CMAKE_BIN = os.path.join(ROOT, "build", "reference_cmake", "src", "examples")
# the reference's -DMULTICORE=ON configuration (-fopenmp -DMULTICORE=1): Makefile- and CMake-built
BIN_MC = os.path.join(ROOT, "build", "reference_mc")
CMAKE_BIN_MC = os.path.join(ROOT, "build", "reference_cmake_mc", "src", "examples")


def require_binary(exe):
    """The binaries are built from the reference's sources, which exist only in the dev container; build/ travels
    with the snapshot.  tests/golden/reference_build_manifest.json (tracked, written by __graft_entry__.build())
    lists what was built: a listed binary that is missing is a FAILURE -- lost evidence must not look like a skip.
    Without a manifest (a checkout that never ran build() next to the reference) there is nothing to run."""
    if os.path.exists(exe):
        return
    import json
    manifest = os.path.join(ROOT, "tests", "golden", "reference_build_manifest.json")
    listed = []
    if os.path.exists(manifest):
        listed = json.load(open(manifest)).get("binaries", [])
    rel = os.path.relpath(exe, ROOT)
    assert rel not in listed, "%s was built by __graft_entry__.build() (manifest) but is missing from this snapshot" % rel
    pytest.skip("%s missing and not in the build manifest: run __graft_entry__.build() where /root/reference exists" % exe)


def run(name, *args, seed="7", bindir=BIN, **extra_env):
    exe = os.path.join(bindir, name)
    require_binary(exe)
    env = dict(os.environ, LSA_SEED=seed, **extra_env)
    return subprocess.run([exe, *args], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600, text=True)


@pytest.mark.parametrize("seed", ["7", "12345"])
def test_cplink_unchanged_source_verifies(seed):
    r = run("cplink", seed=seed)
    assert r.returncode == 0, r.stdout[-2000:]
    assert "NCHUNKS : 1" in r.stdout          # multiExpMA ran (src/utils/globl.h:72)
    assert "Error!" not in r.stdout


def test_hadamard_and_matrixsc_run():
    r = run("hadamard", "5")
    assert r.returncode == 0, r.stdout[-2000:]
    assert "had_lipmaa Prove" in r.stdout and "had_sc" in r.stdout
    r = run("matrixsc", "3")
    assert r.returncode == 0, r.stdout[-2000:]


def test_keygen_matrix_batched_call_equals_reference_column_loop():
    """legosnark_amd/shim/checks/keygen_check.cc: the reference's own mtxmultiexp column loop
    (simplesparsemexp from its unchanged sparsemexp.cc) vs libff::lsa_mtxmultiexp on the CPlink
    relation matrix; the program exits non-zero on any mismatching column."""
    r = run("keygen_check", "10", "128")
    assert r.returncode == 0, r.stdout[-2000:]
    assert '"mismatches": 0' in r.stdout


@pytest.mark.parametrize("bindir", [BIN, BIN_MC], ids=["single", "multicore"])
def test_fr_kernels_equal_the_references_own_templates(bindir):
    """legosnark_amd/shim/checks/fr_check.cc: the reference's unchanged header code -- MultiVPolyT::evalMLE
    (prototools/polytools.h:207-234), CPPoly::prove's recursion (gadgets/poly.h:55-67, observed through its own multiExpMA
    over distinct random bases), DPMle::pushRandomness / getMLEPoly and DPBeta::pushRandomness / getBetaPoly
    (prototools/mle.h), CPSumcheck::make_new_h_poly (gadgets/sumcheck.h:85-106) with one to three tables, with and
    without the beta factor, d = 1 .. 16 -- against lsa_fr_eval_mle / _cppoly_witness / _fold / _scale_upper /
    _sumcheck_round on the same random inputs, byte for byte.  This pins the f3 kernels to the reference itself, not
    only to the oracle's restatement of these loops."""
    import json
    r = run("fr_check", "16", "2", bindir=bindir, OMP_NUM_THREADS="4")
    assert r.returncode == 0, r.stdout[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["mismatching_shapes"] == 0
    assert line["evalMLE"] == 32 and line["cppoly_witness"] == 12 and line["sumcheck_rounds"] > 500 and line["folds"] > 900 and line["suffix_updates"] > 150 and line["eq_tables"] >= 13


@pytest.mark.parametrize("d,inputs", [("8", "random"), ("12", "squares"), ("16", "random")])
def test_resident_prover_emits_the_references_proof(d, inputs):
    """legosnark_amd/shim/checks/resident_prover_check.cc: CPHad's prover written against the C-ABI with a, b, c resident
    on the device (lsa_fr_eval_mle / _cppoly_witness / _eq_table / _sumcheck_round / _fold / _scale_upper,
    lsa_msm_run_segments_async, lsa_commit_run_async) and the unchanged CPHad::prove (gadgets/hadamardsc.cc:54-98,
    sumcheck.cc:12-125) on the same inputs and the same seeded random stream: every proof element equal, and the
    reference's unchanged CPHad::verify accepts the resident proof."""
    import json
    r = run("resident_prover_check", d, inputs)
    assert r.returncode == 0, r.stdout[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])["resident_prover"]
    assert line["proof_equal"] is True and line["differing_elements"] == 0 and line["reference_verifier_accepts"] is True
    assert line["d"] == int(d)


def test_verifier_side_through_the_shim_deferred_equals_call_by_call():
    """legosnark_amd/shim/checks/pairing_check.cc: libff G2_precomp semantics (coefficients, stream format), deferred
    GT values against explicit C-ABI calls bit for bit, the reference's own simple_pairing_check on true and false
    statements, and its unchanged CPPoly::verify returning the same boolean whether the shim defers or not."""
    import json
    outs = []
    for eager in ("0", "1"):
        exe = os.path.join(BIN, "pairing_check")
        require_binary(exe)
        env = dict(os.environ, LSA_SEED="11", LSA_SHIM_EAGER=eager)
        r = subprocess.run([exe, "6"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600, text=True)
        assert r.returncode == 0, r.stdout[-2000:]
        outs.append(json.loads(r.stdout.strip().splitlines()[-1]))
    assert outs[0]["failures"] == 0 and outs[1]["failures"] == 0
    assert outs[0]["eager"] is False and outs[1]["eager"] is True
    assert outs[0]["cppoly_verify"] == outs[1]["cppoly_verify"]


def test_libff_surface_the_examples_do_not_reach():
    """legosnark_amd/shim/checks/shim_check.cc: libff's window_table indexed like the vector of vectors it is upstream
    (rows, columns, the short last row, iteration, batch_exp over the same object), and libfqfft's
    get_evaluation_domain picking the step domain / rounding as upstream does, refusing only sizes beyond the 2-adicity."""
    r = run("shim_check")
    assert r.returncode == 0, r.stdout[-2000:]
    assert "shim_check: 0 failure(s)" in r.stdout and "FAIL" not in r.stdout


@pytest.mark.parametrize("bindir", [BIN, BIN_MC])
def test_domains_that_are_not_powers_of_two_carry_the_lipmaa_gadget(bindir):
    """legosnark_amd/shim/checks/domain_check.cc: libfqfft's step radix-2 domain (m = 2^b + 2^s) through the shim's
    evaluation_domain interface -- FFT / iFFT / cosetFFT / icosetFFT against Horner at get_domain_element(k), the Lagrange
    coefficients -- and the reference's Lipmaa Hadamard gadget (src/gadgets/lipmaa.cc, unchanged) at n = 12 and 768:
    honest proofs accepted, a wrong product rejected.  A larger n = 3 * 2^14 as argument."""
    r = run("domain_check", str(3 << 14), bindir=bindir)
    assert r.returncode == 0, r.stdout[-3000:]
    assert '"failures": 0' in r.stdout and "FAIL" not in r.stdout


def test_cplink_built_by_the_references_own_cmake_verifies():
    """The reference's unchanged CMakeLists.txt with depends/libsnark and depends/fmt replaced by
    legosnark_amd/shim/cmake (targets snark, ff, fmt::fmt-header-only)."""
    r = run("cplink", bindir=CMAKE_BIN)
    assert r.returncode == 0, r.stdout[-2000:]
    assert "NCHUNKS : 1" in r.stdout and "Error!" not in r.stdout
    r = run("hadamard", "4", bindir=CMAKE_BIN)
    assert r.returncode == 0, r.stdout[-2000:]


def test_multiexpma_on_a_chunk_per_process_from_cpp(tmp_path):
    """legosnark_amd/shim/checks/spmd_prover_check.cc: the reference's unchanged multiExpMA called
    on this rank's chunk with a communicator set (file bootstrap) returns the whole sum -- the
    C++ route to more than one GPU.  One rank here (RCCL wants one device per rank); the same
    binary runs with WORLD_SIZE = N, RANK = r, LSA_DEVICE = r on an N-GPU node."""
    exe = os.path.join(BIN, "spmd_prover_check")
    require_binary(exe)
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LSA_COMM_FILE=str(tmp_path / "comm.id"))
    r = subprocess.run([exe, "12"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600, text=True)
    assert r.returncode == 0, r.stdout[-2000:]
    assert '"matches_single_gpu": true' in r.stdout


def test_shim_stats_and_call_trace_account_for_a_run():
    """LSA_SHIM_STATS=1: the shim's exit line (calls / items / ms per kind of library call, the MSM host path's split,
    the process's wall time); LSA_TRACE=1: one line per host-facing call from the library.  On `hadamard 8` the
    two must agree on the number of MSMs, the time inside must be below the wall time, and every MSM of the prover's
    ladder on the 2^8 + 1 generator copies ... is too small for the CRS cache (so no hits are reported)."""
    import json
    exe = os.path.join(BIN, "hadamard")
    require_binary(exe)
    env = dict(os.environ, LSA_SEED="7", LSA_SHIM_STATS="1", LSA_TRACE="1")
    r = subprocess.run([exe, "8"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stderr.splitlines() if l.startswith('{"lsa_shim_stats"')]
    assert len(lines) == 1, r.stderr[-2000:]
    st = json.loads(lines[0])["lsa_shim_stats"]
    traced = [l.split() for l in r.stderr.splitlines() if l.startswith("[lsa] ")]
    n_msm = sum(1 for t in traced if t[1] in ("msm_g1", "msm_g2"))
    assert n_msm == st["msm_g1"]["calls"] + st["msm_g2"]["calls"] > 0
    assert sum(1 for t in traced if t[1] == "pairing_terms") == st["pairing"]["calls"] > 0
    assert 0 < st["inside_ms"] < st["process_ms"]
    hp = st["msm_host_path"]
    assert hp["kernels_ms"] > 0 and hp["cache_hits"] == 0 and hp["on_pre_shifted_copies"] == 0
    assert st["scalar_mul_host"]["calls"] > 0 and st["g2_precompute"]["calls"] > 0
    # the reference's own output is untouched by either switch
    assert "##had_sc TOTAL Prove" in r.stdout and "lsa_shim_stats" not in r.stdout


# ---- the reference's -DMULTICORE=ON configuration (/root/reference/CMakeLists.txt:35-39,57-59,78-80)

@pytest.mark.parametrize("bindir", [BIN_MC, CMAKE_BIN_MC], ids=["makefile", "cmake"])
def test_multicore_cplink_verifies_with_eight_chunks(bindir):
    """multiExpMA passes chunks = omp_get_max_threads() (src/utils/globl.h:67-72) and prints it; the unchanged cplink
    must still pass its own MYREQUIRE(verify): prover MSM (chunks = 8) <-> keygen <-> pairings."""
    r = run("cplink", bindir=bindir, OMP_NUM_THREADS="8")
    assert r.returncode == 0, r.stdout[-2000:]
    assert "NCHUNKS : 8" in r.stdout and "NCHUNKS : 1" not in r.stdout
    assert "Error!" not in r.stdout
    r = run("cplink", bindir=bindir, OMP_NUM_THREADS="3")
    assert r.returncode == 0 and "NCHUNKS : 3" in r.stdout, r.stdout[-2000:]


def test_multicore_hadamard_and_pairing_check():
    """hadamard 12 runs lipmaa.cc's four `#pragma omp parallel for` loops over the shim's Fr on eight threads (its
    Lipmaa verifier and the sumcheck verifiers return their own verdicts); pairing_check's failure count stays 0."""
    import json
    r = run("hadamard", "12", bindir=BIN_MC, OMP_NUM_THREADS="8")
    assert r.returncode == 0, r.stdout[-2000:]
    assert "had_lipmaa Prove" in r.stdout and "NCHUNKS : 8" in r.stdout
    r = run("hadamard", "10", bindir=CMAKE_BIN_MC, OMP_NUM_THREADS="8")
    assert r.returncode == 0, r.stdout[-2000:]
    r = run("matrixsc", "3", bindir=BIN_MC, OMP_NUM_THREADS="8")
    assert r.returncode == 0, r.stdout[-2000:]
    r = run("pairing_check", "6", bindir=BIN_MC, seed="11", OMP_NUM_THREADS="8")
    assert r.returncode == 0, r.stdout[-2000:]
    assert json.loads(r.stdout.strip().splitlines()[-1])["failures"] == 0
    r = run("keygen_check", "10", "128", bindir=BIN_MC, OMP_NUM_THREADS="8")
    assert r.returncode == 0 and '"mismatches": 0' in r.stdout, r.stdout[-2000:]


@pytest.mark.parametrize("bindir,threads", [(BIN_MC, "8"), (BIN_MC, "2"), (BIN, "1")], ids=["multicore-8", "multicore-2", "single"])
def test_omp_check_same_points_whatever_the_thread_count(bindir, threads):
    """legosnark_amd/shim/checks/omp_check.cc: multiExpMA(chunks = threads) returns the point multi_exp(chunks = 1)
    and the host sum return; lipmaa.cc's loop bodies, random_element, host scalar multiplications, one deferred GT value
    read from every thread, and MSMs issued from several threads at once all equal their serial results."""
    import json
    r = run("omp_check", "3000", bindir=bindir, OMP_NUM_THREADS=threads)
    assert r.returncode == 0, r.stdout[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])["omp_check"]
    assert line["failures"] == 0 and line["threads"] == int(threads)
    assert line["multicore"] is (bindir == BIN_MC)
    assert ("NCHUNKS : %s" % threads) in r.stdout

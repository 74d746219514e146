// keygen_check -- CPlink key generation's matrix-in-the-exponent, reference loop vs batched call.
//
// Compiled against LegoSNARK's own headers and liblegobasic.a (the reference's unchanged
// sources, read in place at build time) plus the libff-compatible shim: builds the CPlink
// relation matrix exactly like makeLinkingRel (/root/reference/src/examples/cplink.cc:21-41),
// runs the reference's own column loop (mtxmultiexp -> simplesparsemexp -> sparsemexpG, each a
// tiny multi_exp forwarded to the GPU plus host scalar multiplications) on the first `sample`
// columns, runs libff::lsa_mtxmultiexp (one batched call) on the whole matrix, and compares.
//   usage: keygen_check [log2 N = 10] [sample columns = 64]
#include <chrono>
#include <cstdio>
#include <cstdlib>

#include "globl.h"
#include "matrix.h"
#include "sparsemexp.h"
#include "util.h"

using namespace std;

int main(int argc, char **argv) {
    default_ec_pp::init_public_params();
    const size_t logn = argc > 1 ? atoi(argv[1]) : 10;
    const size_t N = size_t(1) << logn;
    size_t sample = argc > 2 ? atoi(argv[2]) : 64;

    auto rnd_points = [](size_t n) {
        vector<LFr> e(n);
        for (auto &x : e) x = LFr::random_element();
        return cputil::simpleBatchExp<LG1, LFr>(LG1::one(), e);     // fixed-base batch_exp on the GPU
    };
    vector<LG1> h = rnd_points(1), bases1 = rnd_points(N), F = rnd_points(N + 1);

    const size_t nCols = 2 * N + 2;
    vector<ColG1> M(nCols);
    insertAsColMajor(0, 0, h[0], M);
    insertRowAsColMajor(0, 2, bases1, M);
    insertRowAsColMajor(1, 1, F, M);

    vector<LFr> k = {LFr::random_element(), LFr::random_element()};

    auto t0 = chrono::steady_clock::now();
    vector<LG1> batched;
    libff::lsa_mtxmultiexp(batched, k, M);
    const double ms_batched = chrono::duration<double, milli>(chrono::steady_clock::now() - t0).count();

    if (sample > nCols) sample = nCols;
    t0 = chrono::steady_clock::now();
    size_t bad = 0;
    for (size_t j = 0; j < sample; j++) {
        // spread the sample over filled and empty columns
        const size_t col = j * (nCols / sample);
        LG1 ref = simplesparsemexp(M[col], k);
        if (!(ref == batched[col])) bad++;
    }
    const double ms_ref = chrono::duration<double, milli>(chrono::steady_clock::now() - t0).count();

    printf("{\"N\": %zu, \"columns\": %zu, \"batched_ms\": %.3f, \"reference_loop_sampled_columns\": %zu, "
           "\"reference_loop_ms_per_column\": %.4f, \"mismatches\": %zu}\n",
           N, nCols, ms_batched, sample, ms_ref / sample, bad);
    return bad ? 1 : 0;
}

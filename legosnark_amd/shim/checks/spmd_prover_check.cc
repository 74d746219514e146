// spmd_prover_check -- the reference's multiExpMA (/root/reference/src/utils/globl.h:63-78) on
// more than one GPU from C++: one process per GPU, each passing ITS chunk of the vectors.
//
// Compiled against LegoSNARK's own headers (globl.h, util.h) plus the libff-compatible shim.
// Every rank builds the same CRS P (N + 2 points, as SubspaceSnark::prove sees it,
// src/gadgets/subspace.cc:78-85) and witness w, takes the contiguous range libff's multi_exp
// would give chunk `rank` (lsa_shard_range), and calls the UNCHANGED multiExpMA on that chunk:
// inside a libff::lsa_sharded_scope (and with a communicator set) the shim's multi_exp forwards to lsa_g1_msm_sharded (local MSM +
// one RCCL all-gather of the 96-byte partials + sum), so every rank receives the proof element.
// Rank 0 compares it with the single-GPU multiExpMA over the whole vectors.
//
//   one GPU :  spmd_prover_check [log2 N = 12]
//   N GPUs  :  for r in 0..N-1: LSA_DEVICE=r RANK=r WORLD_SIZE=N LSA_COMM_FILE=/tmp/x.id spmd_prover_check 20 &
// (rank 0 writes the 128-byte RCCL id to LSA_COMM_FILE, the others poll for it.)
#include <cstdio>
#include <cstdlib>
#include <unistd.h>

#include "globl.h"
#include "util.h"

using namespace std;

int main(int argc, char **argv) {
    const int rank = getenv("RANK") ? atoi(getenv("RANK")) : 0;
    const int world = getenv("WORLD_SIZE") ? atoi(getenv("WORLD_SIZE")) : 1;
    const size_t logn = argc > 1 ? atoi(argv[1]) : 12;
    const size_t n = (size_t(1) << logn) + 2;
    default_ec_pp::init_public_params();                      // lsa_init(LSA_DEVICE)

    // identical inputs on every rank: scalars from a fixed stream, P_i = e_i * G by batch_exp
    vector<LFr> e(n), w(n);
    LFr x = LFr(7L), y = LFr(11L);
    for (size_t i = 0; i < n; i++) { x = x * x + LFr(3L); y = y * x + LFr(5L); e[i] = x; w[i] = y; }
    w[0] = LFr::zero();                                       // rH = 0 (src/prototools/commit.h:152)
    vector<LG1> P = cputil::simpleBatchExp<LG1, LFr>(LG1::one(), e);

    LG1 whole = multiExpMA<LG1>(P, w);                        // single GPU, no communicator yet

    char path[256];
    snprintf(path, sizeof path, "%s", getenv("LSA_COMM_FILE") ? getenv("LSA_COMM_FILE") : "/tmp/lsa_spmd_check.id");
    if (rank == 0) unlink(path);
    if (lsa_comm_init_file(rank, world, path, 120) != 0) { fprintf(stderr, "comm init failed: %s\n", lsa_last_error()); return 2; }

    size_t lo, hi;
    lsa_shard_range(n, world, rank, &lo, &hi);
    vector<LG1> P_chunk(P.begin() + lo, P.begin() + hi);
    vector<LFr> w_chunk(w.begin() + lo, w.begin() + hi);
    LG1 sharded, again;
    {
        libff::lsa_sharded_scope spmd;                         // inside: this rank's chunk in, the sum over all ranks out
        sharded = multiExpMA<LG1>(P_chunk, w_chunk);          // unchanged reference call
        again = multiExpMA<LG1>(P_chunk, w_chunk);            // second call: CRS cache hit on the chunk
    }
    // outside the scope a multi_exp over replicated inputs stays a whole sum on this GPU, communicator or not
    LG1 replicated = multiExpMA<LG1>(P, w);

    const bool ok = sharded == whole && again == whole && replicated == whole;
    printf("{\"rank\": %d, \"world\": %d, \"n\": %zu, \"chunk\": [%zu, %zu], \"matches_single_gpu\": %s}\n", rank, world, n, lo, hi,
           ok ? "true" : "false");
    lsa_comm_destroy();
    if (rank == 0) unlink(path);
    return ok ? 0 : 1;
}

// tools/native/h2d_vectors.cc -- GPU box: lsa_g1_msm on std::vector scalars the way CPPoly::prove builds them
// (/root/reference/src/gadgets/poly.h:77-86: a new vector per call, halving sizes), host-path split per call.
//   g++ -std=c++17 -O2 -I include tools/native/h2d_vectors.cc -o build/h2d_vectors -Llegosnark_amd -llegosnark_amd -Wl,-rpath,$PWD/legosnark_amd -Wl,-rpath,/opt/rocm/lib
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
#include "legosnark_amd.h"
#include "../../legosnark_amd/csrc/fp.h"
struct Fr32 { uint64_t w[4]; };
struct G1J { uint64_t w[12]; };
int main(int argc, char **argv) {
    const bool keep = argc > 1 && strcmp(argv[1], "keep") == 0;     // keep every vector alive: no munmap between calls
    std::vector<std::vector<Fr32>> held;
    if (lsa_init(0)) { fprintf(stderr, "%s\n", lsa_last_error()); return 1; }
    const size_t N = 1 << 20;
    // bases: copies of the generator (Montgomery (1, 2, 1))
    std::vector<G1J> bases(N);
    {
        const lsa::Fq one = lsa::Fq::one(), two = one + one;     // the generator (1, 2, 1), Montgomery form
        G1J gen;
        memcpy(&gen.w[0], &one, 32);
        memcpy(&gen.w[4], &two, 32);
        memcpy(&gen.w[8], &one, 32);
        for (auto &b : bases) b = gen;
    }
    std::vector<Fr32> v(N);
    for (size_t i = 0; i < N; i++) { v[i].w[0] = i * 0x9E3779B97F4A7C15ull; v[i].w[1] = i; v[i].w[2] = ~i; v[i].w[3] = i & 0xfffffff; }   // < 2^252: reduced
    G1J out;
    for (int rep = 0; rep < 3; rep++) {
        double h2d = 0, fp = 0, bp = 0, ms = 0, tot = 0;
        size_t bytes = 0;
        for (size_t n = N; n >= 4096; n >>= 1) {
            std::vector<Fr32> tmp(n);
            for (size_t i = 0; i < n; i++) tmp[i] = v[i];
            if (lsa_g1_msm(bases.data(), tmp.data(), n, 1, &out)) { fprintf(stderr, "%s\n", lsa_last_error()); return 1; }
            lsa_host_stats st;
            lsa_msm_host_stats(&st);
            if (rep == 2) printf("n=%zu h2d %.3f fp_wait %.3f bases %.3f msm %.3f total %.3f hit %d\n", n, st.h2d_scalars_ms, st.fingerprint_wait_ms, st.bases_prepare_ms, st.msm_ms, st.total_ms, st.cache_hit);
            if (keep) held.push_back(std::move(tmp));
            h2d += st.h2d_scalars_ms; fp += st.fingerprint_wait_ms; bp += st.bases_prepare_ms; ms += st.msm_ms; tot += st.total_ms;
            bytes += n * 32;
        }
        printf("rep %d: %.1f MB scalars, h2d %.2f ms (%.1f GB/s), fingerprint wait %.2f, bases %.2f, kernels %.2f, total %.2f\n", rep, bytes / 1e6, h2d,
               bytes / 1e6 / h2d, fp, bp, ms, tot);
    }
    return 0;
}

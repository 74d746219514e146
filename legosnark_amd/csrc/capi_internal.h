// legosnark_amd/csrc/capi_internal.h -- state shared by the translation units that implement the
// C-ABI (capi.hip: single-GPU entry points; comm.hip: the multi-GPU exchange step).
#pragma once
#include <hip/hip_runtime.h>
#include "msm.h"

namespace lsa {

struct State {
    bool ready = false;
    int device = -1;
    hipStream_t stream = nullptr;
    void *d_result = nullptr;      // 512-byte device slot for MSM / pairing results
    void *h_result = nullptr;      // pinned host mirror
};
extern State g;

#define HIPCHK(x)                                                                      \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            set_error("%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
            return LSA_ERR_HIP;                                                        \
        }                                                                              \
    } while (0)

int require_ready();
void comm_release();               // comm.hip: called by lsa_shutdown

}  // namespace lsa

struct lsa_bases {
    void *d_aff = nullptr;   // prepared bases (msm_base_bytes(group) each); with a table: window-major copies
    size_t n = 0;
    int group = 1;           // 1 = G1, 2 = G2
    size_t table_stride = 0; // n when the pre-shifted window copies 2^(s_k)*P are resident, else 0
};

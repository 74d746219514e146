// dev harness: quad_dbl / quad_add on the device vs the sequential host formulas
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <random>
#include "quad29.h"
using namespace lsa;
__global__ void k(const XYZZ29* in, XYZZ29* out) {
    unsigned q = threadIdx.x & 3;
    if (threadIdx.x >= 4) return;
    XYZZ29 a = in[0], b = in[1];
    XYZZ29 d = quad_dbl(a, q);
    XYZZ29 s = quad_add(a, b, q);
    XYZZ29 d2 = xyzz29_dbl(a);
    XYZZ29 s2 = xyzz29_add(a, b);
    if (q == 0) { out[0] = d; out[1] = s; out[2] = d2; out[3] = s2; }
    if (q == 1) { out[4] = d; out[5] = s; }
    if (q == 3) { out[6] = d; out[7] = s; }
}
static bool eq(const XYZZ29&a, const XYZZ29&b){ for(int i=0;i<9;i++) if(a.X.l[i]!=b.X.l[i]||a.Y.l[i]!=b.Y.l[i]||a.ZZ.l[i]!=b.ZZ.l[i]||a.ZZZ.l[i]!=b.ZZZ.l[i]) return false; return true; }
int main(){
    std::mt19937_64 rng(1);
    Aff29 g{F29::from_mont256(Fq::from_u32(1)).canonical(), F29::from_mont256(Fq::from_u32(2)).canonical()};
    XYZZ29 a = xyzz29_madd(XYZZ29::inf(), g); a = xyzz29_dbl(a); a = xyzz29_madd(a, g);   // 3G
    XYZZ29 b = xyzz29_dbl(xyzz29_dbl(a));                                                 // 12G
    XYZZ29 h[2] = {a, b}, *din, *dout, o[8];
    hipMalloc(&din, sizeof h); hipMalloc(&dout, sizeof o);
    hipMemcpy(din, h, sizeof h, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, din, dout);
    hipMemcpy(o, dout, sizeof o, hipMemcpyDeviceToHost);
    XYZZ29 hd = xyzz29_dbl(a), hs = xyzz29_add(a, b);
    printf("dev seq dbl == host: %d, dev seq add == host: %d\n", eq(o[2], hd), eq(o[3], hs));
    printf("quad dbl q0 == host: %d (q1 %d, q3 %d)\n", eq(o[0], hd), eq(o[4], hd), eq(o[6], hd));
    printf("quad add q0 == host: %d (q1 %d, q3 %d)\n", eq(o[1], hs), eq(o[5], hs), eq(o[7], hs));
    const char* nm[4]={"X","Y","ZZ","ZZZ"};
    for (int c=0;c<4;c++){ const F29* pq=&o[0].X+c; const F29* ph=&hd.X+c; const F29* p3=&o[6].X+c; printf("%s q0:", nm[c]); for(int i=0;i<9;i++) printf(" %08x", pq->l[i]); printf("\n   q3:"); for(int i=0;i<9;i++) printf(" %08x", p3->l[i]); printf("\n host:"); for(int i=0;i<9;i++) printf(" %08x", ph->l[i]); printf("\n"); }
    return 0;
}

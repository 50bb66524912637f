// tools/lab/dpp_combine_repro.hip -- reproduces / pins the ROCm 7.2 DPP-combiner miscompile worked around in
// plonky_amd/csrc/ecz_coop.cuh (quad_bcast_u32): without the empty asm after the v_mov_b32_dpp, the y coordinate
// of xyzzz_dbl_q / xyzzz_add_q (= bcast<0>(r) - bcast<1>(r)) is right only in the lane that owns the subtrahend.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iplonky_amd/csrc -o dpp_repro tools/lab/dpp_combine_repro.hip
// Run on the GPU: prints one byte per lane (low nibble: doubling, high nibble: addition; bit k = coordinate k differs
// from the one-lane arithmetic), all-lanes-active and divergent; all zeros on a healthy build.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include "ecz_coop.cuh"
#include "field_params.cuh"
using namespace plk;
using FP = TweedledeeBaseParams;
__device__ Fz<FP> mk(uint32_t seed) {
    Fz<FP> r;
    for (int i = 0; i < 9; ++i) r.l[i] = (seed * 2654435761u + i * 40503u * seed + i) & 0x1fffffffu;
    r.l[8] &= 0x3fffff;  // < p
    return r;
}
__device__ uint32_t diff(const XyzzZ<FP>& a, const XyzzZ<FP>& b) {
    if (a.inf || b.inf) return a.inf != b.inf;
    const Fz<FP> one = fz_one_rprime<FP>();
    uint32_t bad = 0;
    const Fz<FP>* ca[4] = {&a.x, &a.y, &a.zz, &a.zzz};
    const Fz<FP>* cb[4] = {&b.x, &b.y, &b.zz, &b.zzz};
    for (int k = 0; k < 4; ++k) {
        const Fe<FP> u = fz_to_fe_canonical<FP>(fz_mul<FP>(*ca[k], one)), v = fz_to_fe_canonical<FP>(fz_mul<FP>(*cb[k], one));
        for (int i = 0; i < 8; ++i) if (u.v[i] != v.v[i]) bad |= 1u << k;
    }
    return bad;
}
__global__ void k(uint32_t* out, int divergent) {
    const int tid = threadIdx.x, ql = tid & 3, quad = tid >> 2;
    XyzzZ<FP> a, b;
    a.x = mk(quad * 8 + 1); a.y = mk(quad * 8 + 2); a.zz = mk(quad * 8 + 3); a.zzz = mk(quad * 8 + 4); a.inf = false;
    b.x = mk(quad * 8 + 5); b.y = mk(quad * 8 + 6); b.zz = mk(quad * 8 + 7); b.zzz = mk(quad * 8 + 8); b.inf = false;
    uint32_t res = 0;
    if (!divergent || (quad & 3) == 1) {
        res = diff(xyzzz_dbl_q<FP>(a, ql), xyzzz_dbl<FP>(a));
        res |= diff(xyzzz_add_q<FP>(a, b, ql), xyzzz_add<FP>(a, b)) << 4;
    }
    out[tid] = res;
}
int main() {
    uint32_t* d; hipMalloc(&d, 64 * 4);
    for (int dv = 0; dv < 2; ++dv) {
        k<<<1, 64>>>(d, dv);
        uint32_t h[64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("divergent=%d:", dv);
        for (int i = 0; i < 32; ++i) printf(" %02x", h[i]);
        printf("\n");
    }
    return 0;
}

// Host-side harness: compiles the device field arithmetic (plonky_amd/csrc/fp.cuh, plain C++ when not
// under hipcc) with g++ so the CPU test suite can sweep it against Python integers.
#include <cstddef>
#include "../plonky_amd/csrc/fp.cuh"
using namespace plk;

template <class P> static void run(int op, const uint32_t* a, const uint32_t* b, uint32_t* out, size_t n) {
    for (size_t i = 0; i < n; ++i) {
        Fe<P> x, y, r;
        for (int k = 0; k < P::NL; ++k) { x.v[k] = a[i * P::NL + k]; y.v[k] = b[i * P::NL + k]; }
        switch (op) {
            case 0: r = fe_add<P>(x, y); break;
            case 1: r = fe_sub<P>(x, y); break;
            case 2: r = fe_mul<P>(x, y); break;
            case 3: r = fe_mul_cios<P>(x, y); break;
            case 4: r = fe_sqr<P>(x); break;
            case 5: r = fe_inv<P>(x); break;
            case 6: r = fe_half<P>(x); break;
            case 8: r = fe_inv_eea<P>(x); break;
            default: r = fe_neg<P>(x); break;
        }
        for (int k = 0; k < P::NL; ++k) out[i * P::NL + k] = r.v[k];
    }
}
extern "C" int fp_host_op(int field, int op, const uint32_t* a, const uint32_t* b, uint32_t* out, size_t n) {
    switch (field) {
        case 0: run<TweedledeeBaseParams>(op, a, b, out, n); return 0;
        case 1: run<TweedledumBaseParams>(op, a, b, out, n); return 0;
        case 2: run<Bls12377ScalarParams>(op, a, b, out, n); return 0;
        case 3: run<Bls12377BaseParams>(op, a, b, out, n); return 0;
    }
    return -1;
}

// fieldops.hip -- element-wise field kernels behind plk_field_op: the device counterpart of
// the reference's `test_arithmetic!` sweep (src/field/field.rs:618-780) so that the parity
// tests can run the HIP field arithmetic itself against the oracle on the edge-value inputs.
#include "common.h"
#include "fp.cuh"

namespace plk {

template <class P> __global__ void k_field_op(int op, const uint4* a, const uint4* b, uint4* out, size_t count) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    constexpr int W = P::NL / 4;
    Fe<P> x = fe_load<P>(a + i * W), y = fe_zero<P>(), r;
    if (op <= 2) y = fe_load<P>(b + i * W);
    switch (op) {
        case 0: r = fe_add<P>(x, y); break;
        case 1: r = fe_sub<P>(x, y); break;
        case 2: r = fe_mul<P>(x, y); break;
        case 3: r = fe_neg<P>(x); break;
        case 4: r = fe_sqr<P>(x); break;
        case 5: r = fe_inv<P>(x); break;
        case 6: r = fe_to_canonical<P>(x); break;
        case 7: r = fe_from_canonical<P>(x); break;
        case 8: r = fe_inv_eea<P>(x); break;       // the reference's Euclid (bigint_inverse.rs:6-55)
        default: r = fe_inv_safegcd<P>(x); break;  // what the kernels use
    }
    fe_store<P>(out + i * W, r);
}

template <class P> static int field_op_t(int op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t count) {
    const size_t bytes = count * P::NL * 4;
    DevBuf da, db, dout;
    PLK_TRY(da.alloc(bytes));
    PLK_TRY(db.alloc(bytes));
    PLK_TRY(dout.alloc(bytes));
    PLK_HIP_TRY(hipMemcpy(da.p, a, bytes, hipMemcpyHostToDevice));
    if (op <= 2) PLK_HIP_TRY(hipMemcpy(db.p, b, bytes, hipMemcpyHostToDevice));
    if (count) {
        k_field_op<P><<<(unsigned)((count + 127) / 128), 128>>>(op, (const uint4*)da.p, (const uint4*)db.p, (uint4*)dout.p, count);
        PLK_HIP_TRY(hipGetLastError());
    }
    PLK_HIP_TRY(hipMemcpy(out, dout.p, bytes, hipMemcpyDeviceToHost));
    return PLK_OK;
}

int field_op_impl(int field, int op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t count) {
    if (op < 0 || op > 9) return set_error(PLK_ERR_INVALID_ARG, "bad field op %d", op);
    if (!a || !out || (op <= 2 && !b)) return set_error(PLK_ERR_INVALID_ARG, "null pointer");
    PLK_TRY(ensure_device());
    switch (field) {
        case PLK_FIELD_TWEEDLEDEE_BASE: return field_op_t<TweedledeeBaseParams>(op, a, b, out, count);
        case PLK_FIELD_TWEEDLEDUM_BASE: return field_op_t<TweedledumBaseParams>(op, a, b, out, count);
        case PLK_FIELD_BLS12_377_SCALAR: return field_op_t<Bls12377ScalarParams>(op, a, b, out, count);
        case PLK_FIELD_BLS12_377_BASE: return field_op_t<Bls12377BaseParams>(op, a, b, out, count);
    }
    return set_error(PLK_ERR_INVALID_ARG, "bad field id %d", field);
}

}  // namespace plk

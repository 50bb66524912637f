// fieldops.hip -- element-wise field kernels behind plk_field_op: the device counterpart of
// the reference's `test_arithmetic!` sweep (src/field/field.rs:618-780) so that the parity
// tests can run the HIP field arithmetic itself against the oracle on the edge-value inputs.
#include "common.h"
#include "fp.cuh"
#include "fz.cuh"

namespace plk {

template <class P> __global__ void k_field_op(int op, const uint4* a, const uint4* b, uint4* out, size_t count) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (op == 11) {  // the one-lane inversion: ONE active lane per wave, element = wave index
        if (threadIdx.x & 63) return;
        i >>= 6;
    }
    if (i >= count) return;
    constexpr int W = P::NL / 4;
    Fe<P> x = fe_load<P>(a + i * W), y = fe_zero<P>(), r;
    if (op <= 2 || op >= 12) y = fe_load<P>(b + i * W);
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
        case 10: r = fe_inv_safegcd_var<P>(x); break;  // the data-dependent form (runs of division steps)
        case 11: r = fe_inv_safegcd_one_lane<P>(x); break;  // what the end of an MSM calls: MODE 1 (variable-time, vector unit) in the shipped build - the same routine as op 10, kept as the entry point of the one-lane form; its scalar-unit variants (-DPLK_ONE_LANE_INV_MODE=2 / 3) are tuning builds only
        case 12: {
            // fz_mul_add2 with every limb below the top one at the largest value its callers pass (fz.cuh; the limb patterns of
            // tests/fp_host_harness.cpp op 24, rebuilt from the same input words by tests/test_gpu_parity.py)
            constexpr int NZ = FzCfg<P>::NZ;
            constexpr bool SMALL = NZ <= 9;
            const uint32_t la = SMALL ? (1u << 29) + 7u : (1u << 29) + (1u << 27), lb = SMALL ? 0x80000000u : la, lc = SMALL ? (1u << 30) : la,
                           ld = (1u << 29) - 1u, top = SMALL ? (1u << 25) : 3u;
            Fz<P> a, b, c, d;
            for (int k = 0; k < NZ - 1; ++k) {
                a.l[k] = la - (x.v[0] & 7u);
                b.l[k] = lb - (y.v[0] & 0xffu);
                c.l[k] = lc - (x.v[2] & 0xffu);
                d.l[k] = ld - (y.v[1] & 0xffu);
            }
            a.l[NZ - 1] = top + (x.v[1] & 0xffffu);
            b.l[NZ - 1] = c.l[NZ - 1] = top;
            d.l[NZ - 1] = SMALL ? (1u << 22) : 1u;
            r = fz_to_fe_canonical<P>(fz_mul<P>(fz_mul_add2<P>(a, b, c, d), fz_one_rprime<P>()));
        } break;
        case 13: {  // the same sum through the column accumulators of the quotient numerator (FzWide): a b + c d + (a as a plain value)
            const Fz<P> xz = fz_from_fe<P>(x), yz = fz_from_fe<P>(y);
            FzWide<P> w;
            fz_wide_clear<P>(w);
            fz_wide_mac<P>(w, xz, yz);
            fz_wide_mac<P>(w, fz_add<P>(xz, yz), yz);
            fz_wide_add<P>(w, xz);
            r = fz_to_fe_canonical<P>(fz_mul<P>(fz_wide_reduce<P>(w), fz_one_rprime<P>()));
        } break;
        default: r = fe_inv_safegcd<P>(x); break;      // what the kernels use
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
    if (op <= 2 || op >= 12) PLK_HIP_TRY(hipMemcpy(db.p, b, bytes, hipMemcpyHostToDevice));
    if (count) {
        const size_t lanes = op == 11 ? count * 64 : count;
        k_field_op<P><<<(unsigned)((lanes + 127) / 128), 128>>>(op, (const uint4*)da.p, (const uint4*)db.p, (uint4*)dout.p, count);
        PLK_HIP_TRY(hipGetLastError());
    }
    PLK_HIP_TRY(hipMemcpy(out, dout.p, bytes, hipMemcpyDeviceToHost));
    return PLK_OK;
}

// ---------------------------------------------------------------------------------------------
// Field::batch_multiplicative_inverse / _opt (field.rs:223-278) and ProjectivePoint::batch_to_affine (curve.rs:216-232)
// ---------------------------------------------------------------------------------------------
// Montgomery's trick, one chain of BATCH_PER_LANE elements per lane: prefix products forward, ONE division-step inversion,
// back substitution - 3 multiplications per element plus 1/8 of an inversion (an inversion is ~36 multiplications).
// Sharing one inversion across a wave would not be cheaper: a wave pays for an instruction whether one lane or all 64
// execute it, so 64 private inversions cost the SIMD exactly what one shared inversion would, without the two
// cross-lane product scans (12 more multiplications per lane).  Elements of a lane are strided by the grid so that every load
// and store of a wave is contiguous.  Zero elements (no inverse: `None` in _opt) are replaced by 1 in the chain, come back
// as 0 and are flagged.
constexpr int BATCH_PER_LANE = 8;
template <class P>
__global__ void __launch_bounds__(128) k_batch_inverse(const uint4* __restrict__ x, uint4* __restrict__ out, uint8_t* __restrict__ is_zero,
                                                       unsigned* __restrict__ zero_count, size_t count) {
    constexpr int W = P::NL / 4;
    const size_t lanes = (size_t)gridDim.x * blockDim.x;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    Fe<P> v[BATCH_PER_LANE], pre[BATCH_PER_LANE];
    bool zf[BATCH_PER_LANE];
    Fe<P> run = fe_one<P>();
    unsigned zeros = 0;
#pragma unroll
    for (int k = 0; k < BATCH_PER_LANE; ++k) {
        const size_t i = t + (size_t)k * lanes;
        v[k] = i < count ? fe_load<P>(x + i * W) : fe_one<P>();
        zf[k] = fe_is_zero<P>(v[k]);
        if (zf[k]) {
            v[k] = fe_one<P>();
            ++zeros;
        }
        pre[k] = run;
        run = fe_mul<P>(run, v[k]);
    }
    Fe<P> inv = fe_inv_safegcd<P>(run);
#pragma unroll
    for (int k = BATCH_PER_LANE - 1; k >= 0; --k) {
        const size_t i = t + (size_t)k * lanes;
        const Fe<P> r = fe_mul<P>(inv, pre[k]);
        inv = fe_mul<P>(inv, v[k]);
        if (i < count) {
            fe_store<P>(out + i * W, zf[k] ? fe_zero<P>() : r);
            if (is_zero) is_zero[i] = zf[k] ? 1 : 0;
        }
    }
    if (zeros && zero_count) atomicAdd(zero_count, zeros);
}

// homogeneous projective (X : Y : Z) + zero flag -> affine (x, y) = (X / Z, Y / Z) + zero flag, the reference's own
// ProjectivePoint / AffinePoint (curve.rs:74-78,176-181): z_inv from batch_multiplicative_inverse_opt (curve.rs:219)
template <class P>
__global__ void __launch_bounds__(128) k_batch_to_affine(const uint4* __restrict__ xyz, const uint8_t* __restrict__ pzero, uint4* __restrict__ out_xy,
                                                         uint8_t* __restrict__ out_zero, size_t count) {
    constexpr int W = P::NL / 4;
    const size_t lanes = (size_t)gridDim.x * blockDim.x;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    Fe<P> z[BATCH_PER_LANE], pre[BATCH_PER_LANE];
    bool ident[BATCH_PER_LANE];
    Fe<P> run = fe_one<P>();
#pragma unroll
    for (int k = 0; k < BATCH_PER_LANE; ++k) {
        const size_t i = t + (size_t)k * lanes;
        ident[k] = true;
        z[k] = fe_one<P>();
        if (i < count) {
            z[k] = fe_load<P>(xyz + (i * 3 + 2) * W);
            // `zero` is the reference's flag (curve.rs:222-224); a non-flagged point with Z = 0 would make the reference panic on unwrap()
            ident[k] = (pzero && pzero[i]) || fe_is_zero<P>(z[k]);
            if (ident[k]) z[k] = fe_one<P>();
        }
        pre[k] = run;
        run = fe_mul<P>(run, z[k]);
    }
    Fe<P> inv = fe_inv_safegcd<P>(run);
#pragma unroll
    for (int k = BATCH_PER_LANE - 1; k >= 0; --k) {
        const size_t i = t + (size_t)k * lanes;
        const Fe<P> zi = fe_mul<P>(inv, pre[k]);
        inv = fe_mul<P>(inv, z[k]);
        if (i < count) {
            Fe<P> ax = fe_zero<P>(), ay = fe_zero<P>();  // AffinePoint::ZERO = (0, 0, zero = true), curve.rs:81-85
            if (!ident[k]) {
                ax = fe_mul<P>(fe_load<P>(xyz + (i * 3) * W), zi);
                ay = fe_mul<P>(fe_load<P>(xyz + (i * 3 + 1) * W), zi);
            }
            fe_store<P>(out_xy + (i * 2) * W, ax);
            fe_store<P>(out_xy + (i * 2 + 1) * W, ay);
            out_zero[i] = ident[k] ? 1 : 0;
        }
    }
}

static unsigned batch_blocks(size_t count) {
    const size_t lanes = (count + BATCH_PER_LANE - 1) / BATCH_PER_LANE;
    return (unsigned)((lanes + 127) / 128);
}

int field_batch_inverse_dev_impl(int field, const void* d_x, void* d_out, void* d_is_zero, unsigned* d_zero_count, size_t count, hipStream_t stream) {
    if (count == 0) return PLK_OK;
    if (!d_x || !d_out) return set_error(PLK_ERR_INVALID_ARG, "null device pointer");
    PLK_TRY(ensure_device());
    const unsigned blocks = batch_blocks(count);
    switch (field) {
#define CASE(ID, P) \
    case ID: k_batch_inverse<P><<<blocks, 128, 0, stream>>>((const uint4*)d_x, (uint4*)d_out, (uint8_t*)d_is_zero, d_zero_count, count); break;
        CASE(PLK_FIELD_TWEEDLEDEE_BASE, TweedledeeBaseParams)
        CASE(PLK_FIELD_TWEEDLEDUM_BASE, TweedledumBaseParams)
        CASE(PLK_FIELD_BLS12_377_SCALAR, Bls12377ScalarParams)
        CASE(PLK_FIELD_BLS12_377_BASE, Bls12377BaseParams)
        CASE(PLK_FIELD_PALLAS_BASE, PallasBaseParams)
        CASE(PLK_FIELD_VESTA_BASE, VestaBaseParams)
#undef CASE
        default: return set_error(PLK_ERR_INVALID_ARG, "bad field id %d", field);
    }
    PLK_HIP_TRY(hipGetLastError());
    return PLK_OK;
}

int curve_batch_to_affine_dev_impl(int curve, size_t count, const void* d_xyz, const void* d_zero, void* d_out_xy, void* d_out_zero, hipStream_t stream) {
    if (count == 0) return PLK_OK;
    if (!d_xyz || !d_out_xy || !d_out_zero) return set_error(PLK_ERR_INVALID_ARG, "null device pointer");
    PLK_TRY(ensure_device());
    const unsigned blocks = batch_blocks(count);
    switch (curve) {
#define CASE(ID, P) \
    case ID: k_batch_to_affine<P><<<blocks, 128, 0, stream>>>((const uint4*)d_xyz, (const uint8_t*)d_zero, (uint4*)d_out_xy, (uint8_t*)d_out_zero, count); break;
        CASE(PLK_CURVE_TWEEDLEDEE, TweedledeeBaseParams)
        CASE(PLK_CURVE_TWEEDLEDUM, TweedledumBaseParams)
        CASE(PLK_CURVE_BLS12_377, Bls12377BaseParams)
        CASE(PLK_CURVE_PALLAS, PallasBaseParams)
        CASE(PLK_CURVE_VESTA, VestaBaseParams)
#undef CASE
        default: return set_error(PLK_ERR_INVALID_ARG, "bad curve id %d", curve);
    }
    PLK_HIP_TRY(hipGetLastError());
    return PLK_OK;
}

// ---------------------------------------------------------------------------------------------
// scalar side of an IPA round (halo.rs:63-118): Field::inner_product (field.rs:213-221) and the folds
// halo_a' = u^-1 a_hi + u a_lo, halo_b' = u^-1 b_lo + u b_hi (add_slices of scale_slice, halo.rs:117-118)
// ---------------------------------------------------------------------------------------------
// Field addition is exact, commutative and associative: any summation order gives the reference's value.
template <class P>
__global__ void __launch_bounds__(256) k_inner_product(const uint4* __restrict__ a, const uint4* __restrict__ b, size_t count, uint4* __restrict__ part) {
    constexpr int W = P::NL / 4;
    __shared__ uint4 s_acc[256 * W];
    Fe<P> acc = fe_zero<P>();
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x)
        acc = fe_add<P>(acc, fe_mul<P>(fe_load<P>(a + i * W), fe_load<P>(b + i * W)));
    fe_store<P>(s_acc + threadIdx.x * W, acc);
    __syncthreads();
    for (int d = 128; d >= 1; d >>= 1) {
        if ((int)threadIdx.x < d) {
            acc = fe_add<P>(acc, fe_load<P>(s_acc + (threadIdx.x + d) * W));
            fe_store<P>(s_acc + threadIdx.x * W, acc);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) fe_store<P>(part + blockIdx.x * W, acc);
}
template <class P> __global__ void __launch_bounds__(256) k_sum_parts(const uint4* __restrict__ part, unsigned n, uint4* __restrict__ out) {
    constexpr int W = P::NL / 4;
    __shared__ uint4 s_acc[256 * W];
    Fe<P> acc = fe_zero<P>();
    for (unsigned i = threadIdx.x; i < n; i += 256) acc = fe_add<P>(acc, fe_load<P>(part + i * W));
    fe_store<P>(s_acc + threadIdx.x * W, acc);
    __syncthreads();
    for (int d = 128; d >= 1; d >>= 1) {
        if ((int)threadIdx.x < d) {
            acc = fe_add<P>(acc, fe_load<P>(s_acc + (threadIdx.x + d) * W));
            fe_store<P>(s_acc + threadIdx.x * W, acc);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) fe_store<P>(out, acc);
}
struct FoldScalars {
    uint32_t lo[12], hi[12];
};
template <class P>
__global__ void __launch_bounds__(256) k_fold_slices(const uint4* __restrict__ lo, const uint4* __restrict__ hi, FoldScalars sc, size_t count, uint4* __restrict__ out) {
    constexpr int W = P::NL / 4;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    Fe<P> sl, sh;
#pragma unroll
    for (int k = 0; k < P::NL; ++k) {
        sl.v[k] = sc.lo[k];
        sh.v[k] = sc.hi[k];
    }
    fe_store<P>(out + i * W, fe_add<P>(fe_mul<P>(sl, fe_load<P>(lo + i * W)), fe_mul<P>(sh, fe_load<P>(hi + i * W))));
}

int field_inner_product_dev_impl(int field, const void* d_a, const void* d_b, size_t count, void* d_out, hipStream_t stream) {
    if (!d_out || (count && (!d_a || !d_b))) return set_error(PLK_ERR_INVALID_ARG, "null device pointer");
    PLK_TRY(ensure_device());
    const int L = field_limbs(field);
    if (L < 0) return set_error(PLK_ERR_INVALID_ARG, "bad field id %d", field);
    unsigned blocks = (unsigned)((count + 255) / 256);
    if (blocks > 1024) blocks = 1024;
    if (blocks == 0) blocks = 1;
    void* part = scratch_acquire((size_t)blocks * L * 8, stream);
    if (!part) return PLK_ERR_OOM;
    switch (field) {
#define CASE(ID, P)                                                                                                   \
    case ID:                                                                                                          \
        k_inner_product<P><<<blocks, 256, 0, stream>>>((const uint4*)d_a, (const uint4*)d_b, count, (uint4*)part);   \
        k_sum_parts<P><<<1, 256, 0, stream>>>((const uint4*)part, blocks, (uint4*)d_out);                            \
        break;
        CASE(PLK_FIELD_TWEEDLEDEE_BASE, TweedledeeBaseParams)
        CASE(PLK_FIELD_TWEEDLEDUM_BASE, TweedledumBaseParams)
        CASE(PLK_FIELD_BLS12_377_SCALAR, Bls12377ScalarParams)
        CASE(PLK_FIELD_BLS12_377_BASE, Bls12377BaseParams)
        CASE(PLK_FIELD_PALLAS_BASE, PallasBaseParams)
        CASE(PLK_FIELD_VESTA_BASE, VestaBaseParams)
#undef CASE
    }
    hipError_t e = hipGetLastError();
    scratch_release(part, stream);
    if (e != hipSuccess) return set_error(PLK_ERR_HIP, "inner product launch failed: %s", hipGetErrorString(e));
    return PLK_OK;
}

int field_fold_slices_dev_impl(int field, const void* d_lo, const void* d_hi, const uint64_t* s_lo, const uint64_t* s_hi, size_t count, void* d_out,
                               hipStream_t stream) {
    if (count == 0) return PLK_OK;
    if (!d_lo || !d_hi || !d_out || !s_lo || !s_hi) return set_error(PLK_ERR_INVALID_ARG, "null pointer");
    PLK_TRY(ensure_device());
    const int L = field_limbs(field);
    if (L < 0) return set_error(PLK_ERR_INVALID_ARG, "bad field id %d", field);
    FoldScalars sc;
    for (int k = 0; k < L; ++k) {
        sc.lo[2 * k] = (uint32_t)s_lo[k];
        sc.lo[2 * k + 1] = (uint32_t)(s_lo[k] >> 32);
        sc.hi[2 * k] = (uint32_t)s_hi[k];
        sc.hi[2 * k + 1] = (uint32_t)(s_hi[k] >> 32);
    }
    const unsigned blocks = (unsigned)((count + 255) / 256);
    switch (field) {
#define CASE(ID, P) \
    case ID: k_fold_slices<P><<<blocks, 256, 0, stream>>>((const uint4*)d_lo, (const uint4*)d_hi, sc, count, (uint4*)d_out); break;
        CASE(PLK_FIELD_TWEEDLEDEE_BASE, TweedledeeBaseParams)
        CASE(PLK_FIELD_TWEEDLEDUM_BASE, TweedledumBaseParams)
        CASE(PLK_FIELD_BLS12_377_SCALAR, Bls12377ScalarParams)
        CASE(PLK_FIELD_BLS12_377_BASE, Bls12377BaseParams)
        CASE(PLK_FIELD_PALLAS_BASE, PallasBaseParams)
        CASE(PLK_FIELD_VESTA_BASE, VestaBaseParams)
#undef CASE
    }
    PLK_HIP_TRY(hipGetLastError());
    return PLK_OK;
}

int field_op_impl(int field, int op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t count) {
    if (op < 0 || op > 13) return set_error(PLK_ERR_INVALID_ARG, "bad field op %d", op);
    if (!a || !out || ((op <= 2 || op >= 12) && !b)) return set_error(PLK_ERR_INVALID_ARG, "null pointer");
    PLK_TRY(ensure_device());
    switch (field) {
        case PLK_FIELD_TWEEDLEDEE_BASE: return field_op_t<TweedledeeBaseParams>(op, a, b, out, count);
        case PLK_FIELD_TWEEDLEDUM_BASE: return field_op_t<TweedledumBaseParams>(op, a, b, out, count);
        case PLK_FIELD_BLS12_377_SCALAR: return field_op_t<Bls12377ScalarParams>(op, a, b, out, count);
        case PLK_FIELD_BLS12_377_BASE: return field_op_t<Bls12377BaseParams>(op, a, b, out, count);
        case PLK_FIELD_PALLAS_BASE: return field_op_t<PallasBaseParams>(op, a, b, out, count);
        case PLK_FIELD_VESTA_BASE: return field_op_t<VestaBaseParams>(op, a, b, out, count);
    }
    return set_error(PLK_ERR_INVALID_ARG, "bad field id %d", field);
}

}  // namespace plk

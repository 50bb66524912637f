/* plonky_hip.h -- C ABI of the MI355X-native NTT + MSM hot path of Plonky.
 *
 * The reference (0xPolygonZero/plonky, Rust) has no FFI layer; its hot path is reached through
 * five generic functions re-exported at the crate root (src/lib.rs:14-38).  Each entry point
 * below names the reference function it replaces (file:line into the reference tree).  The Rust
 * side binding a maintainer would add is shown in INTEGRATION.md.
 *
 * Conventions
 *  - every function returns PLK_OK (0) or a negative PLK_ERR_* code; nothing aborts
 *    (the reference panics on contract violations: src/curve/curve_msm.rs:67,106,
 *    src/util.rs:17, src/field/field.rs:430 -- the shim turns a negative code into a panic);
 *  - all buffers are caller-owned; field elements are the reference's in-memory limbs:
 *    little-endian u64 limbs, MONTGOMERY form, fully reduced (src/field/tweedledee_base.rs:14-18),
 *    4 limbs for TweedledeeBase / TweedledumBase / Bls12377Scalar / PallasBase / VestaBase, 6 for Bls12377Base;
 *  - affine points cross the boundary as x limbs followed by y limbs (2*L u64) plus one
 *    "zero" byte per point (AffinePoint.zero, src/curve/curve.rs:74-78); the structs are not
 *    repr(C), so the shim copies fields explicitly;
 *  - MSM results are returned as the unique affine point (ProjectivePoint::to_affine,
 *    src/curve/curve.rs:206-214): x, y in Montgomery limbs + zero flag;
 *  - functions with the _dev suffix take DEVICE pointers (HBM-resident data) and a hipStream_t
 *    passed as void*; the others take HOST pointers and copy through PCIe;
 *  - the library is thread safe: calls may come from many host threads (the reference calls
 *    these paths from Rayon workers, src/plonk_util.rs:173-189);
 *  - a call leaves the calling thread's current HIP device as it found it (the library selects devices
 *    freely inside a call); only plk_init, plk_init_devices and plk_set_thread_device select a device
 *    for the caller;
 *  - device-group contexts (plk_init_devices) live on logical device 0: plk_halo_begin_tabled_dev over such
 *    tables must be called from a thread on that device, and per-stage timings (plk_msm_set_profiling)
 *    exist for one-device contexts only - both are refused with PLK_ERR_INVALID_ARG otherwise.
 */
#ifndef PLONKY_HIP_H
#define PLONKY_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PLK_OK 0
#define PLK_ERR_INVALID_ARG (-1)   /* bad id, null pointer, size out of range            */
#define PLK_ERR_SIZE_MISMATCH (-2) /* scalars.len() != generators.len() (curve_msm.rs:67) */
#define PLK_ERR_NOT_POW2 (-3)      /* log2_strict would panic (util.rs:12-19)             */
#define PLK_ERR_TWO_ADICITY (-4)   /* n_power > TWO_ADICITY (field.rs:430)                */
#define PLK_ERR_HIP (-5)           /* a HIP runtime call failed; see plk_last_error()     */
#define PLK_ERR_NO_DEVICE (-6)     /* no gfx950 device visible                            */
#define PLK_ERR_OOM (-7)

/* field ids (reference: src/field/) */
#define PLK_FIELD_TWEEDLEDEE_BASE 0
#define PLK_FIELD_TWEEDLEDUM_BASE 1
#define PLK_FIELD_BLS12_377_SCALAR 2
#define PLK_FIELD_BLS12_377_BASE 3 /* 6 x u64 per element; curve coordinates, and the plain / zero-padded NTT entry points (no polynomial callers) */
#define PLK_FIELD_PALLAS_BASE 4    /* src/field/pallas_base.rs */
#define PLK_FIELD_VESTA_BASE 5     /* src/field/vesta_base.rs */
/* curve ids (reference: src/curve/) */
#define PLK_CURVE_TWEEDLEDEE 0
#define PLK_CURVE_TWEEDLEDUM 1
#define PLK_CURVE_BLS12_377 2
#define PLK_CURVE_PALLAS 3 /* src/curve/pallas_curve.rs: base PallasBase, scalars VestaBase */
#define PLK_CURVE_VESTA 4  /* src/curve/vesta_curve.rs: base VestaBase, scalars PallasBase */

/* ---- library ---------------------------------------------------------------------------- */
/* Select the ONE device this process uses (one process per GPU; -1: from the environment variable PLK_DEVICE).  Idempotent. */
int plk_init(int device);
/* Run over SEVERAL GPUs from this one process (the reference's callers are one process with Rayon threads,
 * plonk_util.rs:169-231, so the split over the GPUs of a node happens below this ABI).  n_devices > 0: the first n_devices visible
 * devices; 0: the environment variable PLK_NGPU, else every visible device.  From then on
 *   - plk_msm_precompute[_dev] of 2^PLK_MULTI_MIN_LOG_N (default 17) generators or more builds its tables on every device;
 *   - plk_msm_execute / plk_msm_execute_batch / plk_msm_execute_dev over such a context deal the batch out as floor(batch / N)
 *     WHOLE vectors per device plus the remaining batch mod N vectors SHARDED by contiguous base range (a single MSM is the
 *     sharded case alone); every device uploads its own share over its own PCIe link (device-resident vectors travel peer to
 *     peer), the partial results meet on the caller's device in one exchange record per device (plk_msm_partials_bytes) and are
 *     combined there (plk_msm_combine_partials_dev);
 *   - plk_ntt_batch / plk_ntt_padded_batch deal their independent transforms out round-robin (no exchange); single-transform
 *     calls arriving from many host threads take the devices in turn;
 *   - every other entry point runs on the calling thread's device: logical device 0 unless plk_set_thread_device chose another.
 * Results are bit-identical to the one-device path.  PLK_VIRTUAL_DEVICES=k (k >= 2) makes k LOGICAL devices out of the one
 * physical device PLK_DEVICE - separate contexts, worker threads and streams on the same GPU - so that a one-GPU machine runs
 * the whole multi-device path (tests).  Call before creating contexts; contexts do not survive a change of the device group. */
int plk_init_devices(int n_devices);
int plk_device_count(void);                    /* logical devices in use (1 after plk_init) */
/* Copies between two devices of the group since the library was loaded: `peer_copies` went as one hipMemcpyAsync (same physical
 * device, or peer access over xGMI), `staged_copies` through a pinned host buffer (no peer access between the pair, or PLK_PEER_MODE=host,
 * which forces that path for every pair - also between the logical devices of PLK_VIRTUAL_DEVICES).  Either pointer may be NULL.
 * PLK_VERBOSE=1 makes plk_init_devices print how many device pairs have peer access. */
int plk_group_copy_stats(unsigned long long* peer_copies, unsigned long long* staged_copies);
/* The calling thread's current HIP device as THIS library's HIP runtime sees it (>= 0; negative: error); set_to >= 0 selects that device
 * first.  Every other entry point leaves the calling thread's HIP device as it found it - except plk_init, plk_init_devices and
 * plk_set_thread_device, which select the device they were asked for.  For hosts and tests that want to check exactly that. */
int plk_thread_hip_device(int set_to);
/* The plan the fan-out follows, as pure arithmetic (no device needed): slot s < *slots of `device` among `world` devices computes
 * vector vec[s] of a batch of `batch` vectors over n generators, over the generators first[s] .. first[s] + count[s] - 1 - whole
 * vectors first (vector s * world + device, all n generators), then its contiguous share of every sharded vector.  vec / first /
 * count: arrays of at least batch entries, or all three NULL to ask for *slots only. */
int plk_multi_plan(unsigned world, unsigned batch, size_t n, unsigned device, unsigned* slots, unsigned* vec, uint64_t* first, uint64_t* count);
int plk_set_thread_device(int logical_device); /* the _dev entry points of the calling thread run on this logical device */
void plk_shutdown(void);
/* Size gate for the binding: problems below 2^plk_min_gpu_log_n() elements / pairs stay on the reference's
 * own CPU path - the library itself has no CPU path.  Environment PLK_MIN_GPU_LOG_N, default 12: PROVISIONAL - the measured crossover of
 * the host-pointer entry points (profiles/r05_crossover_host_pointer_vs_cpu.txt: GPU ahead from 2^8-2^10) is against the C++ restatement
 * of the reference's algorithm, not against its Rust, which cannot be built in this image; one gate for transforms and MSM pairs (a
 * first MSM call also pays its precomputation).  plk_init(-1) takes the
 * device from the environment variable PLK_DEVICE (default 0). */
unsigned plk_min_gpu_log_n(void);
/* Text of the last error on the calling thread (never NULL). */
const char* plk_last_error(void);
/* u64 limbs per element of a field / per coordinate of a curve; negative on bad id. */
int plk_field_limbs(int field);
int plk_curve_limbs(int curve);
int plk_curve_scalar_field(int curve);

/* ---- NTT  (src/fft.rs) -------------------------------------------------------------------- */
/* fft_precompute (fft.rs:47-59): builds and caches the device twiddle tables for transforms of
 * 2^log_n points over `field` (the reference's FftPrecomputation is plain host data; here the
 * cache is keyed by (device, field, log_n) inside the library, so nothing has to be stored in
 * the Rust struct).  Optional: plk_ntt* build the tables on first use. */
int plk_ntt_precompute(int field, unsigned log_n);
/* Drop every cached table (tests / memory pressure). */
int plk_ntt_clear_cache(void);

/* The CONTENTS of the reference's FftPrecomputation (fft.rs:28-59; serde-visible, embedded in VerificationKey, verifier.rs:23-26):
 * subgroups_rev[i] = reverse_index_bits(cyclic_subgroup_known_order(primitive_root_of_unity(i), 2^i)) for i = 0 ..= log_n, layer
 * after layer: 2^(log_n + 1) - 1 elements, layer i starts at element 2^i - 1.  (The reference builds them serially on the CPU -
 * 2n multiplications - inside CircuitBuilder::build, circuit_builder.rs:1118-1119.) */
int plk_ntt_precompute_table(int field, unsigned log_n, uint64_t* out);
int plk_ntt_precompute_table_dev(int field, unsigned log_n, void* d_out, void* stream);

/* fft_with_precomputation_power_of_2 (fft.rs:103-156) when inverse == 0,
 * ifft_with_precomputation_power_of_2 (fft.rs:82-101) when inverse != 0.
 * in/out: 2^log_n elements, natural order in, natural order out; in == out allowed. */
int plk_ntt(int field, unsigned log_n, int inverse, const uint64_t* in, uint64_t* out);
/* `batch` independent transforms (the 9 wire polynomials, src/plonk_util.rs:169-190). */
int plk_ntt_batch(int field, unsigned log_n, int inverse, unsigned batch, const uint64_t* const* in, uint64_t* const* out);
/* Same on device-resident data: `batch` transforms stored back to back (batch * 2^log_n elements).
 * d_in == d_out allowed.  Asynchronous on `stream`. */
int plk_ntt_dev(int field, unsigned log_n, int inverse, unsigned batch, const void* d_in, void* d_out, void* stream);
/* fft_with_precomputation (fft.rs:61-80): zero-pad n_in <= 2^log_n coefficients, then transform. */
int plk_ntt_padded(int field, unsigned log_n, const uint64_t* in, size_t n_in, uint64_t* out);
/* polynomials_to_values_padded (plonk_util.rs:179-190) / Polynomial::eval_domain (polynomial.rs:135-143):
 * `batch` polynomials of n_in[b] <= 2^log_n coefficients each, zero-padded to the domain and transformed.
 * The padding is never materialised: the first pass reads only the stored coefficients. */
int plk_ntt_padded_batch(int field, unsigned log_n, unsigned batch, const uint64_t* const* in, const size_t* n_in, uint64_t* const* out);
/* Device form: polynomial b starts at d_in + b * in_stride elements and has in_len coefficients; outputs are
 * stored back to back (batch * 2^log_n elements).  d_out must not overlap d_in unless batch == 1 and
 * in_stride == 2^log_n.  Asynchronous on `stream`. */
int plk_ntt_padded_dev(int field, unsigned log_n, unsigned batch, const void* d_in, size_t in_len, size_t in_stride, void* d_out,
                       void* stream);

/* ---- polynomial callers of the NTT  (src/polynomial.rs) ------------------------------------- */
/* Polynomial::divide_by_z_h (polynomial.rs:330-380): coeffs / (X^n - 1), assuming the division is exact
 * (otherwise the result is as meaningless as the reference's).  The result has 2^ceil(log2(degree + 1))
 * coefficients (the reference returns the untrimmed inverse transform); the zero polynomial comes back
 * unchanged with its `len` coefficients.  out_cap: capacity of `out` in elements, at least
 * max(len, 2^ceil(log2(len))) is always enough; *out_len receives the result length.
 * Coset scaling, denominators and their inversion are fused into the two transforms (poly.hip). */
int plk_poly_divide_by_z_h(int field, const uint64_t* coeffs, size_t len, size_t n, uint64_t* out, size_t out_cap, size_t* out_len);
/* Same on device-resident coefficients.  Synchronises `stream` once (the degree decides the domain
 * size, as in the reference); d_out may alias d_coeffs. */
int plk_poly_divide_by_z_h_dev(int field, const void* d_coeffs, size_t len, size_t n, void* d_out, size_t out_cap, size_t* out_len,
                               void* stream);
/* Polynomial::mul (polynomial.rs:208-226): product through three transforms of size
 * 2^ceil(log2(deg a + deg b + 1)), which is also the result length; a zero operand gives the single
 * coefficient 0 (Polynomial::zero(1)).  The pointwise product is fused into the second transform. */
int plk_poly_mul(int field, const uint64_t* a, size_t la, const uint64_t* b, size_t lb, uint64_t* out, size_t out_cap, size_t* out_len);
int plk_poly_mul_dev(int field, const void* d_a, size_t la, const void* d_b, size_t lb, void* d_out, size_t out_cap, size_t* out_len,
                     void* stream);

/* ---- the Plonk quotient numerator  (src/plonk.rs, src/gates/) ------------------------------------ */
/* The 8n-point loop of Prover::vanishing_poly (plonk.rs:392-453): for every point x = g^i of the 8n domain the
 * vanishing terms [L_1(x) (Z(x) - 1), Z(x) f'(x) - g'(x) Z(g x), evaluate_all_constraints(...)] reduced with powers of
 * alpha (plonk_util.rs:27-33).  `field` is the circuit's scalar field C::ScalarField, degree = 2^log_degree, n8 = 8 * degree.
 * Device tables, row-major, n8 elements per row, exactly the prover's own: constants_8n (6 rows, plonk.rs:64-69),
 * wire_values_8n (9 rows), s_sigma_values_8n (6 rows), plonk_z_points_8n (1 row: the LDE of Z, plonk.rs:388-391 =
 * plk_ntt_padded_dev).  Host scalars (4 limbs each, Montgomery): k_is[6] = get_subgroup_shift(0..5) (partition.rs:140-153,
 * ChaCha8 output: an input here), the challenges alpha, beta, gamma, and the two constants through which InnerC enters the
 * gates, InnerC::ZETA (curve_endo.rs:119) and InnerC::A (curve_dbl.rs:57).  d_out: n8 elements; the closing
 * Polynomial::from_evaluations (plonk.rs:455) is plk_ntt_dev(inverse = 1) on it.  Asynchronous on `stream`; the first call
 * for a (field, log_degree) builds and caches the circuit-size tables (L_1 over the domain, powers of g, MDS entries).
 * Table elements are the reference's field elements as stored: Montgomery form, CANONICAL (below p - what every Field value is,
 * field.rs / monty.rs); the kernels convert a row element through a table indexed by its top bits, so a word >= p is outside the contract. */
int plk_plonk_vanishing_points_dev(int field, unsigned log_degree, const void* d_constants_8n, const void* d_wires_8n, const void* d_s_sigma_8n,
                                   const void* d_plonk_z_8n, const uint64_t* k_is, const uint64_t* alpha, const uint64_t* beta, const uint64_t* gamma,
                                   const uint64_t* inner_zeta, const uint64_t* inner_a, void* d_out, void* stream);
/* Same with host tables (copied through PCIe). */
int plk_plonk_vanishing_points(int field, unsigned log_degree, const uint64_t* constants_8n, const uint64_t* wires_8n, const uint64_t* s_sigma_8n,
                               const uint64_t* plonk_z_8n, const uint64_t* k_is, const uint64_t* alpha, const uint64_t* beta, const uint64_t* gamma,
                               const uint64_t* inner_zeta, const uint64_t* inner_a, uint64_t* out);
/* evaluate_all_constraints (gates/mod.rs:46-125) at `count` independent points: constants [count][6], local / right / below
 * wire values [count][9] each, out [count][8] (the unified constraint set: the longest gate has 8 constraints). Host pointers. */
int plk_plonk_evaluate_all_constraints(int field, size_t count, const uint64_t* constants, const uint64_t* local_wires, const uint64_t* right_wires,
                                       const uint64_t* below_wires, const uint64_t* inner_zeta, const uint64_t* inner_a, uint64_t* out);

/* ---- MSM  (src/curve/curve_msm.rs) -------------------------------------------------------- */
typedef struct plk_msm_ctx plk_msm_ctx;

/* msm_precompute (curve_msm.rs:27-52): uploads the n generators and builds the device-side
 * window tables [2^(c*j)] G_i.  `window_bits` = 0 lets the library choose c from n (the result
 * of an MSM does not depend on the window; the reference's w only shapes its own tables).
 * bases_xy: n * 2L limbs (x then y, Montgomery); base_zero: n bytes or NULL (no identity inputs). */
int plk_msm_precompute(int curve, size_t n, const uint64_t* bases_xy, const uint8_t* base_zero, unsigned window_bits,
                       plk_msm_ctx** out_ctx);
/* Same with device-resident bases (n * 2L limbs) and flags (n bytes or NULL). */
int plk_msm_precompute_dev(int curve, size_t n, const void* d_bases_xy, const void* d_base_zero, unsigned window_bits, void* stream,
                           plk_msm_ctx** out_ctx);
/* flags for the _ex forms.  PLK_MSM_TABLE_FREE: do not build the window tables -- every window gets its own
 * buckets and is doubled into place at the end.  For generators that are used once or a few times (msm_parallel,
 * curve_msm.rs:54-61; the IPA rounds of halo.rs:87-91 build a fresh MsmPrecomputation per round): the table build
 * costs about 15 executions.  window_bits <= 16 in this mode.  On the prime-order curves (HaloCurve, curve.rs:65-69) the
 * scalars are split along the endomorphism (2n points, half-length scalars: ~120 dependent doublings instead of ~250);
 * the context's memory comes from the library's scratch pool (plk_ntt_clear_cache returns it to the driver). */
#define PLK_MSM_TABLE_FREE 1u
int plk_msm_precompute_ex(int curve, size_t n, const uint64_t* bases_xy, const uint8_t* base_zero, unsigned window_bits, unsigned flags,
                          plk_msm_ctx** out_ctx);
int plk_msm_precompute_dev_ex(int curve, size_t n, const void* d_bases_xy, const void* d_base_zero, unsigned window_bits, unsigned flags,
                              void* stream, plk_msm_ctx** out_ctx);
int plk_msm_free(plk_msm_ctx* ctx);
size_t plk_msm_ctx_len(const plk_msm_ctx* ctx);       /* number of generators n          */
unsigned plk_msm_ctx_window(const plk_msm_ctx* ctx);  /* window size c actually in use   */

/* msm_execute / msm_execute_parallel (curve_msm.rs:63-157): sum_i scalars[i] * G_i.
 * scalars: n_scalars * 4 limbs, Montgomery form IN THE CURVE'S SCALAR FIELD (to_digits converts,
 * curve_msm.rs:164).  n_scalars must equal the context length (PLK_ERR_SIZE_MISMATCH otherwise).
 * out_xy: 2L limbs, out_zero: 1 byte. */
int plk_msm_execute(plk_msm_ctx* ctx, const uint64_t* scalars, size_t n_scalars, uint64_t* out_xy, uint8_t* out_zero);
/* `batch` scalar vectors against the same generators (commit_polynomials, src/plonk_util.rs:215-231).
 * scalars[b]: n limbs*4 each; out_xy: batch * 2L limbs; out_zero: batch bytes. */
int plk_msm_execute_batch(plk_msm_ctx* ctx, unsigned batch, const uint64_t* const* scalars, size_t n_scalars, uint64_t* out_xy,
                          uint8_t* out_zero);
/* Device-resident: d_scalars = batch * n * 4 limbs back to back; d_out_xy = batch * 2L limbs;
 * d_out_zero = batch bytes.  Asynchronous on `stream`. */
int plk_msm_execute_dev(plk_msm_ctx* ctx, unsigned batch, const void* d_scalars, size_t n_scalars, void* d_out_xy, void* d_out_zero,
                        void* stream);
/* msm_execute_parallel with ITS OWN return type (src/curve/curve_msm.rs:102-157 returns the ProjectivePoint `y`, not normalised;
 * src/curve/curve.rs:175-181: x / z, y / z and a zero flag): out_xyz = x | y | z (3L Montgomery limbs per vector), out_zero = the flag.
 * The affine entry points above end their reduction with a field inversion on one lane (a third of the last kernel, ~40 us of a
 * 1.3 ms MSM); this form ends with six products and leaves the inversion to the caller's to_affine / batch_to_affine
 * (poly_commit.rs:58-66 normalises a whole batch with ONE inversion).  Any representative of the point may come back (z = 1 from comb and
 * device-group contexts); compare on to_affine. */
int plk_msm_execute_projective(plk_msm_ctx* ctx, const uint64_t* scalars, size_t n_scalars, uint64_t* out_xyz, uint8_t* out_zero);
int plk_msm_execute_projective_dev(plk_msm_ctx* ctx, unsigned batch, const void* d_scalars, size_t n_scalars, void* d_out_xyz, void* d_out_zero,
                                   void* stream);
/* The same batch when a vector only covers PART of the generators: result b = sum_{i < count[b]} scalars_b[i] * G[first[b] + i].
 * first / count: host arrays of `batch` entries (first[b] + count[b] <= n); d_scalars: host array of `batch` DEVICE pointers
 * (count[b] * 4 limbs each).  Tabled contexts only.  This is a rank's call in the multi-GPU split of a commitment batch
 * (section "Multi-GPU" below): its whole vectors with (0, n), its share of a sharded vector with its base range; all vectors
 * still share one reduction.  (Against the zero-padded full-length form of the same share it saves the ordering kernels n - count
 * zero scalars to skip: 1 % of a rank's step at 8 ranks - zero scalars were cheap already - and the padded copy of the vector.) */
int plk_msm_execute_parts_dev(plk_msm_ctx* ctx, unsigned batch, const uint64_t* first, const uint64_t* count, const void* const* d_scalars,
                              void* d_out_xy, void* d_out_zero, void* stream);
/* The same with a BUCKET range per vector (round 6): vector b keeps only the entries whose bucket falls into the bucket_part[b]-th of
 * bucket_parts[b] equal ranges of the context's coarse bins (0 or 1 parts: every bucket).  The partial results of the bucket_parts[b]
 * ranges add up to the vector's MSM (plk_msm_combine_partials_dev).  This is the other way to share ONE vector among N devices: every
 * device reads the whole vector (and holds the whole table) but orders, accumulates and reduces an N-th of the entries over an N-th of
 * the buckets at the window a full-size MSM deserves - where a base range of n / N generators pays a whole reduction over all the
 * buckets of its (smaller) window for an N-th of the additions.  Tabled, non-comb contexts; host arrays of `batch` entries. */
int plk_msm_execute_parts_buckets_dev(plk_msm_ctx* ctx, unsigned batch, const uint64_t* first, const uint64_t* count, const void* const* d_scalars,
                                      const uint32_t* bucket_part, const uint32_t* bucket_parts, void* d_out_xy, void* d_out_zero, void* stream);
/* msm_parallel (curve_msm.rs:54-61): precompute + execute + free in one call. */
int plk_msm(int curve, size_t n, const uint64_t* bases_xy, const uint8_t* base_zero, const uint64_t* scalars, uint64_t* out_xy,
            uint8_t* out_zero);

/* Sum of k affine points -> affine: affine_summation_best / _pairwise / _batch_inversion (curve_summations.rs:18-22, 39-68 - one
 * group element whatever the form; identity operands, P = Q and P = -Q handled as :86-92, 113-141 do), and what combines per-GPU
 * partial MSM results after the all-gather (point addition is not an RCCL reduction op).  affine_multisummation_best (:24-35) is
 * one call per list.  Host pointers.  Pinned directly by tests/test_gpu_summations.py (the reference's own two unit tests, :164-184). */
int plk_curve_sum_affine(int curve, size_t k, const uint64_t* pts_xy, const uint8_t* pts_zero, uint64_t* out_xy, uint8_t* out_zero);

/* ---- multi-GPU exchange of partial results (no counterpart in the single-process reference) ----------- */
/* One process per GPU.  A batch of scalar vectors against the same generators (commit_polynomials, plonk_util.rs:215-231) is
 * dealt out as WHOLE vectors, `whole_per_rank` = floor(batch / world) to every rank (vector v belongs to rank v mod world, its
 * slot there is v / world), and the remaining batch mod world vectors are SHARDED by contiguous base range: every rank reduces
 * its slice of them.  A single MSM (batch 1) is the sharded case alone.  The ranks exchange their results with ONE all-gather.
 * The record of a rank holds `slots` = whole_per_rank + (batch - whole_per_rank * world) points: slots * 2L limbs (exactly what
 * plk_msm_execute_dev writes at d_out_xy = record), then slots identity flags (d_out_zero = record + slots * 2L * 8), padded to
 * a multiple of 16 bytes = plk_msm_partials_bytes(curve, slots) - the MSM writes straight into the collective's send buffer. */
size_t plk_msm_partials_bytes(int curve, unsigned slots);
/* d_gathered: `world` records back to back (the all-gather output).  d_out_xy / d_out_zero: the `batch` results in vector
 * order, unique affine form: a whole vector is copied from its owner's slot, a sharded one is the sum of the ranks' partial
 * points (point addition is not an RCCL reduction op).  whole_per_rank = 0: every vector is sharded.  Asynchronous on `stream`. */
int plk_msm_combine_partials_dev(int curve, unsigned world, unsigned batch, unsigned whole_per_rank, const void* d_gathered, void* d_out_xy,
                                 void* d_out_zero, void* stream);

/* ---- the reference's own table contents  (curve_msm.rs:16-52) ---------------------------------- */
/* MsmPrecomputation { powers_per_generator, w } is plain, serde-visible data in the reference (embedded in Circuit,
 * plonk.rs:64-69).  A caller that wants that struct filled from the device gets exactly its contents here:
 * out[i * digits + j] = [2^(w j)] G_i, j < digits = ceil(ScalarField::BITS / w) = plk_msm_table_digits(curve, w),
 * affine Montgomery limbs (n * digits * 2L) plus AffinePoint::zero flags (n * digits bytes). */
int plk_msm_table_digits(int curve, unsigned w);
int plk_msm_precompute_table(int curve, size_t n, const uint64_t* bases_xy, const uint8_t* base_zero, unsigned w, uint64_t* out_xy, uint8_t* out_zero);
int plk_msm_precompute_table_dev(int curve, size_t n, const void* d_bases_xy, const void* d_base_zero, unsigned w, void* d_out_xy, void* d_out_zero,
                                 void* stream);

/* ---- IPA generator fold  (src/halo.rs:119-123) ------------------------------------------------ */
/* out_i = [scalar_lo] lo_i + [scalar_hi] hi_i for i < m: the fold G' = [u^-1] G_lo + [u] G_hi of an inner-product-
 * argument round, which the reference computes as m calls of msm_parallel(&[u_inv, u], &[g_lo_i, g_hi_i], 4).
 * Points are affine (m * 2L limbs, Montgomery) with optional identity flags (m bytes or NULL); the two scalars
 * are 4 limbs each, Montgomery form in the curve's SCALAR field, host pointers in both forms of the call.
 * Results are the unique affine points (+ identity flags), as for the MSM. */
int plk_curve_fold_pairs(int curve, size_t m, const uint64_t* lo_xy, const uint8_t* lo_zero, const uint64_t* hi_xy, const uint8_t* hi_zero,
                         const uint64_t* scalar_lo, const uint64_t* scalar_hi, uint64_t* out_xy, uint8_t* out_zero);
int plk_curve_fold_pairs_dev(int curve, size_t m, const void* d_lo_xy, const void* d_lo_zero, const void* d_hi_xy, const void* d_hi_zero,
                             const uint64_t* scalar_lo, const uint64_t* scalar_hi, void* d_out_xy, void* d_out_zero, void* stream);

/* ---- batch inversion  (src/field/field.rs:223-278, src/curve/curve.rs:216-232) ------------------------ */
/* Field::batch_multiplicative_inverse (field.rs:251-278, Montgomery's trick): out[i] = 1 / x[i].  A zero element has no
 * inverse: the reference panics ("No inverse", field.rs:266) and this returns PLK_ERR_INVALID_ARG (out is then unspecified). */
int plk_field_batch_inverse(int field, const uint64_t* x, uint64_t* out, size_t count);
/* Field::batch_multiplicative_inverse_opt (field.rs:223-249): zero elements give None: is_none[i] = 1 and out[i] = 0. */
int plk_field_batch_inverse_opt(int field, const uint64_t* x, uint64_t* out, uint8_t* is_none, size_t count);
/* Device form of the _opt variant (d_is_none may be NULL: zeros then just come back as 0).  d_x == d_out allowed.  Asynchronous. */
int plk_field_batch_inverse_dev(int field, const void* d_x, void* d_out, void* d_is_none, size_t count, void* stream);
/* ProjectivePoint::batch_to_affine (curve.rs:216-232): count homogeneous projective points (X, Y, Z: 3L limbs each, x = X / Z,
 * y = Y / Z; `zero` flags, count bytes or NULL) -> affine (2L limbs each) + zero flags, AffinePoint::ZERO = (0, 0, true). */
int plk_curve_batch_to_affine(int curve, size_t count, const uint64_t* proj_xyz, const uint8_t* proj_zero, uint64_t* out_xy, uint8_t* out_zero);
int plk_curve_batch_to_affine_dev(int curve, size_t count, const void* d_proj_xyz, const void* d_proj_zero, void* d_out_xy, void* d_out_zero, void* stream);

/* ---- canonical byte encodings  (src/serialization.rs:17-72) ----------------------------------------- */
/* ToBytes / FromBytes for field elements (serialization.rs:17-31): BYTES = 8 * limbs little-endian bytes of the CANONICAL
 * value (field.rs:67-102).  plk_field_from_bytes returns PLK_ERR_INVALID_ARG ("Out of range", field.rs:100) when a record
 * is not below the modulus.  Host pointers; the conversion runs on the device. */
int plk_field_to_bytes(int field, const uint64_t* x, size_t count, uint8_t* out_bytes);
int plk_field_from_bytes(int field, const uint8_t* bytes, size_t count, uint64_t* out);
/* ToBytes / FromBytes for AffinePoint (serialization.rs:33-72): 1 + BYTES per point: mask = zero | (y odd) << 1, then x.
 * Decompression recovers y = sqrt(x^3 + B) (Field::square_root, field.rs:440-472) with the parity of the mask.  status
 * (count bytes, may be NULL): 0 ok, 1 "Out of range", 2 "Invalid x coordinate"; any non-zero status also makes the call
 * return PLK_ERR_INVALID_ARG (the reference returns Err for the point). */
int plk_curve_point_to_bytes(int curve, const uint64_t* xy, const uint8_t* zero, size_t count, uint8_t* out_bytes);
int plk_curve_point_from_bytes(int curve, const uint8_t* bytes, size_t count, uint64_t* out_xy, uint8_t* out_zero, uint8_t* status);

/* r = log_inputs rounds of that fold at once, in the scaled form halo.hip keeps its generators in:
 *     out_i = g_i + sum_{t = 1 .. 2^r - 1} [s_t] g_{i + t n_out},   i < n_out,
 * g = 2^r n_out affine points (+ optional identity flags), the 2^r scalars in DEVICE memory (4 limbs each, Montgomery, scalar
 * field), the scalar of input t at index bitreverse_r(t); entry 0 is not read.  With s_t = the product of u_k^2 over the
 * rounds k whose challenge index t picked the upper half in, [prod u_k^-1] out_i is halo_g_i after r rounds of halo.rs:119-123.
 * One doubling chain per OUTPUT (Straus over its 2^r inputs) instead of one per output of every round.  Curves with the
 * endomorphism only (not BLS12-377: PLK_ERR_INVALID_ARG); 1 <= r <= 4; in place (d_out = d_g) is allowed. */
int plk_curve_fold_multi_dev(int curve, size_t n_out, unsigned log_inputs, const void* d_g_xy, const void* d_g_zero, const void* d_scalars,
                             void* d_out_xy, void* d_out_zero, void* stream);

/* ---- scalar side of an IPA round  (src/halo.rs:63-118) --------------------------------------------------- */
/* Field::inner_product (field.rs:213-221): *d_out = sum_i a[i] b[i] (one element, device memory).  Asynchronous on `stream`.
 * The point side of a round is plk_msm (msm_parallel on fresh generators: L_j, R_j, halo.rs:87-93) and
 * plk_curve_fold_pairs (G' = [u^-1] G_lo + [u] G_hi, halo.rs:119-123). */
int plk_field_inner_product_dev(int field, const void* d_a, const void* d_b, size_t count, void* d_out, void* stream);
/* add_slices(scalar_lo.scale_slice(lo), scalar_hi.scale_slice(hi)) (halo.rs:117-118: halo_a' = u^-1 a_hi + u a_lo,
 * halo_b' = u^-1 b_lo + u b_hi): d_out[i] = scalar_lo * d_lo[i] + scalar_hi * d_hi[i]; the two scalars are host pointers
 * (field limbs, Montgomery).  d_out may alias d_lo.  Asynchronous on `stream`. */
int plk_field_fold_slices_dev(int field, const void* d_lo, const void* d_hi, const uint64_t* scalar_lo, const uint64_t* scalar_hi, size_t count,
                              void* d_out, void* stream);

/* ---- one inner-product argument, round by round  (src/halo.rs:63-124) ---------------------------------------------- */
/* The three vectors of the argument - halo_a, halo_b (n scalars, Montgomery form in the curve's SCALAR field) and halo_g (n
 * affine generators + optional identity flags) - are copied into a context and stay in HBM for the log2(n) rounds; H =
 * pedersen_h and U' = u_prime (2L limbs each, affine, host pointers) ride along.  Per round the caller (who owns the
 * transcript and the RNG, halo.rs:83-114) asks for
 *     L_j = <a_lo, G_hi> + [l_j] H + [<a_lo, b_hi>] U',   R_j = <a_hi, G_lo> + [r_j] H + [<a_hi, b_lo>] U'      (halo.rs:86-93)
 * with its blinding factors - again with fresh ones if the challenge has no square root (the reference's retry loop) - and
 * then folds with the challenge:  halo_a = u^-1 a_hi + u a_lo, halo_b = u^-1 b_lo + u b_hi, halo_g_i = [u^-1] g_lo_i + [u] g_hi_i
 * (halo.rs:117-123).  Nothing is allocated, created or freed per round.  Once 2^freeze_log or fewer generators are left
 * (0 = the library's default, 14) they are not folded any more: window tables are built once and the remaining rounds run
 * as tabled MSMs over that frozen set with challenge-expanded scalars (same group elements, see halo.hip).
 * n must be a power of two (PLK_ERR_NOT_POW2, util.rs:17).  A context is used by one host thread at a time. */
typedef struct plk_halo_ctx plk_halo_ctx;
int plk_halo_begin_dev(int curve, size_t n, const void* d_halo_a, const void* d_halo_b, const void* d_halo_g_xy, const void* d_halo_g_zero,
                       const uint64_t* pedersen_h_xy, const uint64_t* u_prime_xy, unsigned freeze_log, void* stream, plk_halo_ctx** out_ctx);
/* The same argument when the caller still holds the window tables of pedersen_g it committed with (plonk.rs:65
 * `pedersen_g_msm_precomputation`: a tabled plk_msm_ctx of this curve whose first n generators are halo_g; it must outlive the lead
 * rounds and is only read).  The first `lead_rounds` rounds (0 = default 3; <= 4; fewer when the vectors are short; none on
 * BLS12-377) leave the generators untouched: L_j / R_j are one batched MSM over those tables with challenge-expanded scalars, and
 * the generators of all lead rounds are then folded at once (plk_curve_fold_multi_dev).  Same L_j, R_j, halo_a, halo_b, halo_g as
 * plk_halo_begin_dev gives, bit for bit; plk_halo_frozen() is 1 during the lead rounds as well (halo_g cannot be read then).
 * h_index / u_index / u_prime_scalar (optional: PLK_NO_INDEX, PLK_NO_INDEX, NULL): when the tables were built over
 * [pedersen_g .., pedersen_h, U ..] - the circuit's fixed generators (plonk.rs:46-51) - the positions of pedersen_h and of U in
 * them (both >= n) and the scalar x = halo_n(u_scaling bits) with u_prime = [x] U (halo.rs:46-47, 4 limbs, Montgomery, scalar
 * field; the CALLER vouches for that relation).  [l_j] H + [<a, b>] U' then are two more scalars of the same MSM; without them
 * they are computed beside it (one lane each, ~1 ms, mostly hidden). */
#define PLK_NO_INDEX ((size_t)-1)
int plk_halo_begin_tabled_dev(int curve, size_t n, const void* d_halo_a, const void* d_halo_b, const void* d_halo_g_xy, const void* d_halo_g_zero,
                              plk_msm_ctx* pedersen_g_tables, const uint64_t* pedersen_h_xy, const uint64_t* u_prime_xy, size_t h_index, size_t u_index,
                              const uint64_t* u_prime_scalar, unsigned freeze_log, unsigned lead_rounds, void* stream, plk_halo_ctx** out_ctx);
/* Same with host vectors (halo_g_zero may be NULL). */
int plk_halo_begin(int curve, size_t n, const uint64_t* halo_a, const uint64_t* halo_b, const uint64_t* halo_g_xy, const uint8_t* halo_g_zero,
                   const uint64_t* pedersen_h_xy, const uint64_t* u_prime_xy, unsigned freeze_log, plk_halo_ctx** out_ctx);
/* L_j then R_j of the current round, unique affine form: lr_xy 2 x 2L limbs, lr_zero 2 bytes (host).  Blinding factors: 4 limbs
 * each, Montgomery, host.  Waits for the result (the transcript needs it).  The two sums cross PCIe as msm_execute's ProjectivePoints
 * and are normalised by the library on the calling thread, where halo.rs:93-101 calls to_affine() (PLK_HALO_DEVICE_AFFINE=1: on the
 * device). */
int plk_halo_round_lr(plk_halo_ctx* ctx, const uint64_t* l_blinding, const uint64_t* r_blinding, uint64_t* lr_xy, uint8_t* lr_zero);
/* The folds of the round with the challenge u_j and its inverse (4 limbs each, Montgomery, host); halves the length.  Asynchronous. */
int plk_halo_round_fold(plk_halo_ctx* ctx, const uint64_t* u_j, const uint64_t* u_j_inv);
size_t plk_halo_len(const plk_halo_ctx* ctx);      /* current length of the three vectors */
int plk_halo_frozen(const plk_halo_ctx* ctx);      /* 1 once the generators are kept as tables + coefficients */
/* Current halo_a / halo_b (len scalars each; NULL to skip) and halo_g (len points + flags; both or neither).  While the
 * generators are frozen halo_g is only defined again at length 1 (the end of the argument, halo.rs:126-127: one MSM). */
int plk_halo_read(plk_halo_ctx* ctx, uint64_t* halo_a, uint64_t* halo_b, uint64_t* halo_g_xy, uint8_t* halo_g_zero);
int plk_halo_free(plk_halo_ctx* ctx);

/* ---- self-test ------------------------------------------------------------------------------ */
/* Runs the quad-cooperative point arithmetic of the MSM reduction tail (ecz_coop.cuh) against the one-lane
 * arithmetic on the n affine points pts_xy (n * 2L limbs, Montgomery), `quads` quads cycling through 8 cases
 * (addition, doubling, doubling inside an addition, opposite points, identity operands, repeated doubling,
 * wave-wide sum).  mismatches[8] receives the number of disagreements per case: all zero on a healthy build. */
int plk_selftest_quad(int curve, const uint64_t* pts_xy, size_t n, unsigned quads, unsigned* mismatches);

/* ---- checked build --------------------------------------------------------------------------------- */
/* libplonky_hip_checked.so (make -C plonky_amd/csrc checked) is the same library with -DPLK_CHECKED: every index the MSM's
 * ordering and accumulation kernels compute into their work arrays and tables is compared with its bound, a violation is
 * counted and the access skipped.  plk_checked_build() tells the two builds apart (1 / 0); plk_checked_failures() returns
 * the violation counters per guarded site (counts[8]; all zero in the normal build, which has no guards). */
int plk_checked_build(void);
int plk_checked_failures(unsigned* counts);

/* ---- measurement hooks (bench.py's roofline: per-kernel durations from HIP events recorded on the
 *      launch stream around each kernel; no effect on results) ------------------------------- */
/* NTT pass kernel: enable, run transforms, then read the summed duration and the number of launches
 * recorded since the previous read (the call waits for the recorded events). */
int plk_ntt_set_profiling(int enable);
int plk_ntt_get_timings(double* sum_ms, unsigned* launches);
/* MSM pipeline: sum_ms[7] = scalar digits, scan, scatter, bucket accumulation, chunk sums, plane sums,
 * final -- summed over `calls` executions since the previous read. */
/* (a context over few generators with an automatic window is a COMB - comb.hip, <= 2^12 generators: 32 KiB of table per generator,
 * 48 for BLS12-377, i.e. 128-192 MiB at 2^12, ~20 x the window tables; it has no stages: enabling the timings on it returns
 * PLK_ERR_INVALID_ARG.  PLK_MSM_COMB=0 keeps every context on the bucket method; when the comb's table cannot be allocated the
 * library falls back to the bucket method by itself.) */
int plk_msm_set_profiling(plk_msm_ctx* ctx, int enable);
int plk_msm_get_timings(plk_msm_ctx* ctx, double* sum_ms, unsigned* calls);

/* ---- utilities used by the harness and the parity tests (device kernels, not CPU code) ------ */
/* Element-wise field ops on host arrays of `count` elements: op 0 add, 1 sub, 2 mul, 3 neg(a),
 * 4 square(a), 5 inverse(a) by Fermat (0 -> 0), 6 to_canonical(a), 7 from_canonical(a), 8 inverse by the
 * reference's binary Euclid (bigint_inverse.rs:6-55), 9 inverse by division steps (the one the kernels
 * use), 10 the same in its data-dependent form, 11 that form with ONE active lane per wave (the normalisation at the end of an MSM);
 * 12 / 13: the shared-reduction arithmetic of the kernels on fixed limb patterns derived from a and b (sums of two products through one
 * Montgomery reduction at the edge of their column bound; the column accumulators of the quotient numerator) - parity tests only.
 * b ignored for unary. */
int plk_field_op(int field, int op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t count);
/* The integer-ALU ceilings of the GPU the calling thread runs on, measured now (~60 ms; G operations per second over the whole GPU):
 * out[0] v_mad_u64_u32 lane-operations (8 waves per SIMD, 8 independent chains per lane) - the raw issue rate of the instruction a
 * modular multiplication is made of (126 per 9-limb product, 294 per 14-limb product): the hardware-referenced roofline of every
 * kernel here; out[1], out[2] this build's Montgomery product at 4 waves per SIMD on the 9-limb / 14-limb fields; out[3], out[4] the
 * lazy mixed addition of the accumulation (8 M + 2 S, no memory) at the occupancy k_msm_accumulate has on those fields (3 / 2 waves).
 * n_out >= 5.  There is nothing to replace in the reference: bench.py prices its rooflines with it. */
int plk_bench_ceilings(double* out, unsigned n_out);
/* The digit recoding of the MSM's ordering kernels on its own (replaces to_digits, curve_msm.rs:159-180; the device never stores
 * digits, it recomputes them where they are used).  scalars: n * 4 limbs, Montgomery form in the curve's SCALAR field, host memory.
 * *n_digits = ceil((BITS + 1) / window_bits) digits per scalar; digits (may be NULL to query n_digits only): n * *n_digits signed
 * values d_j in [-2^(w-1), 2^(w-1)], least significant window first, sum_j d_j 2^(w j) = the canonical scalar.  The reference's
 * unsigned digits are u_j = d_j - carry_j + 2^w carry_(j+1) with carry_(j+1) = [d_j - carry_j < 0], carry_0 = 0
 * (tests/test_gpu_parity.py pins them to the vector of curve_msm.rs:186-216). */
int plk_msm_debug_digits(int curve, unsigned window_bits, size_t n, const uint64_t* scalars, int32_t* digits, unsigned* n_digits);
/* Synthetic generators B_i = G0 + (first + i) * D, i < n, affine, written to DEVICE memory
 * (n * 2L limbs).  g0_xy / d_xy are host pointers (2L limbs each). */
int plk_curve_gen_bases_dev(int curve, size_t n, uint64_t first, const uint64_t* g0_xy, const uint64_t* d_xy, void* d_out_xy, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PLONKY_HIP_H */

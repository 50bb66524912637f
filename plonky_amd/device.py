"""Device-resident entry points (torch tensors as HBM buffers, raw pointers into the C ABI).

PyTorch is plumbing here: it owns device memory and streams; every computation is a call into
libplonky_hip.so.  Tensors are int64 views of the reference's u64 limbs, shape (..., L).
"""
import ctypes

import numpy as np
import torch

from . import lib as _lib
from .api import _CURVE_LIMBS, _FIELD_LIMBS, MsmPrecomputation, log2_strict


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def init(device_index=None):
    if device_index is None:
        device_index = torch.cuda.current_device()
    torch.cuda.set_device(device_index)
    _lib.check(_lib.load().plk_init(int(device_index)))
    return device_index


def to_device(arr, device="cuda"):
    a = np.ascontiguousarray(arr, dtype=np.uint64)
    return torch.from_numpy(a.view(np.int64)).to(device)


def to_host(t):
    return t.detach().cpu().numpy().view(np.uint64)


def ntt_dev(field, x, inverse=False, out=None):
    """x: (batch, n, L) or (n, L) int64 CUDA tensor, L = 4 (6 for Bls12377Base); returns the transforms (natural order)."""
    assert x.is_cuda and x.dtype == torch.int64 and x.is_contiguous()
    L = _FIELD_LIMBS[field]
    assert x.shape[-1] == L
    n = x.shape[-2]
    batch = x.numel() // (n * L)
    log_n = log2_strict(n)
    if out is None:
        out = torch.empty_like(x)
    _lib.check(_lib.load().plk_ntt_dev(field, log_n, 1 if inverse else 0, batch, ctypes.c_void_p(x.data_ptr()),
                                       ctypes.c_void_p(out.data_ptr()), _stream()))
    return out


def ntt_padded_dev(field, x, log_n, out=None):
    """polynomials_to_values_padded on device-resident coefficients: x (batch, len, 4) or (len, 4) with
    len <= 2^log_n; returns (batch, 2^log_n, 4) evaluations.  The zero padding is never stored.  (6 limbs for Bls12377Base.)"""
    L = _FIELD_LIMBS[field]
    assert x.is_cuda and x.dtype == torch.int64 and x.is_contiguous() and x.shape[-1] == L
    import math
    length = x.shape[-2]
    batch = math.prod(x.shape[:-2])  # independent of the length: (B, 0, 4) is B zero polynomials, every output is written
    n = 1 << log_n
    shape = x.shape[:-2] + (n, L)
    if out is None:
        out = torch.empty(shape, dtype=torch.int64, device=x.device)
    _lib.check(_lib.load().plk_ntt_padded_dev(field, log_n, batch, ctypes.c_void_p(x.data_ptr()), length, length,
                                              ctypes.c_void_p(out.data_ptr()), _stream()))
    return out


def divide_by_z_h_dev(field, coeffs, n, out=None):
    """Polynomial::divide_by_z_h on a device-resident (len, 4) coefficient tensor; returns a view of the
    2^ceil(log2(degree + 1)) result coefficients (one stream synchronisation: the degree picks the domain)."""
    assert coeffs.is_cuda and coeffs.dtype == torch.int64 and coeffs.is_contiguous() and coeffs.shape[-1] == 4
    length = coeffs.shape[0]
    cap = max(length, 1 << max(0, (length - 1).bit_length()))
    if out is None:
        out = torch.empty((cap, 4), dtype=torch.int64, device=coeffs.device)
    assert out.shape[0] >= cap
    out_len = ctypes.c_size_t(0)
    _lib.check(_lib.load().plk_poly_divide_by_z_h_dev(field, ctypes.c_void_p(coeffs.data_ptr()), length, n,
                                                      ctypes.c_void_p(out.data_ptr()), out.shape[0], ctypes.byref(out_len), _stream()))
    return out[: out_len.value]


def poly_mul_dev(field, a, b):
    """Polynomial::mul on device-resident coefficient tensors."""
    for t in (a, b):
        assert t.is_cuda and t.dtype == torch.int64 and t.is_contiguous() and t.shape[-1] == 4
    cap = 1 << max(0, (a.shape[0] + b.shape[0] - 1).bit_length())
    out = torch.empty((cap, 4), dtype=torch.int64, device=a.device)
    out_len = ctypes.c_size_t(0)
    _lib.check(_lib.load().plk_poly_mul_dev(field, ctypes.c_void_p(a.data_ptr()), a.shape[0], ctypes.c_void_p(b.data_ptr()), b.shape[0],
                                            ctypes.c_void_p(out.data_ptr()), cap, ctypes.byref(out_len), _stream()))
    return out[: out_len.value]


def gen_bases_dev(curve, n, g0_xy, d_xy, first=0, device="cuda"):
    """B_i = G0 + (first + i) D on the device: (n, 2, L) int64 tensor."""
    L = _CURVE_LIMBS[curve]
    out = torch.empty((n, 2, L), dtype=torch.int64, device=device)
    g0 = np.ascontiguousarray(g0_xy, dtype=np.uint64)
    d = np.ascontiguousarray(d_xy, dtype=np.uint64)
    _lib.check(_lib.load().plk_curve_gen_bases_dev(curve, n, first, g0.ctypes.data_as(ctypes.c_void_p), d.ctypes.data_as(ctypes.c_void_p),
                                                   ctypes.c_void_p(out.data_ptr()), _stream()))
    return out


def msm_precompute_dev(curve, bases, w=11, zero=None, device_window=0, table_free=False):
    """bases: (n, 2, L) int64 CUDA tensor.  table_free: no window tables (generators used once or a few times)."""
    assert bases.is_cuda and bases.dtype == torch.int64 and bases.is_contiguous()
    n = bases.shape[0]
    ctx = ctypes.c_void_p()
    zp = ctypes.c_void_p(zero.data_ptr()) if zero is not None else None
    _lib.check(_lib.load().plk_msm_precompute_dev_ex(curve, n, ctypes.c_void_p(bases.data_ptr()), zp, device_window, 1 if table_free else 0,
                                                     _stream(), ctypes.byref(ctx)))
    return MsmPrecomputation(curve, ctx, n, w)


def msm_execute_dev(pre, scalars, out_xy=None, out_zero=None, projective=False):
    """scalars: (batch, n, 4) or (n, 4) int64 CUDA tensor.  Returns (out_xy (batch, 2, L), out_zero (batch,)) on device.
    projective: the reference's own return type - ProjectivePoints x | y | z, not normalised ((batch, 3, L); plk_msm_execute_projective_dev)."""
    assert scalars.is_cuda and scalars.dtype == torch.int64 and scalars.is_contiguous()
    n = scalars.shape[-2]
    assert n == pre.n, "powers_per_generator.len() != scalars.len()"
    batch = scalars.numel() // (n * 4) if n else 1
    L = _CURVE_LIMBS[pre.curve]
    if out_xy is None:
        out_xy = torch.empty((batch, 3 if projective else 2, L), dtype=torch.int64, device=scalars.device)
    if out_zero is None:
        out_zero = torch.empty((batch,), dtype=torch.uint8, device=scalars.device)
    if projective:
        assert out_xy.numel() == batch * 3 * L
        _lib.check(_lib.load().plk_msm_execute_projective_dev(pre._ctx, batch, ctypes.c_void_p(scalars.data_ptr()), n, ctypes.c_void_p(out_xy.data_ptr()),
                                                              ctypes.c_void_p(out_zero.data_ptr()), _stream()))
        return out_xy, out_zero
    _lib.check(_lib.load().plk_msm_execute_dev(pre._ctx, batch, ctypes.c_void_p(scalars.data_ptr()), n, ctypes.c_void_p(out_xy.data_ptr()),
                                               ctypes.c_void_p(out_zero.data_ptr()), _stream()))
    return out_xy, out_zero


def msm_execute_parts_dev(pre, parts, out_xy=None, out_zero=None, buckets=None):
    """plk_msm_execute_parts_dev: parts = [(first, scalars)] with scalars an (count, 4) int64 CUDA tensor for the generators
    first .. first + count - 1 of `pre` (a tabled precomputation).  One batched call, one shared reduction -> ((batch, 2, L), (batch,)).
    buckets (optional): [(part, parts)] per vector - the vector keeps only the part-th of `parts` ranges of the coarse bucket bins
    (plk_msm_execute_parts_buckets_dev: a rank's BUCKET share of a sharded vector; (0, 1): every bucket)."""
    batch = len(parts)
    L = _CURVE_LIMBS[pre.curve]
    dev0 = parts[0][1].device
    if out_xy is None:
        out_xy = torch.empty((batch, 2, L), dtype=torch.int64, device=dev0)
    if out_zero is None:
        out_zero = torch.empty((batch,), dtype=torch.uint8, device=dev0)
    first = np.array([p[0] for p in parts], dtype=np.uint64)
    count = np.array([p[1].shape[0] for p in parts], dtype=np.uint64)
    for _, t in parts:
        assert t.is_cuda and t.dtype == torch.int64 and t.is_contiguous() and t.shape[-1] == 4
    ptrs = (ctypes.c_void_p * batch)(*[p[1].data_ptr() for p in parts])
    if buckets is not None:
        assert len(buckets) == batch
        bp = np.array([b[0] for b in buckets], dtype=np.uint32)
        bn = np.array([b[1] for b in buckets], dtype=np.uint32)
        _lib.check(_lib.load().plk_msm_execute_parts_buckets_dev(pre._ctx, batch, first.ctypes.data_as(ctypes.c_void_p), count.ctypes.data_as(ctypes.c_void_p),
                                                                 ptrs, bp.ctypes.data_as(ctypes.c_void_p), bn.ctypes.data_as(ctypes.c_void_p),
                                                                 ctypes.c_void_p(out_xy.data_ptr()), ctypes.c_void_p(out_zero.data_ptr()), _stream()))
        return out_xy, out_zero
    _lib.check(_lib.load().plk_msm_execute_parts_dev(pre._ctx, batch, first.ctypes.data_as(ctypes.c_void_p), count.ctypes.data_as(ctypes.c_void_p), ptrs,
                                                     ctypes.c_void_p(out_xy.data_ptr()), ctypes.c_void_p(out_zero.data_ptr()), _stream()))
    return out_xy, out_zero


def vanishing_points_dev(field, log_degree, constants_8n, wire_values_8n, s_sigma_values_8n, plonk_z_points_8n, k_is, alpha, beta, gamma,
                         inner_zeta, inner_a, out=None):
    """Prover::vanishing_poly's 8n-point loop (plonk.rs:392-453) on device-resident tables: int64 CUDA tensors (6, 8n, 4),
    (9, 8n, 4), (6, 8n, 4), (8n, 4); the scalars are host arrays (4 limbs each; k_is (6, 4))."""
    n8 = 8 << log_degree
    for t, rows in ((constants_8n, 6), (wire_values_8n, 9), (s_sigma_values_8n, 6), (plonk_z_points_8n, 1)):
        assert t.is_cuda and t.dtype == torch.int64 and t.is_contiguous() and t.numel() == rows * n8 * 4
    if out is None:
        out = torch.empty((n8, 4), dtype=torch.int64, device=constants_8n.device)
    ks = np.ascontiguousarray(k_is, dtype=np.uint64).reshape(6, 4)
    sc = [np.ascontiguousarray(x, dtype=np.uint64).reshape(4) for x in (alpha, beta, gamma, inner_zeta, inner_a)]
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    _lib.check(_lib.load().plk_plonk_vanishing_points_dev(field, log_degree, ctypes.c_void_p(constants_8n.data_ptr()),
                                                          ctypes.c_void_p(wire_values_8n.data_ptr()), ctypes.c_void_p(s_sigma_values_8n.data_ptr()),
                                                          ctypes.c_void_p(plonk_z_points_8n.data_ptr()), p(ks), *[p(x) for x in sc],
                                                          ctypes.c_void_p(out.data_ptr()), _stream()))
    return out


# ---- one round of the inner-product argument (halo.rs:63-124) on device-resident vectors ----
def _limbs(x, n=4):
    return np.ascontiguousarray(x, dtype=np.uint64).reshape(n)


def inner_product_dev(field, a, b):
    """Field::inner_product (field.rs:213-221) -> (1, 4) int64 CUDA tensor."""
    assert a.is_cuda and b.is_cuda and a.is_contiguous() and b.is_contiguous() and a.shape == b.shape
    out = torch.empty((1, a.shape[-1]), dtype=torch.int64, device=a.device)
    _lib.check(_lib.load().plk_field_inner_product_dev(field, ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(b.data_ptr()), a.shape[0],
                                                       ctypes.c_void_p(out.data_ptr()), _stream()))
    return out


def fold_slices_dev(field, lo, hi, scalar_lo, scalar_hi):
    """scalar_lo * lo + scalar_hi * hi (add_slices of scale_slice, halo.rs:117-118)."""
    assert lo.is_cuda and hi.is_cuda and lo.is_contiguous() and hi.is_contiguous() and lo.shape == hi.shape
    out = torch.empty_like(lo)
    sl, sh = _limbs(scalar_lo, lo.shape[-1]), _limbs(scalar_hi, lo.shape[-1])
    _lib.check(_lib.load().plk_field_fold_slices_dev(field, ctypes.c_void_p(lo.data_ptr()), ctypes.c_void_p(hi.data_ptr()),
                                                     sl.ctypes.data_as(ctypes.c_void_p), sh.ctypes.data_as(ctypes.c_void_p), lo.shape[0],
                                                     ctypes.c_void_p(out.data_ptr()), _stream()))
    return out


def fold_generators_dev(curve, g_lo, g_hi, scalar_lo, scalar_hi, lo_zero=None, hi_zero=None):
    """G' = [scalar_lo] G_lo + [scalar_hi] G_hi pair by pair (halo.rs:119-123) -> ((m, 2, L), (m,) zero flags) on device."""
    assert g_lo.is_cuda and g_hi.is_cuda and g_lo.is_contiguous() and g_hi.is_contiguous() and g_lo.shape == g_hi.shape
    m = g_lo.shape[0]
    out = torch.empty_like(g_lo)
    oz = torch.empty((m,), dtype=torch.uint8, device=g_lo.device)
    sl, sh = _limbs(scalar_lo), _limbs(scalar_hi)
    zp = lambda z: ctypes.c_void_p(z.data_ptr()) if z is not None else None
    _lib.check(_lib.load().plk_curve_fold_pairs_dev(curve, m, ctypes.c_void_p(g_lo.data_ptr()), zp(lo_zero), ctypes.c_void_p(g_hi.data_ptr()), zp(hi_zero),
                                                    sl.ctypes.data_as(ctypes.c_void_p), sh.ctypes.data_as(ctypes.c_void_p),
                                                    ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(oz.data_ptr()), _stream()))
    return out, oz


def fold_generators_multi_dev(curve, g, scalars, log_inputs, g_zero=None):
    """out_i = g_i + sum_{t >= 1} [s_t] g_{i + t n_out}, n_out = len(g) >> log_inputs (plk_curve_fold_multi_dev): g (n, 2, L) CUDA,
    scalars (2^log_inputs, 4) CUDA int64 (Montgomery, scalar field) with the scalar of input t at index bitreverse(t)."""
    assert g.is_cuda and g.is_contiguous() and scalars.is_cuda and scalars.is_contiguous() and scalars.shape[0] == 1 << log_inputs
    n_out = g.shape[0] >> log_inputs
    assert n_out << log_inputs == g.shape[0]
    out = torch.empty((n_out,) + tuple(g.shape[1:]), dtype=g.dtype, device=g.device)
    oz = torch.empty((n_out,), dtype=torch.uint8, device=g.device)
    _lib.check(_lib.load().plk_curve_fold_multi_dev(curve, n_out, log_inputs, ctypes.c_void_p(g.data_ptr()),
                                                    ctypes.c_void_p(g_zero.data_ptr()) if g_zero is not None else None,
                                                    ctypes.c_void_p(scalars.data_ptr()), ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(oz.data_ptr()), _stream()))
    return out, oz


def halo_round_lr_dev(curve, halo_a, halo_b, halo_g, pedersen_h, u_prime, l_blinding, r_blinding, g_zero=None):
    """L_j = <a_lo, G_hi> + [l_j] H + [<a_lo, b_hi>] U',  R_j = <a_hi, G_lo> + [r_j] H + [<a_hi, b_lo>] U' (halo.rs:86-93).
    halo_a / halo_b: (n, 4) scalars, halo_g: (n, 2, L) affine generators (g_zero: (n,) identity flags or None); pedersen_h /
    u_prime: (2, L) affine host points; the blinding factors are host scalars.  The msm_parallel, the blinding term and the
    inner-product term are ONE table-free MSM over [G_half..., H, U'] each.  Returns ((2, 2, L), (2,)): L_j then R_j, affine."""
    from .api import CURVE_SCALAR_FIELD
    n = halo_a.shape[0]
    m = n // 2
    sf = CURVE_SCALAR_FIELD[curve]
    L = _CURVE_LIMBS[curve]
    extra = to_device(np.stack([np.ascontiguousarray(pedersen_h, dtype=np.uint64).reshape(2, L), np.ascontiguousarray(u_prime, dtype=np.uint64).reshape(2, L)]))
    # L_j and R_j are independent and, below 2^16 points, pure latency (the window-doubling chain of a table-free MSM):
    # they run on two streams
    main = torch.cuda.current_stream()
    side = _side_stream(halo_a.device)
    side.wait_stream(main)
    outs, zeros = [], []
    for k, (a_half, b_half, g_half, gz, blind) in enumerate(((halo_a[:m], halo_b[m:], halo_g[m:], None if g_zero is None else g_zero[m:], l_blinding),
                                                             (halo_a[m:], halo_b[:m], halo_g[:m], None if g_zero is None else g_zero[:m], r_blinding))):
        st = main if k == 0 else side
        with torch.cuda.stream(st):
            ip = inner_product_dev(sf, a_half.contiguous(), b_half.contiguous())
            scal = torch.cat([a_half, to_device(_limbs(blind).reshape(1, 4)), ip], dim=0).contiguous()
            bases = torch.cat([g_half, extra], dim=0).contiguous()
            zf = None
            if gz is not None:
                zf = torch.cat([gz, torch.zeros(2, dtype=torch.uint8, device=gz.device)]).contiguous()
            pre = msm_precompute_dev(curve, bases, zero=zf, table_free=True)
            xy, z = msm_execute_dev(pre, scal)
            pre.free()  # a table-free context hands its memory back in stream order: no synchronisation
            if st is side:
                for t in (halo_a, halo_b, halo_g, extra, g_zero):
                    if t is not None:
                        t.record_stream(side)
                for t in (xy, z):  # allocated on the side stream, read by torch.stack on the main stream below
                    t.record_stream(main)
        outs.append(xy[0])
        zeros.append(z[0])
    main.wait_stream(side)
    return torch.stack(outs), torch.stack(zeros)


_SIDE_STREAMS = {}


def _side_stream(device):
    key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=key)
    return _SIDE_STREAMS[key]


def halo_round_fold_dev(curve, halo_a, halo_b, halo_g, u_j, u_j_inv, g_zero=None):
    """halo_a' = u^-1 a_hi + u a_lo, halo_b' = u^-1 b_lo + u b_hi, G' = [u^-1] G_lo + [u] G_hi (halo.rs:117-123)."""
    from .api import CURVE_SCALAR_FIELD
    m = halo_a.shape[0] // 2
    sf = CURVE_SCALAR_FIELD[curve]
    a2 = fold_slices_dev(sf, halo_a[m:].contiguous(), halo_a[:m].contiguous(), u_j_inv, u_j)
    b2 = fold_slices_dev(sf, halo_b[:m].contiguous(), halo_b[m:].contiguous(), u_j_inv, u_j)
    g2, gz2 = fold_generators_dev(curve, halo_g[:m].contiguous(), halo_g[m:].contiguous(), u_j_inv, u_j,
                                  None if g_zero is None else g_zero[:m].contiguous(), None if g_zero is None else g_zero[m:].contiguous())
    return a2, b2, g2, gz2


class HaloArgument:
    """One inner-product argument (halo.rs:63-124) behind the C ABI (plk_halo_*): halo_a / halo_b / halo_g are copied into a
    library context and stay in HBM for the log2(n) rounds; per round the caller - who owns the transcript and the RNG -
    calls round_lr(l_j, r_j) (possibly again, halo.rs:83-114) and round_fold(u_j, u_j^-1).  halo_a / halo_b: (n, 4) int64
    CUDA tensors (Montgomery, scalar field); halo_g: (n, 2, L); g_zero: (n,) uint8 or None; pedersen_h / u_prime: (2, L) host."""

    def __init__(self, curve, halo_a, halo_b, halo_g, pedersen_h, u_prime, g_zero=None, freeze_log=0, tables=None, lead_rounds=0, h_index=None, u_index=None,
                 u_prime_scalar=None):
        for t in (halo_a, halo_b, halo_g):
            assert t.is_cuda and t.dtype == torch.int64 and t.is_contiguous()
        n = halo_a.shape[0]
        assert halo_b.shape[0] == n and halo_g.shape[0] == n
        self.curve, self.L = curve, _CURVE_LIMBS[curve]
        h = np.ascontiguousarray(pedersen_h, dtype=np.uint64).reshape(2, self.L)
        u = np.ascontiguousarray(u_prime, dtype=np.uint64).reshape(2, self.L)
        ctx = ctypes.c_void_p()
        zp = ctypes.c_void_p(g_zero.data_ptr()) if g_zero is not None else None
        if tables is not None:
            # tables: the MsmPrecomputation (msm_precompute_dev) of pedersen_g the caller commits with - the first rounds run over it
            # h_index / u_index / u_prime_scalar: pedersen_h and the fixed generator U inside those tables, u_prime = [u_prime_scalar] U
            self._tables = tables  # must outlive the lead rounds
            NO = ctypes.c_size_t(-1).value
            inside = u_prime_scalar is not None and h_index is not None and u_index is not None
            xs = _limbs(u_prime_scalar) if inside else None
            _lib.check(_lib.load().plk_halo_begin_tabled_dev(curve, n, ctypes.c_void_p(halo_a.data_ptr()), ctypes.c_void_p(halo_b.data_ptr()),
                                                             ctypes.c_void_p(halo_g.data_ptr()), zp, tables._ctx, h.ctypes.data_as(ctypes.c_void_p),
                                                             u.ctypes.data_as(ctypes.c_void_p), h_index if inside else NO, u_index if inside else NO,
                                                             xs.ctypes.data_as(ctypes.c_void_p) if inside else None, freeze_log, lead_rounds, _stream(),
                                                             ctypes.byref(ctx)))
        else:
            _lib.check(_lib.load().plk_halo_begin_dev(curve, n, ctypes.c_void_p(halo_a.data_ptr()), ctypes.c_void_p(halo_b.data_ptr()),
                                                      ctypes.c_void_p(halo_g.data_ptr()), zp, h.ctypes.data_as(ctypes.c_void_p),
                                                      u.ctypes.data_as(ctypes.c_void_p), freeze_log, _stream(), ctypes.byref(ctx)))
        self._ctx = ctx

    def __len__(self):
        return int(_lib.load().plk_halo_len(self._ctx))

    @property
    def frozen(self):
        return bool(_lib.load().plk_halo_frozen(self._ctx))

    def round_lr(self, l_blinding, r_blinding):
        """(L_j, R_j) as a (2, 2, L) uint64 array + (2,) identity flags, on the host (the transcript's input)."""
        lr = np.empty((2, 2, self.L), dtype=np.uint64)
        z = np.zeros(2, dtype=np.uint8)
        lb, rb = _limbs(l_blinding), _limbs(r_blinding)
        p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        _lib.check(_lib.load().plk_halo_round_lr(self._ctx, p(lb), p(rb), p(lr), p(z)))
        return lr, z

    def round_fold(self, u_j, u_j_inv):
        u, ui = _limbs(u_j), _limbs(u_j_inv)
        _lib.check(_lib.load().plk_halo_round_fold(self._ctx, u.ctypes.data_as(ctypes.c_void_p), ui.ctypes.data_as(ctypes.c_void_p)))

    def read(self, with_g=True):
        """(halo_a, halo_b[, halo_g, g_zero]) on the host; halo_g of frozen generators exists at length 1 only."""
        n = len(self)
        a, b = np.empty((n, 4), dtype=np.uint64), np.empty((n, 4), dtype=np.uint64)
        p = lambda x: x.ctypes.data_as(ctypes.c_void_p)
        if not with_g:
            _lib.check(_lib.load().plk_halo_read(self._ctx, p(a), p(b), None, None))
            return a, b
        g, gz = np.empty((n, 2, self.L), dtype=np.uint64), np.zeros(n, dtype=np.uint8)
        _lib.check(_lib.load().plk_halo_read(self._ctx, p(a), p(b), p(g), p(gz)))
        return a, b, g, gz

    def free(self):
        if self._ctx:
            _lib.load().plk_halo_free(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

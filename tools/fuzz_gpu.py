"""Randomised parity stress on the GPU (not part of the pytest suites: minutes, not seconds): random sizes, windows,
identity flags, duplicate / opposite generators, skewed scalars, tables and table-free, all five curves; random
polynomial divisions and products; random NTT sizes and batches.  Everything against the oracle, bit for bit.
Usage: python tools/fuzz_gpu.py [seconds]"""
import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import plonky_amd as pa
from plonky_amd import synth
from oracle import bigint_ref as br, oracle_lib as ol
from tests.util import ints_to_array

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
# FUZZ_DEVICES=k: the same cases over a device group of k logical devices (virtual ones on a box with fewer GPUs): every tabled MSM of
# at least k generators, every batch of transforms and every padded batch then runs through the fan-out of multi.hip - still against
# the oracle, bit for bit
if os.environ.get("FUZZ_DEVICES"):
    _k = int(os.environ["FUZZ_DEVICES"])
    import torch
    if torch.cuda.device_count() < _k:
        os.environ["PLK_VIRTUAL_DEVICES"] = str(_k)
    os.environ.setdefault("PLK_MULTI_MIN_LOG_N", "0")
    assert pa.init_devices(_k) == _k
# Every case draws from its OWN generator, seeded from (FUZZ_SEED, case index): a failure names its case seed, and
# FUZZ_CASE=<seed> replays exactly that case.  The summary keeps every 100th case seed and the first / last of every family.
MASTER = int(os.environ.get("FUZZ_SEED", "12345"))
REPLAY = os.environ.get("FUZZ_CASE")
CURVES = [br.TWEEDLEDEE, br.TWEEDLEDUM, br.BLS12_377, br.PALLAS, br.VESTA]
FIELDS = [br.TWEEDLEDEE_BASE, br.TWEEDLEDUM_BASE, br.BLS12_377_SCALAR, br.PALLAS_BASE, br.VESTA_BASE]
t_end = time.time() + budget
counts = {"msm": 0, "ntt": 0, "poly": 0, "fold": 0, "plonk": 0, "misc": 0, "halo": 0}
seeds = {k: [] for k in counts}


# FUZZ_GEOMETRY=1: every MSM case also draws the lane length / reduction group size of its context (PLK_MSM_SLICE / PLK_MSM_GLOG, read when a
# context is built) and every transform case a random factorisation of its size into passes (PLK_NTT_PLAN, read when a plan is built; the
# plan cache is dropped first) - the geometries the sizes alone would never produce.  Off by default so that old case seeds replay as they ran.
GEOMETRY = os.environ.get("FUZZ_GEOMETRY") == "1"


def draw_msm_geometry(rng):
    for k in ("PLK_MSM_SLICE", "PLK_MSM_GLOG"):
        os.environ.pop(k, None)
    if not GEOMETRY:
        return None
    sl, gl = rng.choice([None, 2, 3, 5, 8, 24, 50, 96, 1000]), rng.choice([None, None, 0, 1, 2, 3, 4, 5])
    if sl is not None:
        os.environ["PLK_MSM_SLICE"] = str(sl)
    if gl is not None:
        os.environ["PLK_MSM_GLOG"] = str(gl)
    return sl, gl


def draw_ntt_plan(rng, log_n):
    os.environ.pop("PLK_NTT_PLAN", None)
    if not GEOMETRY:
        return None
    plan = None
    if log_n >= 2 and rng.random() < 0.8:
        parts, left = [], log_n
        while left > 0 and len(parts) < 5:
            a = rng.randrange(1, min(left, 10) + 1)
            parts.append(a)
            left -= a
        if left == 0:
            plan = ",".join(str(a) for a in parts)
            os.environ["PLK_NTT_PLAN"] = plan
    pa.lib.check(pa.lib.load().plk_ntt_clear_cache())
    return plan


def mont(f, vals):
    return ints_to_array([f.to_mont(v % f.p) for v in vals], f.n_limbs)


def halo_case(rng):
    """a whole inner-product argument behind the C ABI (plk_halo_*) against the oracle's composition: random freeze point, plain or over
    the caller's tables (stages of virtual rounds + 2^r-to-1 folds on the curves with the endomorphism)"""
    from plonky_amd import device as dev
    c = rng.choice([br.TWEEDLEDEE, br.TWEEDLEDUM, br.BLS12_377])
    f = c.scalar
    n = rng.choice([1, 2, 4, 8, 32, 128, 512])
    G = (c.gx, c.gy)
    pt = lambda P: np.array([c.base.mont_limbs(P[0]), c.base.mont_limbs(P[1])], dtype=np.uint64)
    g = ol.gen_bases(c.curve_id, n, pt(G), pt(br.ec_mul(c, rng.randrange(1, 1 << 40), G))).reshape(n, 2, c.base.n_limbs)
    h, up = pt(br.ec_mul(c, rng.randrange(1, 1 << 40), G)), pt(br.ec_mul(c, rng.randrange(1, 1 << 40), G))
    a, b = ol.rand_field(f.field_id, rng.randrange(1 << 30), n), ol.rand_field(f.field_id, rng.randrange(1 << 30), n)
    rounds = n.bit_length() - 1
    us = ol.rand_field(f.field_id, rng.randrange(1 << 30), max(rounds, 1))
    bl = ol.rand_field(f.field_id, rng.randrange(1 << 30), 2 * max(rounds, 1))
    kw = {}
    if rng.random() < 0.5:
        # over the caller's commitment tables (plk_halo_begin_tabled_dev): extra generators behind halo_g, H / U inside them or beside
        L = c.base.n_limbs
        gens = [g]
        extra = rng.choice([0, 0, 1, 3])
        if extra:
            gens.append(ol.gen_bases(c.curve_id, extra, h, up).reshape(extra, 2, L))
        if rng.random() < 0.5:
            ubase = br.ec_mul(c, rng.randrange(1, 1 << 40), G)
            x_int = rng.randrange(1, f.p)
            up = pt(br.ec_mul(c, x_int, ubase))
            gens += [h.reshape(1, 2, L), pt(ubase).reshape(1, 2, L)]
            kw = dict(h_index=n + extra, u_index=n + extra + 1, u_prime_scalar=np.array(f.mont_limbs(x_int), dtype=np.uint64))
        kw["tables"] = dev.msm_precompute_dev(c.curve_id, dev.to_device(np.concatenate(gens)))
        kw["lead_rounds"] = rng.choice([0, 1, 2, 3, 4])
    arg = dev.HaloArgument(c.curve_id, dev.to_device(a), dev.to_device(b), dev.to_device(g), h, up, freeze_log=rng.choice([0, 1, 2, 3, 5]), **kw)
    gz = np.zeros(n, dtype=np.uint8)
    for j in range(rounds):
        lr, z = arg.round_lr(bl[2 * j], bl[2 * j + 1])
        exp, ez = ol.halo_round_lr(c.curve_id, f.field_id, a, b, g, h, up, bl[2 * j], bl[2 * j + 1])
        assert list(z) == list(ez) and np.array_equal(lr, exp), ("halo lr", c.name, n, j)
        u_inv = ol.field_unop(f.field_id, "inverse", us[j].reshape(1, 4))[0]
        arg.round_fold(us[j], u_inv)
        a, b, g, gz = ol.halo_round_fold(c.curve_id, f.field_id, a, b, g, us[j], u_inv)
    fa, fb, fg, fgz = arg.read()
    arg.free()
    assert np.array_equal(fa, a) and np.array_equal(fb, b) and np.array_equal(fg, g) and np.array_equal(fgz, np.asarray(gz, dtype=np.uint8)), ("halo final", c.name, n)


def one_case(rng):
    kind = rng.choice(["msm", "msm", "ntt", "poly", "fold", "plonk", "misc", "halo"])
    if kind == "halo":
        halo_case(rng)
        return kind
    if kind == "msm":
        c = rng.choice(CURVES)
        n = rng.choice([1, 2, 3, 7, 33, 100, 257, 1000, 3000, 5000])
        G = (c.gx, c.gy)
        D = br.ec_mul(c, rng.randrange(1, 1 << 40), G)
        bases = ol.gen_bases(c.curve_id, n, mont(c.base, [G[0], G[1]]).reshape(2, -1), mont(c.base, [D[0], D[1]]).reshape(2, -1))
        zero = np.zeros(n, dtype=np.uint8)
        for _ in range(rng.randrange(0, 3)):
            zero[rng.randrange(n)] = 1
        if n > 3:
            bases[1] = bases[0]                                  # duplicate
            neg = bases[2].copy()
            neg[1] = ol.field_unop(c.base.field_id, "neg", neg[1:2])[0]
            bases[3] = neg                                       # opposite of bases[2]
        style = rng.choice(["uniform", "small", "equal", "edge"])
        r = c.scalar.p
        if style == "uniform":
            sc = synth.rand_field(c.scalar.field_id, rng.randrange(1 << 30), n)
        elif style == "small":
            sc = mont(c.scalar, [rng.randrange(0, 4) for _ in range(n)])
        elif style == "equal":
            v = rng.randrange(r)
            sc = mont(c.scalar, [v] * n)
        else:
            sc = mont(c.scalar, [rng.choice([0, 1, r - 1, r - 2, (r - 1) // 2, 1 << 200]) for _ in range(n)])
        exp, ez = ol.MsmPrecomputation(c.curve_id, bases, 8, zero=zero, threads=8).execute(sc, parallel=True, threads=8)
        tf = rng.random() < 0.5
        win = rng.choice([0, 0, 3, 5, 8, 11] + ([] if tf else [13, 14, 16, 18, 20]))
        geom = draw_msm_geometry(rng)
        pre = pa.msm_precompute(c.curve_id, bases, 8, zero=zero, device_window=win, table_free=tf)
        got, gz = pa.msm_execute_parallel(pre, sc)
        assert gz == ez and (ez or np.array_equal(got, exp)), ("msm", c.name, n, style, tf, win, geom)
        if rng.random() < 0.3:
            # the same vector inside a batch (shared reduction), next to other vectors
            k = rng.choice([2, 3, 17])
            vecs = [synth.rand_field(c.scalar.field_id, rng.randrange(1 << 30), n) for _ in range(k)]
            at = rng.randrange(k)
            vecs[at] = sc
            bo, bz = pa.msm_execute_batch(pre, np.stack(vecs))
            assert int(bz[at]) == ez and (ez or np.array_equal(bo[at], exp)), ("msm batch", c.name, n, style, tf, win, k)
        pre.free()
    elif kind == "ntt":
        f = rng.choice(FIELDS + [br.BLS12_377_BASE])  # the sixth field: plain and zero-padded transforms only (6 limbs)
        log_n = rng.randrange(0, 17)
        batch = rng.choice([1, 1, 2, 5])
        x = synth.rand_field(f.field_id, rng.randrange(1 << 30), batch << log_n).reshape(batch, 1 << log_n, f.n_limbs)
        opre = ol.FftPrecomputation(f.field_id, 1 << log_n)
        inv = rng.random() < 0.5
        plan = draw_ntt_plan(rng, log_n)  # named in a failure by the case seed (FUZZ_CASE replays it under the same FUZZ_GEOMETRY)
        got = pa.api.fft_batch(f.field_id, x, inverse=inv)
        for b in range(batch):
            want = opre.ifft_with_precomputation_power_of_2(x[b], threads=8) if inv else opre.fft_with_precomputation_power_of_2(x[b], threads=8)
            assert np.array_equal(got[b], want), ("ntt", f.name, log_n, batch, inv, plan)
        if f.n_limbs == 6 and rng.random() < 0.5:
            # fft_with_precomputation (fft.rs:61-80): any length, zero-padded to the next power of two
            ln = rng.randrange(1, (1 << log_n) + 1)
            pre = pa.fft_precompute(f.field_id, ln)
            assert np.array_equal(pa.fft_with_precomputation(x[0][:ln], pre), ol.FftPrecomputation(f.field_id, ln).fft_with_precomputation(x[0][:ln])), ("padded6", log_n, ln)
        if f.n_limbs == 4 and log_n >= 3 and rng.random() < 0.5:
            # padded evaluation of shorter polynomials on the same domain (polynomials_to_values_padded)
            lens = [rng.randrange(1, (1 << log_n) // 8 + 1) for _ in range(batch)]
            polys = [x[b][: lens[b]] for b in range(batch)]
            pv = pa.polynomials_to_values_padded(polys, pa.fft_precompute(f.field_id, 1 << log_n))
            for b in range(batch):
                assert np.array_equal(pv[b], ol.poly_to_values_padded(opre, polys[b], threads=8)), ("padded", f.name, log_n, lens[b])
    elif kind == "poly":
        f = rng.choice(FIELDS)
        la = rng.randrange(1, 6000)
        a = synth.rand_field(f.field_id, rng.randrange(1 << 30), la)
        if rng.random() < 0.5:
            n = rng.choice([1, 2, 3, 8, 64, 100, 1024, rng.randrange(1, 3000)])
            m = np.zeros((la + n, 4), dtype=np.uint64)
            m[n:] = a
            m[:la] = ol.field_binop(f.field_id, "sub", m[:la].copy(), a)
            if rng.random() < 0.3:
                m = np.concatenate([m, np.zeros((rng.randrange(1, 9), 4), dtype=np.uint64)])
            got = pa.polynomial_divide_by_z_h(f.field_id, m, n)
            assert np.array_equal(got, ol.poly_divide_by_z_h(f.field_id, m, n, threads=8)), ("divide", f.name, la, n)
        else:
            b = synth.rand_field(f.field_id, rng.randrange(1 << 30), rng.randrange(1, 3000))
            assert np.array_equal(pa.polynomial_mul(f.field_id, a, b), ol.poly_mul(f.field_id, a, b, threads=8)), ("mul", f.name, la)
    elif kind == "plonk":
        # the vanishing points of Prover::vanishing_poly on random tables (every gate contributes to every point)
        f = rng.choice(FIELDS)
        degree = rng.choice([1, 2, 4, 16, 64, 256])
        n8 = 8 * degree
        seed = rng.randrange(1 << 30)
        mk = lambda rows, k: ol.rand_field(f.field_id, seed + k, rows * n8).reshape(rows, n8, 4)
        sc = ol.rand_field(f.field_id, seed + 9, 11)
        args = (mk(6, 1), mk(9, 2), mk(6, 3), mk(1, 4)[0], sc[:6], sc[6], sc[7], sc[8], sc[9], sc[10])
        if rng.random() < 0.5:
            # half of every table (sometimes the challenges too) replaced by words at the edges of the device's input conversion
            # and lazy limbs: 0, 1, p - 1, every cut boundary +- 1 (tests/test_oracle_plonk.py extreme_words)
            from tests.test_oracle_plonk import extreme_words
            ext = ints_to_array(extreme_words(f), 4)
            nrng = np.random.default_rng(seed)
            for t in args[:4]:
                flat = t.reshape(-1, 4)
                idx = nrng.random(flat.shape[0]) < 0.5
                flat[idx] = ext[nrng.integers(0, len(ext), int(idx.sum()))]
            if rng.random() < 0.5:
                idx = nrng.random(11) < 0.5
                sc[idx] = ext[nrng.integers(0, len(ext), int(idx.sum()))]
        assert np.array_equal(pa.api.vanishing_points(f.field_id, degree, *args), ol.vanishing_points(f.field_id, degree, *args, threads=8)), ("plonk", f.name, degree)
    elif kind == "misc":
        # batch inversion with zeros, byte encodings round trip
        f = rng.choice(FIELDS + [br.BLS12_377_BASE])
        n = rng.choice([1, 7, 8, 9, 100, 1025])
        x = ol.rand_field(f.field_id, rng.randrange(1 << 30), n)
        for _ in range(rng.randrange(0, 3)):
            x[rng.randrange(n)] = 0
        inv, none = pa.api.batch_multiplicative_inverse_opt(f.field_id, x)
        nz = ~x.any(axis=1) == False
        assert np.array_equal(none.astype(bool), ~nz) and (not nz.any() or np.array_equal(inv[nz], ol.batch_inverse(f.field_id, x[nz]))), ("binv", f.name, n)
        b = pa.api.field_to_bytes(f.field_id, x)
        assert np.array_equal(b, ol.field_to_bytes(f.field_id, x)) and np.array_equal(pa.api.field_from_bytes(f.field_id, b), x), ("bytes", f.name, n)
    else:
        c = rng.choice(CURVES)
        m = rng.choice([1, 5, 64])
        G = (c.gx, c.gy)
        D = br.ec_mul(c, rng.randrange(1, 1 << 40), G)
        pts = ol.gen_bases(c.curve_id, 2 * m, mont(c.base, [G[0], G[1]]).reshape(2, -1), mont(c.base, [D[0], D[1]]).reshape(2, -1))
        sa = synth.rand_field(c.scalar.field_id, rng.randrange(1 << 30), 1)[0]
        sb = synth.rand_field(c.scalar.field_id, rng.randrange(1 << 30), 1)[0]
        got, gz = pa.fold_generators(c.curve_id, pts[:m], pts[m:], sa, sb)
        for i in range(m):
            p1, z1 = ol.scalar_mul(c.curve_id, sa, pts[i], 0)
            p2, z2 = ol.scalar_mul(c.curve_id, sb, pts[m + i], 0)
            exp, ez = ol.affine_add(c.curve_id, p1, z1, p2, z2)
            assert int(gz[i]) == ez and (ez or np.array_equal(got[i], exp)), ("fold", c.name, m, i)
    return kind


case_idx = 0
while time.time() < t_end:
    case_seed = int(REPLAY, 0) if REPLAY else (MASTER * 1000003 + case_idx * 7919 + 1) & ((1 << 62) - 1)
    case_idx += 1
    try:
        kind = one_case(random.Random(case_seed))
    except Exception:
        print("FAILED case %d: replay with FUZZ_CASE=%#x" % (case_idx - 1, case_seed), flush=True)
        raise
    counts[kind] += 1
    seeds[kind].append(case_seed)
    if REPLAY:
        break
print("fuzz ok (FUZZ_SEED=%d, %d cases%s):" % (MASTER, case_idx, ", device group of %s" % os.environ["FUZZ_DEVICES"] if os.environ.get("FUZZ_DEVICES") else ""), counts)
for k, v in seeds.items():
    if v:
        print("  %-5s %6d cases  first %#x  last %#x  every 100th: %s" % (k, len(v), v[0], v[-1], " ".join("%#x" % x for x in v[::100][:40])))

"""benchlib.quotient -- bench.py --workload quotient: the two heaviest kernels of the callers either side of the path."""
import json
import time

from .common import HBM_PEAK_GBS, METRIC, Ceilings, gpu_identity, kernel_source_hash, latest_profile


def run_quotient(args):
    """--workload quotient: k_vanishing_points (four launches per call), k_fold_pairs_glv and the 4-to-1 fold k_fold_multi_glv, timed with HIP events on the launch
    stream, priced against the same ceilings as the headline kernels.  One GPU; correctness of both is the GPU suite's business
    (tests/test_gpu_plonk.py, tests/test_gpu_halo.py) - here the fold is checked by its closed form, the numerator by determinism."""
    import numpy as np
    import torch
    from plonky_amd import device as dev, lib, synth
    from plonky_amd.selfcheck import GENERATORS, _add, _mul
    from plonky_amd.synth import MODULI
    assert args.gpus == 1 and torch.cuda.is_available()
    dev.init(0)
    F, CURVE = 1, 0          # the circuit's scalar field is TweedledumBase (Tweedledee's scalar field)
    log_degree = args.log_n                 # a circuit of 2^log_n gates: 8n = 2^(log_n + 3) points, first IPA round = 2^(log_n - 1) pairs
    n8 = 8 << log_degree
    rnd = lambda seed, rows: dev.to_device(synth.rand_field(F, seed, rows * n8)).reshape(rows, n8, 4)
    consts, wires, sigma, z = rnd(1, 6), rnd(2, 9), rnd(3, 6), rnd(4, 1).reshape(n8, 4)
    k_is = synth.rand_field(F, 9, 6)
    alpha, beta, gamma, zeta = synth.rand_field(F, 10, 4)
    a_coeff = np.zeros(4, dtype=np.uint64)
    out = torch.empty((n8, 4), dtype=torch.int64, device="cuda")
    vanish = lambda: dev.vanishing_points_dev(F, log_degree, consts, wires, sigma, z, k_is, alpha, beta, gamma, zeta, a_coeff, out=out)
    p = MODULI[0]
    G = GENERATORS[CURVE]
    D = _mul(p, 424242, G)
    g0 = np.stack([synth.mont(0, G[0]), synth.mont(0, G[1])])
    dd = np.stack([synth.mont(0, D[0]), synth.mont(0, D[1])])
    m = max(1, (1 << log_degree) // 2)
    gens = dev.gen_bases_dev(CURVE, 2 * m, g0, dd)
    u = synth.rand_field(F, 11, 1)[0]
    r = MODULI[F]
    ui = synth.to_int(u) * pow(1 << 256, -1, r) % r
    u_inv = np.array(synth.mont(F, pow(ui, -1, r)), dtype=np.uint64)
    fold = lambda: dev.fold_generators_dev(CURVE, gens[:m].contiguous(), gens[m:].contiguous(), u_inv, u)
    # the 4-to-1 fold of two rounds at once (plk_curve_fold_multi_dev: what the argument behind the C ABI runs for its first two rounds)
    s_ints = [1] + [synth.to_int(row) % r or 1 for row in synth.rand_field(F, 12, 3)]
    rev2 = (0, 2, 1, 3)
    s_multi = np.zeros((4, 4), dtype=np.uint64)
    for t in range(4):
        s_multi[rev2[t]] = synth.mont(F, s_ints[t])
    s_multi_d = dev.to_device(s_multi)
    q = max(1, (2 * m) // 4)
    fold4 = (lambda: dev.fold_generators_multi_dev(CURVE, gens, s_multi_d, 2)) if 2 * m >= 4 else None

    def timed(fn, steps, warm):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / steps

    t0 = time.perf_counter()
    v_ms = timed(vanish, args.steps, args.warmup)
    f_ms = timed(fold, max(1, args.steps // 2), 1)
    f4_ms = timed(fold4, max(1, args.steps // 2), 1) if fold4 else None
    elapsed = time.perf_counter() - t0
    first = out.clone()
    vanish()
    g2, gz2 = fold()
    torch.cuda.synchronize()
    # fold closed form: [u^-1] (G0 + i D) + [u] (G0 + (m + i) D) = [u^-1 + u] G0 + [u^-1 i + u (m + i)] D, checked at i = 0 and i = m - 1
    ok = True
    for i in (0, m - 1):
        exp = _add(p, _mul(p, (pow(ui, -1, r) + ui) % r, G), _mul(p, (pow(ui, -1, r) * i + ui * (m + i)) % r, D))
        got = dev.to_host(g2[i])
        ok = ok and (synth.from_mont(0, got[0]), synth.from_mont(0, got[1])) == exp
    checks = {"fold_closed_form_bit_exact": bool(ok and not gz2.any().item()), "vanishing_points_deterministic": bool(torch.equal(first, out))}
    if fold4:
        # out_i = sum_t s_t (G0 + (i + t q) D), s_0 = 1: [sum s_t] G0 + [sum s_t (i + t q)] D, checked at i = 0 and i = q - 1
        g4, gz4 = fold4()
        torch.cuda.synchronize()
        ok4 = True
        for i in (0, q - 1):
            exp = _add(p, _mul(p, sum(s_ints) % r, G), _mul(p, sum(sv * (i + t * q) for t, sv in enumerate(s_ints)) % r, D))
            got = dev.to_host(g4[i])
            ok4 = ok4 and (synth.from_mont(0, got[0]), synth.from_mont(0, got[1])) == exp
        checks["fold_4_to_1_closed_form_bit_exact"] = bool(ok4 and not gz4.any().item())
    ceil = Ceilings(lib.load(), gpu_identity(torch, 0), 4, "tweedledee")
    pmc = {}
    try:
        with open(latest_profile("pmc_traffic_quotient.json")) as fh:
            pmc = json.load(fh)
    except (OSError, ValueError):
        pass
    src_hash = kernel_source_hash()

    def traffic(k):
        e = pmc.get(k)
        return (e["bytes_per_call"], e.get("source")) if e and pmc.get("kernel_source_sha") == src_hash and e.get("log_n") == args.log_n else (None, None)

    def entry(kernel, modmul_per_unit, units, ms, alg_bytes, note):
        gmm = modmul_per_unit * units / (ms * 1e-3) / 1e9
        tr, trs = traffic(kernel)
        return ceil.entry(kernel, gmm, gmm, ms, {
            "traffic": tr, "traffic_source": trs,
            "hbm": {"achieved": alg_bytes / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "algorithmic_bytes_per_launch": alg_bytes}, "note": note})

    rooflines = {
        "vanishing_points": entry("k_vanishing_points", 151.0, n8, v_ms, 30.0 * 32 * n8,
                                  "four launches per call (launch_ms = the call); 151 multiplication-equivalents per point = the 19 003 multiplier "
                                  "instructions of a point / 126 per product (a squaring counts 0.74; round 3 quoted ~170 for what was 200 by this "
                                  "count; DESIGN.md 4c); algorithmic bytes: "
                                  "29 elements read + 1 written per point"),
        "fold_pairs": entry("k_fold_pairs_glv", 2570.0, m, f_ms, 3.0 * 64 * m,
                            "G' = [u^-1] G_lo + [u] G_hi along the endomorphism (plk_curve_fold_pairs_dev): ~130 doublings (6M + 3S) + ~130 mixed additions "
                            "(8M + 2S) + two inversions per pair; the argument behind the C ABI folds scaled, lo + [u^2] hi: ~65 additions"),
    }
    if fold4:
        rooflines["fold_multi"] = entry("k_fold_multi_glv", 3300.0, q, f4_ms, 5.0 * 64 * q,
                                        "out_i = g_i + sum of three [s_t] g_(i + t q) (plk_curve_fold_multi_dev, two rounds of the argument at once): one chain of "
                                        "~128 doublings (6M + 3S) per OUTPUT + ~64 mixed additions (8M + 2S, half of them with a multiplication by beta) per "
                                        "input + the inversions of the operand preparation and the affine result: ~3300 multiplications per output; "
                                        "units = outputs; launch_ms = digits + preparation + main kernel")
    result = {
        "metric": METRIC, "value": n8 / (v_ms * 1e-3) / 1e6,
        "unit": "M points/s of the quotient numerator (the fold is reported in components)", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": v_ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
        "config": {"workload": "quotient numerator of a 2^%d-gate circuit (8n = 2^%d points) + generator fold of 2^%d pairs" % (log_degree, log_degree + 3, m.bit_length() - 1),
                   "log_n": args.log_n, "curve": "tweedledee", "kernel_source_sha": src_hash, "gpu": gpu_identity(torch, 0)},
        "components": {"vanishing_points_ms": v_ms, "vanishing_mpoints_per_s": n8 / (v_ms * 1e-3) / 1e6, "fold_pairs_ms": f_ms,
                       "fold_mpairs_per_s": m / (f_ms * 1e-3) / 1e6, "fold_4_to_1_ms": f4_ms,
                       "fold_4_to_1_minputs_per_s": (2 * m / (f4_ms * 1e-3) / 1e6) if f4_ms else None, "wall_s": elapsed},
        "checks": checks, "roofline": rooflines["vanishing_points"], "rooflines": rooflines,
    }
    print(json.dumps(result), flush=True)
    assert all(checks.values()), "self-check failed: %r" % checks

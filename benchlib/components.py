"""benchlib.components -- the component timings of the headline line (separate loops after the timed region): each transform /
MSM alone and nine at once, the quotient-path callers, the one-shot msm_parallel, the opening argument, and the drop-in HOST-pointer
entry points (PCIe inside).  Every function takes the run's state `c` (a namespace built by headline.run) and fills c.comp."""
import ctypes
import time

from .common import SEED_NTT


def loop_time(c, fn, iters, warm=3):
    """Warmed with three calls of the same work, then `iters` calls between two synchronisations: a ten-call loop straight after
    host-side preparation measures the GPU's clock ramp, not the kernel (profiles/r04_ntt_harness_reconcile.txt)."""
    iters = max(1, iters)
    for _ in range(warm):
        fn()
    c.sync()
    t_ = time.perf_counter()
    for _ in range(iters):
        fn()
    c.sync()
    return (time.perf_counter() - t_) / iters


def ntt_components(c):
    args, comp, dev, api, synth, np, torch = c.args, c.comp, c.dev, c.api, c.synth, c.np, c.torch
    NTT_FIELD, n, x, y, world, rank = c.NTT_FIELD, c.n, c.x, c.y, c.world, c.rank
    loop = lambda fn, iters, warm=3: loop_time(c, fn, iters, warm)
    tn = loop(lambda: dev.ntt_dev(NTT_FIELD, x, out=y), max(args.steps, 100))
    comp["ntt_ms"] = tn * 1e3
    comp["ntt_melems_per_s"] = world * n / tn / 1e6
    # the prover transforms its 9 wire polynomials together (plonk_util.rs:169-190): same kernels, one call
    xb = x.unsqueeze(0).repeat(9, 1, 1).contiguous()
    yb = torch.empty_like(xb)
    tb = loop(lambda: dev.ntt_dev(NTT_FIELD, xb, out=yb), max(args.steps, 30))
    comp["ntt_batch9_ms"] = tb * 1e3
    comp["ntt_batch9_melems_per_s"] = world * 9 * n / tb / 1e6
    del xb, yb
    # the quotient path either side of the transforms (polynomial.rs:330-380, plonk_util.rs:179-190)
    # at the sizes this n implies: divide_by_z_h of a degree < n polynomial by Z_H of n/8, LDE of 9 wires n/8 -> n
    if args.log_n >= 13:
        nq = n // 8
        # m = q0 * (X^nq - 1) for a random q0 of 7 nq coefficients: m[i] = q0[i - nq] - q0[i]
        q0 = synth.rand_field(NTT_FIELD, SEED_NTT + 100 + rank, 7 * nq)
        zpad = np.zeros((nq, 4), dtype=np.uint64)
        m = dev.to_device(api.field_op(NTT_FIELD, "sub", np.concatenate([zpad, q0]), np.concatenate([q0, zpad])))
        q_out = torch.empty((n, 4), dtype=torch.int64, device="cuda")
        comp["divide_by_z_h_ms"] = loop(lambda: dev.divide_by_z_h_dev(NTT_FIELD, m, nq, out=q_out), args.steps) * 1e3
        w = dev.to_device(synth.rand_field(NTT_FIELD, SEED_NTT + 200 + rank, 9 * nq)).reshape(9, nq, 4)
        ev = torch.empty((9, n, 4), dtype=torch.int64, device="cuda")
        comp["lde9_ms"] = loop(lambda: dev.ntt_padded_dev(NTT_FIELD, w, args.log_n, out=ev), args.steps // 2) * 1e3
        comp["quotient_path_note"] = "divide_by_z_h: degree < 2^%d by Z_H of 2^%d (2 fused transforms); lde9: 9 x 2^%d coefficients -> 2^%d evaluations" % (
            args.log_n, args.log_n - 3, args.log_n - 3, args.log_n)
        if not args.no_check:
            q_host = dev.to_host(q_out)
            comp["_q_check"] = bool(np.array_equal(q_host[: 7 * nq], q0) and not q_host[7 * nq:].any())
        del m, q_out, ev


def msm_components(c):
    args, comp, dev, torch, cv = c.args, c.comp, c.dev, c.torch, c.cv
    n, s, oxy, oz, pre, bases, world, strong, batch, plan, CURVE = c.n, c.s, c.oxy, c.oz, c.pre, c.bases, c.world, c.strong, c.batch, c.plan, c.CURVE
    sync = c.sync
    loop = lambda fn, iters, warm=3: loop_time(c, fn, iters, warm)
    tm = loop(lambda: dev.msm_execute_dev(pre, s, oxy, oz), args.steps)
    pairs = batch * (n if strong else world * n)
    comp["msm_ms"] = tm * 1e3
    comp["msm_mpairs_per_s"] = (plan.pairs_local() if args.emulate_rank else pairs) / tm / 1e6
    if strong:
        return
    # msm_execute_parallel with its OWN return type (curve_msm.rs:102-157 returns the ProjectivePoint, not normalised): the reduction ends
    # with six products instead of the inversion.  Reported beside msm_ms, which stays the affine form of rounds 1-5 (and of `value`).
    pxyz = torch.empty((1, 3, cv["limbs"]), dtype=torch.int64, device="cuda")
    pz = torch.empty((1,), dtype=torch.uint8, device="cuda")
    tp = loop(lambda: dev.msm_execute_dev(pre, s, pxyz, pz, projective=True), args.steps)
    comp["msm_projective_ms"] = tp * 1e3
    if not args.no_check:
        # the same point: x / z, y / z against the affine result of the call above
        sync()
        dev.msm_execute_dev(pre, s, oxy, oz)
        sync()
        f = cv["base_field"]
        P = c.MODULI[f]
        hx = dev.to_host(pxyz)[0]
        x, y, z = (c.synth.from_mont(f, hx[k]) for k in range(3))
        zi = pow(z, -1, P) if z else 0
        ha = dev.to_host(oxy).reshape(-1, 2, cv["limbs"])[0]
        comp["_proj_check"] = bool(int(pz.item()) == 0 and (x * zi % P, y * zi % P) == (c.synth.from_mont(f, ha[0]), c.synth.from_mont(f, ha[1])))
    # commit_polynomials (plonk_util.rs:215-231): the 9 wire polynomials against the same generators, one call
    sb = s.unsqueeze(0).repeat(9, 1, 1).contiguous()
    oxy9 = torch.empty((9, 2, cv["limbs"]), dtype=torch.int64, device="cuda")
    oz9 = torch.empty((9,), dtype=torch.uint8, device="cuda")
    tb = loop(lambda: dev.msm_execute_dev(pre, sb, oxy9, oz9), args.steps // 4, warm=2)
    comp["msm_batch9_ms"] = tb * 1e3
    comp["msm_batch9_mpairs_per_s"] = world * 9 * n / tb / 1e6
    if not args.no_check:
        comp["_b9_check"] = bool(torch.equal(oxy9, oxy.expand(9, 2, cv["limbs"])) and int(oz9.sum().item()) == 0)
    del sb
    # msm_parallel (curve_msm.rs:54-61): generators used once -> precompute included, table-free mode
    sync()
    t1 = time.perf_counter()
    reps = max(1, args.steps // 4)
    for _ in range(reps):
        pre1 = dev.msm_precompute_dev(CURVE, bases, table_free=True)
        dev.msm_execute_dev(pre1, s, oxy9[:1], oz9[:1])
        sync()
        pre1.free()
    comp["msm_parallel_one_shot_ms"] = (time.perf_counter() - t1) / reps * 1e3
    if not args.no_check:
        comp["_os_check"] = bool(torch.equal(oxy9[:1], oxy) and int(oz9[0].item()) == 0)


def ipa_component(c):
    """The opening argument of the same proof (halo.rs:63-124): all log2(n) rounds behind the C ABI, over the prover's
    commitment tables [pedersen_g .., pedersen_h, U] (plk_halo_begin_tabled_dev), full-size challenges."""
    args, comp, dev, synth, np, torch, cv = c.args, c.comp, c.dev, c.synth, c.np, c.torch, c.cv
    n, bases, CURVE, p, G, MODULI, _mul = c.n, c.bases, c.CURVE, c.p, c.G, c.MODULI, c._mul
    SCAL = cv["scalar_field"]
    r_mod = MODULI[SCAL]
    mm = lambda f, v: np.array(synth.mont(f, v), dtype=np.uint64)
    pt = lambda P: np.stack([mm(cv["base_field"], P[0]), mm(cv["base_field"], P[1])])
    x_int = 0x1F3D5B79A2C4E6081F3D5B79A2C4E6081F3D5B79A2C4E608 % r_mod
    UB = _mul(p, 13, G)
    Hh, Uu = _mul(p, 11, G), _mul(p, x_int, UB)
    ha, hb = dev.to_device(synth.rand_field(SCAL, 1, n)), dev.to_device(synth.rand_field(SCAL, 2, n))
    us = [synth.to_int(row) % r_mod or 1 for row in synth.rand_field(SCAL, 3, args.log_n)]
    ums = [(mm(SCAL, u), mm(SCAL, pow(u, -1, r_mod))) for u in us]
    bl = [(mm(SCAL, 100 + j), mm(SCAL, 200 + j)) for j in range(args.log_n)]
    tables = dev.msm_precompute_dev(CURVE, torch.cat([bases, dev.to_device(pt(Hh)[None]), dev.to_device(pt(UB)[None])]))

    def ipa():
        t_0 = time.perf_counter()
        arg = dev.HaloArgument(CURVE, ha, hb, bases, pt(Hh), pt(Uu), tables=tables, h_index=n, u_index=n + 1, u_prime_scalar=mm(SCAL, x_int))
        lrs = []
        for j in range(args.log_n):
            lrs.append(arg.round_lr(*bl[j]))
            arg.round_fold(*ums[j])
        fin = arg.read()
        t_ = time.perf_counter() - t_0
        arg.free()
        return t_, lrs, fin

    ipa()
    t_a, lr_a, fin_a = ipa()
    t_b, lr_b, fin_b = ipa()
    comp["ipa_ms"] = min(t_a, t_b) * 1e3
    comp["ipa_note"] = "all %d rounds of one opening at n = 2^%d over the prover's tables (H, U inside), best of two; L / R back on the host every round" % (args.log_n, args.log_n)
    if not args.no_check:
        comp["_ipa_check"] = bool(all(np.array_equal(a[0], b[0]) for a, b in zip(lr_a, lr_b)) and all(np.array_equal(x_, y_) for x_, y_ in zip(fin_a, fin_b)))
    tables.free()
    del ha, hb


def host_pointer_components(c, do_ntt_c, do_msm_c):
    """The drop-in entry points: HOST pointers, PCIe included - what an unmodified plonk.rs gets (plonk_util.rs:169-231)."""
    args, comp, dev, lib, L, np, cv = c.args, c.comp, c.dev, c.lib, c.L, c.np, c.cv
    NTT_FIELD, n = c.NTT_FIELD, c.n
    x_host, y, s_host, pre, oxy = getattr(c, "x_host", None), getattr(c, "y", None), getattr(c, "s_host", None), getattr(c, "pre", None), getattr(c, "oxy", None)
    PCIE_GBS = 56.0  # measured both ways on this platform (profiles/r03_h2d_probe.txt)
    vp = ctypes.c_void_p
    host = {}
    if do_ntt_c:
        hin = [np.ascontiguousarray(x_host.copy()) for _ in range(9)]
        hout = [np.zeros_like(x_host) for _ in range(9)]   # touched: no first-touch page faults inside the timing
        ins = (vp * 9)(*[a.ctypes.data for a in hin])
        outs = (vp * 9)(*[a.ctypes.data for a in hout])
        lib.check(L.plk_ntt_batch(NTT_FIELD, args.log_n, 0, 9, ins, outs))
        reps = max(2, args.steps // 4)
        t1 = time.perf_counter()
        for _ in range(reps):
            lib.check(L.plk_ntt_batch(NTT_FIELD, args.log_n, 0, 9, ins, outs))
        host["host_ntt9_ms"] = (time.perf_counter() - t1) / reps * 1e3
        host["host_ntt9_pcie_floor_ms"] = 9 * n * 32 / (PCIE_GBS * 1e9) * 1e3   # one direction; the two directions overlap
        lib.check(L.plk_ntt(NTT_FIELD, args.log_n, 0, vp(hin[0].ctypes.data), vp(hout[1].ctypes.data)))
        t1 = time.perf_counter()
        for _ in range(reps):
            lib.check(L.plk_ntt(NTT_FIELD, args.log_n, 0, vp(hin[0].ctypes.data), vp(hout[1].ctypes.data)))
        host["host_ntt_ms"] = (time.perf_counter() - t1) / reps * 1e3
        # the reference's own calling pattern: nine Rayon workers, one transform each (plonk_util.rs:173-176) - nine host threads,
        # each on its own lane of the library (ctypes releases the GIL for the duration of a call)
        import threading

        def nine_threads():
            ts = [threading.Thread(target=lambda b=b: lib.check(L.plk_ntt(NTT_FIELD, args.log_n, 0, vp(hin[b].ctypes.data), vp(hout[b].ctypes.data))))
                  for b in range(9)]
            for t in ts:
                t.start()
            for t in ts:
                t.join()
        nine_threads()
        t1 = time.perf_counter()
        for _ in range(reps):
            nine_threads()
        host["host_ntt9_nine_threads_ms"] = (time.perf_counter() - t1) / reps * 1e3
        if not args.no_check:
            checks_host_ntt = bool(np.array_equal(hout[0], dev.to_host(y)) and np.array_equal(hout[8], hout[0]) and np.array_equal(hout[1], hout[0]))
            host["_ntt_ok"] = checks_host_ntt
        del hin, hout
    if do_msm_c:
        hs = [np.ascontiguousarray(s_host.copy()) for _ in range(9)]
        hxy = np.zeros((9, 2, cv["limbs"]), dtype=np.uint64)
        hz = np.zeros(9, dtype=np.uint8)
        ptrs = (vp * 9)(*[a.ctypes.data for a in hs])
        lib.check(L.plk_msm_execute_batch(pre._ctx, 9, ptrs, n, vp(hxy.ctypes.data), vp(hz.ctypes.data)))
        reps = max(2, args.steps // 4)
        t1 = time.perf_counter()
        for _ in range(reps):
            lib.check(L.plk_msm_execute_batch(pre._ctx, 9, ptrs, n, vp(hxy.ctypes.data), vp(hz.ctypes.data)))
        host["host_commit9_ms"] = (time.perf_counter() - t1) / reps * 1e3
        host["host_commit9_pcie_floor_ms"] = 9 * n * 32 / (PCIE_GBS * 1e9) * 1e3
        if "msm_batch9_ms" in comp:
            host["host_commit9_vs_max_pcie_device"] = host["host_commit9_ms"] / max(host["host_commit9_pcie_floor_ms"], comp["msm_batch9_ms"])
        lib.check(L.plk_msm_execute(pre._ctx, vp(hs[0].ctypes.data), n, vp(hxy.ctypes.data), vp(hz.ctypes.data)))
        t1 = time.perf_counter()
        for _ in range(reps):
            lib.check(L.plk_msm_execute(pre._ctx, vp(hs[0].ctypes.data), n, vp(hxy.ctypes.data), vp(hz.ctypes.data)))
        host["host_msm_ms"] = (time.perf_counter() - t1) / reps * 1e3
        if not args.no_check:
            host["_msm_ok"] = bool(np.array_equal(hxy[0].view(np.int64), oxy[0].cpu().numpy()) and not hz.any())
        del hs
    if "host_ntt9_ms" in host and "ntt_batch9_ms" in comp:
        host["host_ntt9_vs_max_pcie_device"] = host["host_ntt9_ms"] / max(host["host_ntt9_pcie_floor_ms"], comp["ntt_batch9_ms"])
        # both directions carry 9 x 32 MiB.  Since round 5 plk_ntt_batch keeps one stream per direction (uploads back to back, kernels on
        # the main stream, downloads back to back): both directions of the link are busy from the second transform on
        # (profiles/r05_host_ntt9_pipeline_ab.txt: 9.7 -> 7.0 ms same lease; round 3's three-stream probe measured 8.9 ms from pinned memory)
        host["host_ntt9_duplex_floor_ms"] = host["host_ntt9_pcie_floor_ms"] * 10.0 / 9.0  # first upload + nine downloads, the rest overlapped
    host["note"] = "host-pointer C ABI calls on pageable numpy buffers, one caller thread; pcie_floor = bytes one way / 56 GB/s"
    comp["host_pointer"] = host

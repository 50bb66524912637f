#!/usr/bin/env python3
"""bench.py -- the headline benchmark of BASELINE.json on MI355X.

metric  "MSM Mpairs/sec + NTT Melems/sec, 2^20 Tweedledee, 1/2/4/8 GPU"
step    one pass of the hot path over one batch of synthetic input:
          * one forward 2^20-point NTT over TweedledeeBase  (BASELINE configs[1]; fft.rs:103), and
          * one 2^20-pair MSM on Tweedledee                  (BASELINE configs[2]; curve_msm.rs:102)
        both with inputs already resident in HBM, tables/precomputation excluded exactly as
        benches/fft.rs:22-30 and src/bin/msms.rs:25,54-58 exclude them.
value   whole-job units per second, 1 unit = 1 NTT element or 1 MSM scalar-point pair
        (2 * 2^20 units per step per GPU); the two components are reported separately in
        "components" as NTT Melems/s and MSM Mpairs/s - those are the numbers BASELINE.md tracks.
N > 1   one process per GPU (torch.distributed, backend nccl = RCCL).  NTTs are independent units
        (no collective).  The MSM is a global N * 2^20-pair MSM sharded by contiguous base range:
        every rank reduces its own 2^20 pairs, one all-gather of the N affine partial results
        (65 bytes each) and a local point sum give every rank the result.  Weak scaling.

Use --workload ntt|msm to time one component alone.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

LOG_N = 20
NTT_FIELD = 0       # TweedledeeBase
CURVE = 0           # Tweedledee (scalars in TweedledumBase)
SEED_NTT = 0xF70020
SEED_MSM = 0x350020
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)


def cpu_baseline(workload):
    """The oracle (C++ restatement of the reference algorithm) timed on this host's cores, on a
    bounded sample (about 10-30 s of CPU work in all): the NTT at the full 2^20 size and the MSM at 2^16 with the
    reference's w = 11 tables prebuilt, median of 3 runs for every thread count of a small sweep."""
    import numpy as np
    from oracle import bigint_ref as br, oracle_lib as ol
    from plonky_amd import synth
    cores = os.cpu_count() or 1
    sweep = sorted(set(t for t in (1, 8, 32, cores) if t <= cores))
    out = {"kind": "port", "label": "C++ restatement of the reference algorithm (oracle/plk_oracle.cpp), not plonky Rust",
           "host_cores": cores}
    used = []
    if workload in ("both", "ntt"):
        ln = 20
        x = synth.rand_field(NTT_FIELD, SEED_NTT, 1 << ln)
        pre = ol.FftPrecomputation(NTT_FIELD, 1 << ln)
        best = None
        for th in sweep:  # the layer loop forks per layer like Rayon; more threads is not always faster
            pre.fft_with_precomputation_power_of_2(x, threads=th)
            ts = []
            for _ in range(3):
                t0 = time.perf_counter()
                pre.fft_with_precomputation_power_of_2(x, threads=th)
                ts.append(time.perf_counter() - t0)
            t = sorted(ts)[1]
            if best is None or t < best[0]:
                best = (t, th)
        out["ntt_melems_per_s"] = (1 << ln) / best[0] / 1e6
        out["ntt_threads"] = best[1]
        used.append(best[1])
        out["ntt_sample"] = "2^%d TweedledeeBase forward NTT, best of threads %s (= %d), median of 3" % (ln, sweep, best[1])
    if workload in ("both", "msm"):
        lm = 16
        c = br.TWEEDLEDEE
        G = (c.gx, c.gy)
        D = br.ec_mul(c, 424242, G)
        g0 = np.array([c.base.mont_limbs(G[0]), c.base.mont_limbs(G[1])], dtype=np.uint64)
        dd = np.array([c.base.mont_limbs(D[0]), c.base.mont_limbs(D[1])], dtype=np.uint64)
        bases = ol.gen_bases(CURVE, 1 << lm, g0, dd)
        s = synth.rand_field(1, SEED_MSM, 1 << lm)
        pre = ol.MsmPrecomputation(CURVE, bases, 11, threads=min(cores, 64))  # table build excluded, as src/bin/msms.rs:25
        best = None
        for th in sweep:
            ts = []
            for _ in range(3):
                t0 = time.perf_counter()
                pre.execute(s, parallel=True, threads=th)
                ts.append(time.perf_counter() - t0)
            t = sorted(ts)[1]
            if best is None or t < best[0]:
                best = (t, th)
        out["msm_mpairs_per_s"] = (1 << lm) / best[0] / 1e6
        out["msm_threads"] = best[1]
        used.append(best[1])
        out["msm_sample"] = "2^%d Tweedledee msm_execute_parallel, w = 11 tables prebuilt, best of threads %s (= %d), median of 3" % (lm, sweep, best[1])
    out["cores"] = max(used) if used else 1
    n_units, t_units = 0.0, 0.0
    if "ntt_melems_per_s" in out:
        n_units += 1
        t_units += 1.0 / out["ntt_melems_per_s"]
    if "msm_mpairs_per_s" in out:
        n_units += 1
        t_units += 1.0 / out["msm_mpairs_per_s"]
    out["value"] = n_units / t_units  # same definition as the GPU value: units / time for equal unit counts
    out["unit"] = "M units/s (1 unit = 1 NTT element or 1 MSM pair)"
    out["sample"] = "; ".join(out[k] for k in ("ntt_sample", "msm_sample") if k in out)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=["both", "ntt", "msm"], default="both")
    ap.add_argument("--log-n", type=int, default=LOG_N)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--timed-only", action="store_true", help="run only the warm-up + timed region (for rocprofv3 --pmc passes)")
    args = ap.parse_args()

    import ctypes
    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, "--gpus %d but WORLD_SIZE=%d" % (args.gpus, world)
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU path)"
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from plonky_amd import device as dev, lib, parallel, synth
    from plonky_amd.selfcheck import closed_form_msm, _mul, _add
    from plonky_amd.synth import MODULI
    dev.init(local_rank)
    L = lib.load()

    n = 1 << args.log_n
    do_ntt = args.workload in ("both", "ntt")
    do_msm = args.workload in ("both", "msm")

    # ---- synthetic inputs, resident in HBM before the timed region ----
    if do_ntt:
        x_host = synth.rand_field(NTT_FIELD, SEED_NTT + rank, n)
        x = dev.to_device(x_host)
        y = torch.empty_like(x)
        lib.check(L.plk_ntt_precompute(NTT_FIELD, args.log_n))
    if do_msm:
        p = MODULI[0]
        G = (p - 1, 2)  # tweedledee_curve.rs:14-18
        d = synth.to_int(synth.rand_field(1, SEED_MSM, 1)[0]) % MODULI[1]
        D = _mul(p, d, G)
        g0 = np.stack([synth.mont(0, G[0]), synth.mont(0, G[1])])
        dd = np.stack([synth.mont(0, D[0]), synth.mont(0, D[1])])
        first = rank * n  # this rank's contiguous base range of the global N * n MSM
        bases = dev.gen_bases_dev(CURVE, n, g0, dd, first=first)
        s_host = synth.rand_field(1, SEED_MSM + 1 + rank, n)
        s = dev.to_device(s_host)
        pre = dev.msm_precompute_dev(CURVE, bases)
        oxy = torch.empty((1, 2, 4), dtype=torch.int64, device="cuda")
        oz = torch.empty((1,), dtype=torch.uint8, device="cuda")
        g_pair = [None, None]  # the gathered partial points of the base-range shards

    def step():
        if do_ntt:
            dev.ntt_dev(NTT_FIELD, x, out=y)
        if do_msm:
            dev.msm_execute_dev(pre, s, oxy, oz)
            if world > 1:
                # the one exchange step of the path: partial results of the base-range shards
                g_pair[0], g_pair[1] = parallel.all_gather_points(oxy, oz)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()

    # per-kernel durations for the roofline: HIP events on the launch stream around each kernel
    if do_ntt:
        L.plk_ntt_get_timings(None, None)
        L.plk_ntt_set_profiling(1)
    if do_msm:
        L.plk_msm_set_profiling(pre._ctx, 1)

    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    ntt_kernel_ms = msm_stage_ms = None
    if do_ntt:
        sm, cnt = ctypes.c_double(0), ctypes.c_uint(0)
        L.plk_ntt_get_timings(ctypes.byref(sm), ctypes.byref(cnt))
        L.plk_ntt_set_profiling(0)
        ntt_launches = cnt.value
        ntt_kernel_ms = sm.value
    if do_msm:
        arr = (ctypes.c_double * 7)()
        calls = ctypes.c_uint(0)
        L.plk_msm_get_timings(pre._ctx, arr, ctypes.byref(calls))
        L.plk_msm_set_profiling(pre._ctx, 0)
        msm_stage_ms = [v / max(1, calls.value) for v in arr]

    # ---- component timings (separate short loops, same K) so both headline numbers are reported ----
    comp = {}
    if args.timed_only:
        do_ntt_c = do_msm_c = False
    else:
        do_ntt_c, do_msm_c = do_ntt, do_msm
    if do_ntt_c:
        sync()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            dev.ntt_dev(NTT_FIELD, x, out=y)
        sync()
        tn = (time.perf_counter() - t1) / args.steps
        comp["ntt_ms"] = tn * 1e3
        comp["ntt_melems_per_s"] = world * n / tn / 1e6
        # the prover transforms its 9 wire polynomials together (plonk_util.rs:169-190): same kernels, one call
        xb = x.unsqueeze(0).repeat(9, 1, 1).contiguous()
        yb = torch.empty_like(xb)
        dev.ntt_dev(NTT_FIELD, xb, out=yb)
        sync()
        t1 = time.perf_counter()
        for _ in range(max(1, args.steps // 2)):
            dev.ntt_dev(NTT_FIELD, xb, out=yb)
        sync()
        tb = (time.perf_counter() - t1) / max(1, args.steps // 2)
        comp["ntt_batch9_ms"] = tb * 1e3
        comp["ntt_batch9_melems_per_s"] = world * 9 * n / tb / 1e6
        del xb, yb
        # the quotient path either side of the transforms (polynomial.rs:330-380, plonk_util.rs:179-190)
        # at the sizes this n implies: divide_by_z_h of a degree < n polynomial by Z_H of n/8, LDE of 9 wires n/8 -> n
        if args.log_n >= 13:
            nq = n // 8
            from plonky_amd import api as _api
            # m = q0 * (X^nq - 1) for a random q0 of 7 nq coefficients: m[i] = q0[i - nq] - q0[i]
            q0 = synth.rand_field(NTT_FIELD, SEED_NTT + 100 + rank, 7 * nq)
            zpad = np.zeros((nq, 4), dtype=np.uint64)
            m = dev.to_device(_api.field_op(NTT_FIELD, "sub", np.concatenate([zpad, q0]), np.concatenate([q0, zpad])))
            q_out = torch.empty((n, 4), dtype=torch.int64, device="cuda")
            dev.divide_by_z_h_dev(NTT_FIELD, m, nq, out=q_out)
            sync()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                dev.divide_by_z_h_dev(NTT_FIELD, m, nq, out=q_out)
            sync()
            comp["divide_by_z_h_ms"] = (time.perf_counter() - t1) / args.steps * 1e3
            w = dev.to_device(synth.rand_field(NTT_FIELD, SEED_NTT + 200 + rank, 9 * nq)).reshape(9, nq, 4)
            ev = torch.empty((9, n, 4), dtype=torch.int64, device="cuda")
            dev.ntt_padded_dev(NTT_FIELD, w, args.log_n, out=ev)
            sync()
            t1 = time.perf_counter()
            for _ in range(max(1, args.steps // 2)):
                dev.ntt_padded_dev(NTT_FIELD, w, args.log_n, out=ev)
            sync()
            comp["lde9_ms"] = (time.perf_counter() - t1) / max(1, args.steps // 2) * 1e3
            comp["quotient_path_note"] = "divide_by_z_h: degree < 2^%d by Z_H of 2^%d (2 fused transforms); lde9: 9 x 2^%d coefficients -> 2^%d evaluations" % (
                args.log_n, args.log_n - 3, args.log_n - 3, args.log_n)
            if not args.no_check:
                q_host = dev.to_host(q_out)
                comp["_q_check"] = bool(np.array_equal(q_host[: 7 * nq], q0) and not q_host[7 * nq:].any())
            del m, q_out, ev
    if do_msm_c:
        sync()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            dev.msm_execute_dev(pre, s, oxy, oz)
        sync()
        tm = (time.perf_counter() - t1) / args.steps
        comp["msm_ms"] = tm * 1e3
        comp["msm_mpairs_per_s"] = world * n / tm / 1e6
    if do_msm_c:
        # commit_polynomials (plonk_util.rs:215-231): the 9 wire polynomials against the same generators, one call
        sb = s.unsqueeze(0).repeat(9, 1, 1).contiguous()
        oxy9 = torch.empty((9, 2, 4), dtype=torch.int64, device="cuda")
        oz9 = torch.empty((9,), dtype=torch.uint8, device="cuda")
        dev.msm_execute_dev(pre, sb, oxy9, oz9)
        sync()
        t1 = time.perf_counter()
        for _ in range(max(1, args.steps // 4)):
            dev.msm_execute_dev(pre, sb, oxy9, oz9)
        sync()
        tb = (time.perf_counter() - t1) / max(1, args.steps // 4)
        comp["msm_batch9_ms"] = tb * 1e3
        comp["msm_batch9_mpairs_per_s"] = world * 9 * n / tb / 1e6
        if not args.no_check:
            comp["_b9_check"] = bool(torch.equal(oxy9, oxy.expand(9, 2, 4)) and int(oz9.sum().item()) == 0)
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
    if do_msm:
        comp["msm_window_bits"] = pre.window
        comp["msm_stage_ms"] = dict(zip(["digits", "partition", "scan_scatter", "accumulate", "bucket_sum", "planes", "final"],
                                        [round(v, 4) for v in msm_stage_ms]))

    # ---- correctness of what was just timed (not in the timed region) ----
    checks = {}
    if not args.no_check:
        if do_ntt:
            back = dev.to_host(dev.ntt_dev(NTT_FIELD, y, inverse=True))
            checks["ntt_roundtrip_bit_exact"] = bool(np.array_equal(back, x_host))
            if "_q_check" in comp:
                checks["divide_by_z_h_identity"] = comp.pop("_q_check")
        if do_msm:
            got = dev.to_host(oxy).reshape(2, 4)
            exp = closed_form_msm(CURVE, s_host, G, D, first=first)
            gotp = (synth.from_mont(0, got[0]), synth.from_mont(0, got[1]))
            checks["msm_closed_form_bit_exact"] = bool(int(oz.cpu()[0]) == 0 and gotp == exp)
            if "_b9_check" in comp:
                checks["msm_batch9_equals_single"] = comp.pop("_b9_check")
            if "_os_check" in comp:
                checks["msm_one_shot_equals_tabled"] = comp.pop("_os_check")
            if world > 1:
                tot_xy = torch.empty((1, 2, 4), dtype=torch.int64, device="cuda")
                tot_z = torch.empty((1,), dtype=torch.uint8, device="cuda")
                hx, hz = dev.to_host(g_pair[0]).reshape(world, 2, 4), g_pair[1].cpu().numpy().reshape(world)
                from plonky_amd import api
                tot, tz = api.curve_sum_affine(CURVE, hx, hz)
                checks["msm_global_sum_is_point"] = bool(tz == 0)

    units_per_step = (n if do_ntt else 0) + (n if do_msm else 0)
    value = world * units_per_step * args.steps / elapsed / 1e6

    # ---- roofline of the dominant kernel ----
    roofline = None
    rooflines = {}
    if do_ntt and ntt_launches:
        per_launch_ms = ntt_kernel_ms / ntt_launches
        launches_per_ntt = ntt_launches / args.steps
        # algorithmic bytes per transform: 64 B / element (read 32 B + write 32 B once);
        # one launch of the pass kernel handles all n elements once => 64 B * n / launches_per_ntt per launch
        alg_bytes = 64.0 * n / launches_per_ntt
        ach = alg_bytes / (per_launch_ms * 1e-3) / 1e9
        rooflines["ntt_pass"] = {"kernel": "k_ntt_pass", "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": ach / HBM_PEAK_GBS, "traffic": None, "launch_ms": per_launch_ms,
                                 "launches_per_transform": launches_per_ntt,
                                 "modmul_per_s": (n / 2.0 * args.log_n) / (per_launch_ms * launches_per_ntt * 1e-3)}
    if do_msm:
        acc_ms = msm_stage_ms[3]
        alg_bytes = 96.0 * n  # 64 B affine base + 32 B scalar per pair
        ach = alg_bytes / (acc_ms * 1e-3) / 1e9
        windows = (255 + 1 + pre.window - 1) // pre.window
        rooflines["msm_accumulate"] = {"kernel": "k_msm_accumulate", "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                       "frac": ach / HBM_PEAK_GBS, "traffic": None, "launch_ms": acc_ms,
                                       "mixed_adds_per_s": n * windows / (acc_ms * 1e-3),
                                       "note": "int-ALU bound (~10 modmul per mixed add), HBM fraction is expected to be << 1"}
    # HBM traffic per launch from the committed rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE need two
    # separate passes and cannot be sampled from inside this process): profiles/r01_pmc_traffic.json
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as fh:
            pmc = json.load(fh)
        for key, kname in (("ntt_pass", "k_ntt_pass"), ("msm_accumulate", "k_msm_accumulate")):
            if key in rooflines and kname in pmc and pmc[kname].get("log_n") == args.log_n:
                rooflines[key]["traffic"] = pmc[kname]["bytes_per_launch"]
                rooflines[key]["traffic_source"] = pmc[kname]["source"]
    except (OSError, ValueError):
        pass
    if rooflines:
        dom = max(rooflines.values(), key=lambda r: r["launch_ms"] * (r.get("launches_per_transform", 1)))
        roofline = dom

    result = {
        "metric": "MSM Mpairs/sec + NTT Melems/sec, 2^20 Tweedledee, 1/2/4/8 GPU",
        "value": value,
        "unit": "M units/s (1 unit = 1 NTT element or 1 MSM scalar-point pair; components below)",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u32", "data": "synthetic",
        "config": {"workload": {"both": "2^%d TweedledeeBase forward NTT + 2^%d-pair Tweedledee MSM per GPU per step" % (args.log_n, args.log_n),
                                "ntt": "2^%d TweedledeeBase forward NTT per GPU per step" % args.log_n,
                                "msm": "2^%d-pair Tweedledee MSM per GPU per step" % args.log_n}[args.workload],
                   "log_n": args.log_n, "sharding": "independent NTTs; MSM sharded by base range + all-gather of partial points" if world > 1 else "single GPU",
                   "seeds": {"ntt": SEED_NTT, "msm": SEED_MSM}},
        "components": comp,
        "checks": checks,
        "roofline": roofline,
        "rooflines": rooflines,
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(args.workload)
    if rank == 0:
        print(json.dumps(result))
    assert all(checks.values()), "self-check failed: %r" % checks
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

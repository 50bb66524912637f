"""benchlib.cpu -- the CPU baseline: the oracle (C++ restatement of the reference algorithm) timed on the GPU box's host cores.
The ONLY place of the benchmark that touches oracle/ (test infrastructure: never the thing measured as the product)."""
import os
import sys
import time

from .common import METRIC, SEED_MSM, SEED_NTT


def cpu_baseline(workload, cv):
    """The oracle (C++ restatement of the reference algorithm, persistent worker pool) timed on this host's cores on a bounded
    sample: the full 2^20 NTT (T = 1 and the best of a thread sweep, 10 timed runs each after a warm-up run) and the MSM with the
    reference's w = 11 tables prebuilt: 2^20 pairs at the best thread count (10 runs) and 2^18 pairs at T = 1 (10 runs)."""
    import numpy as np
    from oracle import bigint_ref as br, oracle_lib as ol
    from plonky_amd import synth
    cores = os.cpu_count() or 1
    sweep = sorted(set(t for t in (8, 32, 64, cores) if t <= cores))
    out = {"kind": "port", "label": "C++ restatement of the reference algorithm (oracle/plk_oracle.cpp), not plonky Rust",
           "host_cores": cores}

    def timed(fn, runs):
        fn()
        ts = []
        for _ in range(runs):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return sorted(ts)[len(ts) // 2]

    used = []
    if workload in ("both", "ntt"):
        ln = 20
        x = synth.rand_field(cv["ntt_field"], SEED_NTT, 1 << ln)
        pre = ol.FftPrecomputation(cv["ntt_field"], 1 << ln)
        best = None
        for th in sweep:  # a layer is a fork-join over 2000-pair chunks like the reference's; more threads is not always faster
            t = timed(lambda: pre.fft_with_precomputation_power_of_2(x, threads=th), 3)
            if best is None or t < best[0]:
                best = (t, th)
        t_all = timed(lambda: pre.fft_with_precomputation_power_of_2(x, threads=best[1]), 10)
        t_one = timed(lambda: pre.fft_with_precomputation_power_of_2(x, threads=1), 10)
        out["ntt_melems_per_s"] = (1 << ln) / t_all / 1e6
        out["ntt_threads"] = best[1]
        out["ntt_melems_per_s_1_thread"] = (1 << ln) / t_one / 1e6
        used.append(best[1])
        out["ntt_sample"] = "2^%d forward NTT, median of 10: T = %d (best of %s) and T = 1" % (ln, best[1], sweep)
    if workload in ("both", "msm", "commit9"):
        c = br.CURVES[cv["curve"]]
        G = (c.gx, c.gy)
        D = br.ec_mul(c, 424242, G)
        g0 = np.array([c.base.mont_limbs(G[0]), c.base.mont_limbs(G[1])], dtype=np.uint64)
        dd = np.array([c.base.mont_limbs(D[0]), c.base.mont_limbs(D[1])], dtype=np.uint64)
        lm = 20 if cores >= 32 else 16  # the 2^20 table build and 10 executions need a real host (minutes on 8 cores)
        th_all = min(cores, 256)
        bases = ol.gen_bases(cv["curve"], 1 << lm, g0, dd)
        s = synth.rand_field(cv["scalar_field"], SEED_MSM, 1 << lm)
        pre = ol.MsmPrecomputation(cv["curve"], bases, 11, threads=th_all)  # table build excluded, as src/bin/msms.rs:25
        best = None
        for th in sorted(set(t for t in (32, 64, th_all) if t <= th_all)):
            t = timed(lambda: pre.execute(s, parallel=True, threads=th), 1)
            if best is None or t < best[0]:
                best = (t, th)
        t_all = timed(lambda: pre.execute(s, parallel=True, threads=best[1]), 10)
        out["msm_mpairs_per_s"] = (1 << lm) / t_all / 1e6
        out["msm_threads"] = best[1]
        # one thread: ~6 s per 2^20-pair execution on this class of host, so ten timed runs at full size would be a minute of the
        # "10 - 30 s of CPU work" this baseline is bounded to: 2^18 pairs (the rate per pair is flat in n at fixed w), ten runs
        l1 = min(lm, 18)
        pre1 = pre if lm == l1 else ol.MsmPrecomputation(cv["curve"], bases[: 1 << l1], 11, threads=th_all)
        t_one = timed(lambda: pre1.execute(s[: 1 << l1], parallel=True, threads=1), 10)
        out["msm_mpairs_per_s_1_thread"] = (1 << l1) / t_one / 1e6
        used.append(best[1])
        out["msm_sample"] = ("2^%d-pair msm_execute_parallel, w = 11 tables prebuilt, median of 10 at T = %d; T = 1: 2^%d pairs, median of 10 "
                             "(2^20 at T = 1 is ~6 s per run: outside the bounded sample)" % (lm, best[1], l1))
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


# ---- bench.py --workload crossover -----------------------------------------------------------------------------------------
# Where the drop-in (HOST-pointer) entry points cross the CPU path: the reference's bench shapes (benches/fft.rs:10-53 sweeps
# 2^1..2^18 forward + inverse; src/bin/msms.rs:12-44 is 2^14 pairs at w = 11 / 12) as a size sweep 2^8..2^log_n of plk_ntt from /
# to pageable host memory (what fft_with_precomputation_power_of_2 behind the shim costs) and plk_msm_execute over prebuilt tables
# from host scalars (msm_execute_parallel behind the shim), against the oracle - the C++ restatement of the reference algorithm:
# layered NTT with 2000-pair chunks, Yao MSM over w = 11 tables - on the host's cores at its best thread count and at T = 1.
# The table (stderr; one JSON line on stdout) justifies the library's size gate PLK_MIN_GPU_LOG_N (plk_min_gpu_log_n()).
def best_of(fn, reps):
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2]


def crossover_sweep(args):
    """bench.py --workload crossover: see the note above best_of()."""
    import ctypes
    import json
    import numpy as np
    min_log_n, max_log_n = 8, args.log_n
    import plonky_amd as pa
    from plonky_amd import device as dev, lib, synth
    from plonky_amd.selfcheck import GENERATORS, _mul
    from plonky_amd.synth import MODULI
    from oracle import oracle_lib as ol
    L = lib.load()
    lib.check(L.plk_init(0))
    vp = ctypes.c_void_p
    cores = os.cpu_count() or 1
    sweep = sorted(set(t for t in (1, 8, 32, 64, cores) if t <= cores))
    p = MODULI[0]
    G = GENERATORS[0]
    D = _mul(p, 424242, G)
    g0 = np.stack([synth.mont(0, G[0]), synth.mont(0, G[1])])
    dd = np.stack([synth.mont(0, D[0]), synth.mont(0, D[1])])
    lines = ["# tools/crossover_sweep.py: host-pointer C ABI (pageable numpy buffers, PCIe inside) against the oracle's CPU path, %d host threads available" % cores,
             "# times in microseconds, median of the repetitions; cpu_best = best of T in %s" % sweep,
             "%-6s %12s %12s %12s %10s | %12s %12s %10s | %12s %12s %12s %10s" % ("log_n", "ntt_gpu", "ntt_cpu_T1", "ntt_cpu_best", "(threads)", "intt_gpu", "intt_cpu_best", "(threads)",
                                                                             "msm_gpu", "msm_cpu_T1", "msm_cpu_best", "(threads)")]
    for log_n in range(min_log_n, max_log_n + 1):
        n = 1 << log_n
        reps = 30 if log_n <= 14 else 10
        x = np.ascontiguousarray(synth.rand_field(0, 0xF70000 + log_n, n))
        y = np.zeros_like(x)
        lib.check(L.plk_ntt_precompute(0, log_n))
        t_gpu = best_of(lambda: lib.check(L.plk_ntt(0, log_n, 0, vp(x.ctypes.data), vp(y.ctypes.data))), reps)
        pre = ol.FftPrecomputation(0, n)
        exp = pre.fft_with_precomputation_power_of_2(x, threads=1)
        assert np.array_equal(exp, y), "NTT mismatch at 2^%d" % log_n
        cpu = {t: best_of(lambda t=t: pre.fft_with_precomputation_power_of_2(x, threads=t), 5 if log_n >= 18 else reps) for t in sweep}
        tb = min(cpu, key=cpu.get)
        # the inverse transform (benches/fft.rs:32-53 times it beside the forward one): of the forward result, back to x
        z = np.zeros_like(x)
        t_igpu = best_of(lambda: lib.check(L.plk_ntt(0, log_n, 1, vp(y.ctypes.data), vp(z.ctypes.data))), reps)
        assert np.array_equal(z, x) and np.array_equal(pre.ifft_with_precomputation_power_of_2(y, threads=1), x), "inverse NTT mismatch at 2^%d" % log_n
        icpu = {t: best_of(lambda t=t: pre.ifft_with_precomputation_power_of_2(y, threads=t), 5 if log_n >= 18 else reps) for t in sweep}
        itb = min(icpu, key=icpu.get)
        # MSM: tables prebuilt on both sides (src/bin/msms.rs:25 excludes msm_precompute from its timing)
        bases = ol.gen_bases(0, n, g0, dd)
        s = np.ascontiguousarray(synth.rand_field(1, 0x350000 + log_n, n))
        ctx = pa.msm_precompute(0, bases, 11)
        oxy, oz = np.zeros((2, 4), dtype=np.uint64), np.zeros(1, dtype=np.uint8)
        m_gpu = best_of(lambda: lib.check(L.plk_msm_execute(ctx._ctx, vp(s.ctypes.data), n, vp(oxy.ctypes.data), vp(oz.ctypes.data))), reps)
        opre = ol.MsmPrecomputation(0, bases, 11, threads=min(cores, 256))
        exy, ez = opre.execute(s, parallel=True, threads=min(cores, 32))
        assert ez == int(oz[0]) and np.array_equal(exy, oxy), "MSM mismatch at 2^%d" % log_n
        mrep = 3 if log_n >= 18 else (5 if log_n >= 15 else 10)
        mcpu = {t: best_of(lambda t=t: opre.execute(s, parallel=True, threads=t), mrep) for t in sweep if not (t == 1 and log_n > 18)}
        mb = min(mcpu, key=mcpu.get)
        ctx.free()
        lines.append("%-6d %12.1f %12.1f %12.1f %10d | %12.1f %12.1f %10d | %12.1f %12s %12.1f %10d" % (
            log_n, t_gpu * 1e6, cpu[1] * 1e6, cpu[tb] * 1e6, tb, t_igpu * 1e6, icpu[itb] * 1e6, itb,
            m_gpu * 1e6, ("%.1f" % (mcpu[1] * 1e6)) if 1 in mcpu else "-", mcpu[mb] * 1e6, mb))
        print(lines[-1], flush=True)
    ntt_cross = [l for l in lines[3:] if float(l.split()[1]) < float(l.split()[3])]
    intt_cross = [l for l in lines[3:] if float(l.split()[6]) < float(l.split()[7])]
    lines.append("# smallest size at which the GPU path beats the CPU's best: NTT 2^%s, inverse NTT 2^%s; plk_min_gpu_log_n() = %d" % (
        ntt_cross[0].split()[0] if ntt_cross else "-", intt_cross[0].split()[0] if intt_cross else "-", int(L.plk_min_gpu_log_n())))
    sys.stderr.write("\n".join(lines) + "\n")
    rows = [l.split() for l in lines[3:-1]]
    print(json.dumps({"metric": METRIC, "workload": "crossover: host-pointer plk_ntt / plk_msm_execute against the oracle's CPU path, 2^%d..2^%d" % (min_log_n, max_log_n),
                      "unit": "microseconds per call", "host_cores": cores, "min_gpu_log_n": int(L.plk_min_gpu_log_n()),
                      "rows": [{"log_n": int(r[0]), "ntt_gpu_us": float(r[1]), "ntt_cpu_1_thread_us": float(r[2]), "ntt_cpu_best_us": float(r[3]), "ntt_cpu_best_threads": int(r[4]),
                                "intt_gpu_us": float(r[6]), "intt_cpu_best_us": float(r[7]), "intt_cpu_best_threads": int(r[8]),
                                "msm_gpu_us": float(r[10]), "msm_cpu_1_thread_us": None if r[11] == "-" else float(r[11]), "msm_cpu_best_us": float(r[12]),
                                "msm_cpu_best_threads": int(r[13])} for r in rows]}), flush=True)



#!/usr/bin/env python3
"""bench.py -- the headline benchmark of BASELINE.json on MI355X.

metric  "MSM Mpairs/sec + NTT Melems/sec, 2^20 Tweedledee, 1/2/4/8 GPU"
step    one pass of the hot path over one batch of synthetic input (--workload):
          both    (default) one forward 2^20-point NTT over TweedledeeBase (BASELINE configs[1]; fft.rs:103) and one 2^20-pair
                  MSM on Tweedledee (configs[2]; curve_msm.rs:102) per GPU.  N > 1: weak scaling - the NTTs are independent
                  units (no collective); the MSM is a global N * 2^20-pair MSM sharded by contiguous base range: every rank
                  reduces its own 2^20 pairs, ONE packed all-gather of the N affine partial results and a local point sum.
          ntt / msm   one component alone.
          quotient  the two heaviest kernels of the callers either side of the path, with their own
                  roofline entries: the 8n-point loop of Prover::vanishing_poly (plonk.rs:392-453) for a circuit of 2^log_n gates
                  (2^(log_n + 3) points) and the generator fold of the first IPA round (halo.rs:119-123) over 2^(log_n - 1) pairs.
          commit9 BASELINE configs[3]: the 9-wire commitment batch of poly_commit.rs:52-66 - nine 2^20 scalar vectors against
                  the same 2^20 generators.  N > 1: STRONG scaling - the generators are sharded by base range (each rank holds
                  2^20 / N of them and the matching slice of every vector), one packed all-gather of 9 partial points.
        --curve bls12_377 --log-n 22 --workload msm  is BASELINE configs[4] (the reference has BLS12-377, not -381);
                  with --shard it is the STRONG-scaling form: one fixed 2^log_n MSM, generators sharded by base range.
        --emulate-rank r/N  runs rank r's shard of a strong-scaling problem (commit9, msm --shard) alone on one GPU: the
                  per-rank time from which DESIGN.md section 6 predicts the N-GPU curve without an N-GPU node.
launch  python bench.py --gpus N            spawns its N ranks itself (one process per GPU, torch.multiprocessing), or
        python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 bench.py --gpus N   (the driver's form:
        RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment).  Backend nccl (= RCCL over xGMI); --same-device puts all
        ranks on GPU 0 over gloo (the world-size-2 test on a one-GPU box).
        Inputs are resident in HBM before the timed region; tables / precomputation are excluded exactly as
        benches/fft.rs:22-30 and src/bin/msms.rs:25,54-58 exclude them.
value   whole-job units per second, 1 unit = 1 NTT element or 1 MSM scalar-point pair; the components are reported
        separately in "components" as NTT Melems/s and MSM Mpairs/s - those are the numbers BASELINE.md tracks.
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

LOG_N = 20
SEED_NTT = 0xF70020
SEED_MSM = 0x350020
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8 TB/s spec
# integer-ALU ceilings are read from the newest profiles/rNN_field_op_costs.json (tools/bench_field.hip on this round's arithmetic headers,
# tagged with the hash of those headers): fz_mul at 4 waves / SIMD per field and the raw v_mad_u64_u32 issue rate
def latest_profile(suffix):
    """profiles/rNN_<suffix> of the highest round present (the files are named per round)."""
    import glob
    found = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r[0-9][0-9]_" + suffix)))
    return found[-1] if found else os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r00_" + suffix)


CEILINGS_FILE = latest_profile("field_op_costs.json")
MADS_PER_MODMUL = {4: 126, 6: 294}   # v_mad_u64_u32 per fz_mul: 9 limbs 81 + 45, 14 limbs 196 + 98 (fz.cuh)
CURVES = {"tweedledee": dict(curve=0, ntt_field=0, scalar_field=1, base_field=0, limbs=4, scalar_bits=255, pair_bytes=96),
          "bls12_377": dict(curve=2, ntt_field=2, scalar_field=2, base_field=3, limbs=6, scalar_bits=253, pair_bytes=128)}
STAGES = ["order_count", "order_scatter", "order_buckets", "accumulate", "assemble_lines", "planes", "final"]


def kernel_source_hash():
    """Identifies the kernels a PMC traffic figure was measured on: sha256 over the DEVICE sources (capi.hip, multi.hip and their
    two headers hold no kernel: staging, the C ABI and the fan-out over devices do not change what a kernel moves)."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "plonky_amd", "csrc")
    host_only = ("capi.hip", "multi.hip", "common.h", "host_lane.h")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".cuh", ".h")) and name not in host_only:
            with open(os.path.join(d, name), "rb") as fh:
                h.update(name.encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def arith_source_hash():
    """sha256 over the arithmetic headers a measured ceiling belongs to (fp / fp29 / fz / ec / ecz + parameters)."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "plonky_amd", "csrc")
    for name in ("fp.cuh", "fp29.cuh", "fz.cuh", "ec.cuh", "ecz.cuh", "field_params.cuh"):
        with open(os.path.join(d, name), "rb") as fh:
            h.update(name.encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def load_ceilings():
    """{"fz_mul_gops": {"tweedledee": .., "bls12_377": ..}, "mad_u64_u32_glaneops": .., "arith_source_sha": ..} or {}."""
    try:
        with open(CEILINGS_FILE) as fh:
            c = json.load(fh)
    except (OSError, ValueError):
        return {}
    c["stale"] = c.get("arith_source_sha") != arith_source_hash()
    return c


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


def gpu_identity(torch, index):
    """Which physical GPU a number comes from and what its clocks were: a 5-10 % kernel difference between two leases cannot be
    told from box-to-box spread without it (round-3 review).  uuid from the HIP runtime; clocks / serial from rocm-smi when present."""
    import subprocess
    info = {}
    try:
        pr = torch.cuda.get_device_properties(index)
        info.update(name=pr.name, uuid=str(getattr(pr, "uuid", "")), cus=pr.multi_processor_count, hbm_gib=round(pr.total_memory / 2 ** 30, 1))
    except Exception as e:  # noqa: BLE001
        info["error"] = str(e)
    try:
        out = subprocess.run(["rocm-smi", "-d", str(index), "--showuniqueid", "--showserial", "--showclocks", "--showperflevel", "--showpower"],
                             stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=30).stdout
        for line in out.splitlines():
            if not line.startswith("GPU["):
                continue
            low = line.lower()
            for key, tag in (("unique id", "unique_id"), ("serial number", "serial"), ("sclk clock level", "sclk"), ("mclk clock level", "mclk"),
                             ("performance level", "perf_level"), ("average graphics package power", "power_w"), ("current socket graphics package power", "power_w")):
                if key in low and tag not in info:
                    info[tag] = line.split(":")[-1].strip()
    except Exception:  # noqa: BLE001
        pass
    return info


def strong_case(curve_name, log_n, batch, world, rank, steps, warmup, gloo, solo):
    """One STRONG-scaling problem - `batch` scalar vectors of 2^log_n against the same 2^log_n generators, split over `world` ranks
    by parallel.BatchPlan (whole vectors + a base-range-sharded remainder; batch 1: the sharded case alone) - set up, timed for
    `steps` steps after `warmup` and checked against the closed form of the WHOLE problem.  solo: this process runs the whole
    problem alone as the world = 1 form (rank 0 measuring T_1 inside a multi-rank run; the other ranks wait at the caller's barrier).
    Returns {"ms": per step (this rank), "ok": bool}.  BASELINE configs 4 (commit9) and 5 (one 2^22 BLS12-377 G1 MSM)."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from plonky_amd import device as dev, parallel, synth
    from plonky_amd.selfcheck import GENERATORS, closed_form_msm, _mul
    from plonky_amd.synth import MODULI
    cv = CURVES[curve_name]
    CURVE = cv["curve"]
    n = 1 << log_n
    w_, r_ = (1, 0) if solo else (world, rank)
    p = MODULI[cv["base_field"]]
    G = GENERATORS[CURVE]
    D = _mul(p, synth.to_int(synth.rand_field(cv["scalar_field"], SEED_MSM, 1)[0]) % MODULI[cv["scalar_field"]], G)
    g0 = np.stack([synth.mont(cv["base_field"], G[0]), synth.mont(cv["base_field"], G[1])])
    dd = np.stack([synth.mont(cv["base_field"], D[0]), synth.mont(cv["base_field"], D[1])])
    plan = parallel.BatchPlan(batch, w_, r_, n)
    s_host = np.stack([synth.rand_field(cv["scalar_field"], SEED_MSM + 0x900 + k, n) for k in range(batch)])
    s = dev.to_device(plan.local_scalars(s_host))
    bases = dev.gen_bases_dev(CURVE, plan.n_local, g0, dd, first=plan.first)
    pre = dev.msm_precompute_dev(CURVE, bases)
    ex = parallel.PartialExchange(CURVE, batch, "cuda", whole_per_rank=plan.whole, world=w_, rank=r_, solo=solo)
    parts = plan.parts(s) if (plan.full_context and plan.sharded) else None

    def step():
        if parts is not None:
            dev.msm_execute_parts_dev(pre, parts, ex.out_xy, ex.out_zero)
        else:
            dev.msm_execute_dev(pre, s, ex.out_xy, ex.out_zero)
        ex.gather()
        return ex.combine()

    def sync():
        torch.cuda.synchronize()
        if not solo and world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        gxy, gz = step()
    sync()
    ms = (time.perf_counter() - t0) / steps * 1e3
    if not solo and world > 1:
        t = torch.tensor([ms], dtype=torch.float64, device="cpu" if gloo else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ms = float(t.item())
    got, gzh = dev.to_host(gxy), gz.cpu().numpy()
    ok = not gzh.any()
    for v in range(batch):
        ok = ok and (synth.from_mont(cv["base_field"], got[v][0]), synth.from_mont(cv["base_field"], got[v][1])) == closed_form_msm(CURVE, s_host[v], G, D, first=0)
    pre.free()
    del bases, s, ex
    torch.cuda.empty_cache()
    return {"ms": ms, "ok": bool(ok)}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", choices=["both", "ntt", "msm", "commit9", "quotient"], default="both")
    ap.add_argument("--curve", choices=sorted(CURVES), default="tweedledee")
    ap.add_argument("--log-n", type=int, default=LOG_N)
    ap.add_argument("--shard", action="store_true", help="--workload msm: strong scaling - ONE 2^log_n MSM, generators sharded by base range")
    ap.add_argument("--emulate-rank", default=None, metavar="r/N", help="run rank r's shard of the N-rank strong-scaling problem alone on one GPU")
    ap.add_argument("--same-device", action="store_true", help="all ranks on GPU 0, gloo backend (world-size-2 test on a one-GPU box)")
    ap.add_argument("--no-parts", action="store_true", help="strong scaling: a rank's share of a sharded vector as a zero-padded full-length vector (round-3 start) instead of its base range")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--timed-only", action="store_true", help="run only the warm-up + timed region (for rocprofv3 --pmc passes)")
    ap.add_argument("--single-process", action="store_true",
                    help="N GPUs from ONE process through the host-pointer C ABI (plk_init_devices: what an untouched plonk.rs gets): commit9, one sharded MSM and "
                         "a transform batch over --gpus N devices against the same calls on one device")
    ap.add_argument("--virtual-devices", action="store_true", help="--single-process on a box with fewer GPUs: PLK_VIRTUAL_DEVICES logical devices on GPU 0")
    return ap.parse_args(argv)


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _spawned_rank(rank, argv, world, port):
    os.environ.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    run(parse_args(argv))


def spawn_ranks(args, argv):
    """`python bench.py --gpus N` without a launcher: one process per GPU, rank 0 prints the JSON line."""
    import torch
    import torch.multiprocessing as mp
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    need = 1 if args.same_device else args.gpus
    if have < need:
        sys.stderr.write("bench.py --gpus %d: needs %d GPU(s), %d visible (there is no CPU path)\n" % (args.gpus, need, have))
        sys.exit(3)
    mp.spawn(_spawned_rank, args=(argv, args.gpus, _free_port()), nprocs=args.gpus, join=True)


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if args.single_process:
        return run_single_process(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args, argv)
    return run(args)


def single_process_case(L, lib, n_devices, log_n, reps, virtual=False):
    """The in-library multi-GPU path (plonky_amd/csrc/multi.hip) through the HOST-POINTER entry points - exactly what an untouched
    plonk.rs / poly_commit.rs reaches through the shim of INTEGRATION.md: plk_msm_precompute once, then per step nine commitments
    in one plk_msm_execute_batch (BASELINE config 4), one plk_msm_execute (a single MSM, sharded by base range) and nine transforms
    in one plk_ntt_batch; PCIe is inside every number.  Runs the same calls on ONE device first (plk_init) and then on the group
    (plk_init_devices(n_devices)); results must agree bit for bit.  Returns the timings of both and their ratios."""
    import ctypes
    import numpy as np
    from plonky_amd import api, synth
    from plonky_amd.selfcheck import GENERATORS, _mul
    from plonky_amd.synth import MODULI
    vp = ctypes.c_void_p
    n = 1 << log_n
    p = MODULI[0]
    G = GENERATORS[0]
    D = _mul(p, 0x51761E, G)
    # generators as HOST data (the reference's pedersen_g): G + i D built by the device once, read back
    g0 = np.stack([synth.mont(0, G[0]), synth.mont(0, G[1])])
    dd = np.stack([synth.mont(0, D[0]), synth.mont(0, D[1])])
    lib.check(L.plk_init(0))
    import torch
    from plonky_amd import device as dev
    bases = dev.to_host(dev.gen_bases_dev(0, n, g0, dd)).reshape(n, 2, 4).copy()
    vecs = [np.ascontiguousarray(synth.rand_field(1, SEED_MSM + 0x900 + k, n)) for k in range(9)]
    polys = [np.ascontiguousarray(synth.rand_field(0, SEED_NTT + k, n)) for k in range(9)]
    outs = [np.zeros_like(polys[0]) for _ in range(9)]
    sptr = (vp * 9)(*[a.ctypes.data for a in vecs])
    iptr = (vp * 9)(*[a.ctypes.data for a in polys])
    optr = (vp * 9)(*[a.ctypes.data for a in outs])

    def measure():
        ctx = vp()
        t0 = time.perf_counter()
        lib.check(L.plk_msm_precompute(0, n, vp(bases.ctypes.data), None, 0, ctypes.byref(ctx)))
        t_pre = (time.perf_counter() - t0) * 1e3
        xy9, z9 = np.zeros((9, 2, 4), dtype=np.uint64), np.zeros(9, dtype=np.uint8)
        xy1, z1 = np.zeros((2, 4), dtype=np.uint64), np.zeros(1, dtype=np.uint8)
        res = {"precompute_ms": t_pre}
        for name, fn in (("commit9_ms", lambda: lib.check(L.plk_msm_execute_batch(ctx, 9, sptr, n, vp(xy9.ctypes.data), vp(z9.ctypes.data)))),
                         ("msm_single_ms", lambda: lib.check(L.plk_msm_execute(ctx, vp(vecs[4].ctypes.data), n, vp(xy1.ctypes.data), vp(z1.ctypes.data)))),
                         ("ntt9_ms", lambda: lib.check(L.plk_ntt_batch(0, log_n, 0, 9, iptr, optr)))):
            fn()
            fn()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            res[name] = (time.perf_counter() - t0) / reps * 1e3
        lib.check(L.plk_msm_free(ctx))
        return res, (xy9.copy(), z9.copy(), xy1.copy(), z1.copy(), [o.copy() for o in outs])

    one, r_one = measure()
    L.plk_shutdown()
    if virtual:
        os.environ["PLK_VIRTUAL_DEVICES"] = str(n_devices)
    lib.check(L.plk_init_devices(n_devices))
    assert int(L.plk_device_count()) == n_devices
    grp, r_grp = measure()
    L.plk_shutdown()
    same = (np.array_equal(r_one[0], r_grp[0]) and np.array_equal(r_one[1], r_grp[1]) and np.array_equal(r_one[2], r_grp[2]) and np.array_equal(r_one[3], r_grp[3])
            and all(np.array_equal(a, b) for a, b in zip(r_one[4], r_grp[4])) and np.array_equal(r_grp[2], r_grp[0][4]) and not r_grp[1].any())
    # the same checks the spawned-rank form makes: every commitment against the closed form of the WHOLE problem
    # (sum s_i (G + i D) = [sum s_i] G + [sum i s_i] D on Python integers), the transforms through an inverse transform on one device
    from plonky_amd.selfcheck import closed_form_msm
    closed = all((synth.from_mont(0, r_grp[0][v][0]), synth.from_mont(0, r_grp[0][v][1])) == closed_form_msm(0, vecs[v], G, D, first=0) for v in range(9))
    lib.check(L.plk_init(0))
    back = np.zeros_like(polys[0])
    lib.check(L.plk_ntt(0, log_n, 1, vp(r_grp[4][8].ctypes.data), vp(back.ctypes.data)))
    roundtrip = bool(np.array_equal(back, polys[8]))
    L.plk_shutdown()
    out = {"devices": n_devices, "virtual": bool(virtual), "log_n": log_n, "one_device": one, "group": grp, "bit_identical_to_one_device": bool(same),
           "msm_closed_form_bit_exact": bool(closed), "ntt_roundtrip_bit_exact": roundtrip,
           "note": "host-pointer C ABI (pageable numpy buffers, PCIe inside): plk_msm_execute_batch of nine 2^log_n vectors, one plk_msm_execute, "
                   "plk_ntt_batch of nine transforms; efficiency = T_one_device / (N T_group)"}
    for k in ("commit9_ms", "msm_single_ms", "ntt9_ms"):
        out["efficiency_" + k[:-3]] = one[k] / (n_devices * grp[k])
    return out


def single_process_child(n_devices, log_n, steps, timeout_s=900, extra=()):
    """single_process_case in a process of its own (no launcher variables in its environment), its JSON line parsed; a time-out, a
    crash or a failed self-check comes back as {"error": ...} - never as an exception, never as a hang of the caller."""
    import subprocess
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "GROUP_WORLD_SIZE", "ROLE_RANK",
                        "ROLE_WORLD_SIZE", "ROLE_NAME") and not k.startswith("TORCHELASTIC") and not k.startswith("TORCH_NCCL")}
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", str(n_devices), "--single-process", "--log-n", str(log_n), "--steps", str(steps)] + list(extra)
    try:
        p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout_s)
    except subprocess.TimeoutExpired:
        return {"error": "no result within %d s (child stopped)" % timeout_s}
    except OSError as e:
        return {"error": "could not start: %s" % e}
    for line in reversed(p.stdout.splitlines()):
        if line.startswith("{"):
            try:
                r = json.loads(line)["components"]
                r["exit_code"] = p.returncode
                return r
            except (ValueError, KeyError):
                break
    return {"error": "exit code %d, no JSON line; stderr tail: %s" % (p.returncode, p.stderr[-300:])}


def run_single_process(args):
    """python bench.py --gpus N --single-process [--virtual-devices]: ONE JSON line for the in-library multi-GPU path."""
    import torch
    from plonky_amd import lib
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU path)"
    have = torch.cuda.device_count()
    virtual = args.virtual_devices or have < args.gpus
    L = lib.load()
    r = single_process_case(L, lib, args.gpus, args.log_n, max(2, args.steps // 4), virtual)
    pairs = 9 * (1 << args.log_n)
    result = {"metric": "MSM Mpairs/sec + NTT Melems/sec, 2^20 Tweedledee, 1/2/4/8 GPU", "value": pairs / (r["group"]["commit9_ms"] * 1e-3) / 1e6,
              "unit": "M pairs/s of the nine-vector commitment batch from HOST memory (PCIe inside), one process, %d devices" % args.gpus,
              "n_gpus": args.gpus, "steps": max(2, args.steps // 4), "warmup": 2, "ms_per_step": r["group"]["commit9_ms"], "higher_is_better": True,
              "scaling": "strong", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
              "config": {"workload": "single process, plk_init_devices(%d)%s: nine 2^%d-pair commitments per step through plk_msm_execute_batch (host pointers)"
                                     % (args.gpus, " on virtual devices of GPU 0" if virtual else "", args.log_n), "log_n": args.log_n, "curve": "tweedledee",
                         "gpu": gpu_identity(torch, 0)},
              "components": r, "checks": {"bit_identical_to_one_device": r["bit_identical_to_one_device"], "msm_closed_form_bit_exact": r["msm_closed_form_bit_exact"],
                                         "ntt_roundtrip_bit_exact": r["ntt_roundtrip_bit_exact"]}}
    print(json.dumps(result), flush=True)
    assert all(result["checks"].values()), "self-check failed: %r" % result["checks"]


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
    ceil = load_ceilings()
    ceil_ok = bool(ceil) and not ceil.get("stale")
    peak = ceil.get("fz_mul_gops", {}).get("tweedledee") if ceil_ok else None
    mad_peak = ceil.get("mad_u64_u32_glaneops") if ceil_ok else None
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
        return {"kernel": kernel, "bound": "valu", "achieved": gmm, "peak": peak, "unit": "G modmul/s", "frac": gmm / peak if peak else None,
                "mad_issue_frac": gmm * 126 / mad_peak if mad_peak else None, "launch_ms": ms, "traffic": tr, "traffic_source": trs,
                "hbm": {"achieved": alg_bytes / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        "algorithmic_bytes_per_launch": alg_bytes}, "note": note}

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
        "metric": "MSM Mpairs/sec + NTT Melems/sec, 2^20 Tweedledee, 1/2/4/8 GPU", "value": n8 / (v_ms * 1e-3) / 1e6,
        "unit": "M points/s of the quotient numerator (the fold is reported in components)", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": v_ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
        "config": {"workload": "quotient numerator of a 2^%d-gate circuit (8n = 2^%d points) + generator fold of 2^%d pairs" % (log_degree, log_degree + 3, m.bit_length() - 1),
                   "log_n": args.log_n, "curve": "tweedledee", "kernel_source_sha": src_hash},
        "components": {"vanishing_points_ms": v_ms, "vanishing_mpoints_per_s": n8 / (v_ms * 1e-3) / 1e6, "fold_pairs_ms": f_ms,
                       "fold_mpairs_per_s": m / (f_ms * 1e-3) / 1e6, "fold_4_to_1_ms": f4_ms,
                       "fold_4_to_1_minputs_per_s": (2 * m / (f4_ms * 1e-3) / 1e6) if f4_ms else None, "wall_s": elapsed},
        "checks": checks, "roofline": rooflines["vanishing_points"], "rooflines": rooflines,
    }
    print(json.dumps(result), flush=True)
    assert all(checks.values()), "self-check failed: %r" % checks


def run(args):
    if args.workload == "quotient":
        return run_quotient(args)
    cv = CURVES[args.curve]
    CURVE, NTT_FIELD = cv["curve"], cv["ntt_field"]

    import ctypes
    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, "--gpus %d but WORLD_SIZE=%d (launch with `python bench.py --gpus N` or torchrun --nproc-per-node N)" % (args.gpus, world)
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU path)"
    device_index = 0 if args.same_device else local_rank
    assert device_index < torch.cuda.device_count(), "rank %d needs GPU %d, %d visible" % (rank, device_index, torch.cuda.device_count())
    torch.cuda.set_device(device_index)
    gloo = bool(args.same_device)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if gloo:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device_index))

    from plonky_amd import api, device as dev, lib, parallel, synth
    from plonky_amd.selfcheck import closed_form_msm, _mul
    from plonky_amd.synth import MODULI
    dev.init(device_index)
    L = lib.load()

    n = 1 << args.log_n
    commit9 = args.workload == "commit9"
    do_ntt = args.workload in ("both", "ntt")
    do_msm = args.workload in ("both", "msm", "commit9")
    batch = 9 if commit9 else 1
    strong = commit9 or (args.workload == "msm" and args.shard)
    # the shard this process computes: its own rank, or (emulation) rank r of N on a single GPU
    shard_rank, shard_world = rank, world
    if args.emulate_rank:
        assert world == 1 and strong, "--emulate-rank needs a strong-scaling workload (commit9, msm --shard) and --gpus 1"
        shard_rank, shard_world = (int(v) for v in args.emulate_rank.split("/"))
        assert 0 <= shard_rank < shard_world

    # ---- synthetic inputs, resident in HBM before the timed region ----
    if do_ntt:
        x_host = synth.rand_field(NTT_FIELD, SEED_NTT + rank, n)
        x = dev.to_device(x_host)
        y = torch.empty_like(x)
        lib.check(L.plk_ntt_precompute(NTT_FIELD, args.log_n))
    if do_msm:
        from plonky_amd.selfcheck import GENERATORS
        p = MODULI[cv["base_field"]]
        G = GENERATORS[CURVE]  # tweedledee_curve.rs:14-18 / bls12_377_curve.rs:16-33
        d = synth.to_int(synth.rand_field(cv["scalar_field"], SEED_MSM, 1)[0]) % MODULI[cv["scalar_field"]]
        D = _mul(p, d, G)
        g0 = np.stack([synth.mont(cv["base_field"], G[0]), synth.mont(cv["base_field"], G[1])])
        dd = np.stack([synth.mont(cv["base_field"], D[0]), synth.mont(cv["base_field"], D[1])])
        if strong:
            # whole vectors per rank + the remainder sharded by base range (parallel.BatchPlan); one MSM: the sharded case alone
            plan = parallel.BatchPlan(batch, shard_world, shard_rank, n)
            first, n_local = plan.first, plan.n_local
            s_host = np.stack([synth.rand_field(cv["scalar_field"], SEED_MSM + 0x900 + k, n) for k in range(batch)])
            s = dev.to_device(plan.local_scalars(s_host))
            slots, whole = plan.slots, plan.whole
        else:
            first, n_local = rank * n, n                         # this rank's contiguous range of the global N * n MSM
            s_host = synth.rand_field(cv["scalar_field"], SEED_MSM + 1 + rank, n)
            s = dev.to_device(s_host)
            slots, whole, plan = 1, 0, None
        bases = dev.gen_bases_dev(CURVE, n_local, g0, dd, first=first)
        pre = dev.msm_precompute_dev(CURVE, bases)
        # the exchange step of the sharded MSM, every buffer allocated once: the MSM writes its results straight into the send
        # record of the ONE all-gather; whole vectors are handed over, sharded ones added up, on the device
        ex = parallel.PartialExchange(CURVE, batch, "cuda", whole_per_rank=whole, world=shard_world if strong else world,
                                      rank=shard_rank if strong else rank)
        oxy, oz = ex.out_xy, ex.out_zero
        exchange = world > 1 or shard_world > 1

    # a rank that holds whole vectors AND a share of a sharded one passes the share with its base range (plk_msm_execute_parts_dev)
    msm_parts = plan.parts(s) if (do_msm and plan is not None and plan.full_context and plan.sharded and not args.no_parts) else None

    def step():
        if do_ntt:
            dev.ntt_dev(NTT_FIELD, x, out=y)
        if do_msm:
            if msm_parts is not None:
                dev.msm_execute_parts_dev(pre, msm_parts, oxy, oz)
            else:
                dev.msm_execute_dev(pre, s, oxy, oz)
            if exchange:
                ex.gather()
                ex.combine()

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()

    # ---- the timed region: EXACTLY K steps, nothing of the harness inside (no per-kernel events) ----
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if gloo else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- the same K steps once more with HIP events on the launch stream around each kernel: the durations the rooflines use.
    # The events cost launches of their own (8 per MSM, 2 per NTT pass): this region is reported as ms_per_step_profiled, never as
    # the headline.  A batched MSM (commit9) shares ONE reduction among its vectors, which the per-stage events would split up:
    # its stage times come from single executions further down.
    msm_live_profile = do_msm and batch == 1
    if do_ntt:
        L.plk_ntt_get_timings(None, None)
        L.plk_ntt_set_profiling(1)
    if msm_live_profile:
        L.plk_msm_set_profiling(pre._ctx, 1)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    elapsed_profiled = time.perf_counter() - t0

    ntt_kernel_ms = msm_stage_ms = None
    ntt_launches = 0
    if do_ntt:
        sm, cnt = ctypes.c_double(0), ctypes.c_uint(0)
        L.plk_ntt_get_timings(ctypes.byref(sm), ctypes.byref(cnt))
        L.plk_ntt_set_profiling(0)
        ntt_launches = cnt.value
        ntt_kernel_ms = sm.value
    if do_msm:
        if not msm_live_profile:
            L.plk_msm_set_profiling(pre._ctx, 1)
            for _ in range(max(1, args.steps // 4)):
                dev.msm_execute_dev(pre, s, oxy, oz)
            sync()
        arr = (ctypes.c_double * 7)()
        calls = ctypes.c_uint(0)
        L.plk_msm_get_timings(pre._ctx, arr, ctypes.byref(calls))
        L.plk_msm_set_profiling(pre._ctx, 0)
        msm_stage_ms = [v / max(1, calls.value) for v in arr]   # per MSM (a profiled batch runs its MSMs one by one)

    # ---- component timings (separate loops) so both headline numbers are reported ----
    # Every loop is warmed with three calls of the same work and runs until it has lasted a few milliseconds: a ten-call loop
    # straight after host-side preparation measures the GPU's clock ramp, not the kernel (profiles/r04_ntt_harness_reconcile.txt:
    # 8.5-9.2 G elements/s in such a loop against 10.4 in steady state, same call).
    comp = {}
    do_ntt_c, do_msm_c = (False, False) if args.timed_only else (do_ntt, do_msm)

    def loop_time(fn, iters, warm=3):
        iters = max(1, iters)
        for _ in range(warm):
            fn()
        sync()
        t_ = time.perf_counter()
        for _ in range(iters):
            fn()
        sync()
        return (time.perf_counter() - t_) / iters

    if do_ntt_c:
        tn = loop_time(lambda: dev.ntt_dev(NTT_FIELD, x, out=y), max(args.steps, 100))
        comp["ntt_ms"] = tn * 1e3
        comp["ntt_melems_per_s"] = world * n / tn / 1e6
        # the prover transforms its 9 wire polynomials together (plonk_util.rs:169-190): same kernels, one call
        xb = x.unsqueeze(0).repeat(9, 1, 1).contiguous()
        yb = torch.empty_like(xb)
        tb = loop_time(lambda: dev.ntt_dev(NTT_FIELD, xb, out=yb), max(args.steps, 30))
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
            comp["divide_by_z_h_ms"] = loop_time(lambda: dev.divide_by_z_h_dev(NTT_FIELD, m, nq, out=q_out), args.steps) * 1e3
            w = dev.to_device(synth.rand_field(NTT_FIELD, SEED_NTT + 200 + rank, 9 * nq)).reshape(9, nq, 4)
            ev = torch.empty((9, n, 4), dtype=torch.int64, device="cuda")
            comp["lde9_ms"] = loop_time(lambda: dev.ntt_padded_dev(NTT_FIELD, w, args.log_n, out=ev), args.steps // 2) * 1e3
            comp["quotient_path_note"] = "divide_by_z_h: degree < 2^%d by Z_H of 2^%d (2 fused transforms); lde9: 9 x 2^%d coefficients -> 2^%d evaluations" % (
                args.log_n, args.log_n - 3, args.log_n - 3, args.log_n)
            if not args.no_check:
                q_host = dev.to_host(q_out)
                comp["_q_check"] = bool(np.array_equal(q_host[: 7 * nq], q0) and not q_host[7 * nq:].any())
            del m, q_out, ev
    if do_msm_c:
        tm = loop_time(lambda: dev.msm_execute_dev(pre, s, oxy, oz), args.steps)
        pairs = batch * (n if strong else world * n)
        comp["msm_ms"] = tm * 1e3
        comp["msm_mpairs_per_s"] = (plan.pairs_local() if args.emulate_rank else pairs) / tm / 1e6
    if do_msm_c and not strong:
        # commit_polynomials (plonk_util.rs:215-231): the 9 wire polynomials against the same generators, one call
        sb = s.unsqueeze(0).repeat(9, 1, 1).contiguous()
        oxy9 = torch.empty((9, 2, cv["limbs"]), dtype=torch.int64, device="cuda")
        oz9 = torch.empty((9,), dtype=torch.uint8, device="cuda")
        tb = loop_time(lambda: dev.msm_execute_dev(pre, sb, oxy9, oz9), args.steps // 4, warm=2)
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
    # ---- the opening argument of the same proof (halo.rs:63-124): all log2(n) rounds behind the C ABI, over
    # the prover's commitment tables [pedersen_g .., pedersen_h, U] (plk_halo_begin_tabled_dev), full-size challenges ----
    if do_msm_c and not strong and world == 1 and args.curve == "tweedledee" and args.log_n >= 12:
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
    # ---- the drop-in entry points: HOST pointers, PCIe included - what an unmodified plonk.rs gets (plonk_util.rs:169-231) ----
    if not args.timed_only and not strong and world == 1 and args.log_n >= 16:
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
            # both directions carry 9 x 32 MiB; the link's two directions overlap only partly on this platform: nine pinned uploads
            # + downloads on three streams take 8.9 ms (profiles/r03_h2d_probe.txt), 1.65 x the one-way time
            host["host_ntt9_pinned_duplex_floor_ms"] = 8.9
        host["note"] = "host-pointer C ABI calls on pageable numpy buffers, one caller thread; pcie_floor = bytes one way / 56 GB/s"
        comp["host_pointer"] = host
    if do_msm:
        comp["msm_window_bits"] = pre.window
        comp["msm_stage_ms"] = dict(zip(STAGES, [round(v, 4) for v in msm_stage_ms]))

    # ---- correctness of what was just timed (not in the timed region) ----
    checks = {}
    if not args.no_check:
        if do_ntt:
            back = dev.to_host(dev.ntt_dev(NTT_FIELD, y, inverse=True))
            checks["ntt_roundtrip_bit_exact"] = bool(np.array_equal(back, x_host))
            if "_q_check" in comp:
                checks["divide_by_z_h_identity"] = comp.pop("_q_check")
            if "_ntt_ok" in comp.get("host_pointer", {}):
                checks["host_pointer_ntt_equals_device"] = comp["host_pointer"].pop("_ntt_ok")
        if do_msm:
            dev.msm_execute_dev(pre, s, oxy, oz)
            if exchange:
                ex.gather()
                gxy, gz = ex.combine()
            torch.cuda.synchronize()
            got = dev.to_host(oxy).reshape(slots, 2, cv["limbs"])
            ok = int(oz.sum().item()) == 0
            for k in range(slots):
                if not strong:
                    exp = closed_form_msm(CURVE, s_host, G, D, first=first)          # this rank's range of the global MSM
                elif k < whole:
                    exp = closed_form_msm(CURVE, s_host[plan.own[k]], G, D, first=0)  # a whole vector of this rank
                else:
                    exp = closed_form_msm(CURVE, s_host[plan.rem[k - whole], plan.lo:plan.hi], G, D, first=plan.lo)  # its slice of a sharded one
                gotp = (synth.from_mont(cv["base_field"], got[k][0]), synth.from_mont(cv["base_field"], got[k][1]))
                ok = ok and gotp == exp
            checks["msm_closed_form_bit_exact"] = bool(ok)
            if "_b9_check" in comp:
                checks["msm_batch9_equals_single"] = comp.pop("_b9_check")
            if "_os_check" in comp:
                checks["msm_one_shot_equals_tabled"] = comp.pop("_os_check")
            if "_ipa_check" in comp:
                checks["ipa_deterministic"] = comp.pop("_ipa_check")
            if "_msm_ok" in comp.get("host_pointer", {}):
                checks["host_pointer_msm_equals_device"] = comp["host_pointer"].pop("_msm_ok")
            if world > 1:
                # the device results of the exchange against the host-pointer point sum over the gathered records and (strong
                # scaling) the closed form of the WHOLE problem: every vector against all 2^log_n generators
                hx, hz = ex.partials()
                hx, hz = dev.to_host(hx), hz.cpu().numpy()
                tot_dev, tz_dev = dev.to_host(gxy), gz.cpu().numpy()
                ok = True
                for v in range(batch):
                    if v < whole * world:
                        tot, tz = hx[v % world, v // world], int(hz[v % world, v // world])
                    else:
                        sl = whole + (v - whole * world)
                        tot, tz = api.curve_sum_affine(CURVE, hx[:, sl], hz[:, sl])
                    ok = ok and tz == 0 and int(tz_dev[v]) == 0 and np.array_equal(tot, tot_dev[v])
                    if strong:
                        exp = closed_form_msm(CURVE, s_host[v], G, D, first=0)
                        ok = ok and (synth.from_mont(cv["base_field"], tot[0]), synth.from_mont(cv["base_field"], tot[1])) == exp
                checks["msm_global_sum_closed_form" if strong else "msm_global_sum_is_point"] = bool(ok)

    # ---- N > 1, the driver's default line: the STRONG-scaling configurations of BASELINE.json as well (configs 4 and 5), each with
    # its one-GPU time measured by rank 0 in this same run, and the number of ranks the RCCL communicator actually carries ----
    multi = {}
    if world > 1 and args.workload == "both" and not args.timed_only:
        k_strong = max(3, args.steps // 4)
        cases = {"commit9_strong": ("tweedledee", args.log_n, 9), "bls12_377_2p22_shard": ("bls12_377", min(22, args.log_n + 2), 1)}
        for name, (cname, lg, bt) in cases.items():
            rN = strong_case(cname, lg, bt, world, rank, k_strong, 2, gloo, solo=False)
            r1 = strong_case(cname, lg, bt, world, rank, k_strong, 2, gloo, solo=True) if rank == 0 else None
            sync()
            multi[name + "_ms"] = rN["ms"]
            checks[name + "_closed_form"] = rN["ok"]
            if r1 is not None:
                multi[name + "_one_gpu_ms"] = r1["ms"]
                multi[name + "_efficiency"] = r1["ms"] / (world * rN["ms"])
                checks[name + "_one_gpu_closed_form"] = r1["ok"]
            multi[name + "_problem"] = "%d x 2^%d pairs, %s" % (bt, lg, cname)
        ones = torch.ones(1, dtype=torch.int32, device="cpu" if gloo else "cuda")
        dist.all_reduce(ones)
        multi["rccl_ranks" if not gloo else "gloo_ranks"] = int(ones.item())
        multi["backend"] = dist.get_backend()
        # the in-library form of the same split: rank 0 alone drives all N GPUs from its one process through the host-pointer
        # C ABI (plk_init_devices) while the other ranks wait; skipped when the ranks share a GPU
        if not gloo and torch.cuda.device_count() >= world:
            # the other ranks wait on the process group's key-value store, on the CPU: a collective barrier would park an RCCL
            # kernel on the very GPUs rank 0 is about to measure
            store = dist.distributed_c10d._get_default_store()
            sync()
            if rank == 0:
                # in a CHILD process with a time limit (`bench.py --gpus N --single-process`, the form the GPU suite runs on virtual
                # devices): this path has never met a real multi-GPU node, and neither a hang nor a crash in it may take the
                # spawned-rank numbers above with it
                multi["single_process"] = single_process_child(world, args.log_n, args.steps)
                if "error" not in multi["single_process"]:
                    checks["single_process_bit_identical"] = multi["single_process"]["bit_identical_to_one_device"]
                    checks["single_process_msm_closed_form"] = multi["single_process"]["msm_closed_form_bit_exact"]
                store.set("plk_single_process_done", "1")
            else:
                store.wait(["plk_single_process_done"])
        sync()
        comp["multi_gpu"] = multi

    units_per_step = (n if do_ntt else 0) + (((plan.pairs_local() if args.emulate_rank else batch * n)) if do_msm else 0)
    value = (1 if strong else world) * units_per_step * args.steps / elapsed / 1e6

    # ---- rooflines: every kernel entry carries the integer-ALU figure (what binds these kernels) and the HBM figure ----
    rooflines = {}
    src_hash = kernel_source_hash()
    pmc = {}
    try:
        with open(latest_profile("pmc_traffic.json")) as fh:
            pmc = json.load(fh)
    except (OSError, ValueError):
        pass

    def traffic_of(kname):
        e = pmc.get(kname)
        # a PMC figure is only valid for the kernels it was measured on, at the size and window it was measured at
        if not e or pmc.get("kernel_source_sha") != src_hash or e.get("log_n") != args.log_n or pmc.get("curve", "tweedledee") != args.curve:
            return None, None
        return e["bytes_per_launch"], e.get("source")

    ceil = load_ceilings()
    ceil_ok = bool(ceil) and not ceil.get("stale")
    valu_peak = ceil.get("fz_mul_gops", {}).get(args.curve) if ceil_ok else None        # G modmul/s: this round's fz_mul at 4 waves / SIMD
    mad_peak = ceil.get("mad_u64_u32_glaneops") if ceil_ok else None                     # G lane-ops/s: raw v_mad_u64_u32 issue rate
    mads = MADS_PER_MODMUL[cv["limbs"]]
    ceil_src = ("profiles/%s (arith_source_sha %s, gpu %s)" % (os.path.basename(CEILINGS_FILE), ceil.get("arith_source_sha"), ceil.get("gpu_uuid"))) if ceil_ok else None

    def valu_entry(kernel, gmm, executed_gmm, launch_ms, extra):
        e = {"kernel": kernel, "bound": "valu", "achieved": gmm, "peak": valu_peak, "unit": "G modmul/s",
             "frac": gmm / valu_peak if valu_peak else None, "peak_source": ceil_src,
             # executed multiplier instructions against the measured raw issue rate: independent of this repo's fz_mul
             "mad_issue_frac": executed_gmm * mads / mad_peak if mad_peak else None, "launch_ms": launch_ms}
        e.update(extra)
        return e

    if do_ntt and ntt_launches:
        per_launch_ms = ntt_kernel_ms / ntt_launches
        launches_per_ntt = ntt_launches / args.steps
        # algorithmic bytes per transform: 64 B / element (read 32 B + write 32 B once);
        # one launch of the pass kernel handles all n elements once => 64 B * n / launches_per_ntt per launch
        alg_bytes = 64.0 * n / launches_per_ntt
        ach = alg_bytes / (per_launch_ms * 1e-3) / 1e9
        t_ntt = per_launch_ms * launches_per_ntt * 1e-3
        gmm = (n / 2.0 * args.log_n) / t_ntt / 1e9   # algorithmic: n/2 log n multiplications
        gmm_exec = (n * 9.75) / t_ntt / 1e9           # executed: 9.75 per element at 2^20
        tr, src = traffic_of("k_ntt_pass")
        rooflines["ntt_pass"] = valu_entry("k_ntt_pass", gmm, gmm_exec, per_launch_ms, {
            "traffic": tr, "traffic_source": src,
            "hbm": {"achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": alg_bytes},
            "launches_per_transform": launches_per_ntt,
            "note": "VALU-bound (DESIGN.md section 4): achieved counts the algorithmic n/2 log n multiplications; the kernel executes more (inter-pass twiddles)"})
    if do_msm:
        acc_ms = msm_stage_ms[3]
        n_acc = (plan.hi - plan.lo) if (strong and plan.whole == 0) else n_local   # pairs of the MSM the stage times belong to (slot 0)
        alg_bytes = float(cv["pair_bytes"]) * n_acc   # affine base + 32 B scalar per pair, one MSM
        ach = alg_bytes / (acc_ms * 1e-3) / 1e9
        windows = (cv["scalar_bits"] + 1 + pre.window - 1) // pre.window
        adds = n_acc * windows
        gmm = adds * 10.0 / (acc_ms * 1e-3) / 1e9        # a mixed XYZZ addition = 8 M + 2 S
        tr, src = traffic_of("k_msm_accumulate")
        rooflines["msm_accumulate"] = valu_entry("k_msm_accumulate", gmm, gmm, acc_ms, {
            "traffic": tr, "traffic_source": src,
            "hbm": {"achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": alg_bytes},
            "mixed_adds_per_s": adds / (acc_ms * 1e-3),
            "note": "integer-ALU bound by construction (~2 modmul per algorithmic byte); the HBM fraction is reported because the metric asks for it"})
    roofline = None
    if rooflines:
        roofline = max(rooflines.values(), key=lambda r: r["launch_ms"] * (r.get("launches_per_transform", 1)))

    wl = {"both": "2^%d %s forward NTT + 2^%d-pair %s MSM per GPU per step" % (args.log_n, "TweedledeeBase" if args.curve == "tweedledee" else "Bls12377Scalar", args.log_n, args.curve),
          "ntt": "2^%d forward NTT per GPU per step" % args.log_n,
          "msm": ("ONE 2^%d-pair %s MSM per step, generators sharded by base range over the GPUs" if strong else "2^%d-pair %s MSM per GPU per step") % (args.log_n, args.curve),
          "commit9": "9-wire commitment batch: nine 2^%d-pair %s MSMs against the same generators per step (generators sharded by base range over the GPUs)" % (args.log_n, args.curve)}[args.workload]
    result = {
        "metric": "MSM Mpairs/sec + NTT Melems/sec, 2^20 Tweedledee, 1/2/4/8 GPU",
        "value": value,
        "unit": "M units/s (1 unit = 1 NTT element or 1 MSM scalar-point pair; components below)",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "ms_per_step_profiled": elapsed_profiled / args.steps * 1e3,
        "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
        "dtype": "u32", "data": "synthetic",
        "config": {"workload": wl, "log_n": args.log_n, "curve": args.curve,
                   "sharding": ("whole vectors per rank, the remainder sharded by base range; one packed all-gather + device point sum" if strong else
                                "independent NTTs; MSM sharded by base range + one packed all-gather of partial points + device point sum") if world > 1 else "single GPU",
                   "backend": ("gloo (ranks share GPU 0)" if gloo else "nccl (RCCL)") if world > 1 else None,
                   "seeds": {"ntt": SEED_NTT, "msm": SEED_MSM}, "kernel_source_sha": src_hash, "gpu": gpu_identity(torch, device_index)},
        "components": comp,
        "checks": checks,
        "roofline": roofline,
        "rooflines": rooflines,
    }
    if args.emulate_rank:
        # one rank of the N-rank problem alone on one GPU: every rank does the same amount of work, the exchange is one
        # all-gather of batch x (2L + 1) words per rank (latency-bound: ~20-40 us over xGMI), so the N-GPU step time is this
        # rank's time plus that, and the predicted whole-job rate is the global unit count over it
        result["emulated_rank"] = {"rank": shard_rank, "of": shard_world, "n_local": n_local, "whole_vectors": plan.whole, "sharded_vectors": plan.sharded,
                                   "pairs_local": plan.pairs_local(),
                                   "predicted_global_units_per_s_M": batch * n * args.steps / elapsed / 1e6,
                                   "note": "value / ms_per_step are THIS rank's share (its whole vectors + its base range of the sharded ones) incl. the local copy standing in for the all-gather and the point sum"}
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.emulate_rank:
        result["cpu_baseline"] = cpu_baseline(args.workload, cv)
    if rank == 0:
        print(json.dumps(result), flush=True)
    assert all(checks.values()), "self-check failed: %r" % checks
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""benchlib.headline -- the default line of bench.py: K steps of (one forward 2^20 NTT + one 2^20-pair MSM) per GPU between two
synchronisations, then the same steps with per-kernel HIP events for the rooflines, the component loops, the self-checks, the
multi-GPU section and the ONE JSON line."""
import json
import os
import time
from types import SimpleNamespace

from . import components, multi
from .common import (CURVES, HBM_PEAK_GBS, METRIC, SEED_MSM, SEED_NTT, STAGES, Ceilings, gpu_identity, kernel_source_hash, latest_profile)
from .cpu import cpu_baseline
from .quotient import run_quotient


def run(args):
    if args.workload == "quotient":
        return run_quotient(args)
    if args.steps < 1:
        raise SystemExit("--steps must be at least 1")
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
            plan = parallel.BatchPlan(batch, shard_world, shard_rank, n, bucket_shard=args.bucket_shard)
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
    msm_buckets = plan.buckets() if msm_parts is not None else None

    def step():
        if do_ntt:
            dev.ntt_dev(NTT_FIELD, x, out=y)
        if do_msm:
            if msm_parts is not None:
                dev.msm_execute_parts_dev(pre, msm_parts, oxy, oz, buckets=msm_buckets)
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


    # ---- component timings (separate loops, benchlib/components.py) so both headline numbers are reported ----
    comp = {}
    two_streams_same = None
    do_ntt_c, do_msm_c = (False, False) if args.timed_only else (do_ntt, do_msm)
    c = SimpleNamespace(args=args, cv=cv, comp=comp, dev=dev, api=api, lib=lib, L=L, synth=synth, np=np, torch=torch, sync=sync, n=n, world=world, rank=rank,
                        NTT_FIELD=NTT_FIELD, CURVE=CURVE, strong=strong, batch=batch, MODULI=MODULI, _mul=_mul)
    if do_ntt:
        c.x, c.y, c.x_host = x, y, x_host
    if do_msm:
        c.s, c.s_host, c.oxy, c.oz, c.pre, c.bases, c.plan, c.p, c.G = s, s_host, oxy, oz, pre, bases, plan, p, G
    if do_ntt_c:
        components.ntt_components(c)
    if do_msm_c:
        components.msm_components(c)
    if do_msm_c and not strong and world == 1 and args.curve == "tweedledee" and args.log_n >= 12:
        components.ipa_component(c)
    if not args.timed_only and not strong and world == 1 and args.log_n >= 16:
        components.host_pointer_components(c, do_ntt_c, do_msm_c)
    # (last of the component loops: when it ran first, the transform's own 100-call loop read 0.125 instead of 0.113 ms on three leases;
    # the cause was not isolated - tools/null_stream_probe.py does not reproduce it outside this harness - so nothing is measured after it)
    if do_ntt and do_msm and world == 1 and not strong and not args.timed_only:
        # The step's two calls are independent tasks (plonk.rs runs its transforms and commitments under Rayon): with the transform on a
        # second stream it runs under the MSM's reduction tail (a handful of workgroups for ~0.2 ms).  Reported beside the headline, which
        # keeps both calls on ONE stream so that its per-kernel durations are the ones the rocprofv3 trace of the same command shows.
        side = torch.cuda.Stream()
        y_side = torch.empty_like(y)

        def step2():
            with torch.cuda.stream(side):
                dev.ntt_dev(NTT_FIELD, x, out=y_side)
            dev.msm_execute_dev(pre, s, oxy, oz)
        for _ in range(args.warmup):
            step2()
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step2()
        sync()
        comp["step_two_streams_ms"] = (time.perf_counter() - t0) / args.steps * 1e3
        two_streams_same = bool(torch.equal(y, y_side))
        del y_side
    if do_msm:
        comp["msm_window_bits"] = pre.window
        comp["msm_stage_ms"] = dict(zip(STAGES, [round(v, 4) for v in msm_stage_ms]))

    # ---- correctness of what was just timed (not in the timed region) ----
    checks = {}
    if not args.no_check:
        if do_ntt:
            back = dev.to_host(dev.ntt_dev(NTT_FIELD, y, inverse=True))
            checks["ntt_roundtrip_bit_exact"] = bool(np.array_equal(back, x_host))
            if two_streams_same is not None:
                checks["ntt_on_second_stream_equals_first"] = two_streams_same
            if "_q_check" in comp:
                checks["divide_by_z_h_identity"] = comp.pop("_q_check")
            if "_ntt_ok" in comp.get("host_pointer", {}):
                checks["host_pointer_ntt_equals_device"] = comp["host_pointer"].pop("_ntt_ok")
        if do_msm:
            if msm_buckets is not None:
                dev.msm_execute_parts_dev(pre, msm_parts, oxy, oz, buckets=msm_buckets)
            else:
                dev.msm_execute_dev(pre, s, oxy, oz)
            if exchange:
                ex.gather()
                gxy, gz = ex.combine()
            torch.cuda.synchronize()
            got = dev.to_host(oxy).reshape(slots, 2, cv["limbs"])
            ok = int(oz.sum().item()) == 0
            if msm_buckets is not None:
                # a bucket range has no closed form of its own: the shares of ALL the ranks (computed here, one after the other) must add
                # up to the whole vector's closed form, and this rank's share must be the one the timed region produced
                for j in range(plan.sharded):
                    row = s[whole + j]
                    pts, zs = [], []
                    for r in range(shard_world):
                        pxy, pz = dev.msm_execute_parts_dev(pre, [(0, row)], buckets=[(r, shard_world)])
                        pts.append(dev.to_host(pxy)[0])
                        zs.append(int(pz.cpu()[0]))
                    ok = ok and np.array_equal(pts[shard_rank], got[whole + j])
                    tot, tz = api.curve_sum_affine(CURVE, np.stack(pts), np.array(zs, dtype=np.uint8))
                    exp = closed_form_msm(CURVE, s_host[plan.rem[j]], G, D, first=0)
                    ok = ok and tz == 0 and (synth.from_mont(cv["base_field"], tot[0]), synth.from_mont(cv["base_field"], tot[1])) == exp
            for k in range(slots):
                if msm_buckets is not None and k >= whole:
                    continue  # checked above
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
            if "_proj_check" in comp:
                checks["msm_projective_is_the_same_point"] = comp.pop("_proj_check")
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

    # ---- N > 1, the driver's default line: the STRONG-scaling configurations of BASELINE.json as well (benchlib/multi.py) ----
    if world > 1 and args.workload == "both" and not args.timed_only:
        comp["multi_gpu"] = multi.multi_gpu_section(args, world, rank, gloo, sync, checks)

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

    gpu = gpu_identity(torch, device_index)
    ceil = Ceilings(L, gpu, cv["limbs"], args.curve)   # measured in this process, on this GPU, now
    valu_entry = ceil.entry

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
            # the same lazy mixed addition with no memory at all, at the kernel's occupancy (plk_bench_ceilings, this process)
            "lazy_madd_no_memory_adds_per_s": ceil.lazy_madd * 1e9,
            "frac_of_lazy_madd_no_memory": adds / (acc_ms * 1e-3) / (ceil.lazy_madd * 1e9),
            "note": "integer-ALU bound by construction (~2 modmul per algorithmic byte); the HBM fraction is reported because the metric asks for it"})
    roofline = None
    if rooflines:
        roofline = max(rooflines.values(), key=lambda r: r["launch_ms"] * (r.get("launches_per_transform", 1)))

    wl = {"both": "2^%d %s forward NTT + 2^%d-pair %s MSM per GPU per step" % (args.log_n, "TweedledeeBase" if args.curve == "tweedledee" else "Bls12377Scalar", args.log_n, args.curve),
          "ntt": "2^%d forward NTT per GPU per step" % args.log_n,
          "msm": ("ONE 2^%d-pair %s MSM per step, generators sharded by base range over the GPUs" if strong else "2^%d-pair %s MSM per GPU per step") % (args.log_n, args.curve),
          "commit9": "9-wire commitment batch: nine 2^%d-pair %s MSMs against the same generators per step (generators sharded by base range over the GPUs)" % (args.log_n, args.curve)}[args.workload]
    result = {
        "metric": METRIC,
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
                   "seeds": {"ntt": SEED_NTT, "msm": SEED_MSM}, "kernel_source_sha": src_hash, "gpu": gpu},
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

"""benchlib.multi -- more than one GPU: the strong-scaling cases of BASELINE.json (configs 4, 5) inside a multi-rank run, the
spawned-rank launcher, and the single-process form (plk_init_devices: N GPUs behind the host-pointer C ABI)."""
import json
import os
import sys
import time

from .common import BENCH_PY, CURVES, METRIC, SEED_MSM, SEED_NTT, gpu_identity, parse_args


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


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _spawned_rank(rank, argv, world, port):
    os.environ.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    from .headline import run
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
    cmd = [sys.executable, BENCH_PY, "--gpus", str(n_devices), "--single-process", "--log-n", str(log_n), "--steps", str(steps)] + list(extra)
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
    result = {"metric": METRIC, "value": pairs / (r["group"]["commit9_ms"] * 1e-3) / 1e6,
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


def multi_gpu_section(args, world, rank, gloo, sync, checks):
    """N > 1, the driver's default line: BASELINE configs 4 and 5 as STRONG-scaling cases, each with its one-GPU time measured by rank 0 in
    this same run, the number of ranks the RCCL communicator actually carries, and the single-process form.  Fills `checks`."""
    import torch
    import torch.distributed as dist
    multi = {}
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
    return multi

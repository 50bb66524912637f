"""BASELINE.json's full-size configurations through the C ABI, checked by the closed form of the arithmetic-progression
generators (sum s_i (G0 + i D) = [sum s_i] G0 + [sum i s_i] D, two scalar multiplications on Python integers):
config 4 (the 9-wire commit batch at 2^20), config 5 (a 2^22 BLS12-377 G1 MSM), and the sharded MSM of the multi-GPU path over
the HIP code with the real RCCL backend at world size 1."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import bigint_ref as br
from plonky_amd import synth
from plonky_amd.selfcheck import closed_form_msm
from tests.test_oracle_kats import from_mont_arr
from tests.util import limbs_to_int


def _pt(c, P):
    return np.array([c.base.mont_limbs(P[0]), c.base.mont_limbs(P[1])], dtype=np.uint64)


def test_commit9_2p20_closed_form():
    """BASELINE config 4: nine 2^20 scalar vectors against the same 2^20 generators, one batched call; every result is
    checked against its own closed form."""
    torch = pytest.importorskip("torch")
    from plonky_amd import device as dev
    dev.init(0)
    c = br.TWEEDLEDEE
    n = 1 << 20
    G = (c.gx, c.gy)
    D = br.ec_mul(c, 0x350920, G)
    bases = dev.gen_bases_dev(0, n, _pt(c, G), _pt(c, D))
    pre = dev.msm_precompute_dev(0, bases)
    sv = np.stack([synth.rand_field(1, 0x350920 + k, n) for k in range(9)])
    oxy, oz = dev.msm_execute_dev(pre, dev.to_device(sv))
    torch.cuda.synchronize()
    got = dev.to_host(oxy)
    assert not oz.cpu().numpy().any()
    for k in range(9):
        assert tuple(from_mont_arr(c.base, got[k])) == closed_form_msm(0, sv[k], G, D), k
    pre.free()


def test_bls12_377_2p22_closed_form():
    """BASELINE config 5 (BLS12-377 G1, the curve the reference actually has): one 2^22-pair MSM."""
    torch = pytest.importorskip("torch")
    from plonky_amd import device as dev
    dev.init(0)
    c = br.BLS12_377
    n = 1 << 22
    G = (c.gx, c.gy)
    D = br.ec_mul(c, 0x350022, G)
    bases = dev.gen_bases_dev(2, n, _pt(c, G), _pt(c, D))
    pre = dev.msm_precompute_dev(2, bases)
    s = synth.rand_field(2, 0x350022, n)
    oxy, oz = dev.msm_execute_dev(pre, dev.to_device(s))
    torch.cuda.synchronize()
    assert int(oz.cpu()[0]) == 0
    assert tuple(from_mont_arr(c.base, dev.to_host(oxy).reshape(2, 6))) == closed_form_msm(2, s, G, D)
    pre.free()


def test_halo_argument_2p20_two_routes_and_closed_form():
    """The inner-product argument of one 2^20 opening (halo.rs:63-124) at BASELINE's size, twice: the plain
    context (stage of two virtual rounds over the explicit generators, pairwise folds, frozen generators) and the one that starts over
    the prover's commitment tables with pedersen_h and U inside them.  Every L_j / R_j and the final halo_a / halo_b / halo_g of the
    two routes must agree bit for bit, and the final generator must be the closed form <s, G> = [sum s_i] G0 + [sum i s_i] D,
    s_i = prod_j u_j^(+-1) by the bits of i (sum s_i = prod (u_j + u_j^-1); sum i s_i by the same product with one factor replaced)."""
    torch = pytest.importorskip("torch")
    from plonky_amd import device as dev
    from plonky_amd.selfcheck import _add, _mul
    dev.init(0)
    c = br.TWEEDLEDEE
    f = c.scalar
    log_n = 20
    n = 1 << log_n
    p, r = c.base.p, f.p
    G = (c.gx, c.gy)
    kd = 0x51ED270B
    D = br.ec_mul(c, kd, G)
    gens = dev.gen_bases_dev(c.curve_id, n + 2, _pt(c, G), _pt(c, D))      # pedersen_g (n), pedersen_h, U
    g = gens[:n].contiguous()
    x_int = 0x1F3D5B79A2C4E6081F3D5B79A2C4E6081F3D5B79A2C4E608 % r
    h_pt, u_pt = _pt(c, _mul(p, (1 + n * kd) % r, G)), _pt(c, _mul(p, x_int * (1 + (n + 1) * kd) % r, G))   # u_prime = [x] U
    a, b = dev.to_device(synth.rand_field(f.field_id, 41, n)), dev.to_device(synth.rand_field(f.field_id, 42, n))
    u_ints = [limbs_to_int(row) % r or 1 for row in synth.rand_field(f.field_id, 43, log_n)]
    m = lambda v: np.array(f.mont_limbs(v), dtype=np.uint64)
    us = [(m(u), m(pow(u, -1, r))) for u in u_ints]
    bl = [(m(1000 + j), m(2000 + j)) for j in range(log_n)]
    tables = dev.msm_precompute_dev(c.curve_id, gens)

    def run(**kw):
        arg = dev.HaloArgument(c.curve_id, a, b, g, h_pt, u_pt, **kw)
        lrs = []
        for j in range(log_n):
            lrs.append(arg.round_lr(*bl[j]))
            arg.round_fold(*us[j])
        fin = arg.read()
        arg.free()
        return lrs, fin

    plain_lr, plain_fin = run()
    tab_lr, tab_fin = run(tables=tables, h_index=n, u_index=n + 1, u_prime_scalar=m(x_int))
    for j, ((lr1, z1), (lr2, z2)) in enumerate(zip(plain_lr, tab_lr)):
        assert np.array_equal(lr1, lr2) and np.array_equal(z1, z2) and not z1.any(), "round %d" % j
    for x1, x2 in zip(plain_fin, tab_fin):
        assert np.array_equal(x1, x2)
    # closed form of the final generator: round j (j = 0 first) splits on bit log_n - 1 - j of the index; hi takes u_j, lo u_j^-1
    inv = [pow(u, -1, r) for u in u_ints]
    sum_s = 1
    for u, ui in zip(u_ints, inv):
        sum_s = sum_s * (u + ui) % r
    sum_is = 0
    for j, u in enumerate(u_ints):
        term = (1 << (log_n - 1 - j)) * u % r
        for k in range(log_n):
            if k != j:
                term = term * (u_ints[k] + inv[k]) % r
        sum_is = (sum_is + term) % r
    exp = _add(p, _mul(p, sum_s, G), _mul(p, sum_is * kd % r, G))
    fa, fb, fg, fgz = tab_fin
    assert int(fgz[0]) == 0 and tuple(from_mont_arr(c.base, fg[0])) == exp


def test_sharded_msm_hip_nccl_world_size_1():
    """plonky_amd.parallel.msm_sharded_hip over the HIP path with backend nccl (= RCCL): base range of this rank, one packed
    all-gather, plk_curve_sum_affine.  One rank here; the CPU suite runs the same plumbing at world size 2 on gloo."""
    torch = pytest.importorskip("torch")
    import torch.distributed as dist
    from plonky_amd import device as dev, parallel
    dev.init(0)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29577")
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        c = br.TWEEDLEDEE
        n = 1 << 14
        G = (c.gx, c.gy)
        D = br.ec_mul(c, 4242, G)
        lo, hi = parallel.shard_bounds(n, dist.get_rank(), dist.get_world_size())
        bases = dev.gen_bases_dev(0, hi - lo, _pt(c, G), _pt(c, D), first=lo)
        pre = dev.msm_precompute_dev(0, bases)
        sv = np.stack([synth.rand_field(1, 0x77 + k, n) for k in range(3)])
        xy, zero = parallel.msm_sharded_hip(pre, dev.to_device(sv[:, lo:hi]))
        for k in range(3):
            assert zero[k] == 0 and tuple(from_mont_arr(c.base, xy[k])) == closed_form_msm(0, sv[k], G, D), k
    finally:
        dist.destroy_process_group()


# ---- N > 1 over the HIP path: two ranks share the one GPU of the box (gloo, the ~200-byte payload staged through the host) ----
def _rank_hip(rank, world, port, n, batch, q):
    import torch
    import torch.distributed as dist
    from plonky_amd import device as dev, parallel
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev.init(0)
        c = br.TWEEDLEDEE
        G = (c.gx, c.gy)
        D = br.ec_mul(c, 4242, G)
        lo, hi = parallel.shard_bounds(n, rank, world)
        bases = dev.gen_bases_dev(0, hi - lo, _pt(c, G), _pt(c, D), first=lo)
        pre = dev.msm_precompute_dev(0, bases)
        sv = np.stack([synth.rand_field(1, 0x77 + k, n) for k in range(batch)])
        ex = parallel.PartialExchange(0, batch, "cuda")
        for _ in range(2):  # the buffers are reused from step to step
            xy, zero = parallel.msm_sharded_hip(pre, dev.to_device(np.ascontiguousarray(sv[:, lo:hi])), exchange=ex)
        part_xy, part_z = ex.partials()
        # the same three vectors through the batch plan: one whole vector per rank, the third sharded by base range
        plan = parallel.BatchPlan(batch, world, rank, n)
        bases_all = dev.gen_bases_dev(0, n, _pt(c, G), _pt(c, D))
        pre_all = dev.msm_precompute_dev(0, bases_all)
        ex2 = parallel.PartialExchange(0, batch, "cuda", whole_per_rank=plan.whole)
        dev.msm_execute_dev(pre_all, dev.to_device(plan.local_scalars(sv)), ex2.out_xy, ex2.out_zero)
        ex2.gather()
        hxy, hz = ex2.combine()
        q.put((rank, xy, zero, dev.to_host(part_xy), part_z.cpu().numpy(), dev.to_host(hxy), hz.cpu().numpy()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_sharded_msm_hip_world_size_2_one_gpu():
    """parallel.msm_sharded_hip at world size 2: both ranks run the HIP kernels on GPU 0 over their own base range, ONE
    all-gather of the packed records, plk_msm_combine_partials_dev; every rank must hold the closed form of the GLOBAL MSM."""
    pytest.importorskip("torch")
    import socket
    import torch.multiprocessing as mp
    n, batch, world = 1 << 14, 3, 2
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_hip, args=(r, world, port, n, batch, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    c = br.TWEEDLEDEE
    G = (c.gx, c.gy)
    D = br.ec_mul(c, 4242, G)
    sv = np.stack([synth.rand_field(1, 0x77 + k, n) for k in range(batch)])
    from plonky_amd import parallel
    for rank, xy, zero, part_xy, part_z, hxy, hz in res:
        assert part_xy.shape == (world, batch, 2, 4) and not part_z.any()
        for k in range(batch):
            assert zero[k] == 0 and tuple(from_mont_arr(c.base, xy[k])) == closed_form_msm(0, sv[k], G, D), (rank, k)
            # whole vectors handed over by their owner + the sharded remainder: the same global results
            assert int(hz[k]) == 0 and np.array_equal(hxy[k], xy[k]), (rank, k)
            for r in range(world):  # the gathered records hold every rank's partial result, in rank order
                lo, hi = parallel.shard_bounds(n, r, world)
                assert tuple(from_mont_arr(c.base, part_xy[r, k])) == closed_form_msm(0, sv[k, lo:hi], G, D, first=lo), (rank, k, r)


@pytest.mark.parametrize("flags", [["--workload", "commit9", "--log-n", "15"],
                                   ["--workload", "msm", "--shard", "--curve", "bls12_377", "--log-n", "14"],
                                   ["--workload", "both", "--log-n", "14"]])
def test_bench_spawns_its_ranks(flags):
    """`python bench.py --gpus 2` with no launcher and no WORLD_SIZE: spawns two ranks (here both on GPU 0), prints ONE JSON
    line, every self-check (per-rank closed form, device point sum against the host sum, global closed form) true."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--same-device", "--steps", "2", "--warmup", "1",
                          "--no-cpu-baseline"] + flags, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["checks"] and all(r["checks"].values()), r["checks"]
    assert r["scaling"] == ("weak" if "both" in flags else "strong")
    if r["scaling"] == "strong":
        assert r["checks"]["msm_global_sum_closed_form"]
    else:
        # the driver's default form at N > 1 also carries BASELINE's strong-scaling configurations with their one-GPU times
        # measured by rank 0 in the same run, and the number of ranks the communicator carried
        m = r["components"]["multi_gpu"]
        for case in ("commit9_strong", "bls12_377_2p22_shard"):
            assert m[case + "_ms"] > 0 and m[case + "_one_gpu_ms"] > 0 and 0 < m[case + "_efficiency"] < 1.5, m
            assert r["checks"][case + "_closed_form"] and r["checks"][case + "_one_gpu_closed_form"]
        assert m["gloo_ranks"] == 2 and m["backend"] == "gloo"


def test_bench_single_process_virtual_devices():
    """`python bench.py --gpus 2 --single-process --virtual-devices`: the in-library device group (plk_init_devices) through the
    host-pointer C ABI - nine commitments in one call, one sharded MSM, nine transforms - bit-identical to one device."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "PLK_VIRTUAL_DEVICES")}
    env["PLK_MULTI_MIN_LOG_N"] = "10"
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--single-process", "--virtual-devices", "--log-n", "16", "--steps", "8"],
                         env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    c = r["components"]
    assert r["n_gpus"] == 2 and c["devices"] == 2 and c["virtual"] and c["bit_identical_to_one_device"]
    for k in ("commit9_ms", "msm_single_ms", "ntt9_ms"):
        assert c["one_device"][k] > 0 and c["group"][k] > 0


def test_bench_single_process_eight_virtual_devices():
    """BASELINE configs 4 / 5 are quoted on 8 GPUs: `python bench.py --gpus 8 --single-process --virtual-devices` at 2^16 runs the plan an
    8-GPU node would run (one whole vector per device + eighths of the ninth, a single MSM in eighths, nine transforms over eight
    devices) - on a real node only the physical devices behind the eight logical ones differ."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "PLK_VIRTUAL_DEVICES")}
    env["PLK_MULTI_MIN_LOG_N"] = "10"
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--single-process", "--virtual-devices", "--log-n", "16", "--steps", "8"],
                         env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    c = r["components"]
    assert r["n_gpus"] == 8 and c["devices"] == 8 and c["virtual"] and c["bit_identical_to_one_device"] and c["msm_closed_form_bit_exact"] and c["ntt_roundtrip_bit_exact"]


def test_bench_single_process_child_form():
    """The same case the way the driver's N > 1 line runs it: bench.single_process_child (a child process with a time limit,
    launcher variables stripped), here on two virtual devices."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    saved = {k: os.environ.get(k) for k in ("RANK", "WORLD_SIZE", "PLK_MULTI_MIN_LOG_N")}
    os.environ.update({"RANK": "0", "WORLD_SIZE": "2", "PLK_MULTI_MIN_LOG_N": "10"})
    try:
        c = bench.single_process_child(2, 16, 8, extra=["--virtual-devices"])
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    assert "error" not in c, c
    assert c["exit_code"] == 0 and c["devices"] == 2 and c["virtual"] and c["bit_identical_to_one_device"] and c["msm_closed_form_bit_exact"], c


def test_bench_nccl_every_visible_gpu():
    """`python bench.py --gpus N` with N = every visible GPU over the nccl backend (RCCL): skips on a one-GPU box - the only
    place where RCCL carries more than one rank (the driver's 8-GPU node, a developer's multi-GPU box)."""
    torch = pytest.importorskip("torch")
    import json
    import subprocess
    import sys
    count = torch.cuda.device_count()
    if count < 2:
        pytest.skip("one GPU visible: RCCL with more than one rank needs a multi-GPU node")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(count), "--steps", "4", "--warmup", "2", "--no-cpu-baseline", "--log-n", "18"],
                         env=env, capture_output=True, text=True, timeout=1800)
    assert out.returncode == 0, out.stderr[-2000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    m = r["components"]["multi_gpu"]
    assert r["n_gpus"] == count and m["rccl_ranks"] == count and m["backend"] == "nccl" and all(r["checks"].values()), (m, r["checks"])
    assert m["single_process"].get("bit_identical_to_one_device"), m["single_process"]


def test_bench_emulated_rank():
    """--emulate-rank r/N: one rank's shard of the strong-scaling problem alone (the per-rank time behind DESIGN.md section 6)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "commit9", "--log-n", "16", "--emulate-rank", "3/8", "--steps", "2",
                          "--warmup", "1"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    e = r["emulated_rank"]
    # nine vectors on eight ranks: one whole vector and an eighth of the ninth per rank (parallel.BatchPlan)
    assert e["rank"] == 3 and e["whole_vectors"] == 1 and e["sharded_vectors"] == 1 and e["pairs_local"] == (1 << 16) + (1 << 16) // 8
    assert all(r["checks"].values())

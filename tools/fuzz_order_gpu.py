"""Randomised parity stress for the round-6 ordering / reduction paths (tools/fuzz_gpu.py draws mostly small sizes, which never select them):
2^16 .. ~2^17.2 generators at a forced window of 19 / 20 / 21 bits, random skews of the scalar vector (uniform, few distinct values, small
values, single windows, zero runs, mixtures), whole vectors, generator sub-ranges and bucket ranges, the projective result - every case
against the oracle's msm_execute_parallel, bit for bit.  Usage: python tools/fuzz_order_gpu.py [seconds]   (FUZZ_SEED, FUZZ_CASE as fuzz_gpu.py)"""
import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import plonky_amd as pa
from plonky_amd import synth, device as dev
from oracle import bigint_ref as br, oracle_lib as ol
from tests.test_oracle_kats import mont_arr

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
MASTER = int(os.environ.get("FUZZ_SEED", "6060"))
REPLAY = os.environ.get("FUZZ_CASE")
THREADS = min(64, os.cpu_count() or 1)
CURVES = [br.TWEEDLEDEE, br.TWEEDLEDUM, br.BLS12_377, br.PALLAS, br.VESTA]
dev.init(0)


def vector(rng, nprng, c, n):
    f = c.scalar
    rnd = synth.rand_field(f.field_id, rng.getrandbits(40), n)
    m = lambda vals: mont_arr(f, [int(v) % f.p for v in vals])
    kind = rng.choice(["uniform", "distinct", "small", "window", "zeros", "mix", "equal"])
    if kind == "uniform":
        return kind, rnd
    if kind == "equal":
        return kind, np.repeat(rnd[:1], n, axis=0)
    if kind == "distinct":
        k = rng.choice([2, 16, 300])
        return kind, rnd[:k][nprng.integers(0, k, size=n)]
    if kind == "small":
        bits = rng.choice([1, 8, 19, 20, 21, 40])
        return kind, m([rng.getrandbits(bits) for _ in range(n)])
    if kind == "window":
        sh = rng.choice([0, 20, 100, 220, 240])
        return kind, m([rng.getrandbits(rng.choice([5, 14, 20])) << sh for _ in range(n)])
    if kind == "zeros":
        v = rnd.copy()
        v[nprng.random(n) < rng.choice([0.5, 0.9, 0.999])] = 0
        return kind, v
    v = rnd.copy()
    a, b = sorted(rng.sample(range(n), 2))
    v[:a] = m([1])[0]
    v[a:b] = m([f.p - 1])[0]
    return kind, v


cases, t_end, case = 0, time.time() + budget, 0
while time.time() < t_end:
    seed = int(REPLAY, 0) if REPLAY else (MASTER << 20) + case
    case += 1
    rng, nprng = random.Random(seed), np.random.default_rng(seed & 0xFFFFFFFF)
    c = rng.choice(CURVES)
    n = rng.randint(1 << 16, 150000)
    window = rng.choice([19, 20, 20, 20, 21])
    G = (c.gx, c.gy)
    pt = lambda P: np.array([c.base.mont_limbs(P[0]), c.base.mont_limbs(P[1])], dtype=np.uint64)
    bases = ol.gen_bases(c.curve_id, n, pt(G), pt(br.ec_mul(c, rng.getrandbits(60) | 1, G)))
    if rng.random() < 0.3:
        bases[rng.randrange(n)] = bases[rng.randrange(n)]  # a duplicated generator
    kind, s = vector(rng, nprng, c, n)
    opre = ol.MsmPrecomputation(c.curve_id, bases, 13, threads=THREADS)
    exp, ez = opre.execute(s, parallel=True, threads=THREADS)
    db, ds = dev.to_device(bases.reshape(n, 2, -1)), dev.to_device(s)
    pre = dev.msm_precompute_dev(c.curve_id, db, device_window=window)
    tag = (hex(seed), c.name, n, window, kind)
    mode = rng.choice(["whole", "parts", "buckets", "projective", "batch"])
    if mode == "whole":
        oxy, oz = dev.msm_execute_dev(pre, ds)
        assert int(oz.cpu()[0]) == ez and np.array_equal(dev.to_host(oxy)[0].reshape(exp.shape), exp), tag
    elif mode == "batch":
        _, s2 = vector(rng, nprng, c, n)
        e2, z2 = opre.execute(s2, parallel=True, threads=THREADS)
        oxy, oz = dev.msm_execute_dev(pre, dev.to_device(np.stack([s, s2, s])))
        got, gz = dev.to_host(oxy), oz.cpu().numpy()
        assert [int(v) for v in gz] == [ez, z2, ez] and np.array_equal(got[0].reshape(exp.shape), exp) and np.array_equal(got[1].reshape(exp.shape), e2) and np.array_equal(got[2], got[0]), tag
    elif mode == "parts":
        a, b = sorted(rng.sample(range(n + 1), 2))
        v = np.zeros_like(s)
        v[a:b] = s[a:b]
        e2, z2 = opre.execute(v, parallel=True, threads=THREADS)
        oxy, oz = dev.msm_execute_parts_dev(pre, [(0, ds), (a, ds[a:b].contiguous())])
        got, gz = dev.to_host(oxy), oz.cpu().numpy()
        assert int(gz[0]) == ez and np.array_equal(got[0].reshape(exp.shape), exp) and int(gz[1]) == z2 and np.array_equal(got[1].reshape(exp.shape), e2), tag + (a, b)
    elif mode == "buckets":
        world = rng.choice([2, 3, 5, 8])
        pts, zs = [], []
        for r in range(world):
            oxy, oz = dev.msm_execute_parts_dev(pre, [(0, ds)], buckets=[(r, world)])
            pts.append(dev.to_host(oxy)[0]); zs.append(int(oz.cpu()[0]))
        tot, tz = pa.curve_sum_affine(c.curve_id, np.stack(pts), np.array(zs, dtype=np.uint8))
        assert tz == ez and np.array_equal(tot.reshape(exp.shape), exp), tag + (world,)
    else:
        oxyz, oz = dev.msm_execute_dev(pre, ds, projective=True)
        f = c.base
        xyz = dev.to_host(oxyz)[0]
        if ez:
            assert int(oz.cpu()[0]) == 1, tag
        else:
            x, y, z = (f.from_mont(synth.to_int(xyz[k])) for k in range(3))
            zi = pow(z, -1, f.p)
            assert int(oz.cpu()[0]) == 0 and (x * zi % f.p, y * zi % f.p) == (f.from_mont(synth.to_int(exp[0])), f.from_mont(synth.to_int(exp[1]))), tag
    pre.free()
    cases += 1
    if cases % 20 == 0:
        print("  %d cases, last %s" % (cases, tag), flush=True)
    if REPLAY:
        break
print("fuzz_order ok (FUZZ_SEED=%d, %d cases)" % (MASTER, cases))

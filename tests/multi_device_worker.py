"""Body of tests/test_gpu_multi.py, run in a process of its own (the device group is process-wide state).

    python tests/multi_device_worker.py <logical devices> <log_n> <out.npz>

With ONE physical GPU the logical devices are virtual (PLK_VIRTUAL_DEVICES): separate contexts, worker threads and streams on the
same GPU - the code path of a multi-GPU node.  With as many physical GPUs as logical devices they are real.
Everything goes through the HOST-POINTER entry points, i.e. what an untouched plonk.rs / poly_commit.rs reaches through the shim
of INTEGRATION.md: plk_msm_precompute, plk_msm_execute, plk_msm_execute_batch, plk_ntt_batch, plk_ntt_padded_batch, plk_ntt.
Results of the one-device library and of the device group are written side by side; the test compares them with each other
and with the oracle.
"""
import os
import sys
import threading

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def main():
    world, log_n, out_path = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    import torch
    if torch.cuda.device_count() < world:
        os.environ["PLK_VIRTUAL_DEVICES"] = str(world)
    os.environ.setdefault("PLK_MULTI_MIN_LOG_N", "8")
    import plonky_amd as pa
    from plonky_amd import api, device as dev, lib, synth
    from plonky_amd.selfcheck import _mul
    from plonky_amd.synth import MODULI
    L = lib.load()
    n = 1 << log_n
    curve = 0
    p = MODULI[0]
    G = (p - 1, 2)
    D = _mul(p, 0x5EED5EED1234567, G)
    g0 = np.stack([synth.mont(0, G[0]), synth.mont(0, G[1])])
    dd = np.stack([synth.mont(0, D[0]), synth.mont(0, D[1])])
    res = {}

    def run(tag):
        bases = dev.to_host(dev.gen_bases_dev(curve, n, g0, dd)).reshape(n, 2, 4)  # generators G + i D, as host data
        if tag == "one":
            res["bases"] = bases
        # nine scalar vectors: uniform ones, a sparse one (Z = 1 style), edge scalars 0 / 1 / r - 1, an all-zero one
        vecs = np.stack([synth.rand_field(1, 0x350920 + v, n) for v in range(9)])
        vecs[3, n // 3:] = 0
        vecs[4, :] = 0
        vecs[5, 0] = synth.mont(1, MODULI[1] - 1)
        vecs[5, 1] = synth.mont(1, 1)
        vecs[5, 2] = 0
        res["vecs"] = vecs
        pre = pa.msm_precompute(curve, bases, 11)
        for batch in (9, 8, 3, 2, 1):  # whole + sharded, whole only (when the world divides it), fewer vectors than devices, one
            xy, z = pa.msm_execute_batch(pre, vecs[:batch])
            res["%s_batch%d_xy" % (tag, batch)] = xy
            res["%s_batch%d_z" % (tag, batch)] = z
        xy, z = pa.msm_execute_parallel(pre, vecs[7])  # a single MSM: sharded by base range
        res[tag + "_single_xy"], res[tag + "_single_z"] = xy, np.array([z], dtype=np.uint8)
        # the same context from nine host threads at once (each a single MSM)
        outs = [None] * 9

        def one(t):
            outs[t] = pa.msm_execute_parallel(pre, vecs[t])
        th = [threading.Thread(target=one, args=(t,)) for t in range(9)]
        [t.start() for t in th]
        [t.join() for t in th]
        res[tag + "_threads_xy"] = np.stack([o[0] for o in outs])
        res[tag + "_threads_z"] = np.array([o[1] for o in outs], dtype=np.uint8)
        # device-resident vectors against the same context (they travel peer to peer)
        dxy, dz = dev.msm_execute_dev(pre, dev.to_device(vecs[:5]))
        torch.cuda.synchronize()
        res[tag + "_dev_xy"] = dev.to_host(dxy).reshape(5, 2, 4)
        res[tag + "_dev_z"] = dz.cpu().numpy()
        pre.free()
        # transforms: a batch dealt out over the devices, padded transforms, single transforms from many threads
        polys = np.stack([synth.rand_field(0, 0xF70020 + v, n) for v in range(9)])
        res["polys"] = polys
        res[tag + "_ntt"] = api.fft_batch(0, polys)
        res[tag + "_intt"] = api.fft_batch(0, polys[:4], inverse=True)
        pre_f = pa.fft_precompute(0, n)
        short = [polys[v][: n // 8 - v] for v in range(9)]
        res[tag + "_lde"] = pa.polynomials_to_values_padded(short, pre_f)
        outs = [None] * 9

        def one_ntt(t):
            outs[t] = pa.fft_with_precomputation_power_of_2(polys[t], pre_f)
        th = [threading.Thread(target=one_ntt, args=(t,)) for t in range(9)]
        [t.start() for t in th]
        [t.join() for t in th]
        res[tag + "_ntt_threads"] = np.stack(outs)

    def hip_device(set_to=-1):
        """The calling thread's current HIP device, from the runtime the library itself is linked to (torch keeps its own idea of it)."""
        d = int(L.plk_thread_hip_device(set_to))
        assert d >= 0, d
        return d

    def copy_stats():
        import ctypes
        a, b = ctypes.c_ulonglong(0), ctypes.c_ulonglong(0)
        lib.check(L.plk_group_copy_stats(ctypes.byref(a), ctypes.byref(b)))
        return np.array([a.value, b.value], dtype=np.uint64)

    if len(sys.argv) > 4 and sys.argv[4] == "stress":
        # 200 fan-out calls with the workers arriving out of step (PLK_TEST_WORKER_JITTER_US, set by the test): single MSMs sharded by
        # base range, device-resident vectors, every tenth call a hybrid batch - every result must equal the first of its kind
        got = pa.init_devices(world)
        assert got == world
        bases = dev.to_host(dev.gen_bases_dev(curve, n, g0, dd)).reshape(n, 2, 4)
        vecs = np.stack([synth.rand_field(1, 0x350A20 + v, n) for v in range(world + 1)])
        pre = pa.msm_precompute(curve, bases, 11)
        first_single = pa.msm_execute_parallel(pre, vecs[0])
        first_batch = pa.msm_execute_batch(pre, vecs)
        dvec = dev.to_device(vecs[:1])
        bad = 0
        for it in range(200):
            if it % 10 == 9:
                xy, z = pa.msm_execute_batch(pre, vecs)
                bad += int(not (np.array_equal(xy, first_batch[0]) and np.array_equal(z, first_batch[1])))
            elif it % 10 == 4:
                dxy, dz = dev.msm_execute_dev(pre, dvec)
                torch.cuda.synchronize()
                bad += int(not (np.array_equal(dev.to_host(dxy).reshape(2, 4), first_single[0]) and int(dz.cpu()[0]) == first_single[1]))
            else:
                xy, z = pa.msm_execute_parallel(pre, vecs[0])
                bad += int(not (np.array_equal(xy, first_single[0]) and z == first_single[1]))
        assert np.array_equal(first_batch[0][0], first_single[0])
        pre.free()
        res.update(bad=np.array([bad]), copy_stats=copy_stats(), world=np.array([got, torch.cuda.device_count()]))
        L.plk_shutdown()
        np.savez(out_path, **res)
        print("multi_device_worker stress: %d mismatches in 200 calls, copies peer / staged %s" % (bad, res["copy_stats"]))
        return
    lib.check(L.plk_init(0))
    run("one")
    L.plk_shutdown()
    got = pa.init_devices(world)
    assert got == world, (got, world)
    res["world"] = np.array([got, torch.cuda.device_count()])
    run("multi")
    # A host shim (or torch) allocates on "the current device" between two calls: every public call must leave the calling thread's
    # HIP device as it found it.  The thread parks on the LAST visible GPU - with several physical GPUs that is not
    # the device the library's logical device 0 lives on - and makes host-pointer calls of every fan-out form (numpy only: torch
    # would move the current device itself).
    park = torch.cuda.device_count() - 1
    seen = [hip_device(park)]
    small = res["bases"][: 1 << 10]
    pre2 = pa.msm_precompute(curve, small, 11)
    seen.append(hip_device())
    pa.msm_execute_batch(pre2, res["vecs"][:3, : 1 << 10])
    seen.append(hip_device())
    pa.msm_execute_parallel(pre2, res["vecs"][0, : 1 << 10])
    seen.append(hip_device())
    api.fft_batch(0, res["polys"][:3])
    seen.append(hip_device())
    pa.fft_with_precomputation_power_of_2(res["polys"][0], pa.fft_precompute(0, n))
    seen.append(hip_device())
    pre2.free()
    seen.append(hip_device())
    res["hip_device"] = np.array([park] + seen)
    res["copy_stats"] = copy_stats()
    L.plk_shutdown()
    np.savez(out_path, **res)
    print("multi_device_worker: ok, %d logical devices on %d GPU(s), n = 2^%d" % (world, torch.cuda.device_count(), log_n))


if __name__ == "__main__":
    main()

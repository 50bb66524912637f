"""GPU parity: independent calls on DIFFERENT streams at the same time.

plonk.rs runs its transforms and commitments as independent Rayon tasks; behind the C ABI that is two (or more) HIP streams in flight at
once (`bench.py` reports the step that way as `components.step_two_streams_ms`).  Everything a call borrows from the library - scratch
buffers, twiddle tables, an MSM context's workspace - must therefore belong to the stream (or the context) that uses it.  Here a
forward NTT, an inverse NTT and an MSM run concurrently on three streams, many times over, and every result must equal the one the same
call gives alone - and the oracle's."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from plonky_amd import synth
from oracle import bigint_ref as br
from oracle import oracle_lib as ol


def test_ntt_intt_and_msm_on_three_streams_at_once():
    torch = pytest.importorskip("torch")
    from plonky_amd import device as dev
    dev.init(0)
    c = br.TWEEDLEDEE
    log_n, n_msm = 16, 1 << 15
    n = 1 << log_n
    x_host = synth.rand_field(0, 0x57EA01, n)
    x = dev.to_device(x_host)
    G = (c.gx, c.gy)
    pt = lambda P: np.array([c.base.mont_limbs(P[0]), c.base.mont_limbs(P[1])], dtype=np.uint64)
    g0, dd = pt(G), pt(br.ec_mul(c, 0x57EA02, G))
    bases = dev.gen_bases_dev(0, n_msm, g0, dd)
    s_host = synth.rand_field(1, 0x57EA03, n_msm)
    s = dev.to_device(s_host)
    pre = dev.msm_precompute_dev(0, bases)
    # alone, one after the other
    y_ref = dev.ntt_dev(0, x)
    z_ref = dev.ntt_dev(0, y_ref, inverse=True)
    oxy_ref, oz_ref = dev.msm_execute_dev(pre, s)
    torch.cuda.synchronize()
    assert np.array_equal(dev.to_host(z_ref), x_host)
    assert np.array_equal(dev.to_host(y_ref), ol.FftPrecomputation(0, n).fft_with_precomputation_power_of_2(x_host))
    exp, ez = ol.MsmPrecomputation(0, ol.gen_bases(0, n_msm, g0, dd), 8, threads=8).execute(s_host, parallel=True, threads=8)
    assert int(oz_ref.cpu()[0]) == ez and np.array_equal(dev.to_host(oxy_ref).reshape(2, 4), exp)
    # at the same time, 40 rounds without a synchronisation in between
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    ys = [torch.empty_like(x) for _ in range(2)]
    zs = [torch.empty_like(x) for _ in range(2)]
    oxy = [torch.empty_like(oxy_ref) for _ in range(2)]
    oz = [torch.empty_like(oz_ref) for _ in range(2)]
    main = torch.cuda.current_stream()
    s1.wait_stream(main)
    s2.wait_stream(main)
    for k in range(40):
        with torch.cuda.stream(s1):
            dev.ntt_dev(0, x, out=ys[k & 1])
        with torch.cuda.stream(s2):
            dev.ntt_dev(0, y_ref, inverse=True, out=zs[k & 1])
        dev.msm_execute_dev(pre, s, oxy[k & 1], oz[k & 1])
    torch.cuda.synchronize()
    for k in range(2):
        assert torch.equal(ys[k], y_ref), "forward NTT beside other streams"
        assert torch.equal(zs[k], z_ref), "inverse NTT beside other streams"
        assert torch.equal(oxy[k], oxy_ref) and torch.equal(oz[k], oz_ref), "MSM beside other streams"
    pre.free()

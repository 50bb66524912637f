"""The hot-path calls of one Plonk proof (src/plonk.rs:85-200) strung together on the device, data resident in HBM
between them: 9 wire iNTTs (values_to_polynomials), 9 LDEs to 8n (polynomials_to_values_padded), 9 blinded commitments
(commit_polynomials), Z: one iNTT + LDE + commitment, the vanishing polynomial (vanishing_poly: the 8n-point constraint
evaluation + one 8n iNTT), the quotient t = vanishing / Z_H (divide_by_z_h) and its 7 chunk commitments.
The witness is an honest one (ArithmeticGate / ConstantGate rows with random selector constants, identity wiring, hence Z = 1), so
the numerator is a REAL vanishing polynomial and the division by Z_H is checked to be exact (q * Z_H == vanishing).
Witness generation and the transcript are not part of this probe.  With `ipa` the rounds of the opening's inner-product argument
(halo.rs:63-124) follow, over the SAME tables - built over the circuit's fixed generators [pedersen_g, pedersen_h, U] (plonk.rs:46-51)
- with stand-in challenges.  Usage (GPU box): python tools/prover_pipeline_probe.py [log_n] [ipa]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from plonky_amd import api, device as dev, synth
from plonky_amd.selfcheck import _mul
from plonky_amd.synth import MODULI

log_n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1] != "ipa" else 20
WITH_IPA = "ipa" in sys.argv
n = 1 << log_n
F, CURVE = 1, 0   # wires live in the scalar field of Tweedledee = TweedledumBase
dev.init(0)
p = MODULI[0]
G = (p - 1, 2)
D = _mul(p, 0xC0FFEE, G)
g0 = np.stack([synth.mont(0, G[0]), synth.mont(0, G[1])]); dd = np.stack([synth.mont(0, D[0]), synth.mont(0, D[1])])
gens = dev.gen_bases_dev(CURVE, n + 2, g0, dd)            # pedersen_g (n) followed by pedersen_h and U
t0 = time.perf_counter(); pre = dev.msm_precompute_dev(CURVE, gens); torch.cuda.synchronize()
print("setup: msm_precompute of %d generators %.1f ms (once per circuit)" % (n + 2, (time.perf_counter() - t0) * 1e3))

# ---- an honest witness: ArithmeticGate rows (arithmetic.rs:32-46): w3 = c0 w0 w1 + c1 w2, selector constants 1001 c0 c1 ----
ONE, ZERO = synth.mont(F, 1), synth.mont(F, 0)
w = synth.rand_field(F, 1, 9 * n).reshape(9, n, 4)
c0, c1 = synth.rand_field(F, 7, n), synth.rand_field(F, 8, n)
w[3] = api.field_op(F, "add", api.field_op(F, "mul", api.field_op(F, "mul", c0, w[0]), w[1]), api.field_op(F, "mul", c1, w[2]))
consts = np.stack([np.tile(ONE, (n, 1)), np.tile(ZERO, (n, 1)), np.tile(ZERO, (n, 1)), np.tile(ONE, (n, 1)), c0, c1])
# odd rows: ConstantGate (constant.rs:30-39), selector constants 10110 c, w0 = c -- so that the selector polynomials are not constant
consts[2, 1::2], consts[3, 1::2], consts[4, 1::2] = ONE, ONE, ZERO
w[0, 1::2] = c1[1::2]
k_is = synth.rand_field(F, 9, 6)                                                   # get_subgroup_shift(0..5): inputs
alpha, beta, gamma = synth.rand_field(F, 10, 3)
ZETA = np.array([7605997034305223424, 3132214451552427455, 3308921103222877309, 2709928666517121162], dtype=np.uint64)  # tweedledum_curve.rs:37-44
# circuit-constant 8n tables (circuit_builder.rs:1135-1160): constants, S_sigma_j = k_j X (identity wiring)
t0 = time.perf_counter()
const_coeffs = dev.ntt_dev(F, dev.to_device(consts), inverse=True)
consts_8n = dev.ntt_padded_dev(F, const_coeffs, log_n + 3)
sig = np.zeros((6, 2, 4), dtype=np.uint64); sig[:, 1] = k_is
sigma_8n = dev.ntt_padded_dev(F, dev.to_device(sig), log_n + 3)
torch.cuda.synchronize()
print("setup: constants_8n / s_sigma_values_8n %.1f ms (once per circuit)" % ((time.perf_counter() - t0) * 1e3))
wires = dev.to_device(w)
blind = torch.cat([dev.to_device(synth.rand_field(F, 2, 9 + 1 + 7)).reshape(-1, 1, 4), torch.zeros((17, 1, 4), dtype=torch.int64, device="cuda")], dim=1)  # [r] H + [0] U
zvals = dev.to_device(np.tile(ONE, (n, 1)))                                         # identity wiring: Z = 1
t_out = torch.empty((8 * n, 4), dtype=torch.int64, device="cuda")
ev8 = torch.empty((9, 8 * n, 4), dtype=torch.int64, device="cuda")
pts = torch.empty((8 * n, 4), dtype=torch.int64, device="cuda")
t7 = torch.zeros((7 * n, 4), dtype=torch.int64, device="cuda")


def run():
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(9)]
    marks[0].record()
    polys = dev.ntt_dev(F, wires, inverse=True)                                     # values_to_polynomials
    marks[1].record()
    dev.ntt_padded_dev(F, polys, log_n + 3, out=ev8)                                # polynomials_to_values_padded
    marks[2].record()
    c_wires = dev.msm_execute_dev(pre, torch.cat([polys, blind[:9]], dim=1).contiguous())   # commit_polynomials (+ [r] H)
    marks[3].record()
    zpoly = dev.ntt_dev(F, zvals, inverse=True)
    z_8n = dev.ntt_padded_dev(F, zpoly, log_n + 3)                                  # plonk.rs:388-391
    c_z = dev.msm_execute_dev(pre, torch.cat([zpoly, blind[9]], dim=0).contiguous())
    marks[4].record()
    dev.vanishing_points_dev(F, log_n, consts_8n, ev8, sigma_8n, z_8n, k_is, alpha, beta, gamma, ZETA, ZERO, out=pts)   # plonk.rs:392-453
    marks[5].record()
    vanishing = dev.ntt_dev(F, pts, inverse=True)                                   # Polynomial::from_evaluations, plonk.rs:455
    t = dev.divide_by_z_h_dev(F, vanishing, n, out=t_out)                           # quotient, plonk.rs:178-181
    marks[6].record()
    t7.zero_()
    t7[: min(t.shape[0], 7 * n)] = t[: 7 * n]                                      # plonk.rs:182: pad to 7n, split
    chunks = t7.reshape(7, n, 4)
    c_t = dev.msm_execute_dev(pre, torch.cat([chunks, blind[10:17]], dim=1).contiguous())
    marks[7].record()
    torch.cuda.synchronize()
    names = ["9 wire iNTT (n)", "9 LDE n -> 8n", "9 wire commitments", "Z: iNTT + LDE + commitment", "vanishing points (8n, 10 gates)",
             "iNTT (8n) + divide_by_z_h", "7 quotient-chunk commitments"]
    ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(7)]
    return names, ms, t, vanishing


run()
names, ms, t, vanishing = run()
q = dev.to_host(t7)
v = dev.to_host(vanishing)
zpad = np.zeros((n, 4), dtype=np.uint64)
back = api.field_op(F, "sub", np.concatenate([zpad, q]), np.concatenate([q, zpad]))   # q * (X^n - 1)
ok = bool(v.any() and not dev.to_host(t)[7 * n:].any() and np.array_equal(back, v))
for nm, tv in zip(names, ms):
    print("  %-34s %8.3f ms" % (nm, tv))
print("hot-path device time per proof at n = 2^%d: %.2f ms   (real numerator; q * Z_H == vanishing polynomial, deg q < 7n: %s)" % (log_n, sum(ms), ok))
if WITH_IPA:
    r = MODULI[F]
    m1 = lambda v: np.array(synth.mont(F, v), dtype=np.uint64)
    x_int = 0x1F3D5B79A2C4E6081F3D5B79A2C4E6081F3D5B79A2C4E608 % r          # halo_n(u_scaling bits): u_prime = [x] U
    k_u, k_h = (1 + (n + 1) * 0xC0FFEE) % r, (1 + n * 0xC0FFEE) % r            # generator i is G0 + i D = [1 + i 0xC0FFEE] G0
    pm = lambda P: np.stack([synth.mont(0, P[0]), synth.mont(0, P[1])])
    Hm, Upm = pm(_mul(p, k_h, G)), pm(_mul(p, x_int * k_u % r, G))
    us = [synth.to_int(row) % r or 1 for row in synth.rand_field(F, 31, log_n)]
    ums = [(m1(u), m1(pow(u, -1, r))) for u in us]
    bl = [(m1(100 + j), m1(200 + j)) for j in range(log_n)]
    polys = dev.ntt_dev(F, wires, inverse=True)
    halo_a, halo_b = polys[0].contiguous(), dev.to_device(synth.rand_field(F, 32, n))
    g_only = gens[:n].contiguous()
    def ipa():
        t0 = time.perf_counter()
        arg = dev.HaloArgument(CURVE, halo_a, halo_b, g_only, Hm, Upm, tables=pre, h_index=n, u_index=n + 1, u_prime_scalar=m1(x_int))
        for j in range(log_n):
            t1 = time.perf_counter()
            arg.round_lr(*bl[j])
            t2 = time.perf_counter()
            arg.round_fold(*ums[j])
            if os.environ.get("PROBE_ROUNDS"):
                torch.cuda.synchronize()
                print("    round at length %8d: L/R %.3f ms  fold %.3f ms" % (len(arg) * 2, (t2 - t1) * 1e3, (time.perf_counter() - t2) * 1e3))
        fin = arg.read()
        t = time.perf_counter() - t0
        arg.free()
        return t, fin
    ipa()
    t_ipa, _ = ipa()
    print("  %-34s %8.3f ms   (%d rounds over the same tables; wall clock of the loop, every round ends on the host)" % ("inner-product argument", t_ipa * 1e3, log_n))
    print("hot path of one proof incl. the opening: %.2f ms" % (sum(ms) + t_ipa * 1e3))

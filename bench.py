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
          crossover  the drop-in HOST-pointer entry points against the CPU path (the oracle) as a size sweep 2^8..2^log_n: where the
                  GPU path starts to pay (the size gate PLK_MIN_GPU_LOG_N); table on stderr, one JSON line on stdout.
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
code    this file is the command line; the work is in benchlib/: headline.py (the timed region, checks, rooflines, the JSON line),
        components.py (component loops, host-pointer entry points), multi.py (N > 1: strong-scaling cases, spawned ranks, the
        single-process device group), quotient.py (--workload quotient), cpu.py (the oracle as CPU baseline), common.py (constants,
        GPU identity, the in-process ceilings of plk_bench_ceilings and the roofline entry).
"""
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from benchlib.common import arith_source_hash, gpu_identity, kernel_source_hash, latest_profile, parse_args  # noqa: E402,F401  (tools/ import these from here)
from benchlib.multi import single_process_child  # noqa: E402,F401  (tests/ call it)


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if args.single_process:
        from benchlib.multi import run_single_process
        return run_single_process(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        from benchlib.multi import spawn_ranks
        return spawn_ranks(args, argv)
    if args.workload == "crossover":
        from benchlib.cpu import crossover_sweep
        return crossover_sweep(args)
    from benchlib.headline import run
    return run(args)


if __name__ == "__main__":
    main()

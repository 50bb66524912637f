"""benchlib.common -- constants, the physical GPU's identity, source hashes, the in-process ceilings and the roofline entry shared by
every workload of bench.py."""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
BENCH_PY = os.path.join(ROOT, "bench.py")
METRIC = "MSM Mpairs/sec + NTT Melems/sec, 2^20 Tweedledee, 1/2/4/8 GPU"

LOG_N = 20
SEED_NTT = 0xF70020
SEED_MSM = 0x350020
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8 TB/s spec
NOMINAL_GHZ = 2.4           # MI355X_MICROARCH.md: peak engine clock
MADS_PER_MODMUL = {4: 126, 6: 294}   # v_mad_u64_u32 per fz_mul: 9 limbs 81 + 45, 14 limbs 196 + 98 (fz.cuh)
CURVES = {"tweedledee": dict(curve=0, ntt_field=0, scalar_field=1, base_field=0, limbs=4, scalar_bits=255, pair_bytes=96),
          "bls12_377": dict(curve=2, ntt_field=2, scalar_field=2, base_field=3, limbs=6, scalar_bits=253, pair_bytes=128)}
STAGES = ["order_count", "order_scatter", "order_buckets", "accumulate", "assemble_lines", "planes", "final"]


def latest_profile(suffix):
    """profiles/rNN_<suffix> of the highest round present (the files are named per round)."""
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_" + suffix)))
    return found[-1] if found else os.path.join(ROOT, "profiles", "r00_" + suffix)


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


def measure_ceilings(L):
    """plk_bench_ceilings on the GPU of THIS process, now (~60 ms): {"mad_u64_u32_glaneops": raw v_mad_u64_u32 issue rate at 8 waves per
    SIMD, "fz_mul_gops": {curve: this build's product at 4 waves per SIMD}, "lazy_madd_gops": {curve: the accumulation's mixed addition
    with no memory at the kernel's occupancy}}, all in G operations per second over the whole GPU."""
    import ctypes
    from plonky_amd import lib
    out = (ctypes.c_double * 5)()
    lib.check(L.plk_bench_ceilings(out, 5))
    return {"mad_u64_u32_glaneops": out[0], "fz_mul_gops": {"tweedledee": out[1], "bls12_377": out[2]},
            "lazy_madd_gops": {"tweedledee": out[3], "bls12_377": out[4]}}


def rocprof_kernel_avgs():
    """{kernel: {"avg_us", "source"}} from the newest committed profiles/rNN_rocprofv3_kernel_stats.txt - another run, under the profiler
    (lower clocks), usually another lease: printed beside the live event time so that a reader can reproduce `frac` from profiles/."""
    path = latest_profile("rocprofv3_kernel_stats.txt")
    out = {}
    try:
        with open(path) as fh:
            for line in fh:
                t = line.split()
                if len(t) >= 5 and t[0].startswith("k_"):
                    try:
                        out[t[0].split("<")[0]] = {"avg_us": float(t[3]), "source": "profiles/" + os.path.basename(path)}
                    except ValueError:
                        pass
    except OSError:
        pass
    return out


class Ceilings:
    """The integer-ALU rooflines of one bench process: measured once, on the bench GPU, by the library itself.
    roofline.frac is HARDWARE-referenced: algorithmic modular multiplications per second over the measured v_mad_u64_u32 issue
    rate / the multiplier instructions a product cannot go below (126 for nine 29-bit limbs, 294 for fourteen).  Beside it:
    frac_nominal (the same at 16 lanes per clock per SIMD and the guide's 2.4 GHz), frac_own (this build's fz_mul at its best,
    what rounds 1-4 reported as `frac`), executed_frac (the multiplications the kernel really executes)."""

    def __init__(self, L, gpu, limbs, curve_name):
        self.raw = measure_ceilings(L)
        self.mads = MADS_PER_MODMUL[limbs]
        self.simds = 4 * int(gpu.get("cus") or 256)
        self.mad_peak = self.raw["mad_u64_u32_glaneops"]
        self.mad_nominal = 16.0 * self.simds * NOMINAL_GHZ
        self.own = self.raw["fz_mul_gops"][curve_name]
        self.lazy_madd = self.raw["lazy_madd_gops"][curve_name]
        self.gpu_uuid = gpu.get("uuid")
        self.prof = rocprof_kernel_avgs()

    def entry(self, kernel, gmm, executed_gmm, launch_ms, extra):
        peak = self.mad_peak / self.mads
        e = {"kernel": kernel, "bound": "valu", "achieved": gmm, "peak": peak, "unit": "G modmul/s", "frac": gmm / peak,
             "peak_source": "plk_bench_ceilings in this process on gpu %s: v_mad_u64_u32 at %.2f T lane-ops/s / %d multiplier instructions per product"
                            % (self.gpu_uuid, self.mad_peak / 1e3, self.mads),
             "executed_frac": executed_gmm / peak,
             "peak_nominal": self.mad_nominal / self.mads, "frac_nominal": gmm / (self.mad_nominal / self.mads),
             "peak_nominal_source": "16 lanes/clk/SIMD x %d SIMDs x %.1f GHz (MI355X_MICROARCH.md) / %d" % (self.simds, NOMINAL_GHZ, self.mads),
             "peak_own": self.own, "frac_own": gmm / self.own, "peak_own_source": "this build's fz_mul at 4 waves per SIMD, same call",
             "launch_ms": launch_ms, "launch_ms_source": "HIP events on the launch stream, this run"}
        pa = self.prof.get(kernel)
        if pa:
            e["rocprof_avg_ms"] = pa["avg_us"] / 1e3
            e["rocprof_source"] = pa["source"]
            e["frac_at_rocprof_avg"] = e["frac"] * launch_ms / e["rocprof_avg_ms"] if launch_ms else None
        e.update(extra)
        return e


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



def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", choices=["both", "ntt", "msm", "commit9", "quotient", "crossover"], default="both")
    ap.add_argument("--curve", choices=sorted(CURVES), default="tweedledee")
    ap.add_argument("--log-n", type=int, default=LOG_N)
    ap.add_argument("--shard", action="store_true", help="--workload msm: strong scaling - ONE 2^log_n MSM, generators sharded by base range")
    ap.add_argument("--emulate-rank", default=None, metavar="r/N", help="run rank r's shard of the N-rank strong-scaling problem alone on one GPU")
    ap.add_argument("--same-device", action="store_true", help="all ranks on GPU 0, gloo backend (world-size-2 test on a one-GPU box)")
    ap.add_argument("--no-parts", action="store_true", help="strong scaling: a rank's share of a sharded vector as a zero-padded full-length vector (round-3 start) instead of its base range")
    ap.add_argument("--bucket-shard", action="store_true", help="strong scaling: a sharded vector is shared by BUCKET range (every rank reads the whole vector and keeps its N-th of the coarse bins) instead of by base range")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--timed-only", action="store_true", help="run only the warm-up + timed region (for rocprofv3 --pmc passes)")
    ap.add_argument("--single-process", action="store_true",
                    help="N GPUs from ONE process through the host-pointer C ABI (plk_init_devices: what an untouched plonk.rs gets): commit9, one sharded MSM and "
                         "a transform batch over --gpus N devices against the same calls on one device")
    ap.add_argument("--virtual-devices", action="store_true", help="--single-process on a box with fewer GPUs: PLK_VIRTUAL_DEVICES logical devices on GPU 0")
    return ap.parse_args(argv)

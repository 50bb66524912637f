#!/usr/bin/env python3
"""Builds and runs tools/lab/field_ceilings.hip on the GPU box and writes the ceilings bench.py reads:
gpurun_out/<round>_field_op_costs.txt (the raw lines) and .json ({"fz_mul_gops": {curve: G modmul/s at 4 waves per SIMD},
"mad_u64_u32_glaneops": raw issue rate at 8 waves per SIMD, "arith_source_sha": hash of the arithmetic headers}).
Copy both into profiles/ (bench.py reads the newest round's file and ignores it when the headers have changed since).  The file
records which physical GPU it was measured on (uuid / rocm-smi unique id): a ceiling belongs to a box.
Usage: python tools/measure_ceilings.py [out_dir] [round tag, default r04]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import arith_source_hash  # noqa: E402

out_dir = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out")
RN = sys.argv[2] if len(sys.argv) > 2 else "r04"
os.makedirs(out_dir, exist_ok=True)
exe = os.path.join(ROOT, "build", "field_ceilings")
os.makedirs(os.path.dirname(exe), exist_ok=True)
if not os.path.exists(exe) or os.path.getmtime(exe) < os.path.getmtime(os.path.join(ROOT, "tools", "lab", "field_ceilings.hip")):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "plonky_amd", "csrc"),
                           "-o", exe, os.path.join(ROOT, "tools", "lab", "field_ceilings.hip")])
txt = subprocess.check_output([exe], text=True)
open(os.path.join(out_dir, RN + "_field_op_costs.txt"), "w").write(txt)
res = {"arith_source_sha": arith_source_hash(), "fz_mul_gops": {}, "fz_sqr_gops": {}, "lazy_madd_gops": {}, "all": []}
for line in txt.splitlines():
    kv = dict(t.split("=") for t in line.split()[1:])
    if line.startswith("OP"):
        res["all"].append(kv)
        if kv["waves_per_simd"] == "4":
            key = {"fz_mul": "fz_mul_gops", "fz_sqr": "fz_sqr_gops", "lazy_madd": "lazy_madd_gops"}.get(kv["op"])
            if key:
                res[key][kv["field"]] = float(kv["gops"])
    elif line.startswith("MAD") and kv["waves_per_simd"] == "8":
        res["mad_u64_u32_glaneops"] = float(kv["glaneops"])
    elif line.startswith("INFO"):
        res["max_clock_khz"] = int(kv["max_clock_khz"])
res["source"] = "tools/lab/field_ceilings.hip: Gop/s over the whole GPU at 4 waves per SIMD; v_mad_u64_u32 lane-ops/s at 8 waves per SIMD"
try:
    import torch
    from bench import gpu_identity
    ident = gpu_identity(torch, 0)
    res["gpu_uuid"], res["gpu"] = ident.get("uuid"), ident
except Exception as e:  # noqa: BLE001
    res["gpu_uuid"] = None
json.dump(res, open(os.path.join(out_dir, RN + "_field_op_costs.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if k != "all"}, indent=1))

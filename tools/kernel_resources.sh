#!/bin/bash
# kernel_resources.sh <object or .so> -- VGPR / AGPR / SGPR / scratch / LDS per kernel from the code object's metadata
set -e
f=$(readlink -f "$1"); d=$(mktemp -d); cd "$d"
/opt/rocm/lib/llvm/bin/llvm-objcopy --dump-section=.hip_fatbin=fat.bin "$f" 2>/dev/null || cp "$f" fat.bin
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --input=fat.bin --list > targets.txt 2>/dev/null || true
t=$(grep gfx950 targets.txt | head -1)
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --input=fat.bin --targets="$t" --output=k.co --unbundle
/opt/rocm/lib/llvm/bin/llvm-readelf --notes k.co | python3 -c '
import sys,re
cur={}
rows=[]
for line in sys.stdin:
    m=re.match(r"\s+\.?-?\s*\.(\w+):\s+(.*)",line)
    if not m: continue
    k,v=m.group(1),m.group(2).strip()
    if k=="name" and v.startswith("_Z") or (k=="name" and v.startswith("k_")):
        cur["name"]=v
    if k in("vgpr_count","agpr_count","sgpr_count","private_segment_fixed_size","group_segment_fixed_size","vgpr_spill_count"): cur[k]=v
    if k=="wavefront_size":
        rows.append(cur); cur={}
import subprocess
for r in rows:
    n=r.get("name","?")
    try: n=subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt",n],capture_output=True,text=True).stdout.strip().split("(")[0]
    except Exception: pass
    print("%-60s vgpr %4s agpr %4s sgpr %4s scratch %5s lds %6s spill %s"%(n[:60],r.get("vgpr_count"),r.get("agpr_count"),r.get("sgpr_count"),r.get("private_segment_fixed_size"),r.get("group_segment_fixed_size"),r.get("vgpr_spill_count")))
'
rm -rf "$d"

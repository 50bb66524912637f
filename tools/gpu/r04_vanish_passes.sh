#!/bin/bash
# per-launch durations of the quotient numerator's five passes (the kernel stats merge the template instantiations)
O=gpurun_out/r4j; mkdir -p $O
export TMPDIR=/tmp; R=$PWD; cd /tmp; rm -rf $R/$O/prof
rocprofv3 --kernel-trace -d $R/$O/prof -o q -- python $R/bench.py --workload quotient --steps 4 --warmup 1 > /dev/null 2>&1
cd $R
python - <<'PY'
import sqlite3, glob, re, collections
db = glob.glob("gpurun_out/r4j/prof/**/*.db", recursive=True)[0]
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [x for x in tabs if x.startswith("rocpd_kernel_dispatch")][0]
ks = [x for x in tabs if x.startswith("rocpd_info_kernel_symbol")][0]
agg = collections.defaultdict(list)
for name, st, en, vg, acc, scr in c.execute(f"select s.kernel_name, d.start, d.end, s.arch_vgpr_count, s.accum_vgpr_count, d.private_segment_size from {kd} d join {ks} s on d.kernel_id = s.id"):
    if "k_vanishing_points" in name:
        m = re.search(r"Li(\d)EE", name)
        agg[(m.group(1) if m else name[-40:], vg, acc, scr)].append((en - st) / 1e3)
for k, v in sorted(agg.items()):
    print("pass %s: arch vgpr %s accum vgpr %s scratch %s B: %d launches, avg %.1f us, min %.1f us" % (k[0], k[1], k[2], k[3], len(v), sum(v) / len(v), min(v)))
PY
rm -rf $O/prof

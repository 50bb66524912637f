#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (bench_results.db): per-kernel count / total / average /
min / max duration, registers, LDS, and - when the run collected PMC counters - the per-dispatch
average of every counter.  Usage: python tools/rocpd_summary.py <results.db> [> profiles/xxx.txt]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    m = re.match(r"(?:plk::)?(k_[a-z0-9_]+)<plk::([A-Za-z0-9]+)", name)
    if m:
        return "%s<%s>" % (m.group(1), m.group(2))
    m = re.match(r"(?:plk::)?(k_[a-z0-9_]+)", name)
    if m:
        return m.group(1)
    return name[:70]


def main(path):
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [x for x in tabs if x.startswith("rocpd_kernel_dispatch")][0]
    ks = [x for x in tabs if x.startswith("rocpd_info_kernel_symbol")][0]
    pm = [x for x in tabs if x.startswith("rocpd_pmc_event")][0]
    pi = [x for x in tabs if x.startswith("rocpd_info_pmc")][0]
    rows = c.execute(f"select s.display_name, s.kernel_name, d.start, d.end, s.arch_vgpr_count, s.sgpr_count, d.group_segment_size, "
                     f"d.private_segment_size, d.grid_size_x, d.workgroup_size_x, d.event_id from {kd} d join {ks} s on d.kernel_id = s.id").fetchall()
    pmc = {}
    for ev, name, val in c.execute(f"select e.event_id, i.name, e.value from {pm} e join {pi} i on e.pmc_id = i.id"):
        pmc.setdefault(ev, {}).setdefault(name, 0.0)
        pmc[ev][name] += val
    agg = {}
    for disp, kname, st, en, vg, sg, lds, scr, gx, wx, ev in rows:
        k = short(disp or kname)
        a = agg.setdefault(k, {"n": 0, "tot": 0, "min": 1 << 62, "max": 0, "vgpr": vg, "sgpr": sg, "lds": lds, "scratch": scr, "grid": gx, "wg": wx, "pmc": {}})
        dur = en - st
        a["n"] += 1
        a["tot"] += dur
        a["min"] = min(a["min"], dur)
        a["max"] = max(a["max"], dur)
        for name, val in pmc.get(ev, {}).items():
            a["pmc"][name] = a["pmc"].get(name, 0.0) + val
    total = sum(a["tot"] for a in agg.values()) or 1
    print("%-44s %7s %12s %10s %10s %10s %6s %5s %5s %7s %8s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "%", "vgpr", "sgpr", "lds_B", "scratch"))
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1]["tot"]):
        print("%-44s %7d %12.1f %10.2f %10.2f %10.2f %6.2f %5s %5s %7s %8s" % (k, a["n"], a["tot"] / 1e3, a["tot"] / a["n"] / 1e3, a["min"] / 1e3,
                                                                             a["max"] / 1e3, 100.0 * a["tot"] / total, a["vgpr"], a["sgpr"], a["lds"], a["scratch"]))
    if any(a["pmc"] for a in agg.values()):
        print("\nPMC counters, average per dispatch (FETCH_SIZE / WRITE_SIZE are in KiB; see MI355X_MICROARCH.md: FETCH_SIZE reads 1/2 of a wide coalesced stream on gfx950)")
        for k, a in sorted(agg.items(), key=lambda kv: -kv[1]["tot"]):
            if a["pmc"]:
                print("%-44s " % k + "  ".join("%s=%.1f" % (n, v / a["n"]) for n, v in sorted(a["pmc"].items())))


if __name__ == "__main__":
    main(sys.argv[1])

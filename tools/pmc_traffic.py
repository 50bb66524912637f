#!/usr/bin/env python3
"""Turns the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, --kernel-trace only) into
profiles/rNN_pmc_traffic.json: HBM-side bytes per launch for the dominant kernels.
Corrections follow MI355X_MICROARCH.md: the counters are in KiB (x1024); on gfx950 FETCH_SIZE reads
exactly 1/2 of a wide coalesced 16-byte-per-lane stream - that is the NTT pass kernel's access
pattern, so its fetch figure is doubled; the MSM accumulation gathers 64-byte points at random
(4 x 16 B per lane), an uncalibrated pattern, so its raw figure is kept and flagged.
The output carries the hash of the device sources it was measured on (bench.py's kernel_source_hash): bench.py only reports
a traffic figure while the kernels are the ones that were measured.
Usage: python tools/pmc_traffic.py <fetch.db> <write.db> <log_n> <out.json>"""
import json
import sqlite3
import sys


def per_kernel(path, counter):
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [x for x in tabs if x.startswith("rocpd_kernel_dispatch")][0]
    ks = [x for x in tabs if x.startswith("rocpd_info_kernel_symbol")][0]
    pm = [x for x in tabs if x.startswith("rocpd_pmc_event")][0]
    pi = [x for x in tabs if x.startswith("rocpd_info_pmc")][0]
    out = {}
    q = (f"select s.display_name, sum(e.value), count(distinct d.id) from {kd} d join {ks} s on d.kernel_id = s.id "
         f"join {pm} e on e.event_id = d.event_id join {pi} i on e.pmc_id = i.id where i.name = ? group by s.display_name")
    for name, total, n in c.execute(q, (counter,)):
        out[name] = (total, n)
    return out


def main(fetch_db, write_db, log_n, out_path, mode="headline"):
    f = per_kernel(fetch_db, "FETCH_SIZE")
    w = per_kernel(write_db, "WRITE_SIZE")
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import kernel_source_hash
    res = {"kernel_source_sha": kernel_source_hash(), "curve": "tweedledee"}
    if mode == "quotient":
        # --workload quotient: the vanishing points are FIVE launches per call (all instantiations of k_vanishing_points summed,
        # divided by the number of calls = dispatches / 5); the fold is one launch per call.  Both read wide coalesced streams
        # (16 bytes per lane from row-major tables / point arrays): FETCH_SIZE is doubled, as for the NTT pass kernel.
        for short, pattern, per_call, note in (("k_vanishing_points", "k_vanishing_points", 5, "sum over the five launches of a call"),
                                               ("k_fold_pairs_glv", "k_fold_pairs_glv", 1, "one launch per call"),
                                               ("k_fold_multi_glv", "k_fold_multi", 3, "digits + preparation + main kernel of one call summed")):
            fk = [k for k in f if pattern in k]
            wk = [k for k in w if pattern in k]
            if not fk or not wk:
                continue
            calls = sum(f[k][1] for k in fk) / per_call
            fb = sum(f[k][0] for k in fk) / calls * 1024.0 * 2.0
            wb = sum(w[k][0] for k in wk) / (sum(w[k][1] for k in wk) / per_call) * 1024.0
            res[short] = {"log_n": int(log_n), "fetch_bytes_per_call": fb, "write_bytes_per_call": wb, "bytes_per_call": fb + wb, "calls_sampled": calls,
                          "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), FETCH_SIZE x2 (wide coalesced streams, gfx950 correction); " + note}
        json.dump(res, open(out_path, "w"), indent=1)
        print(json.dumps(res, indent=1))
        return
    for short, fetch_factor, note in (("k_ntt_pass", 2.0, "FETCH_SIZE x2 (wide coalesced stream, gfx950 correction)"),
                                      ("k_msm_accumulate", 1.0, "FETCH_SIZE uncorrected (random 64-byte gathers, uncalibrated pattern)")):
        fk = [k for k in f if short in k]
        wk = [k for k in w if short in k]
        if not fk or not wk:
            continue
        fb = f[fk[0]][0] / f[fk[0]][1] * 1024.0 * fetch_factor
        wb = w[wk[0]][0] / w[wk[0]][1] * 1024.0
        res[short] = {"log_n": int(log_n), "fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb, "bytes_per_launch": fb + wb,
                      "launches_sampled": f[fk[0]][1], "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), " + note}
    json.dump(res, open(out_path, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:6])

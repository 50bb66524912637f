#!/usr/bin/env python3
"""VALU instruction histogram of the innermost loop (largest backward-branch span) of every kernel of an AMDGPU assembly file whose
name matches a pattern.  Usage: python tools/isa_count.py file.s [name-substring ...]   (hipcc --cuda-device-only -S ... -o file.s)"""
import collections
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
pats = sys.argv[2:]
starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
starts.append((len(lines), "END"))
for (a, name), (b, _) in zip(starts, starts[1:]):
    if pats and not all(p in name for p in pats):
        continue
    body = lines[a:b]
    labels = {l.split(":")[0]: i for i, l in enumerate(body) if re.match(r"^\.LBB\w+:", l)}
    best = None
    for i, l in enumerate(body):
        m = re.search(r"s_cbranch_\w+\s+(\.LBB\w+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            span = (labels[m.group(1)], i)
            if best is None or span[1] - span[0] > best[1] - best[0]:
                best = span
    region = body[best[0]:best[1]] if best else body
    ops = collections.Counter()
    for l in region:
        m = re.match(r"^\s+([vs]_[a-z0-9_]+|global_\w+|ds_\w+|scratch_\w+|buffer_\w+)", l)
        if m:
            ops[m.group(1)] += 1
    valu = sum(v for k, v in ops.items() if k.startswith("v_"))
    vgpr = [l for l in body if ".vgpr_count" in l or "NumVgprs" in l]
    print("%s\n  %s: VALU %d, mad_u64_u32 %d, other VALU %d%s" % (name, "largest loop" if best else "whole kernel", valu, ops["v_mad_u64_u32"],
                                                                   valu - ops["v_mad_u64_u32"], ("  " + vgpr[0].strip()) if vgpr else ""))
    print("  " + "  ".join("%s %d" % kv for kv in sorted(ops.items(), key=lambda kv: -kv[1])[:18]))

#!/usr/bin/env python3
"""Summarise a gfx950 .s file produced by `hipcc -save-temps`: per-kernel registers,
scratch, LDS and an instruction histogram of each kernel body (dev tool)."""
import collections
import re
import sys

path = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else None
txt = open(path).read()
# metadata
for m in re.finditer(r"\.name:\s+(\S+)\n(.*?)\.wavefront_size", txt, re.S):
    pass
kern = collections.OrderedDict()
cur = None
for line in txt.splitlines():
    m = re.match(r"^(_Z\w+):", line)
    if m:
        cur = m.group(1)
        kern[cur] = collections.Counter()
        continue
    if line.startswith("\t.end_amdhsa_kernel") or line.startswith(".Lfunc_end"):
        cur = None
    if cur and line.startswith("\t") and not line.startswith("\t.") and not line.startswith("\t;"):
        if not line.split():
            continue
        op = line.split()[0]
        kern[cur][op] += 1
meta = {}
for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", txt, re.S):
    body = m.group(2)
    g = lambda k: (re.search(r"\.amdhsa_" + k + r"\s+(\S+)", body) or [None, "?"])[1]
    meta[m.group(1)] = dict(vgpr=g("next_free_vgpr"), sgpr=g("next_free_sgpr"), lds=g("group_segment_fixed_size"),
                            scratch=g("private_segment_fixed_size"), accum=g("accum_offset"),
                            dn32=g("float_denorm_mode_32"), dn64=g("float_denorm_mode_16_64"))
for k, c in kern.items():
    if want and want not in k:
        continue
    print(k, meta.get(k, {}))
    tot = sum(c.values())
    print("   total instr", tot, " ".join(f"{o}:{n}" for o, n in c.most_common(40)))

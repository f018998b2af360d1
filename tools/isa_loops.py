#!/usr/bin/env python3
"""Dev tool: list the big basic blocks (unrolled hot loops) of one kernel in a -save-temps .s"""
import collections, re, sys
txt = open(sys.argv[1]).read()
kern = sys.argv[2]
minlen = int(sys.argv[3]) if len(sys.argv) > 3 else 150
dump = len(sys.argv) > 4
names = [m.group(1) for m in re.finditer(r"^(_Z\w+):", txt, re.M) if kern in m.group(1)]
for name in names:
    start = txt.index(name + ":")
    end = txt.index(".end_amdhsa_kernel", start)
    m = re.search(r"\.amdhsa_next_free_vgpr (\d+)", txt[start:end])
    print(name, "vgpr", m.group(1) if m else "?")
    blocks, cur, label = [], [], "entry"
    for l in txt[start:end].splitlines():
        mm = re.match(r"^(\.LBB\d+_\d+):", l)
        if mm:
            blocks.append((label, cur)); label, cur = mm.group(1), []
        else:
            cur.append(l)
    blocks.append((label, cur))
    for label, b in blocks:
        ins = [l.split()[0] for l in b if l.startswith("\t") and len(l.split()) and not l.split()[0].startswith((".", ";"))]
        if len(ins) >= minlen:
            c = collections.Counter(ins)
            print("  ", label, len(ins), dict(c.most_common(30)))
            if dump:
                print("\n".join(b))

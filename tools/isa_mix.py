"""Instruction mix of the big basic blocks (the time loops) of one kernel in an assembly listing (hipcc -S)."""
import re, collections, sys
path, pat = sys.argv[1], sys.argv[2]
lines = open(path).read().splitlines()
start = next(i for i, l in enumerate(lines) if l.startswith("_ZN") and pat in l.split(":")[0])
end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith(".Lfunc_end"))
body = lines[start:end]
blocks, cur, name = [], [], "entry"
for l in body:
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        blocks.append((name, cur)); cur, name = [], m.group(1)
    elif l.startswith("\t") and not l.strip().startswith((".", ";")):
        cur.append(l.strip().split()[0])
blocks.append((name, cur))
print(lines[start].split(":")[0], "instructions", sum(len(b) for _, b in blocks))
for name, ins in blocks:
    if len(ins) < int(sys.argv[3]) if len(sys.argv) > 3 else 80:
        continue
    c = collections.Counter()
    for x in ins:
        if re.match(r"v_(exp|rcp|log|rsq|sqrt|sin|cos)", x): c["trans"] += 1
        elif x.startswith("v_pk_"): c["v_pk"] += 1
        elif x.startswith("v_cvt"): c["v_cvt"] += 1
        elif x.startswith("v_"): c["valu"] += 1
        elif x.startswith("ds_"): c["lds"] += 1
        elif x.startswith(("global_", "buffer_", "scratch_", "flat_")): c["vmem"] += 1
        elif x.startswith("s_waitcnt"): c["wait"] += 1
        elif x.startswith("s_"): c["salu"] += 1
        else: c["other"] += 1
    print(f"  {name}: {len(ins)} instr  {dict(c)}")
    print("     ", collections.Counter(ins).most_common(16))

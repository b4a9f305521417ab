"""Instruction mix of the outermost loops of one kernel in an assembly listing (hipcc -S): every backward branch defines a
loop [target label .. branch]; prints the mix of each maximal one."""
import re, collections, sys
path, pat = sys.argv[1], sys.argv[2]
lines = open(path).read().splitlines()
start = next(i for i, l in enumerate(lines) if l.startswith("_ZN") and pat in l.split(":")[0])
end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith(".Lfunc_end"))
body = lines[start:end]
labels = {}
ins = []
for l in body:
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        labels[m.group(1)] = len(ins)
    elif l.startswith("\t") and not l.strip().startswith((".", ";")):
        ins.append(l.strip())
loops = []
for i, x in enumerate(ins):
    m = re.match(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)", x)
    if m:
        t = labels.get(m.group(1) or m.group(2))
        if t is not None and t <= i:
            loops.append((t, i))
loops = [lp for lp in loops if not any(o != lp and o[0] <= lp[0] and lp[1] <= o[1] for o in loops)]
def cls(x):
    x = x.split()[0]
    if re.match(r"v_(exp|rcp|log|rsq|sqrt|sin|cos)", x): return "trans"
    if x.startswith("v_mfma"): return "mfma"
    if re.match(r"v_(mul_lo|mul_hi|mad_u64|mad_i64)", x): return "valu_quarter"
    if x.startswith("v_accvgpr"): return "accvgpr"
    if x.startswith("v_"): return "valu"
    if x.startswith("ds_"): return "lds"
    if x.startswith(("global_", "buffer_", "scratch_", "flat_")): return "vmem"
    if x.startswith("s_waitcnt"): return "wait"
    if x.startswith("s_barrier"): return "barrier"
    return "salu"
print(lines[start].split(":")[0], "instructions", len(ins))
for t, i in loops:
    c = collections.Counter(cls(x) for x in ins[t:i + 1])
    print(f"  loop [{t}..{i}] {i - t + 1} instr  {dict(c)}")
    print("     ", collections.Counter(x.split()[0] for x in ins[t:i + 1]).most_common(14))

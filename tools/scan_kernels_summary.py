"""Per-kernel average durations of the segm:: kernels in a rocprofv3 kernel-trace database (rocpd sqlite)."""
import sqlite3, sys, re, collections
con = sqlite3.connect(sys.argv[1])
rows = con.execute("select name, end - start from kernels").fetchall()
agg = collections.defaultdict(list)
for n, d in rows:
    if "segm" in n:
        agg[re.sub(r"\s+", " ", n)[:110]].append(d)
lines = []
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    v2 = sorted(v)[len(v) // 5: len(v) - len(v) // 5] or v          # trimmed mean (first launches include warm-up)
    lines.append(f"{sum(v2) / len(v2) / 1e3:10.1f} us avg  n={len(v):4d}  {k}")
print("\n".join(lines))
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write("\n".join(lines) + "\n")

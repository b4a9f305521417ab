"""Summarise the last training step of a rocprofv3 kernel-trace database (rocpd sqlite)."""
import sqlite3, re, collections, sys
db = sys.argv[1]; out = sys.argv[2] if len(sys.argv) > 2 else None
con = sqlite3.connect(db); cur = con.cursor()
rows = cur.execute("select name, start, end, grid_x, workgroup_x from kernels order by start").fetchall()
opt = [r for r in rows if 'multi_tensor_apply' in r[0]]
clusters = []
for r in opt:
    if not clusters or r[1] - clusters[-1][1] > 50_000_000: clusters.append([r[1], r[2]])
    else: clusters[-1][1] = r[2]
t0, t1 = clusters[-2][1], clusters[-1][1]
agg = collections.defaultdict(lambda: [0, 0])
for n, s, e, g, w in rows:
    if s >= t0 and e <= t1:
        k = re.sub(r"\s+", " ", n); agg[k][0] += e - s; agg[k][1] += 1
tot = sum(v[0] for v in agg.values())
lines = [f"last training step: wall {(t1-t0)/1e6:.1f} ms, kernel time {tot/1e6:.1f} ms, {sum(v[1] for v in agg.values())} dispatches"]
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
    lines.append(f"{v[0]/1e6:8.2f} ms {100*v[0]/tot:5.1f}% n={v[1]:4d} avg={v[0]/v[1]/1e3:9.1f}us {k[:130]}")
print("\n".join(lines))
if out: open(out, "w").write("\n".join(lines) + "\n")
big = [(n, s, e, g, w) for n, s, e, g, w in rows if s >= t0 and e <= t1 and e - s > 5_000_000]
print("--- kernels > 5 ms (time since step start)")
for n, s, e, g, w in big: print(f"t={(s-t0)/1e6:8.1f} dur={(e-s)/1e6:8.2f} ms grid={g} wg={w} {re.sub(chr(10),' ',n)[:110]}")

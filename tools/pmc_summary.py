"""Per-kernel FETCH_SIZE / WRITE_SIZE (KB per dispatch, median) from rocprofv3 --pmc databases (rocpd sqlite)."""
import sqlite3, sys, re, collections, statistics
out = {}
for path in sys.argv[1:]:
    con = sqlite3.connect(path)
    rows = con.execute("select kernel_name, counter_name, value from counters_collection").fetchall()
    agg = collections.defaultdict(list)
    for k, c, v in rows:
        if "segm" in k:
            agg[(re.sub(r"\s+", " ", k)[:90], c)].append(float(v))
    for (k, c), v in agg.items():
        out.setdefault(k, {})[c] = (statistics.median(v), len(v))
for k, d in sorted(out.items()):
    print(f"{k:92s} " + "  ".join(f"{c} {m / 1024:9.2f} MB (n={n})" for c, (m, n) in sorted(d.items())))

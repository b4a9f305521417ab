#!/bin/bash
# Round 6: rocprofv3 --kernel-trace --stats of the bench command's roofline leg (python bench.py with the step shortened: the scan
# kernels' durations per launch, to set beside the HIP-event figures `roofline.ms` / `roofline.backward.ms` of the same run)
mkdir -p gpurun_out/prof; R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof/r06_scan_stats -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-configs --no-dropin > $R/gpurun_out/prof/r06_scan_stats.log 2>&1
tail -1 $R/gpurun_out/prof/r06_scan_stats.log | python3 -c "
import json, sys
d = json.loads(sys.stdin.read()); r = d['roofline']
print('bench line (HIP events): scan fwd %.4f ms per direction launch, frac %.4f; bwd %.4f ms, frac %.4f' % (r['ms'], r['frac'], r['backward']['ms'], r['backward']['frac']))
print('kernels per launch (bench):', json.dumps(r.get('kernels', r.get('kernel_ms', {})))[:400])
"
python3 - $R/gpurun_out/prof/r06_scan_stats <<'PY'
import csv, glob, sys, collections, statistics
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "scan_" in n:
        d[(n[:100], r["Grid_Size_X"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("# rocprofv3 --kernel-trace: scan kernels of the whole run, by name and grid (us: median / mean, launches)")
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    print(f"{statistics.median(v):9.1f} {statistics.mean(v):9.1f}  n={len(v):4d}  grid {k[1]:>9s}  {k[0]}")
PY
f=$(ls $R/gpurun_out/prof/r06_scan_stats/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && { echo "# rocprofv3 --stats (top 12)"; head -13 $f | cut -c1-200; }

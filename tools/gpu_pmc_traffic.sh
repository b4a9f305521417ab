#!/bin/bash
# HBM-side traffic of one selective-scan forward launch at the roofline shape -> gpurun_out/scan_traffic.json (copy it to
# profiles/scan_traffic.json, where bench.py reads `roofline.traffic` from).  rocprofv3 PMC, FETCH_SIZE and WRITE_SIZE in
# SEPARATE passes (the TCC block has 4 counter slots), no tracing besides --kernel-trace.  Calibration as the guide
# (/opt/skills/guides/MI355X_MICROARCH.md, HBM section) prescribes: on gfx950 FETCH_SIZE under-reports wide reads by 2x and
# other widths are uncalibrated, so the same passes also run two kernels with KNOWN byte counts in the scan's own access
# pattern (conv1d_fwd: one 16-bit element per lane and row, reads and writes 100.7 MB) and in 16-byte accesses
# (transpose_add); read / write factors = known bytes / counter of conv1d_fwd.
#   usage (from the repo root on the GPU box):  COMMIT=<git sha> bash tools/gpu_pmc_traffic.sh
mkdir -p gpurun_out/prof; R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
cat > /tmp/pmc_traffic.py <<'PY'
import sys, torch
sys.path.insert(0, sys.argv[1])
from segmamba_amd import lib as L, ops_raw
hip = L.get_lib()
dev = "cuda"
B, D, N, Lq = 2, 96, 16, 64 ** 3
for dt in (torch.bfloat16, torch.float32):
    g = torch.Generator(device=dev).manual_seed(0)
    rn = lambda *s: torch.randn(*s, device=dev, generator=g).to(dt)
    u, z = rn(B, Lq, D), rn(B, Lq, D)
    delta = (0.5 * torch.rand(B, Lq, D, device=dev, generator=g)).to(dt)
    A = -0.5 * torch.rand(D, N, device=dev, generator=g)
    Bm, Cm = rn(B, Lq, N), rn(B, Lq, N)
    Dv = torch.randn(D, device=dev, generator=g); db = 0.5 * torch.rand(D, device=dev, generator=g)
    w = torch.randn(D, 4, device=dev, generator=g)
    for _ in range(3):
        ops_raw.scan_fwd(hip, u, delta, A, Bm, Cm, Dv, z, db, True, channel_last=True, need_out=True, need_ckpt=True)
        ops_raw.transpose_add(hip, u)                                   # calibration, 16-byte accesses: reads + writes B L D e
        ops_raw.conv1d_fwd(hip, u, w, None, True, channel_last=True)    # calibration, the scan's pattern: reads + writes B L D e
torch.cuda.synchronize()
PY
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/prof/traffic_$c -o pmc -- python /tmp/pmc_traffic.py $R > $R/gpurun_out/prof/traffic_$c.log 2>&1
  echo "$c rc=$?"
done
python3 - $R <<'PY'
import csv, glob, json, os, sys, collections, statistics, datetime
R = sys.argv[1]
val = collections.defaultdict(dict)        # kernel -> counter -> median KB
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{R}/gpurun_out/prof/traffic_{c}/**/*counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == c:
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        val[k][c] = statistics.median(v)
def pick(sub, tag):
    ks = [k for k in val if sub in k and tag in k]
    return ks
B, D, L = 2, 96, 64 ** 3
out = {"commit": os.environ.get("COMMIT", "unknown"), "date": datetime.date.today().isoformat(),
       "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, KB x 1024), medians over 3 launches, scaled by read / "
                 "write factors = known bytes / counter of conv1d_fwd (same 16-bit-per-lane row pattern) in the same passes; "
                 "the guide's 2x for wide reads is listed as `bytes_guide_2x`", "kernels": {}}
tags = {"bf16": ("DF16b", 2), "fp32": ("IfL", 4)}
for name, (tag, es) in tags.items():
    known = B * L * D * es
    cal = [k for k in val if "conv1d_fwd" in k and (tag in k or (tag == "IfL" and "<float" in k))]
    if not cal:
        continue
    rf = known / (val[cal[0]]["FETCH_SIZE"] * 1024)
    wf = known / (val[cal[0]]["WRITE_SIZE"] * 1024)
    tot = tot2 = 0.0
    ks = {}
    for k in val:
        if ("scan_fwd" in k or "scan_carry" in k) and (tag in k or (tag == "IfL" and "<float" in k) or "scan_carry" in k):
            rd, wr = val[k].get("FETCH_SIZE", 0) * 1024, val[k].get("WRITE_SIZE", 0) * 1024
            n = 0.5 if "scan_carry" in k else 1.0     # the carry kernels run once per dtype pass but are not dtype-tagged: both passes counted
            ks[k[:80]] = {"fetch_raw": rd, "write_raw": wr}
            tot += rd * rf + wr * wf
            tot2 += rd * 2 + wr
    out[name] = {"bytes": int(tot), "bytes_guide_2x": int(tot2), "read_factor": round(rf, 3), "write_factor": round(wf, 3),
                 "algorithmic_bytes": es * B * L * (5 * D + 2 * 16)}
    out["kernels"][name] = ks
json.dump(out, open(f"{R}/gpurun_out/scan_traffic.json", "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "kernels"}, indent=1))
PY

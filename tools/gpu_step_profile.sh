#!/bin/bash
# rocprofv3 kernel trace of two training steps; prints the per-kernel totals per step grouped by owner (own kernels / BLAS /
# MIOpen / ATen) and writes the table to gpurun_out/<tag>_step_kernels.txt:   tools/gpu_step_profile.sh <tag>
TAG=${1:-step}; mkdir -p gpurun_out/prof; R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof/$TAG -o bench -- python $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-roofline --no-configs --no-graph > $R/gpurun_out/prof/$TAG.log 2>&1
echo "rc=$?"; tail -1 $R/gpurun_out/prof/$TAG.log | cut -c1-200
python3 - $R/gpurun_out/prof/$TAG $R/gpurun_out/${TAG}_step_kernels.txt <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last third of the trace = the two timed steps (warm-up and autotuning come first); normalise per step by counting CE kernels
names = [r["Kernel_Name"] for r in rows]
ce = [i for i, n in enumerate(names) if "sgd_coef" in n]     # one launch per optimizer step (the last phase of a training step)
steps = len(ce)
start = ce[-3] + 1 if steps >= 3 else 0           # after the third-last step's coefficient kernel: two whole steps follow (shifted by the update launches)
sel = rows[start:]
nsteps = 2 if steps >= 3 else max(steps, 1)
def owner(n):
    if "segm::" in n or "_ZN4segm" in n: return "own (segm::)"
    if n.startswith("Cijk_") or "rocblas" in n.lower() or "hipblaslt" in n.lower(): return "BLAS"
    if "miopen" in n.lower() or "Im3d2Col" in n or "Col2Im" in n or "ck::" in n or "kernel_grouped_conv" in n or "naive_conv" in n: return "MIOpen / CK"
    if "at::native" in n or "elementwise" in n or "at_cuda" in n: return "ATen"
    return "other"
tot = collections.defaultdict(float); per = collections.defaultdict(lambda: [0.0, 0])
for r in sel:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot[owner(r["Kernel_Name"])] += d
    k = r["Kernel_Name"][:100]; per[k][0] += d; per[k][1] += 1
out = [f"# kernel time per training step (2 x 4 x 128^3, bf16), {nsteps} steps averaged; launches per step = {len(sel) / nsteps:.0f}"]
all_ms = sum(tot.values()) / nsteps / 1e3
for o, v in sorted(tot.items(), key=lambda kv: -kv[1]):
    out.append(f"{v / nsteps / 1e3:8.2f} ms  {100 * v / nsteps / 1e3 / all_ms:5.1f} %  {o}")
out.append(f"{all_ms:8.2f} ms  total")
out.append("")
for k, (d, n) in sorted(per.items(), key=lambda kv: -kv[1][0])[:45]:
    out.append(f"{d / nsteps / 1e3:8.3f} ms  n={n / nsteps:6.1f}  {k}")
print("\n".join(out[:30]))
open(sys.argv[2], "w").write("\n".join(out) + "\n")
PY

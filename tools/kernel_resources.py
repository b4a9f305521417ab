"""Compile one kernel translation unit for gfx950 and print per-kernel register / LDS / scratch usage."""
import re, subprocess, sys
src = sys.argv[1]; filt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-munsafe-fp-atomics",
       "-c", src, "-o", "/tmp/_kr.o", "-Rpass-analysis=kernel-resource-usage"] + sys.argv[3:]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
if "error" in out: print(out[-3000:]); sys.exit(1)
cur = {}
for line in out.splitlines():
    m = re.search(r"remark:\s+(.*?)\s*\[-Rpass", line)
    if not m: continue
    t = m.group(1)
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}
    else:
        k, _, v = t.partition(":"); cur[k.strip()] = v.strip()
        if k.strip().startswith("LDS Size"):
            name = subprocess.run(["c++filt", cur["name"]], capture_output=True, text=True).stdout.strip()
            name = name.replace("segm::", "").replace("(segm::ScanDev)", "").replace("(ConvDev)", "").replace("void ", "")
            if filt in name:
                print(f"{name[:70]:70s} vgpr {cur.get('VGPRs'):>4} agpr {cur.get('AGPRs'):>4} scratch {cur.get('ScratchSize [bytes/lane]'):>5} occ {cur.get('Occupancy [waves/SIMD]'):>2} lds {cur.get('LDS Size [bytes/block]'):>6}")

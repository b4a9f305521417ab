"""Ablation builds of the forward apply kernel (experiments only, nothing here ships): each variant removes ONE ingredient of the
step from a copy of scan_fwd_fast.hip and links a full library with it, so that tools/gpu_scan_occupancy.py can time the kernel in
its throughput-bound regime (6 waves per SIMD of work).  The difference to the unmodified build is what the ingredient costs.
    python tools/build_scan_ablations.py   ->  build/variants/abl_<name>.so"""
import os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from segmamba_amd import build as B

SRC = os.path.join(B.CSRC, "scan_fwd_fast.hip")
text = open(SRC).read()

def sub(t, old, new, count=1):
    assert old in t, old[:60]
    return t.replace(old, new, count)

ST_OUT = "            if (has_out) BufIO<T>::st(op.rs, op.voff, oso, y);\n"
ST_OZ = "            if (has_z) BufIO<T>::st(ozp.rs, ozp.voff, ozso, y * zz * sigmoidf(zz));\n"
variants = {
    "base": lambda t: t,
    # the two row stores of a step: results summed into a register that is stored once per sub-tile instead
    "nostore": lambda t: sub(sub(sub(t, ST_OUT, "            keep += y;\n"), ST_OZ, "            keep += y * zz * sigmoidf(zz);\n"),
                             "        buf ^= 1;\n    }\n    (void)tau0;", "        if (has_z) BufIO<T>::st(ozp.rs, ozp.voff, ozso, keep);\n        buf ^= 1;\n    }\n    (void)tau0;"),
    # no gate arithmetic (sigmoid: exp + rcp) - out_z = y
    "nogate": lambda t: sub(t, ST_OZ, "            if (has_z) BufIO<T>::st(ozp.rs, ozp.voff, ozso, y + zz);\n"),
    # no checkpoint stores
    "nockpt": lambda t: sub(t, "        if (P.ckpt && (s & 1) == 0) {                      // kCkpt = 2 sub-tiles", "        if (false) {"),
    # B / C rows read from LDS once per sub-tile (step 0's rows for every step) instead of once per step
    "nolds": lambda t: sub(sub(t, "                bq[q] = reinterpret_cast<const float4*>(lb + j * kFS)[q];\n                cq[q] = reinterpret_cast<const float4*>(lc + j * kFS)[q];",
                               "                if (j == 0) { bq[q] = reinterpret_cast<const float4*>(lb)[q]; cq[q] = reinterpret_cast<const float4*>(lc)[q]; }"),
                           "            float4 bq[4], cq[4];\n#pragma unroll\n            for (int q = 0; q < 4; ++q) {\n                if (j == 0)", "#pragma unroll\n            for (int q = 0; q < 4; ++q) {\n                if (j == 0)"),
    # no refill of the row rings (the prologue's rows are reused)
    "noload": lambda t: sub(sub(sub(t, "            nu[j] = BufIO<T>::ld(up.rs, up.voff, su);\n            nd[j] = BufIO<T>::ld(dp.rs, dp.voff, sd);\n            nz[j] = BufIO<T>::ld(zp.rs, zp.voff, sz);",
                                    "            SEGM_PIN_F32(nu[j]); SEGM_PIN_F32(nd[j]); SEGM_PIN_F32(nz[j]);"), "XXXX", "XXXX", 0), "YYYY", "YYYY", 0),
    # the output sum y = C h not formed (8 packed multiply-adds): the state sum of h is stored instead
    "noy": lambda t: sub(sub(t, "                ya = c0 * h[2 * q] + ya;\n                yb = c1 * h[2 * q + 1] + yb;", "                ya = h[2 * q] + c0;\n                yb = yb + c1;"), "XXXX", "XXXX", 0),
}

def prep(name, t):
    if name in ("nostore",):
        t = sub(t, "    int buf = 0;\n    const int nsub = gm.chunk / kFT;\n    for (int s = 0; s < nsub; ++s) {\n        float* lb = &s_bc[buf][wave][it.gi][0][0];",
                "    int buf = 0;\n    float keep = 0.f;\n    const int nsub = gm.chunk / kFT;\n    for (int s = 0; s < nsub; ++s) {\n        float* lb = &s_bc[buf][wave][it.gi][0][0];")
    if name == "nolds":
        t = sub(t, "        uint32_t oso = (uint32_t)Uc * (uint32_t)op.stb, ozso", "        float4 bq[4], cq[4];\n        uint32_t oso = (uint32_t)Uc * (uint32_t)op.stb, ozso")
    return t

def sub0(t, old, new, count):
    return t
import builtins
_orig_sub = sub
def sub(t, old, new, count=1):                     # count == 0: placeholder no-op used above to keep the lambdas regular
    if count == 0:
        return t
    return _orig_sub(t, old, new, count)

out_dir = os.path.join(ROOT, "build", "abl_src")
os.makedirs(out_dir, exist_ok=True)
os.makedirs(os.path.join(ROOT, "build", "variants"), exist_ok=True)
names = sys.argv[1:] or list(variants)
for name in names:
    t = prep(name, variants[name](text))
    src = os.path.join(out_dir, f"scan_fwd_fast_{name}.hip")
    open(src, "w").write(t)
    obj = os.path.join(out_dir, f"scan_fwd_fast_{name}.o")
    r = subprocess.run([B.HIPCC, *B.FLAGS, "-I", B.CSRC, "-I", os.path.join(ROOT, "include"), "-c", src, "-o", obj], capture_output=True, text=True)
    if r.returncode != 0:
        print(name, "FAILED\n", r.stderr[-2000:]); continue
    B.build(verbose=False)
    objs = [os.path.join(B.OBJ_DIR, os.path.basename(s).replace(".hip", ".o")) for s in B._sources() if not s.endswith("scan_fwd_fast.hip")] + [obj]
    so = os.path.join(ROOT, "build", "variants", f"abl_{name}.so")
    r = subprocess.run([B.HIPCC, "--offload-arch=" + B.ARCH, "-shared", "-fPIC", *objs, "-o", so], capture_output=True, text=True)
    print(name, "->", so if r.returncode == 0 else r.stderr[-1000:])

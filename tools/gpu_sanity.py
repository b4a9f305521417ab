"""Quick on-GPU sanity (not a test): small parity numbers + scan / conv timings at SegMamba's stage shapes."""
import sys, os, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import helpers as H
from segmamba_amd import lib as L, ops_raw
from oracle.ref_ops import algorithmic_bytes_scan
hip = L.SegmLib(os.environ["SEGM_LIB"]) if os.environ.get("SEGM_LIB") else L.get_lib()
print("lib", hip.path)
print(torch.cuda.get_device_name(0))
for cl in (True, False):
    c = H.scan_case(2, 96, 16, 512)
    ref = H.scan_oracle(c)
    res = H.run_scan(hip, c, "cuda", cl)
    print("cl", cl, {k: float((res[k].float().cpu() - ref[k].float()).abs().max()) for k in res})
def tm(fn, it=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); e1.synchronize(); return e0.elapsed_time(e1)/it
out = []
for dtype in (torch.float32, torch.bfloat16):
    for (D, Lq) in ((96, 262144), (192, 32768), (384, 4096), (768, 512)):
        B, N = 2, 16
        g = torch.Generator(device="cuda").manual_seed(0)
        rn = lambda *s: torch.randn(*s, device="cuda", generator=g).to(dtype)
        u, z, dout = rn(B, Lq, D), rn(B, Lq, D), rn(B, Lq, D)
        delta = (0.5*torch.rand(B, Lq, D, device="cuda", generator=g)).to(dtype)
        A = -0.5*torch.rand(D, N, device="cuda", generator=g); Bm, Cm = rn(B, Lq, N), rn(B, Lq, N)
        Dv = torch.randn(D, device="cuda", generator=g); db = 0.5*torch.rand(D, device="cuda", generator=g)
        for chunk in ((0,) if os.environ.get("SEGM_QUICK") else (0, 64, 128, 512)):
            if chunk > Lq: continue
            f = ops_raw.scan_fwd(hip, u, delta, A, Bm, Cm, Dv, z, db, True, channel_last=True, need_out=True, need_ckpt=True, chunk=chunk)
            ms_f = tm(lambda: ops_raw.scan_fwd(hip, u, delta, A, Bm, Cm, Dv, z, db, True, channel_last=True, need_out=True, need_ckpt=True, chunk=chunk))
            ms_b = tm(lambda: ops_raw.scan_bwd(hip, u, delta, A, Bm, Cm, Dv, z, db, dout, f["out"], f["ckpt"], True, channel_last=True, chunk=f["chunk"]), 5)
            es = u.element_size()
            r = dict(dtype=str(dtype), D=D, L=Lq, chunk=f["chunk"], fwd_ms=round(ms_f,4), bwd_ms=round(ms_b,4),
                     fwd_GBps=round(algorithmic_bytes_scan(B,D,Lq,N,es)/ms_f*1e-6,1), bwd_GBps=round(algorithmic_bytes_scan(B,D,Lq,N,es,backward=True)/ms_b*1e-6,1))
            print(json.dumps(r), flush=True); out.append(r)
        w = torch.randn(D, 4, device="cuda"); bb = torch.randn(D, device="cuda")
        ms_cf = tm(lambda: ops_raw.conv1d_fwd(hip, u, w, bb, True, channel_last=True))
        ms_cb = tm(lambda: ops_raw.conv1d_bwd(hip, u, w, bb, dout, True, channel_last=True))
        print(json.dumps(dict(conv=1, dtype=str(dtype), D=D, L=Lq, fwd_ms=round(ms_cf,4), bwd_ms=round(ms_cb,4),
              fwd_GBps=round(2*es*B*D*Lq/ms_cf*1e-6,1), bwd_GBps=round(3*es*B*D*Lq/ms_cb*1e-6,1))), flush=True)
json.dump(out, open("gpurun_out/sanity_%s.json" % os.environ.get("SEGM_TAG", "a"),"w"))

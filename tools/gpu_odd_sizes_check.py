"""forward + backward of the network under bf16 autocast at input sizes whose lower levels are / are not multiples of 8 (the cube
kernels take 8 | D, H, W; everything else falls back to the row kernels / vendor routes), batch 1 and 2: loss and gradient norm with the
cube routing on and off must agree to bf16 accuracy.      python tools/gpu_odd_sizes_check.py"""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from segmamba_amd import conv3d as C
from model_segmamba.segmamba import SegMamba

torch.manual_seed(0)
net = SegMamba(in_chans=4, out_chans=4, depths=[2, 2, 2, 2], feat_size=[48, 96, 192, 384]).cuda()
for B, S in ((1, 96), (2, 64), (1, 128), (1, 160)):
    x = torch.rand(B, 4, S, S, S, device="cuda")
    y = torch.randint(0, 4, (B, S, S, S), device="cuda")
    res = []
    for cube in (True, False):
        C._CUBE = cube; C._CUBE_WGRAD = cube
        for p in net.parameters():
            p.grad = None
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = torch.nn.functional.cross_entropy(net(x).float(), y)
        loss.backward()
        gn = torch.sqrt(sum((p.grad.float() ** 2).sum() for p in net.parameters() if p.grad is not None))
        res.append((float(loss), float(gn)))
    (l1, g1), (l0, g0) = res
    ok = abs(l1 - l0) <= 2e-2 * abs(l0) and abs(g1 - g0) <= 5e-2 * g0
    print("B=%d %d^3 (lowest level %d^3): cube loss %.5f grad norm %.4f | round-5 routing loss %.5f grad norm %.4f  %s" % (
        B, S, S // 16, l1, g1, l0, g0, "ok" if ok else "MISMATCH"), flush=True)

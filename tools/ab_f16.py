import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F, sdflabel_amd
from sdflabel_amd.fixtures import ASSET, K_for
dev = "cuda"
for prec in (torch.float32, torch.float16):
    dec, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision=prec); dec = dec.to(dev)
    grid = sdflabel_amd.Grid3D(40, dev); lat = F.normalize(torch.tensor([0.3, -0.5, 0.8], device=dev), dim=0)
    for B in (1, 16):
        inp = torch.cat([lat.expand(grid.points.size(0), -1), grid.points], 1).repeat(B, 1).contiguous()
        with torch.no_grad():
            for _ in range(3): dec(inp)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): dec(inp)
            e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 10
        macs = dec.handle(torch.device(dev, 0)).macs
        print(prec, "B=%d" % B, "%.3f ms  %.1f TFLOP/s" % (t, 2 * macs * inp.shape[0] / t / 1e9))
    br = sdflabel_amd.BatchRenderer(dec, 40, K_for(256, 256), (256, 256), 1, device=dev)
    br.set_params(torch.tensor([0.6], device=dev), torch.tensor([[0., 0., 3.5]], device=dev), torch.tensor([[0.3, -0.5, 0.8]], device=dev))
    o3 = torch.ones(1, 3, 256, 256, device=dev); ox = torch.ones(1, br.cap, 3, device=dev)
    for _ in range(3): br.forward(); br.backward(g_color=o3, g_xyzf=ox)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): br.forward(); br.backward(g_color=o3, g_xyzf=ox)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 20
    print(prec, "crop-iteration %.3f ms -> %.1f Mrays/s" % (t, 65536 / t / 1e3))

"""Run only the decoder forward (G=64000) a few times -- target for rocprofv3 PMC passes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
import sdflabel_amd
from sdflabel_amd.fixtures import ASSET
dev = "cuda"
dec, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision=torch.float32); dec = dec.to(dev)
grid = sdflabel_amd.Grid3D(40, dev)
lat = F.normalize(torch.tensor([0.3, -0.5, 0.8], device=dev), dim=0)
inputs = torch.cat([lat.expand(grid.points.size(0), -1), grid.points], 1).contiguous()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dec.mlp_precision = {"f32": torch.float32, "f16": torch.float16, "split": "float32_split"}[sys.argv[2] if len(sys.argv) > 2 else "f32"]
with torch.no_grad():
    for _ in range(n):
        dec(inputs)
torch.cuda.synchronize()

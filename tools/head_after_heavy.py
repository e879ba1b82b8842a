"""Why do the bag-level f32 GEMMs take 2-3x longer inside a step than alone?  The chain decoder [128x512x512] -> GRU input
projection [128x3072x512] -> recurrent [64x3072x1024] is traced (a) alone, (b) right after a streaming elementwise pass over 1 GB,
(c) right after an encoder panel GEMM (268 MB in, 268 MB out, matrix cores busy)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from murcl_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
x = torch.randn((128, 512), device=dev)
Wd, Wih, Whh = torch.randn((512, 512), device=dev), torch.randn((3072, 512), device=dev), torch.randn((3072, 1024), device=dev)
h = torch.randn((64, 1024), device=dev)
b512, b3072 = torch.zeros(512, device=dev), torch.zeros(3072, device=dev)
big = torch.empty(256 << 20, dtype=torch.float32, device=dev)
X = (torch.randn(262144, 512, device=dev) * 0.5).bfloat16()
W = (torch.randn(512, 512, device=dev) * 0.04).bfloat16()
mark0 = torch.zeros(1024, device=dev, dtype=torch.float64)      # start marker: FillFunctor<double>
mark1 = torch.zeros(1024, device=dev, dtype=torch.int16)        # end marker: FillFunctor<short>

def chain():
    y = ops.gemm_nt(x, Wd, epi=ops.EPI_BIAS_RELU, bias=b512)
    g = ops.gemm_nt(y, Wih, epi=ops.EPI_BIAS, bias=b3072)
    return ops.gemm_nt(h, Whh, epi=ops.EPI_BIAS, bias=b3072)

for mode in ("alone", "after_stream", "after_panel", "after_3_panels"):
    for rep in range(6):
        if mode == "after_stream": big.add_(1.0)
        elif mode == "after_panel": ops.panel_gemm(X, W, ops.PG_BIAS_RELU, bias=b512, want_bitmask=True)
        elif mode == "after_3_panels":
            for _ in range(3): ops.panel_gemm(X, W, ops.PG_BIAS_RELU, bias=b512, want_bitmask=True)
        mark0.fill_(1.0)
        chain()
        mark1.fill_(2)
        torch.cuda.synchronize()
print("MODES alone after_stream after_panel after_3_panels")

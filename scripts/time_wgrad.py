"""Per-layer timing of the CostRegNet weight-gradient kernel at the training shape
(640x512 images -> 160x128 feature maps, D=192): python scripts/time_wgrad.py"""
import os, sys
import torch
sys.path.insert(0, os.getcwd())
from mvs_amd import ops

# (name, Cin, Cout, stride, input-grid divisor, transposed)
LAYERS = [("conv0", 32, 8, 1, 1, 0), ("conv1", 8, 16, 2, 1, 0), ("conv2", 16, 16, 1, 2, 0),
          ("conv3", 16, 32, 2, 2, 0), ("conv4", 32, 32, 1, 4, 0), ("conv5", 32, 64, 2, 4, 0),
          ("conv6", 64, 64, 1, 8, 0), ("conv7", 64, 32, 2, 4, 1), ("conv9", 32, 16, 2, 2, 1),
          ("conv11", 16, 8, 2, 1, 1), ("prob", 8, 1, 1, 1, 0)]

def main():
    dev = torch.device("cuda:0")
    D, H, W = 192, 128, 160
    tot = 0.0
    for name, cin, cout, s, div, tr in LAYERS:
        d, h, w = D // div, H // div, W // div      # the fine grid of the layer
        if tr:   # transposed: weight grad = wgrad(x=grad_out on the fine grid, g=layer input on the coarse grid)
            x = torch.randn(1, d, h, w, cout, device=dev); g = torch.randn(1, d // 2, h // 2, w // 2, cin, device=dev)
        else:
            x = torch.randn(1, d, h, w, cin, device=dev)
            g = torch.randn(1, (d - 1) // s + 1, (h - 1) // s + 1, (w - 1) // s + 1, cout, device=dev)
        for _ in range(2):
            ops.conv3d_wgrad(x, g, s)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            ops.conv3d_wgrad(x, g, s)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        gf = 2.0 * 27 * cin * cout * g.shape[1] * g.shape[2] * g.shape[3] / 1e9
        tot += ms
        print(f"{name:7s} x{tuple(x.shape[1:])} g{tuple(g.shape[1:])} s{s}: {ms:7.3f} ms  {gf / ms:7.1f} TFLOP/s useful")
    print(f"total {tot:.3f} ms")

if __name__ == "__main__":
    main()

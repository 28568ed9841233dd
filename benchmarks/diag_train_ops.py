"""fp32 backward of single library ops on the GPU against float64 on the GPU: batch norm (training mode), conv3x3."""
import torch, torch.nn.functional as F
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
x = torch.randn((128, 64, 9, 9), generator=g).to(dev); dy = torch.randn((128, 64, 9, 9), generator=g).to(dev)
w = (torch.randn((64, 64, 3, 3), generator=g) * 0.05).to(dev); gam = torch.rand(64, generator=g).to(dev) + 0.5; bet = torch.randn(64, generator=g).to(dev)
def rel(a, b): return ((a.double() - b).abs().max() / b.abs().max()).item()
for enabled in (True, False):
    with torch.backends.cudnn.flags(enabled=enabled):
        res = {}
        for dt in (torch.float32, torch.float64):
            xx = x.to(dt).clone().requires_grad_(True); gg = gam.to(dt).clone().requires_grad_(True); bb = bet.to(dt).clone().requires_grad_(True)
            y = F.batch_norm(xx, None, None, gg, bb, True, 0.0, 1e-5)
            y.backward(dy.to(dt))
            ww = w.to(dt).clone().requires_grad_(True); x2 = x.to(dt).clone().requires_grad_(True)
            y2 = F.conv2d(x2, ww, None, 1, 1); y2.backward(dy.to(dt))
            res[dt] = (y.detach(), xx.grad, gg.grad, bb.grad, y2.detach(), x2.grad, ww.grad)
        names = ["bn fwd", "bn dx", "bn dgamma", "bn dbeta", "conv fwd", "conv dx", "conv dw"]
        print("library kernels (MIOpen)" if enabled else "native kernels", {n: "%.1e" % rel(a, b) for n, a, b in zip(names, res[torch.float32], res[torch.float64])})

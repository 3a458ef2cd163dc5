"""What a plain copy / fill of one 64 x 1024^2 int32 image achieves on the box (calibration for the label passes)."""
import torch

def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

x = torch.randint(0, 1000, (64, 1024, 1024), dtype=torch.int32, device="cuda")
y = torch.empty_like(x)
z = torch.empty_like(x)
mb = x.numel() * 4 / 1e6
us = t(lambda: y.copy_(x)); print("copy  %.0f MB -> %.0f MB: %.1f us, %.2f TB/s (read + write)" % (mb, mb, us, 2 * mb / us / 1e6 * 1e6 / 1e6))
us = t(lambda: y.fill_(7)); print("fill  %.0f MB: %.1f us, %.2f TB/s" % (mb, us, mb / us))
us = t(lambda: x.sum());  print("read  %.0f MB (sum): %.1f us, %.2f TB/s" % (mb, us, mb / us))
us = t(lambda: torch.add(x, 1, out=z)); print("add   read %.0f MB write %.0f MB: %.1f us, %.2f TB/s" % (mb, mb, us, 2 * mb / us))
xs = [torch.randint(0, 1000, (64, 1024, 1024), dtype=torch.int32, device="cuda") for _ in range(4)]
def rot():
    for i in range(4): torch.add(xs[i], 1, out=xs[(i + 1) % 4])
us = t(rot, 5) / 4; print("add over 4 rotating buffers (no cache reuse): %.1f us each, %.2f TB/s" % (us, 2 * mb / us))

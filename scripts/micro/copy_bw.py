import torch, time
x = torch.empty(8 << 30, dtype=torch.uint8, device="cuda")
y = torch.empty_like(x)
x.fill_(3)
for _ in range(2): y.copy_(x)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): y.copy_(x)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print("DtoD copy of 8 GiB: %.3f ms = %.0f GB/s read + write counted" % (ms, 2 * x.numel() / ms / 1e6))
e0.record()
for _ in range(10): y.fill_(7)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print("fill of 8 GiB: %.3f ms = %.0f GB/s written" % (ms, x.numel() / ms / 1e6))
e0.record()
for _ in range(10): s = x.view(torch.int64).sum()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print("sum of 8 GiB: %.3f ms = %.0f GB/s read" % (ms, x.numel() / ms / 1e6))

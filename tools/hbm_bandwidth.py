"""Practical HBM bandwidth of the box (torch device copy / reduction of 8 GiB of f64): the ceiling the row-sum pass of the
staged assembly is compared with (confirm the vendor figure with a device copy).  Runs on the GPU box."""
import torch, time
x=torch.empty(1<<30, dtype=torch.float64, device='cuda'); y=torch.empty_like(x)
x.fill_(1.0)
for _ in range(3): y.copy_(x)
torch.cuda.synchronize()
a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10): y.copy_(x)
b.record(); torch.cuda.synchronize()
ms=a.elapsed_time(b)/10
print("copy 8 GiB: %.3f ms, %.2f TB/s (read+write)"%(ms, 2*x.numel()*8/ms/1e9))
a.record()
for _ in range(10): s=x.sum()
b.record(); torch.cuda.synchronize()
ms=a.elapsed_time(b)/10
print("read 8 GiB: %.3f ms, %.2f TB/s"%(ms, x.numel()*8/ms/1e9))
a.record()
for _ in range(10): y.add_(x)
b.record(); torch.cuda.synchronize()
ms=a.elapsed_time(b)/10
print("y += x (2 reads + 1 write): %.3f ms, %.2f TB/s"%(ms, 3*x.numel()*8/ms/1e9))

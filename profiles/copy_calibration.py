import torch, time
n = 65536*36*256
a = torch.randn(n, device='cuda'); b = torch.empty_like(a)
for _ in range(3): b.copy_(a)
torch.cuda.synchronize()
e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): b.copy_(a)
e1.record(); torch.cuda.synchronize()
ms=e0.elapsed_time(e1)/20
print('torch copy', ms, 'ms', 2*n*4/ms/1e6, 'GB/s')

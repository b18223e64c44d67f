"""What the memory system of this box sustains for plain streams (torch elementwise kernels over 2.49 GB, the size of one
cfg2 spectrum pass): the practical ceiling the kernels' GB/s are read against next to the 8 TB/s peak.
usage (GPU box): python3 scripts/hbm_microbench.py > gpurun_out/<tag>_hbm_microbench.txt"""
import torch, time
x = torch.empty(2_490_000_000 // 4, dtype=torch.float32, device="cuda")
y = torch.empty_like(x)
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
b = x.numel() * 4
ms = t(lambda: x.zero_()); print("memset  %.3f ms %.2f TB/s" % (ms, b / ms / 1e9))
ms = t(lambda: x.fill_(1.5)); print("fill    %.3f ms %.2f TB/s" % (ms, b / ms / 1e9))
ms = t(lambda: y.copy_(x)); print("copy    %.3f ms %.2f TB/s (r+w)" % (ms, 2 * b / ms / 1e9))
ms = t(lambda: x.sum()); print("sum     %.3f ms %.2f TB/s" % (ms, b / ms / 1e9))
ms = t(lambda: torch.add(x, 1.0, out=y)); print("add     %.3f ms %.2f TB/s (r+w)" % (ms, 2 * b / ms / 1e9))

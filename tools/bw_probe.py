import torch, time
n = 700_000*1000
x = torch.empty(n, dtype=torch.float64, device='cuda')
y = torch.empty(n, dtype=torch.float64, device='cuda')
def t(f, it=10):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/it
ms = t(lambda: x.fill_(1.5)); print("fill 5.6GB: %.3f ms  %.2f TB/s" % (ms, n*8/ms/1e9))
ms = t(lambda: y.copy_(x)); print("copy 5.6GB: %.3f ms  %.2f TB/s (r+w)" % (ms, 2*n*8/ms/1e9))
ms = t(lambda: torch.mul(x, 2.0, out=y)); print("mul  5.6GB: %.3f ms  %.2f TB/s (r+w)" % (ms, 2*n*8/ms/1e9))
z = torch.empty(100_000*1000, dtype=torch.uint8, device='cuda')
ms = t(lambda: z.zero_()); print("zero 100MB u8: %.4f ms" % ms)

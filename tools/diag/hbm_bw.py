import torch, time
def timed(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for mb in (64, 256, 524, 1024, 4096):
    n = mb * 1000 * 1000 // 4
    x = torch.empty(n, device="cuda"); y = torch.empty(n, device="cuda")
    t = timed(lambda: x.fill_(1.0))
    print("fill  %5d MB: %.3f ms  %.2f TB/s written" % (mb, t, mb / 1e3 / t))
    t = timed(lambda: y.copy_(x))
    print("copy  %5d MB: %.3f ms  %.2f TB/s read + %.2f TB/s written" % (mb, t, mb / 1e3 / t, mb / 1e3 / t))
    t = timed(lambda: x.sum())
    print("sum   %5d MB: %.3f ms  %.2f TB/s read" % (mb, t, mb / 1e3 / t))

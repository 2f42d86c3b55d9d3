"""hbm_probe.py -- what this box's HBM sustains for plain streams (torch copy / fill / read-reduce on 2 GiB):
the yardstick for the HBM-bound kernels (Winograd input transform, decode)."""
import torch, time

def timed(f, n=20):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n

N = 1 << 29                                  # 2 GiB of fp32
a = torch.empty(N, device="cuda"); b = torch.randn(N, device="cuda")
t = timed(lambda: a.copy_(b));  print("copy  (read 2 GiB + write 2 GiB): %.2f TB/s" % (2 * 4 * N / t / 1e12))
t = timed(lambda: a.fill_(1.0)); print("fill  (write 2 GiB):              %.2f TB/s" % (4 * N / t / 1e12))
t = timed(lambda: b.sum());      print("sum   (read 2 GiB):               %.2f TB/s" % (4 * N / t / 1e12))

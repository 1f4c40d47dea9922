"""Write bandwidth the chip gives a plain fill / a strided tile-store pattern (context for the Gram kernel's 4.3 TB/s)."""
import torch, time
n = 4 * 1024 ** 3   # doubles: 32 GiB
x = torch.empty(n, dtype=torch.float64, device="cuda")
for name, fn in (("fill_", lambda: x.fill_(1.5)), ("zero_", lambda: x.zero_()), ("mul_ (read+write)", lambda: x.mul_(1.0001))):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print("%-18s %7.2f ms  %6.2f TB/s written" % (name, ms, n * 8 / ms * 1e-9))

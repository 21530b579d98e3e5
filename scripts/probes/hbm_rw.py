"""Probe: what one MI355X sustains for write-only, read-only and copy streams over buffers far larger than the 256 MB
Infinity Cache (torch's own elementwise kernels; measurement tool, not part of the product path).  The short-K / wide-output
1x1 launches of the train step (64 -> 256 at 128^2: 67 MB read, 268 MB written) are 80 % stores: which roof do they sit under?"""
import torch

dev = torch.device("cuda:0")
n = 1 << 30  # 1 Gi elements of bf16 = 2 GiB
a = torch.empty(n, device=dev, dtype=torch.bfloat16)
b = torch.empty(n, device=dev, dtype=torch.bfloat16)


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


gb = n * 2 / 1e9
t = timed(lambda: a.fill_(1.0))
print("write-only  fill_ 2 GiB bf16        {:7.1f} GB/s".format(gb / t))
t = timed(lambda: a.zero_())
print("write-only  zero_ (memset path)     {:7.1f} GB/s".format(gb / t))
t = timed(lambda: torch.sum(a.view(torch.int16), dtype=torch.int64))
print("read-only   sum over 2 GiB          {:7.1f} GB/s".format(gb / t))
t = timed(lambda: b.copy_(a))
print("copy        2 GiB -> 2 GiB          {:7.1f} GB/s (read + write)".format(2 * gb / t))
t = timed(lambda: torch.add(a, a, out=b))
print("1 read : 1 write  add(a, a)         {:7.1f} GB/s".format(2 * gb / t))
q = a[: n // 4]
t = timed(lambda: torch.cat([q, q, q, q], out=b))
print("1 read : 4 writes (cat of a quarter){:7.1f} GB/s (0.5 + 2 GiB... read mostly from cache)".format((gb / 4 + gb) / t))
c = torch.empty(n // 4, device=dev, dtype=torch.bfloat16)
t = timed(lambda: torch.sum(a.view(4, n // 4).float(), 0, out=None) if False else c.copy_(a[: n // 4]))
print("copy        0.5 GiB                 {:7.1f} GB/s (read + write)".format(2 * gb / 4 / t))

"""GPU probe: what HBM streams of the shapes the Winograd transforms produce can reach on this box -- a pure write stream (12 GB fill), a copy (6 GB ->
6 GB) and a 1 : 3.4 read : write stream (the input transform's mix) through torch's own elementwise kernels (16 B per lane).  The transforms of
csrc/winograd.hip are priced against these, not against the 8 TB/s spec.   usage: python tools/hbm_ceiling.py"""
import torch

dev = "cuda"


def timed(fn, n=5):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / n


a = torch.empty(3 * 1024 ** 3, device=dev)          # 12 GiB of float32
ms = timed(lambda: a.zero_())
print(f"| write only (12.9 GB fill) | {ms:.3f} ms | {a.numel() * 4 / ms / 1e9:.2f} TB/s |")
b = a[: a.numel() // 2]
c = a[a.numel() // 2:]
ms = timed(lambda: c.copy_(b))
print(f"| copy (6.4 GB read + 6.4 GB written) | {ms:.3f} ms | {2 * b.numel() * 4 / ms / 1e9:.2f} TB/s |")
src = a[: a.numel() // 4].view(-1, 1)
dst = a[a.numel() // 4:].view(-1, 3)
ms = timed(lambda: dst.copy_(src.expand(-1, 3)))
print(f"| 1 read : 3 written (3.2 GB + 9.7 GB, broadcast copy) | {ms:.3f} ms | {(src.numel() + dst.numel()) * 4 / ms / 1e9:.2f} TB/s |")
ms = timed(lambda: torch.sum(a))
print(f"| read only (12.9 GB reduction) | {ms:.3f} ms | {a.numel() * 4 / ms / 1e9:.2f} TB/s |")

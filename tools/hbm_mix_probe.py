"""What a plain streaming kernel gets from HBM on this box, by read : write mix (torch elementwise kernels on 1 GiB arrays):
   the practical ceiling beside which project_bwd_adam (1.78 GB read, 1.45 GB written per launch) should be read."""
import torch
dev = torch.device("cuda:0")
n = 256 << 20          # floats: 1 GiB per array
a = torch.empty(n, device=dev).normal_()
b = torch.empty(n, device=dev).normal_()
c = torch.empty(n, device=dev).normal_()
def timed(f, reps=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
GB = n * 4 / 1e9
for name, f, r, w in (("read only   (sum)", lambda: a.sum(), 1, 0),
                      ("write only  (fill)", lambda: a.fill_(1.0), 0, 1),
                      ("1R : 1W     (copy)", lambda: a.copy_(b), 1, 1),
                      ("2R : 1W     (add out)", lambda: torch.add(a, b, out=c), 2, 1),
                      ("3R : 3W-ish (addcmul_ in place: 3R 1W)", lambda: a.addcmul_(b, c, value=0.5), 3, 1)):
    t = timed(f)
    print(f"{name:45s} {(r + w) * GB / t / 1e3:6.2f} TB/s  ({t * 1e6:7.1f} us for {(r + w) * GB:.2f} GB)")
# Adam-like: p, m, v read and written, g read (foreach-free single fused op is not in torch: three in-place ops as a proxy is not
# comparable; instead a 3R : 3W copy of three arrays through one kernel)
x = torch.empty(3, n // 4, device=dev).normal_(); y = torch.empty_like(x)
t = timed(lambda: y.copy_(x))
print(f"{'copy of 0.75 GiB':45s} {2 * x.numel() * 4 / 1e9 / t / 1e3:6.2f} TB/s")

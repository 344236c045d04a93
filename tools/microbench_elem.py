"""Achieved HBM bandwidth of the memory-bound BatchNorm kernels against a plain device copy (same tensor sizes).
Usage: tools/microbench_elem.py [npix C]"""
import sys
import time

import torch

sys.path.insert(0, '/root/repo')
import awr_amd
from awr_amd import _lib as L

dev = torch.device('cuda:0')
npix, C = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (128 * 128 * 128, 256)
n = npix * C
g, y, add, out = (torch.randn(n, device=dev) for _ in range(4))
mean, invstd, gam = torch.randn(C, device=dev), torch.rand(C, device=dev) + 0.5, torch.rand(C, device=dev) + 0.5
sc, sh = torch.rand(C, device=dev), torch.randn(C, device=dev)
sums = torch.zeros(16, 2, C, device=dev, dtype=torch.float64)
coef = torch.zeros(3, C, device=dev)
dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)


def timeit(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


s = L.stream()
pass
gb = n * 4 / 1e9
t = timeit(lambda: out.copy_(g))
print("copy                      : %7.1f us  %5.2f TB/s (read + write %.2f GB)" % (t * 1e6, 2 * gb / t / 1e3, 2 * gb))
t = timeit(lambda: torch.add(g, y, out=out))
print("torch add (2 reads 1 write): %7.1f us  %5.2f TB/s" % (t * 1e6, 3 * gb / t / 1e3))
t = timeit(lambda: L.call("awr_bn_apply", L.ptr(g), L.ptr(sc), L.ptr(sh), None, 1, L.ptr(out), npix, C, s))
print("bn_apply (1 read 1 write)  : %7.1f us  %5.2f TB/s" % (t * 1e6, 2 * gb / t / 1e3))
t = timeit(lambda: L.call("awr_bn_apply", L.ptr(g), L.ptr(sc), L.ptr(sh), L.ptr(y), 1, L.ptr(out), npix, C, s))
print("bn_apply + res (2 r 1 w)   : %7.1f us  %5.2f TB/s" % (t * 1e6, 3 * gb / t / 1e3))
for name, a, nb in (("bn_bwd_apply (2 r 1 w)", None, 3), ("bn_bwd_apply + dy_add (3 r 1 w)", add, 4)):
    t = timeit(lambda: L.call("awr_bn_bwd_apply", L.ptr(g), None, L.ptr(y), L.ptr(mean), L.ptr(invstd), L.ptr(gam), L.ptr(sc), L.ptr(sh), L.ptr(sums),
                              L.ptr(coef), npix, C, L.ptr(out), L.ptr(a) if a is not None else None, None, L.ptr(dg), L.ptr(db), 0, 0, s))
    print("%-32s: %7.1f us  %5.2f TB/s" % (name, t * 1e6, nb * gb / t / 1e3))
t = timeit(lambda: L.call("awr_bn_bwd_reduce", L.ptr(g), None, L.ptr(y), L.ptr(mean), L.ptr(invstd), L.ptr(sc), L.ptr(sh), npix, C, L.ptr(sums), 0, s))
print("bn_bwd_reduce (2 reads)    : %7.1f us  %5.2f TB/s" % (t * 1e6, 2 * gb / t / 1e3))

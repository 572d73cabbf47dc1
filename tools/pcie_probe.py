#!/usr/bin/env python3
"""Host<->device copy rates of the box (pinned and pageable, one stream and two streams at once): the ceiling of every host-buffer path."""
import time
import torch
n = 256 << 20
d = torch.empty(n, dtype=torch.uint8, device="cuda:0")
d2 = torch.empty(n, dtype=torch.uint8, device="cuda:0")
hp = torch.empty(n, dtype=torch.uint8).pin_memory()
hp2 = torch.empty(n, dtype=torch.uint8).pin_memory()
hq = torch.empty(n, dtype=torch.uint8)
def rate(f, reps=5):
    f(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        f()
    torch.cuda.synchronize()
    return reps * n / (time.perf_counter() - t) / 1e9
print("H2D pinned   %.1f GB/s" % rate(lambda: d.copy_(hp, non_blocking=True)))
print("D2H pinned   %.1f GB/s" % rate(lambda: hp.copy_(d, non_blocking=True)))
print("H2D pageable %.1f GB/s" % rate(lambda: d.copy_(hq)))
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def two():
    with torch.cuda.stream(s1):
        d.copy_(hp, non_blocking=True)
    with torch.cuda.stream(s2):
        d2.copy_(hp2, non_blocking=True)
print("H2D pinned, two streams at once %.1f GB/s in total" % (2 * rate(two)))
def both():
    with torch.cuda.stream(s1):
        d.copy_(hp, non_blocking=True)
    with torch.cuda.stream(s2):
        hp2.copy_(d2, non_blocking=True)
print("H2D + D2H at once %.1f GB/s in total" % (2 * rate(both)))

#!/usr/bin/env python3
"""Raw PCIe rates of the box (pinned memory, 256 MiB copies): H2D alone, D2H alone, both at once --
the ceiling for ntt_poly_multiply_stream (1 MiB in and 1 MiB out per product at n = 2^17)."""
import time
import torch

N = 256 << 20
h1 = torch.empty(N, dtype=torch.uint8).pin_memory()
h2 = torch.empty(N, dtype=torch.uint8).pin_memory()
d1 = torch.empty(N, dtype=torch.uint8, device="cuda")
d2 = torch.empty(N, dtype=torch.uint8, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def run(h2d, d2h, reps=8):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        if h2d:
            with torch.cuda.stream(s1):
                d1.copy_(h1, non_blocking=True)
        if d2h:
            with torch.cuda.stream(s2):
                h2.copy_(d2, non_blocking=True)
    torch.cuda.synchronize()
    return reps * N / (time.perf_counter() - t0) / 1e9


run(True, True, 2)
print("H2D alone  %.1f GB/s" % run(True, False))
print("D2H alone  %.1f GB/s" % run(False, True))
print("both       %.1f GB/s each way" % run(True, True))

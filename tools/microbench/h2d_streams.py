#!/usr/bin/env python3
"""How fast can frames reach the device?  H2D of 92.4 MB (256 frames of 752x480) from pinned host memory: one copy, the same bytes split over 2 / 4 streams,
and beside a D2H of 19 MB (the previous batch's results) -- the host-input clock of bench.py is bound by this link."""
import time
import torch

N = 256 * 752 * 480
h = torch.empty(N, dtype=torch.uint8).pin_memory()
d = torch.empty(N, dtype=torch.uint8, device="cuda")
hr = torch.empty(19 << 20, dtype=torch.uint8).pin_memory()
dr = torch.empty(19 << 20, dtype=torch.uint8, device="cuda")
streams = [torch.cuda.Stream() for _ in range(4)]
back = torch.cuda.Stream()


def run(k, with_d2h, reps=20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        step = (N + k - 1) // k
        for i in range(k):
            with torch.cuda.stream(streams[i]):
                d[i * step:(i + 1) * step].copy_(h[i * step:(i + 1) * step], non_blocking=True)
        if with_d2h:
            with torch.cuda.stream(back):
                hr.copy_(dr, non_blocking=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    return N / dt / 1e9, dt * 1e3


for k in (1, 2, 4):
    for w in (False, True):
        run(k, w, 3)
        g, ms = run(k, w)
        print(f"{k} stream(s){' + D2H 19 MB' if w else '':14s}: {g:6.1f} GB/s H2D, {ms:.3f} ms per 92.4 MB")

#!/usr/bin/env python3
"""Does a device -> host copy slow down an HBM-bound kernel that runs beside it?  One elementwise torch kernel over 1 GiB on stream A, timed alone and while
stream B copies 16 MiB blocks to pinned host memory of three flavours (hipHostMalloc default / non-coherent / write-combined).  Prints one line per case."""
import ctypes as C
import time
import torch

hip = C.CDLL("libamdhip64.so")
hip.hipHostMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
D2H = 2
torch.cuda.init()
x = torch.ones(256 << 20, dtype=torch.float32, device="cuda")   # 1 GiB
src = torch.ones(16 << 20, dtype=torch.uint8, device="cuda")
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def kernel_ms(n=6):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(sa):
        e0.record()
        for _ in range(n):
            x.mul_(1.0)
        e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / n


torch.cuda.synchronize()
kernel_ms(2)
base = kernel_ms()
print(f"kernel alone: {base:.3f} ms per 2 GiB of traffic ({2.147 / base:.0f} GB/s... x1e3)")
for name, flags in (("default (coherent)", 0x0), ("non-coherent", 0x80000000 | 0x0), ("write-combined", 0x4), ("numa-user|mapped", 0x2)):
    p = C.c_void_p()
    rc = hip.hipHostMalloc(C.byref(p), 16 << 20, flags)
    if rc != 0:
        print(name, "hipHostMalloc failed", rc)
        continue
    torch.cuda.synchronize()
    # copy alone
    t0 = time.perf_counter()
    for _ in range(20):
        hip.hipMemcpyAsync(p, C.c_void_p(src.data_ptr()), 16 << 20, D2H, C.c_void_p(sb.cuda_stream))
    sb.synchronize()
    copy_alone = (time.perf_counter() - t0) / 20 * 1e3
    # kernel beside the copies
    for _ in range(40):
        hip.hipMemcpyAsync(p, C.c_void_p(src.data_ptr()), 16 << 20, D2H, C.c_void_p(sb.cuda_stream))
    k = kernel_ms()
    sb.synchronize()
    print(f"{name:22s}: copy alone {copy_alone:.3f} ms per 16 MiB ({16.78 / copy_alone:.1f} GB/s), kernel beside copies {k:.3f} ms ({k / base:.2f}x)")

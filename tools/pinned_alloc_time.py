#!/usr/bin/env python3
"""What hipHostMalloc / hipMalloc of the BGZF writer's staging sizes cost on this stack (ms):  python tools/pinned_alloc_time.py"""
import ctypes as C
import time
import torch  # noqa: F401  (loads the HIP runtime this process shares with the library)

torch.cuda.init()
hip = C.CDLL("libamdhip64.so")
for mb in (4, 16, 64, 64, 256):
    p = C.c_void_p()
    t0 = time.perf_counter()
    rc = hip.hipHostMalloc(C.byref(p), C.c_size_t(mb << 20), C.c_uint(0))
    t1 = time.perf_counter()
    hip.hipHostFree(p)
    t2 = time.perf_counter()
    d = C.c_void_p()
    rc2 = hip.hipMalloc(C.byref(d), C.c_size_t(mb << 20))
    t3 = time.perf_counter()
    hip.hipFree(d)
    t4 = time.perf_counter()
    print(f"{mb:4d} MB: hipHostMalloc {1e3 * (t1 - t0):7.2f} ms (rc {rc}), hipHostFree {1e3 * (t2 - t1):6.2f} ms; hipMalloc {1e3 * (t3 - t2):6.2f} ms (rc {rc2}), hipFree {1e3 * (t4 - t3):6.2f} ms")

#!/usr/bin/env python3
"""hipMalloc cost by size (big pan tables): python tools/malloc_time.py"""
import time, torch, ctypes
hip = ctypes.CDLL("libamdhip64.so")
torch.cuda.init(); torch.zeros(1, device="cuda")
def one(gb):
    p = ctypes.c_void_p()
    t0 = time.perf_counter(); rc = hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(int(gb * (1 << 30)))); t1 = time.perf_counter()
    hip.hipMemset(p, 0, ctypes.c_size_t(int(gb * (1 << 30)))); hip.hipDeviceSynchronize(); t2 = time.perf_counter()
    return p, t1 - t0, t2 - t1, rc
for gb in (116, 64, 80, 96, 128, 200):
    p, tm, ts, rc = one(gb)
    t3 = time.perf_counter(); hip.hipFree(p); t4 = time.perf_counter()
    print(f"{gb:4d} GB: hipMalloc {tm:.3f} s (rc {rc}), first memset {ts:.3f} s, hipFree {t4-t3:.3f} s")
ps = []
t0 = time.perf_counter()
for i in range(4):
    ps.append(one(29)[0])
print(f"4 x 29 GB: {time.perf_counter()-t0:.3f} s")

cat /sys/fs/cgroup/cpu.max 2>/dev/null; nproc; python - <<'PY'
import os, time, zlib, threading
print("affinity", len(os.sched_getaffinity(0)))
data = os.urandom(1<<16) * 4  # 256 KB incompressible-ish repeated
import numpy as np
rng = np.random.default_rng(0)
buf = (rng.integers(0, 4, 1 << 22, dtype=np.uint8) * 85).tobytes()
def work(n):
    for _ in range(n):
        zlib.compress(buf, 6)
for T in (1, 8, 16, 32, 64, 128):
    th = [threading.Thread(target=work, args=(4,)) for _ in range(T)]
    t0 = time.perf_counter(); [t.start() for t in th]; [t.join() for t in th]; dt = time.perf_counter() - t0
    print(T, "threads:", T * 4 * len(buf) / dt / 1e6, "MB/s")
PY

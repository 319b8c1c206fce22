"""diagnostic: what hipMalloc / hipFree cost by size and by count (step 4 allocates ~45 arrays per call)"""
import ctypes, time
hip = ctypes.CDLL("/opt/rocm/lib/libamdhip64.so")
hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]; hip.hipFree.argtypes = [ctypes.c_void_p]
def alloc(n):
    p = ctypes.c_void_p(); t = time.perf_counter(); rc = hip.hipMalloc(ctypes.byref(p), n); dt = time.perf_counter() - t; assert rc == 0, rc; return p, dt
p, _ = alloc(1 << 20); hip.hipFree(p)
for sz in (64 << 20, 256 << 20, 1 << 30, 4 << 30, 12 << 30):
    p, dt = alloc(sz); t = time.perf_counter(); hip.hipFree(p); df = time.perf_counter() - t
    print(f"{sz / 2**20:8.0f} MiB: malloc {1e3 * dt:7.2f} ms, free {1e3 * df:7.2f} ms")
ps = []; t = time.perf_counter()
for i in range(45): ps.append(alloc(256 << 20)[0])
print(f"45 x 256 MiB: malloc {1e3 * (time.perf_counter() - t):.1f} ms", end=""); t = time.perf_counter()
for p in ps: hip.hipFree(p)
print(f", free {1e3 * (time.perf_counter() - t):.1f} ms")

"""micro: what pinning caller memory in place costs (hipHostRegister / Unregister of 4 MiB blocks and of one large range) against
the memcpy into pinned staging it would replace."""
import ctypes as C, time, numpy as np
hip = C.CDLL("libamdhip64.so")
hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]; hip.hipHostUnregister.argtypes = [C.c_void_p]
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]; hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
hip.hipHostMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
n, bsz = 256, 4 << 20
blocks = [np.random.randint(0, 255, bsz, dtype=np.uint8) for _ in range(n)]
d = C.c_void_p(); assert hip.hipMalloc(C.byref(d), n * bsz) == 0
pin = C.c_void_p(); assert hip.hipHostMalloc(C.byref(pin), n * bsz, 0) == 0
for rep in range(2):
    t0 = time.perf_counter()
    for b in blocks: assert hip.hipHostRegister(b.ctypes.data, bsz, 0) == 0
    t1 = time.perf_counter()
    for i, b in enumerate(blocks): hip.hipMemcpy(d.value + i * bsz, b.ctypes.data, bsz, 1)
    t2 = time.perf_counter()
    for b in blocks: hip.hipHostUnregister(b.ctypes.data)
    t3 = time.perf_counter()
    for i, b in enumerate(blocks): C.memmove(pin.value + i * bsz, b.ctypes.data, bsz)
    t4 = time.perf_counter()
    hip.hipMemcpy(d.value, pin.value, n * bsz, 1)
    t5 = time.perf_counter()
    print("per 4 MiB block: register %.0f us, H2D from registered %.0f us, unregister %.0f us | memcpy to staging (1 thread) %.0f us, H2D from staging %.0f us"
          % ((t1 - t0) / n * 1e6, (t2 - t1) / n * 1e6, (t3 - t2) / n * 1e6, (t4 - t3) / n * 1e6, (t5 - t4) / n * 1e6))
big = np.random.randint(0, 255, n * bsz, dtype=np.uint8)
t0 = time.perf_counter(); assert hip.hipHostRegister(big.ctypes.data, n * bsz, 0) == 0; t1 = time.perf_counter()
hip.hipMemcpy(d.value, big.ctypes.data, n * bsz, 1); t2 = time.perf_counter(); hip.hipHostUnregister(big.ctypes.data); t3 = time.perf_counter()
print("one 1 GiB range: register %.1f ms, H2D %.1f ms, unregister %.1f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))

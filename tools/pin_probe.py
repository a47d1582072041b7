import ctypes, time, numpy as np
hip = ctypes.CDLL("/opt/rocm/lib/libamdhip64.so")
hip.hipMalloc.argtypes=[ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
hip.hipMemcpy.argtypes=[ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
hip.hipHostRegister.argtypes=[ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint]
hip.hipHostUnregister.argtypes=[ctypes.c_void_p]
hip.hipHostMalloc.argtypes=[ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_uint]
hip.hipHostFree.argtypes=[ctypes.c_void_p]
n = 256 << 20
d = ctypes.c_void_p(); assert hip.hipMalloc(ctypes.byref(d), n) == 0
hip.hipDeviceSynchronize()
for rep in range(3):
    a = np.empty(n // 8); t0 = time.perf_counter(); r = hip.hipMemcpy(a.ctypes.data, d, n, 2); t1 = time.perf_counter()
    print("fresh pageable D2H 256 MB: %.1f ms (%.1f GB/s) rc=%d" % ((t1-t0)*1e3, n/(t1-t0)/1e9, r))
    t0 = time.perf_counter(); r = hip.hipMemcpy(a.ctypes.data, d, n, 2); t1 = time.perf_counter()
    print("touched pageable D2H: %.1f ms (%.1f GB/s)" % ((t1-t0)*1e3, n/(t1-t0)/1e9))
    b = np.empty(n // 8); t0 = time.perf_counter(); r1 = hip.hipHostRegister(b.ctypes.data, n, 0); t1 = time.perf_counter()
    r2 = hip.hipMemcpy(b.ctypes.data, d, n, 2); t2 = time.perf_counter(); r3 = hip.hipHostUnregister(b.ctypes.data); t3 = time.perf_counter()
    print("register %.1f ms + copy %.1f ms (%.1f GB/s) + unregister %.1f ms rc=%d %d %d" % ((t1-t0)*1e3, (t2-t1)*1e3, n/(t2-t1)/1e9, (t3-t2)*1e3, r1, r2, r3))
    p = ctypes.c_void_p(); t0 = time.perf_counter(); r1 = hip.hipHostMalloc(ctypes.byref(p), n, 0); t1 = time.perf_counter()
    r2 = hip.hipMemcpy(p, d, n, 2); t2 = time.perf_counter(); hip.hipHostFree(p); t3 = time.perf_counter()
    print("hipHostMalloc %.1f ms + copy %.1f ms (%.1f GB/s) + free %.1f ms" % ((t1-t0)*1e3, (t2-t1)*1e3, n/(t2-t1)/1e9, (t3-t2)*1e3))
# the asynchronous form on a non-blocking stream, as the library issues it
st = ctypes.c_void_p(); hip.hipStreamCreateWithFlags.argtypes=[ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint]; hip.hipStreamCreateWithFlags(ctypes.byref(st), 1)
hip.hipMemcpyAsync.argtypes=[ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
hip.hipStreamSynchronize.argtypes=[ctypes.c_void_p]
for rep in range(3):
    a = np.empty(n // 8); t0 = time.perf_counter(); hip.hipMemcpyAsync(a.ctypes.data, d, n, 2, st); hip.hipStreamSynchronize(st); t1 = time.perf_counter()
    print("fresh pageable D2H, hipMemcpyAsync + sync: %.1f ms (%.1f GB/s)" % ((t1-t0)*1e3, n/(t1-t0)/1e9))
    for piece in (32 << 20,):
        a = np.empty(n // 8); t0 = time.perf_counter()
        for o in range(0, n, piece): hip.hipMemcpyAsync(a.ctypes.data + o, d.value + o, piece, 2, st)
        hip.hipStreamSynchronize(st); t1 = time.perf_counter()
        print("fresh pageable D2H in 32 MB pieces (async): %.1f ms (%.1f GB/s)" % ((t1-t0)*1e3, n/(t1-t0)/1e9))

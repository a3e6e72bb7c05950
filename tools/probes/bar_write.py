#!/usr/bin/env python3
"""Probe: can the host store straight into device memory (large BAR)?  hipExtMallocWithFlags(fine-grained) / hipMalloc + a host
memmove into the returned pointer, in a CHILD process (a fault there is a crash of the child only)."""
import ctypes as C, subprocess, sys, time
if len(sys.argv) > 1:
    flag = int(sys.argv[1])
    hip = C.CDLL("libamdhip64.so")
    p = C.c_void_p()
    if flag < 0:
        rc = hip.hipMalloc(C.byref(p), 1 << 20)
    else:
        rc = hip.hipExtMallocWithFlags(C.byref(p), C.c_size_t(1 << 20), C.c_uint(flag))
    print("alloc rc", rc, hex(p.value or 0), flush=True)
    import numpy as np
    src = np.arange(2048, dtype=np.int64)
    t0 = time.perf_counter()
    for _ in range(100):
        C.memmove(p.value, src.ctypes.data, 16384)
    dt = (time.perf_counter() - t0) / 100 * 1e6
    print(f"host memmove of 16 KB into the allocation: {dt:.2f} us", flush=True)
    back = np.zeros(2048, dtype=np.int64)
    rc = hip.hipMemcpy(C.c_void_p(back.ctypes.data), p, C.c_size_t(16384), C.c_int(2))
    print("copy back rc", rc, "equal", bool((back == src).all()), flush=True)
else:
    for flag in (-1, 1, 3):
        r = subprocess.run([sys.executable, __file__, str(flag)], capture_output=True, text=True, timeout=120)
        print(f"flag {flag}: exit {r.returncode}\n{r.stdout}{r.stderr[-300:]}")

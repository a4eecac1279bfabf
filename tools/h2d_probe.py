"""probe: H2D copy time of one frame's worth of bytes from different kinds of host memory (HIP events on a copy stream)"""
import ctypes as C
import sys
import time

import torch

hip = C.CDLL("libamdhip64.so")
hip.hipHostMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
hip.hipStreamCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
hip.hipStreamSynchronize.argtypes = [C.c_void_p]
hip.hipEventCreate.argtypes = [C.POINTER(C.c_void_p)]
hip.hipEventRecord.argtypes = [C.c_void_p, C.c_void_p]
hip.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), C.c_void_p, C.c_void_p]
hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]

torch.cuda.init()
n = 640 * 480 * 4
dst = torch.empty(2 * n, dtype=torch.uint8, device="cuda")
s = C.c_void_p()
assert hip.hipStreamCreateWithFlags(C.byref(s), 1) == 0
e0, e1 = C.c_void_p(), C.c_void_p()
hip.hipEventCreate(C.byref(e0)); hip.hipEventCreate(C.byref(e1))


def timed(src_ptr, nbytes, reps=20):
    best = 1e9
    for _ in range(reps):
        hip.hipEventRecord(e0, s)
        hip.hipMemcpyAsync(dst.data_ptr(), src_ptr, nbytes, 1, s)
        hip.hipEventRecord(e1, s)
        hip.hipStreamSynchronize(s)
        ms = C.c_float()
        hip.hipEventElapsedTime(C.byref(ms), e0, e1)
        best = min(best, ms.value)
    return best * 1e3


p = C.c_void_p()
assert hip.hipHostMalloc(C.byref(p), 2 * n, 0) == 0
print("hipHostMalloc default     1.2 MB: %.1f us   2.4 MB: %.1f us" % (timed(p.value, n), timed(p.value, 2 * n)))
t = torch.empty(2 * n, dtype=torch.uint8).pin_memory()
print("torch pin_memory()        1.2 MB: %.1f us   2.4 MB: %.1f us" % (timed(t.data_ptr(), n), timed(t.data_ptr(), 2 * n)))
big = torch.empty((50, 2 * n), dtype=torch.uint8).pin_memory()
print("torch pinned, row 37      1.2 MB: %.1f us   2.4 MB: %.1f us" % (timed(big[37].data_ptr(), n), timed(big[37].data_ptr(), 2 * n)))
u = torch.empty(2 * n, dtype=torch.uint8)
print("pageable                  1.2 MB: %.1f us   2.4 MB: %.1f us" % (timed(u.data_ptr(), n), timed(u.data_ptr(), 2 * n)))
t0 = time.perf_counter()
for _ in range(50):
    dst.copy_(t, non_blocking=True)
torch.cuda.synchronize()
print("torch copy_ non_blocking 2.4 MB: %.1f us each (host wall)" % (1e6 * (time.perf_counter() - t0) / 50))

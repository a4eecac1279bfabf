"""probe: does the H2D rate of a 1.2 MB copy depend on WHICH stream (copy engine) carries it, and is it stable per stream?"""
import ctypes as C
import torch

hip = C.CDLL("libamdhip64.so")
hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
hip.hipStreamCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
hip.hipStreamSynchronize.argtypes = [C.c_void_p]
hip.hipEventCreate.argtypes = [C.POINTER(C.c_void_p)]
hip.hipEventRecord.argtypes = [C.c_void_p, C.c_void_p]
hip.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), C.c_void_p, C.c_void_p]

torch.cuda.init()
n = 640 * 480 * 4
dst = torch.empty(4 * n, dtype=torch.uint8, device="cuda")
src = torch.empty((64, n), dtype=torch.uint8).pin_memory()
streams = []
for i in range(10):
    s = C.c_void_p()
    assert hip.hipStreamCreateWithFlags(C.byref(s), 1) == 0
    streams.append(s)
e0, e1 = C.c_void_p(), C.c_void_p()
hip.hipEventCreate(C.byref(e0)); hip.hipEventCreate(C.byref(e1))


def timed(s, reps=16):
    ts = []
    for r in range(reps):
        hip.hipEventRecord(e0, s)
        hip.hipMemcpyAsync(dst.data_ptr(), src[r].data_ptr(), n, 1, s)
        hip.hipEventRecord(e1, s)
        hip.hipStreamSynchronize(s)
        ms = C.c_float()
        hip.hipEventElapsedTime(C.byref(ms), e0, e1)
        ts.append(ms.value * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[0], ts[-1]


for rnd in range(3):
    print("round", rnd, " ".join("s%d:%.0f/%.0f/%.0f" % ((i,) + timed(s)) for i, s in enumerate(streams)))
# two streams at once: the pairs (0,1), (2,3) ...
for a in range(0, 10, 2):
    sa, sb = streams[a], streams[a + 1]
    import time
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for r in range(32):
        hip.hipMemcpyAsync(dst.data_ptr(), src[r].data_ptr(), n, 1, sa)
        hip.hipMemcpyAsync(dst.data_ptr() + 2 * n, src[32 + r].data_ptr(), n, 1, sb)
    hip.hipStreamSynchronize(sa); hip.hipStreamSynchronize(sb)
    dt = time.perf_counter() - t0
    print("pair (%d,%d): %.1f us per 2 x 1.2 MB -> %.1f GB/s" % (a, a + 1, 1e6 * dt / 32, 2 * n * 32 / dt / 1e9))

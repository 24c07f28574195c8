"""How fast are strided device-to-host copies (one TRGSW row set of a circuit-bootstrap batch: 1024 rows of 32 KiB at a pitch of 256 KiB) against one contiguous copy?"""
import ctypes as C, time
hip = C.CDLL("libamdhip64.so")
def chk(r):
    assert r == 0, r
N = 1024; row = 32768; pitch = 8 * row
d = C.c_void_p(); h = C.c_void_p(); st = C.c_void_p()
chk(hip.hipMalloc(C.byref(d), C.c_size_t(N * pitch))); chk(hip.hipHostMalloc(C.byref(h), C.c_size_t(N * pitch), 0)); chk(hip.hipStreamCreate(C.byref(st)))
hip.hipMemcpy2DAsync.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p]
hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
def t(f, reps=5):
    f(); chk(hip.hipStreamSynchronize(st))
    t0 = time.perf_counter()
    for _ in range(reps): f()
    chk(hip.hipStreamSynchronize(st))
    return (time.perf_counter() - t0) / reps * 1e3
for rows in (256, 1024):
    ms = t(lambda: chk(hip.hipMemcpy2DAsync(h, pitch, d, pitch, row, rows, 2, st)))
    print("2D  %4d rows of 32 KiB, pitch 256 KiB: %.3f ms = %.1f GB/s" % (rows, ms, rows * row / ms / 1e6))
    ms = t(lambda: chk(hip.hipMemcpyAsync(h, d, rows * row, 2, st)))
    print("1D  %4d x 32 KiB contiguous:           %.3f ms = %.1f GB/s" % (rows, ms, rows * row / ms / 1e6))
ms = t(lambda: chk(hip.hipMemcpyAsync(h, d, N * pitch, 2, st)))
print("1D  whole 268 MB: %.3f ms = %.1f GB/s" % (ms, N * pitch / ms / 1e6))

"""A/B of library builds on ONE box: wall time per filtered frame (C3 workload) for each given .so.
usage: ab_frame.py lib1.so lib2.so ...   (alternates between them, several rounds)"""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
pkg = entry.load_package()
W, H, n = int(os.environ.get("AB_W", 1920)), int(os.environ.get("AB_H", 1080)), int(os.environ.get("AB_N", 100_000_000))
scene = os.environ.get("AB_SCENE", "room_shell")
poses = [np.ascontiguousarray(pkg.orbit_projection(k, W, H), dtype=np.float32).reshape(16) for k in range(120)]
libs = []
for path in sys.argv[1:]:
    L = C.CDLL(os.path.abspath(path))
    L.rtr_create.argtypes = [C.POINTER(C.c_void_p), C.c_int]
    L.rtr_generate_synthetic.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64]
    L.rtr_set_resolution.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.rtr_render.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.rtr_synchronize.argtypes = [C.c_void_p]
    L.rtr_destroy.argtypes = [C.c_void_p]
    ctx = C.c_void_p()
    assert L.rtr_create(C.byref(ctx), 0) == 0
    assert L.rtr_generate_synthetic(ctx, 1 if scene == "room_shell" else 0, 0xC0FFEE03, 0, n, n) == 0
    if os.environ.get("AB_REORDER"):
        L.rtr_reorder_points.argtypes = [C.c_void_p]
        assert L.rtr_reorder_points(ctx) == 0
    assert L.rtr_set_resolution(ctx, W, H) == 0
    libs.append((os.path.basename(path), L, ctx))
for rnd in range(3):
    for name, L, ctx in libs:
        for k in range(10):
            L.rtr_render(ctx, poses[k].ctypes.data_as(C.c_void_p), 1)
        L.rtr_synchronize(ctx)
        t0 = time.perf_counter()
        for k in range(100):
            L.rtr_render(ctx, poses[10 + k].ctypes.data_as(C.c_void_p), 1)
        L.rtr_synchronize(ctx)
        dt = time.perf_counter() - t0
        print("round %d %-28s %.4f ms/frame" % (rnd, name, dt * 10), flush=True)
for name, L, ctx in libs:
    L.rtr_destroy(ctx)

"""Timing-experiment aid (RTR_LIB_VARIANT=xp): device time stamps inside T1's epilogue and three tile workgroups."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
pkg = entry.load_package()
W, H, n = 1920, 1080, int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
FILT = (sys.argv[2] != "0") if len(sys.argv) > 2 else True
p = pkg.Projector(0); p.set_resolution(W, H); p.generate_synthetic("room_shell", 0xC0FFEE03, 0, n, n)
lib = p._lib
lib.rtr_debug_stamps.argtypes = [C.c_void_p, C.c_void_p]
for k in range(8):
    p.render(pkg.orbit_projection(k, W, H), FILT)
p.synchronize()
out = np.zeros(64, np.uint64)
for k in range(8, 12):
    p.render(pkg.orbit_projection(k, W, H), FILT); p.synchronize()
    lib.rtr_debug_stamps(p._ctx, out.ctypes.data_as(C.c_void_p))
    t = out.astype(np.int64)
    us = lambda a, b: round((t[b] - t[a]) / 100.0, 2)
    print("T1 tail: wg-done->epi start %s, consts %s, first loads %s, stores+rest of batches %s, stats barrier %s, rest %s  | total %s us" %
          (us(0, 1), us(1, 5), us(5, 6), us(6, 2), us(2, 3), us(3, 4), us(0, 4)))
    u = out  # T1's waves: first start, first / average / last end, long chunks per wave
    if u[60]:
        start = int(~u[59] & np.uint64(0xFFFFFFFFFFFFFFFF)); first_end = int(~u[56] & np.uint64(0xFFFFFFFFFFFFFFFF))
        print("T1 waves %d: first end %.1f us, average end %.1f, last end %.1f after the first start | long chunks per wave: average %.2f, most %d" %
              (int(u[60]), (first_end - start) / 100.0, (int(u[58]) / int(u[60]) - start) / 100.0, (int(u[57]) - start) / 100.0,
               int(u[62]) / int(u[60]), int(u[61])))
        print("   one wave in eight by its own duration (<60, 60-80, ... 180-200, >=200 us):", [int(v) for v in u[40:48]])
    for name, b in (("wg5", 24), ("wg1900", 32)):  # k_filter4 (the copy of these words is asynchronous: read after the sync below)
        print("  F2 %s: LUT + frame loads requested + levels staged %s, partials folded + barrier %s, three up steps %s, final step %s | total %s" %
              (name, us(b, b + 1), us(b + 1, b + 2), us(b + 2, b + 3), us(b + 3, b + 4), us(b, b + 4)))
    for name, b in (("wg5", 8),):
        print("     %s detail: loads issued->first arrives %s, ->all %s, min-merge+atomics %s | acc body %s, barrier %s, overflow check+barrier %s" %
              (name, us(b + 1, b + 11), us(b + 11, b + 12), us(b + 12, b + 2), us(b + 3, b + 9), us(b + 9, b + 10), us(b + 10, b + 4)))
        print("  T4 %s: start(after T1 end) %s | record+seg+init %s, min pass %s, barrier %s, acc %s, writeout %s, image %s, pyramid %s | total %s" %
              (name, us(4, b), us(b, b + 1), us(b + 1, b + 2), us(b + 2, b + 3), us(b + 3, b + 4), us(b + 4, b + 5), us(b + 5, b + 6), us(b + 6, b + 7), us(b, b + 7)))
p.close()

#!/usr/bin/env python3
"""GPU box: times the read-only stream probe variants (see k_stream_probe)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
pkg = entry.load_package()
p = pkg.Projector(0)
n, W, H = 100_000_000, 1920, 1080
p.set_resolution(W, H)
p.generate_synthetic("room_shell", 0xC0FFEE03, 0, n, n)
poses = [pkg.orbit_projection(k, W, H) for k in range(30)]
for rnd in range(2):
    for v in (0, 1, 2, 3, 4):
        p.set_option("probe_variant", v)
        for k in range(3):
            p.stream_probe(poses[k])
        p.synchronize(); p.timing_enable(True); p.timing_reset()
        for k in range(20):
            p.stream_probe(poses[3 + k])
        ms, cnt = p.timing()["probe"]
        p.timing_enable(False)
        print(json.dumps({"variant": v, "us": round(ms / cnt * 1e3, 1), "TBps": round(1.2e9 / (ms / cnt * 1e-3) / 1e12, 2)}), flush=True)

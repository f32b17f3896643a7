"""Entry statistics of the tile kernel on the C3 workload (GPU box): per pose entries, heaviest tile, split tiles.
usage: tile_stats.py [points]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
pkg = entry.load_package()
W, H = 1920, 1080
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
p = pkg.Projector(0)
p.generate_synthetic("room_shell", 0xC0FFEE03, 0, n, n)
p.set_resolution(W, H)
for k in range(0, 110, 10):
    p.render(pkg.orbit_projection(k, W, H), True)
    st = p.frame_stats()
    print(k, {a: st[a] for a in ("entries", "heaviest_tile", "split_tiles", "split_items", "colour_chunks")}, flush=True)
p.close()

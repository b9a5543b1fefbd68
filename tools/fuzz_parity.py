"""Fuzz against the CPU oracle for a given time (needs a GPU; the oracle is the checker here exactly as in tests/).
    python tools/fuzz_parity.py [seed] [seconds]            see tests/fuzz_common.py for what a frame is
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import __graft_entry__ as g
from oracle import oracle as orc, ref_loader
import fuzz_common

r = g.load_package()
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
with r.Context(0) as ctx:
    n_frames, n_path, n_dormant, worst = fuzz_common.run(r, orc, ref_loader, ctx, seed, seconds, verbose=True,
                                                         many_samples=bool(int(os.environ.get("FUZZ_MANY_SAMPLES", "0"))))
print(f"ok: {n_frames} frames ({n_path} path-traced, {n_dormant} with single triangles / orthographic rays), worst colour difference {worst:.2e}")

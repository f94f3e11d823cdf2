"""Failure envelope of the overlap stage under camera rotation and zoom: for each motion the overlap
ratio found (CPU oracle, which the device equals bit for bit -- tests/test_overlap_gpu.py) against the ratio of the TRUE
homography pushed through the same overlapArea.  Runs without a GPU.
    python tools/overlap_envelope.py [rows cols]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import _oracle
from uwimageproc_amd import synth

_a = [a for a in sys.argv[1:] if not a.startswith('--')]
rows, cols = (int(_a[0]), int(_a[1])) if len(_a) > 1 else (1080, 1920)
orc = _oracle.load()
upright = "--upright" in sys.argv
print("descriptor:", "upright (round 2)" if upright else "oriented")
print(f"{'theta':>6} {'scale':>6} | {'ratio':>8} {'truth':>8} {'err':>7} | kp_cur kp_key good inl | verdict")
for theta, scale in [(0, 1), (0.5, 1), (1, 1.01), (2, 1), (5, 1), (10, 1), (15, 1), (20, 1), (30, 1), (45, 1), (90, 1), (180, 1),
                     (0, 0.8), (0, 0.9), (0, 1.1), (0, 1.25), (10, 1.1), (20, 0.9), (45, 1.25),
                     # the reference's SURF spans four octaves -- chart the zoom range an altitude change covers
                     (0, 0.5), (0, 0.67), (0, 1.5), (0, 2.0), (20, 0.5), (20, 0.67), (20, 1.5), (20, 2.0)]:
    key, cur, H = synth.uw_motion_pair(rows, cols, theta, scale)
    Hw = synth.to_working_homography(H, cols)
    truth, _ = orc.overlapArea(Hw, 640, 480)
    r, info, Hest = orc.calcOverlap(key, cur, 640, 480, seed=1, upright=upright)
    ok = "ok" if abs(r - truth) <= 0.01 else "FAIL"
    print(f"{theta:6.1f} {scale:6.2f} | {r:8.4f} {truth:8.4f} {r - truth:+7.4f} | {info[0]:5d} {info[1]:6d} {info[2]:4d} {info[3]:3d} | {ok}", flush=True)

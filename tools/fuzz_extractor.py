"""One-off fuzz of the extractor against the oracle over unusual constructor parameters / image sizes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ceres_mono_orb_slam2_amd import synth, ORBextractor
from ceres_mono_orb_slam2_amd._lib import OrbHipError
from oracle import pyoracle as po
bad = 0; errs = {}
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 80):
    rng = np.random.default_rng(7000 + seed)
    w = int(rng.choice([46, 47, 64, 97, 131, 320, 641, 1000, 2047])); h = int(rng.choice([46, 50, 77, 120, 241, 480, 700]))
    nf = int(rng.choice([1, 5, 30, 100, 777, 2500])); scale = float(rng.choice([1.05, 1.2, 1.41, 2.0])); nl = int(rng.integers(1, 9))
    ini = int(rng.choice([5, 20, 40])); mn = int(rng.choice([1, 5, 7])); mn = min(mn, ini)
    fam = ["blocks", "checker", "flat"][seed % 3]
    img = synth.make_frame(9000 + seed, w, h, fam)
    tag = (w, h, nf, scale, nl, ini, mn, fam)
    try:
        E = po.OracleExtractor(nf, scale, nl, ini, mn)
        ok, od = E.extract(img)
    except Exception as e:
        print("oracle failed", tag, e); continue
    try:
        ex = ORBextractor(nf, scale, nl, ini, mn)
        k, d = ex(img)
    except OrbHipError as e:
        msg = str(e).split(":")[-1][:60]
        errs[msg] = errs.get(msg, 0) + 1; print("GPU error", tag, len(ok), msg)
        continue
    same = len(k) == len(ok) and all(np.array_equal(k[f], ok[f]) for f in ("x", "y", "size", "response", "octave", "class_id", "angle")) and np.array_equal(d, od)
    if not same:
        bad += 1; print("MISMATCH", tag, len(k), len(ok))
print("mismatches", bad, "loud errors", errs)

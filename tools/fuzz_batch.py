"""Longer random sweep of the BATCH extractor path (k_resize_mfma / k_resize, k_blur7_mfma with its edge chunks, strided rows) against
the oracle: python tools/fuzz_batch.py [first_seed=100] [count=100].  The body is tests/test_gpu_extractor.py's
test_random_configurations_batches_bit_exact with other seeds."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pyoracle
pyoracle.lib()
from tests.test_gpu_extractor import test_random_configurations_batches_bit_exact as body
first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
t0 = time.time(); bad = 0
for s in range(first, first + count):
    try:
        body(pyoracle, s)
    except AssertionError as e:
        bad += 1; print("seed", s, "FAILED:", str(e)[:300], flush=True)
print("fuzz_batch: %d seeds from %d, %d failures, %.0f s" % (count, first, bad, time.time() - t0))
sys.exit(1 if bad else 0)

"""More seeds of tests/test_gpu_parity.py::test_random_scenes_fuzz than the test suite runs (on the GPU box): python tools/fuzz_more.py [first] [last]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tests.test_gpu_parity as t
bad = 0
first = int(sys.argv[1]) if len(sys.argv) > 1 else 6
last = int(sys.argv[2]) if len(sys.argv) > 2 else 60
for seed in range(first, last):
    try:
        t.test_random_scenes_fuzz(seed)
    except Exception as e:
        bad += 1
        print("seed", seed, "FAILED:", str(e)[:300], flush=True)
print("done, failures:", bad)

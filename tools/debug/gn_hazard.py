"""One arm of tools/debug/gn_hazard.sh: the scenario of tests/test_gpu_concurrency.py (GroupNorm backward beside the weight-gradient GEMM /
the big-tile conv of a second context), repeated; prints how many of the noisy runs differ from the quiet one."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import test_gpu_concurrency as T
for neighbour in ("wgrad", "conv_big"):
    bad = []
    for rep in range(3):
        bad += T._scenario(neighbour)
    print(f"hazard arm {sys.argv[1]} nth {sys.argv[2]} neighbour {neighbour}: {len(bad)} of 27 noisy runs differ", bad[:6], flush=True)

"""The randomised parity tests of tests/test_gpu_fuzz.py over OTHER seeds than the committed ones (a one-off campaign on a
GPU box, not part of the suite): every `np.random.default_rng(seed)` of a round draws from seed + 1000003 * round.
usage: fuzz_campaign.py [rounds] [first_round] [substring of the test names to run]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "fp8-emulation-toolkit_amd"))
import conftest  # noqa: E402,F401  (puts the oracle on the path as the tests do)
import test_gpu_fuzz as fz  # noqa: E402
from mpemu.pytquant.hip import _binding as hip  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
first = int(sys.argv[2]) if len(sys.argv) > 2 else 1
hip.lib()
orig = np.random.default_rng
only = sys.argv[3] if len(sys.argv) > 3 else ""
tests = [getattr(fz, n) for n in dir(fz) if n.startswith("test_") and only in n]
for rnd in range(first, first + rounds):
    np.random.default_rng = lambda s=None, _r=rnd: orig((0 if s is None else int(s)) + 1000003 * _r)
    for t in tests:
        t0 = time.time()
        try:
            t(hip)
        finally:
            hip.lib().fpemu_test_set_max_blocks_(0)
        print("round %d %s ok (%.1f s)" % (rnd, t.__name__, time.time() - t0), flush=True)
np.random.default_rng = orig
print("campaign: %d rounds x %d tests, all equal to the oracle" % (rounds, len(tests)))

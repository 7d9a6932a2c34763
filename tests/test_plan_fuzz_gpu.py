"""tools/plan_fuzz.py as a test: random input sizes (the streaming stem's multiples of 4 and the others) and batches, the default plan
against the least fused fp32-MFMA plan and the uint8 entry against the float one -- the reference's graph is one function whatever
kernels run it (facerec_test.py:120)."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_default_plan_equals_the_unfused_plan_on_random_shapes():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "plan_fuzz.py"), "16", "11"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    assert "worst relative difference" in r.stdout and "over 16 cases" in r.stdout

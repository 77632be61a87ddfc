"""CPU: the tight binning's geometry (gaussian-splatting-wgpu_amd/csrc/gs_tight.h -- the row items of the product path, compiled for the host)
against the oracle's exact contribution masks (oracle.instance_masks): no instance that contributes to a pixel under the
canonical arithmetic may be dropped, and every kept instance's sub-block mask must cover its contributing blocks.  The GPU
tests prove the same for the kernels (tests/gpu_checks.py::check_product_lists); this one runs without a GPU."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("case", [(20000, 640, 360, 16, 3, 1.0), (15000, 320, 200, 8, 5, 1.0), (15000, 640, 480, 32, 9, 1.0),
                                  (6000, 256, 256, 16, 40, 2.5)])
def test_row_items_drop_nothing_that_contributes(case):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "tight_check", "run.py")] + [str(x) for x in case],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    m = re.search(r"reference instances (\d+) .*\(tight total (\d+)\).*dropped-but-contributing (\d+) \| mask misses (\d+)", out.stdout)
    assert m, out.stdout[-500:]
    ref, kept, dropped, misses = (int(x) for x in m.groups())
    assert dropped == 0 and misses == 0
    assert 0 < kept <= ref  # an ordered subset of the reference's instances (and a real reduction on these scenes)

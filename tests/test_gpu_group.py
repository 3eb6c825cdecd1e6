"""Single-process multi-device (``devices=[...]`` -> pcr_group_*, csrc/group.hip) on the one GPU of the test box: N contexts of
device 0 (the C ABI lets a device id repeat for exactly this).  The checks live in tests/group_check.py and run in a fresh
process per N, because N streams that wait for each other inside kernels need N hardware queues (GPU_MAX_HW_QUEUES, read when
the HIP runtime starts).  Reference seam: the single-process call order of registration.py:28,71 / demo_matching.py:147-152."""
import os
import subprocess
import sys

import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [1, 2, 4, 8])
def test_group_n_contexts_one_gpu(n):
    env = dict(os.environ, GPU_MAX_HW_QUEUES=str(4 if n <= 2 else 4 * n), PYTHONPATH=REPO)
    r = subprocess.run([sys.executable, os.path.join(REPO, "tests", "group_check.py"), str(n)], env=env, capture_output=True,
                       text=True, timeout=900)
    print(r.stdout[-3000:])
    assert r.returncode == 0 and "GROUP_CHECK_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])

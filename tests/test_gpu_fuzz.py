"""Short runs of the randomised campaigns under benchmarks/ (the long runs are recorded in
profiles/r01_fuzz_*_summary.jsonl): random models, shapes, error scales, prior boxes, batch
sizes, ensemble sizes, thinning / chunking -- against the oracle and against host replays."""

import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('script,extra', [('fuzz_parity.py', ['--cases', '200', '--seed', '21']),
                                          ('fuzz_parity.py', ['--cases', '100', '--seed', '22', '--widen', '2']),
                                          ('fuzz_sampler.py', ['--cases', '60', '--seed', '21']),
                                          ('fuzz_batch.py', ['--cases', '40', '--seed', '21']),
                                          ('extreme_shapes.py', ['--sizes', '100,1024,4096'])])
def test_randomised_campaign(script, extra):
    run = subprocess.run([sys.executable, os.path.join(ROOT, 'benchmarks', script)] + extra,
                         capture_output=True, text=True, timeout=600)
    last = run.stdout.strip().splitlines()[-1] if run.stdout.strip() else ''
    assert run.returncode == 0, (last, run.stderr[-2000:])
    assert '"summary": true' in last

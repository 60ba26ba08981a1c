"""CPU oracle for the log-probability hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this package; nothing under ``bisip_amd/`` does.
"""
from .oracle import (OracleProblem, build_oracle, logprob, forward, log_prior,  # noqa: F401
                     log_likelihood, max_threads, numpy_logprob, numpy_forward)

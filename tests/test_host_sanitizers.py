"""AddressSanitizer + UndefinedBehaviorSanitizer over the library's host-only C++ (the
long-double and binary128 operand precompute with its Householder QR and the probing that picks a reduced kernel, the
NumPy-order random stream, the spectrum-file parser on well-formed and hostile files, the thread helper).
GPU sanitizers are not available on the target pool; this is the CPU build the task's
environment notes ask for."""

import os
import shutil
import subprocess

import pytest

from conftest import ROOT


@pytest.mark.skipif(shutil.which('g++') is None, reason='needs g++')
def test_host_code_under_asan_ubsan(tmp_path):
    exe = tmp_path / 'sanitize_host'
    src = [os.path.join(ROOT, 'tests', 'native', 'sanitize_host.cpp'),
           os.path.join(ROOT, 'bisip_amd', 'csrc', 'host_precompute.cpp'),
           os.path.join(ROOT, 'bisip_amd', 'csrc', 'host_rng.cpp'),
           os.path.join(ROOT, 'bisip_amd', 'csrc', 'host_ingest.cpp')]
    build = subprocess.run(['g++', '-std=gnu++17', '-O1', '-g', '-fno-omit-frame-pointer',
                            '-fsanitize=address,undefined', '-fno-sanitize-recover=all', '-pthread',
                            '-o', str(exe)] + src + ['-lquadmath'], capture_output=True, text=True)
    assert build.returncode == 0, build.stderr
    env = dict(os.environ, ASAN_OPTIONS='detect_leaks=1:abort_on_error=0', UBSAN_OPTIONS='print_stacktrace=1')
    files = tmp_path / 'files'
    files.mkdir()
    run = subprocess.run([str(exe), str(files)], capture_output=True, text=True, env=env, timeout=300)
    assert run.returncode == 0, run.stdout + run.stderr
    assert 'sanitize_host: ok' in run.stdout


@pytest.mark.skipif(shutil.which('g++') is None, reason='needs g++')
def test_host_threads_under_tsan(tmp_path):
    """The same driver under ThreadSanitizer with four host threads: the blocks of spectra of a batch
    context and the files of a survey are handled by threads that must share nothing but their inputs."""
    exe = tmp_path / 'sanitize_host_tsan'
    src = [os.path.join(ROOT, 'tests', 'native', 'sanitize_host.cpp'),
           os.path.join(ROOT, 'bisip_amd', 'csrc', 'host_precompute.cpp'),
           os.path.join(ROOT, 'bisip_amd', 'csrc', 'host_rng.cpp'),
           os.path.join(ROOT, 'bisip_amd', 'csrc', 'host_ingest.cpp')]
    build = subprocess.run(['g++', '-std=gnu++17', '-O1', '-g', '-fsanitize=thread', '-pthread', '-o', str(exe)] + src + ['-lquadmath'],
                           capture_output=True, text=True)
    if build.returncode != 0 and 'tsan' in build.stderr.lower():
        pytest.skip('no ThreadSanitizer runtime here')
    assert build.returncode == 0, build.stderr
    files = tmp_path / 'files'
    files.mkdir()
    run = subprocess.run([str(exe), str(files)], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, BISIP_HOST_THREADS='4', TSAN_OPTIONS='halt_on_error=1'))
    if 'FATAL: ThreadSanitizer' in run.stderr and 'unexpected memory mapping' in run.stderr:
        pytest.skip('ThreadSanitizer cannot run in this sandbox')
    assert run.returncode == 0, run.stdout + run.stderr
    assert 'sanitize_host: ok' in run.stdout and 'WARNING: ThreadSanitizer' not in run.stderr

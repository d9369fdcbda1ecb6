"""AddressSanitizer + UndefinedBehaviorSanitizer over the HOST side of the C-ABI library, in this
container, without a GPU: argument checking, mode switches, workspace sizing and carving are host code that runs before any
kernel is launched.  The library is rebuilt with -fsanitize=address,undefined for the host pass only
(joint_tensorf_amd/build.py: build_sanitized -> tests/lib/libjt_render_asan.so; device-side sanitizers are not available on the
GPU pool and this build never travels there as the product) and a child python with the sanitizer runtime preloaded drives
every entry point through tests/sanitizer_driver.py."""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(torch.cuda.is_available(), reason="sanitizer runs belong to the CPU container, never to a GPU box")
def test_host_side_of_the_library_under_asan_and_ubsan():
    import importlib.util
    spec = importlib.util.spec_from_file_location("jt_build", os.path.join(ROOT, "joint_tensorf_amd", "build.py"))
    jb = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(jb)
    lib = jb.build_sanitized()
    env = dict(os.environ, LD_PRELOAD=jb.asan_runtime(), JT_LIB_PATH=lib,
               ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=0:exitcode=97",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1:exitcode=98")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "sanitizer_driver.py")], env=env, capture_output=True,
                       text=True, timeout=600)
    tail = (r.stdout[-1500:] + "\n" + r.stderr[-4000:])
    assert r.returncode == 0, tail
    assert "sanitizer driver:" in r.stdout and "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, tail
    print(r.stdout.strip().splitlines()[-1])

#!/usr/bin/env python3
"""Runs the GPU tests of the given files with the order of the tests inside each file REVERSED (one pytest process per file): the suite
shares one engine per file (fixture `eng`), and a test that only passes because an earlier one left the engine in some state hides a bug
a fresh host would hit.   tools/run_tests_reversed.py tests/test_gpu_render.py [...]   (arguments that start with '-' go to pytest)"""
import subprocess
import sys

opts = [a for a in sys.argv[1:] if a.startswith("-")]
paths = [a for a in sys.argv[1:] if not a.startswith("-")]
if not paths:
    sys.exit("usage: tools/run_tests_reversed.py [pytest options] tests/test_gpu_x.py [...]")
rc = 0
for path in paths:
    ids = [l.strip() for l in subprocess.run([sys.executable, "-m", "pytest", path, "-m", "gpu", "--collect-only", "-q"], capture_output=True, text=True).stdout.splitlines() if "::" in l]
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-m", "gpu", "-p", "no:cacheprovider"] + opts + ids[::-1])
    rc |= r.returncode
sys.exit(rc)

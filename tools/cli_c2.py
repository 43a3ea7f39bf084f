#!/usr/bin/env python3
"""Runs bin/pathtracer on the C2 stand-in with the README flags (README.md:29-39) and prints its log.
No GPU is touched by this process; the CLI is the child that renders."""
import os
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pathtracer_amd import _ffi, scenes  # noqa: E402

with tempfile.TemporaryDirectory() as d:
    gltf = scenes.write_many_lights_standin(d)
    out = os.path.join(d, "out")
    argv = [os.path.join(_ffi.REPO_ROOT, "bin", "pathtracer"), "--path-gltf", gltf, "--dir-output", out,
            "--output-resolution-x", "1024", "--output-resolution-y", "1024", "--sampling-samples", sys.argv[1] if len(sys.argv) > 1 else "200",
            "--sampling-samples-every-update", "50", "--sampling-seed", "42", "--output-exposure", "0",
            "--world-clear-color", "0 0 0", "--world-clear-color-texture", "", "--logger-log-level", os.environ.get("PT_CLI_LOG_LEVEL", "3"), "--logger-log-stdout", "0",
            "--default-camera-position", "0 0 5", "--default-camera-target", "0 0 0", "--default-camera-updir", "0 1 0",
            "--default-camera-yfov", "0.7"]
    r = subprocess.run(argv, capture_output=True, text=True)
    print("rc", r.returncode, r.stderr[-500:])
    print(open(os.path.join(out, "log.txt")).read()[-3500:])
    print("render.png bytes", os.path.getsize(os.path.join(out, "render.png")))

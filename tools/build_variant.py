#!/usr/bin/env python3
"""Build the CURRENT tree into tools/abl/<name>/ for a same-box A/B (tools/ab_lib.sh):  python tools/build_variant.py <name> [extra hipcc flags]
The variant carries the tree's source hash (so the loader accepts it) and a copy of every csrc file it was built from."""
import os, shutil, subprocess, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from speaker_embedding_torch_amd import _build
name, extra = sys.argv[1], sys.argv[2:]
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "abl", name)
os.makedirs(out, exist_ok=True)
digest = _build.source_hash()
cmd = ["/opt/rocm/bin/hipcc"] + _build.FLAGS + extra + [f'-DGE2E_SOURCE_HASH="{digest}"', "-o", os.path.join(out, "libge2e_hip.so")]
cmd += [os.path.join(_build.CSRC, s) for s in _build.SOURCES]
subprocess.run(cmd, check=True, cwd=_build.CSRC, stderr=subprocess.DEVNULL)
os.makedirs(os.path.join(out, "csrc"), exist_ok=True)
for f in _build.dependencies()[:-1]:
    shutil.copy(f, os.path.join(out, "csrc"))
print(out, digest)

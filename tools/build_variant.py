#!/usr/bin/env python3
"""Build the CURRENT tree into tools/abl/<name>/ for a same-box A/B (tools/ab_lib.sh):  python tools/build_variant.py <name> [extra hipcc flags]
The variant's hash covers the extra flags too, so it never equals the tree's: it can only be loaded through GE2E_LIB_OVERRIDE."""
import hashlib, os, subprocess, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from speaker_embedding_torch_amd import _build
name, extra = sys.argv[1], sys.argv[2:]
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "abl", name)
os.makedirs(out, exist_ok=True)
digest = hashlib.sha256((_build.source_hash() + " variant " + name + " " + " ".join(extra)).encode()).hexdigest()[:32]
cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + _build.FLAGS + extra + [f'-DGE2E_SOURCE_HASH="{digest}"', "-o", os.path.join(out, "libge2e_hip.so")]
cmd += [os.path.join(_build.CSRC, s) for s in _build.SOURCES]
subprocess.run(cmd, check=True, cwd=_build.CSRC, stderr=subprocess.DEVNULL)
print(out, digest)

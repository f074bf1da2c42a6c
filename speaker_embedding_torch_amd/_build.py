"""Compile libge2e_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU).

The dependency list is whatever csrc/ holds (every *.hip / *.cuh) plus the C header; a SHA-256 over those files is
compiled into the library (`ge2e_source_hash()`) and written beside it, so a binary that does not match the sources it
sits next to is rebuilt here and refused by `_lib.load()` -- the .so is git-ignored but travels to the GPU box, where a
stale one would otherwise pass the tests against the wrong code."""
import glob
import hashlib
import os
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libge2e_hip.so")
STAMP = LIB + ".hash"
HEADER = os.path.normpath(os.path.join(PKG, "..", "include", "ge2e_hip.h"))
SOURCES = ["ge2e_capi.hip"]
# -fno-slp-vectorize: left on, the compiler pairs neighbouring fp32 multiplies / adds into v_pk_mul_f32 / v_pk_fma_f32, which cost more
# issue time than the two scalar instructions they replace when MFMAs share the SIMD (MI355X_MICROARCH.md, "packed f32 VALU ... an
# anti-lever beside MFMAs"): same-box A/B of the training step 3.647 -> 3.607 ms.  The flags are part of the source hash.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-fno-slp-vectorize"]


def dependencies():
    """Every file the library is built from, in a stable order."""
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.cuh"))) + [HEADER]


def source_hash():
    h = hashlib.sha256()
    h.update(" ".join(FLAGS).encode() + b"\0")
    for path in dependencies():
        h.update(os.path.basename(path).encode() + b"\0")
        with open(path, "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    return h.hexdigest()[:32]


def built_hash():
    try:
        with open(STAMP) as f:
            return f.read().strip()
    except OSError:
        return None


def needs_build():
    return not os.path.exists(LIB) or built_hash() != source_hash()


def build(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 -shared; returns the library path."""
    if not force and not needs_build():
        return LIB
    digest = source_hash()
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + [f'-DGE2E_SOURCE_HASH="{digest}"', "-o", LIB + ".tmp"]
    cmd += [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True, cwd=CSRC)
    os.replace(LIB + ".tmp", LIB)
    with open(STAMP, "w") as f:
        f.write(digest + "\n")
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))

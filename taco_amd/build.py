"""Build libtaco_env.so (the HIP product library) in-tree for gfx950.

    python -m taco_amd.build [--force]

hipcc cross-compiles without a GPU.  -ffp-contract=off is part of the kernel's numerical contract (see
csrc/taco_math.hpp): fused multiply-adds appear only where the source writes fma().  -fno-slp-vectorize keeps the
compiler from pairing scalar fp32 ops into v_pk_* (no faster on gfx950, costs ~30 VGPRs in register-pair shuffles).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libtaco_env.so")
SOURCES = ["taco_capi.hip"]
DEPS = ["taco_capi.hip", "taco_step.hpp", "taco_math.hpp", "taco_rollout.hpp", "taco_policy.hpp", os.path.join("..", "..", "include", "taco_env.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize",
         "-Wall", "-Wno-unused-function"]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in DEPS)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    cmd = [HIPCC] + FLAGS + ["-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))

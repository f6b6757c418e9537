"""Build libtaco_env.so (the HIP product library) in-tree for gfx950.

    python -m taco_amd.build [--force] [--test-hooks]

hipcc cross-compiles without a GPU.  -ffp-contract=off is part of the kernel's numerical contract (see
csrc/taco_math.hpp): fused multiply-adds appear only where the source writes fma().  -fno-slp-vectorize keeps the
compiler from pairing scalar fp32 ops into v_pk_* (no faster on gfx950, costs ~30 VGPRs in register-pair shuffles).

-mllvm -amdgpu-kernarg-preload-count=11: gfx950 delivers the first user SGPRs' worth of kernel arguments in registers at wavefront start; the step
kernel's eleven leading scalar arguments (taco_step.hpp StepKernelArgs) are what its up-front loads need, so they are issued without the round
trip to the argument segment.  (On firmware without the feature the kernels' compatibility prologue loads the same registers: correct, no gain.)

Every build embeds a hash of its sources (taco_source_hash()); `_lib.load()` compares it with the sources on disk, so an edited
csrc/ can never be run through a stale binary.  The output is written to a temporary file and renamed into place under a file lock:
concurrent ranks of a fresh checkout (torchrun) build once.

--test-hooks builds libtaco_env_testhooks.so (-DTACO_TEST_HOOKS): the same library plus taco_test_slow_battery_server, for
tests/test_parity_gpu.py::test_battery_mailbox_wait_path.  The product library contains no test hook and reads no environment variable.
"""
import fcntl
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libtaco_env.so")
LIB_HOOKS = os.path.join(HERE, "libtaco_env_testhooks.so")
SOURCES = ["taco_capi.hip"]
DEPS = ["taco_capi.hip", "taco_step.hpp", "taco_math.hpp", "taco_rollout.hpp", "taco_policy.hpp", "taco_fused.hpp", os.path.join("..", "..", "include", "taco_env.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize", "-falign-loops=64",
         "-mllvm", "-amdgpu-kernarg-preload-count=11", "-Wall", "-Wno-unused-function"]


def source_hash():
    h = hashlib.sha256()
    for d in DEPS:
        with open(os.path.join(CSRC, d), "rb") as f:
            h.update(f.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()[:16]


def embedded_hash(path):
    """the hash a built library carries (read from the file: no dlopen, works without a GPU runtime)"""
    try:
        with open(path, "rb") as f:
            blob = f.read()
    except OSError:
        return None
    i = blob.find(b"taco-src-hash:")
    return blob[i + 14:i + 30].decode("ascii", "replace") if i >= 0 else None


def needs_build(path=LIB):
    return embedded_hash(path) != source_hash()


def build(force=False, verbose=False, test_hooks=False):
    out = LIB_HOOKS if test_hooks else LIB
    if not force and not needs_build(out):
        return out
    with open(os.path.join(HERE, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and not needs_build(out):  # another rank built it while we waited
            return out
        tmp = f"{out}.{os.getpid()}.tmp"
        cmd = [HIPCC] + FLAGS + [f'-DTACO_SOURCE_HASH="{source_hash()}"'] + (["-DTACO_TEST_HOOKS"] if test_hooks else []) + \
              ["-o", tmp] + [os.path.join(CSRC, s) for s in SOURCES]
        if verbose:
            print(" ".join(cmd))
        try:
            subprocess.check_call(cmd, cwd=CSRC)
            os.replace(tmp, out)
        finally:
            if os.path.exists(tmp):
                os.remove(tmp)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, test_hooks="--test-hooks" in sys.argv))

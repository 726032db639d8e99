"""Build tests/emu/_build/librhip_emu.so: the engine sources of croaring_amd/csrc compiled for the host
against the hipemu shim (tests/emu/shim/hip/hip_runtime.h).  TEST INFRASTRUCTURE ONLY -- see the shim header.
The croaring_amd package never loads this library."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "croaring_amd", "csrc")
OUT = os.path.join(HERE, "_build")
LIB = os.path.join(OUT, "librhip_emu.so")
CXX = "/opt/rocm/lib/llvm/bin/clang++"


def _newest(paths):
    return max(os.path.getmtime(p) for p in paths)


def sources():
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    deps += [os.path.join(HERE, f) for f in ("hipemu_core.cpp", "prims_emu.cpp", "build_emu.py")]
    deps += [os.path.join(HERE, "shim", "hip", "hip_runtime.h"), os.path.join(ROOT, "include", "roaring_hip.h"),
             os.path.join(ROOT, "include", "roaring_hip_compat.h")]
    return deps


def asan_runtime() -> str:
    r = subprocess.run([CXX, "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True)
    return r.stdout.strip()


def build(force: bool = False, verbose: bool = False, asan: bool = False) -> str:
    """asan=True builds librhip_emu_asan.so: AddressSanitizer (out-of-bounds global / LDS / stack accesses of the
    kernels abort) plus alignment traps (a misaligned typed access -- which x86 would forgive and the GPU would
    not -- raises SIGILL); the process using it must run with LD_PRELOAD=asan_runtime()."""
    if asan:
        return _build(os.path.join(OUT, "librhip_emu_asan.so"), ["-fsanitize=address,alignment",
                                                                  "-fsanitize-trap=alignment",
                                                                  "-fno-omit-frame-pointer", "-shared-libasan"], force, verbose)
    return _build(LIB, [], force, verbose)


def _build(lib: str, extra, force: bool, verbose: bool) -> str:
    if not os.path.exists(CXX):
        raise RuntimeError(f"{CXX} not found: hipemu needs clang (ext_vector_type, nontemporal builtins)")
    os.makedirs(OUT, exist_ok=True)
    if not force and os.path.exists(lib) and os.path.getmtime(lib) >= _newest(sources()):
        return lib
    cmd = [CXX, "-std=c++17", "-O1", "-g", "-fPIC", "-shared", "-pthread", "-DRHIP_EMU=1", *extra,
           "-Wno-unused-value", "-Wno-deprecated-declarations",
           "-I", os.path.join(HERE, "shim"), "-I", CSRC, "-I", os.path.join(ROOT, "include"),
           "-x", "c++", os.path.join(CSRC, "rhip_engine.hip"),
           os.path.join(HERE, "hipemu_core.cpp"), os.path.join(HERE, "prims_emu.cpp"),
           "-o", lib]
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipemu build failed:\n" + r.stdout[-4000:] + r.stderr[-8000:])
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, asan="--asan" in sys.argv))

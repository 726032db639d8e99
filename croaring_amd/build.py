"""Build libroaring_hip.so (gfx950) in-tree with hipcc.  `python -m croaring_amd.build`"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
OUT = os.path.join(PKG, "libroaring_hip.so")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-pthread", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function"]
SOURCES = ["rhip_prims.hip", "rhip_engine.hip"]
DEPS = ["rhip_kernels.h", "rhip_common.h", "rhip_plan.h", "rhip_bitset.h", "rhip_array.h", "rhip_grouped.h", "rhip_runs.h", "rhip_classes.h", "rhip_block.h", "rhip_many.h", "rhip_many_host.inc", "rhip_heap.h", "rhip_heap_host.inc", "rhip_sharded.inc", "rhip_synth.inc", "rhip_prims.h", "roaring_compat.inc", "rhip_poolops.h", "rhip_pool_ops.inc", "rhip_wemit.inc", "rhip_serial.h", "rhip_deser.h", "rhip_frozen.h", "rhip_values.h", "rhip_flip.h",
        os.path.join("..", "..", "include", "roaring_hip.h"), os.path.join("..", "..", "include", "roaring_hip_compat.h")]


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libroaring_hip.so can only be built with the ROCm toolchain")


def _stale(obj: str, src: str) -> bool:
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    paths = [src] + [os.path.join(CSRC, d) for d in DEPS]
    return any(os.path.exists(p) and os.path.getmtime(p) > t for p in paths)


def build(force: bool = False, verbose: bool = False) -> str:
    hipcc = _hipcc()
    extra = os.environ.get("RHIP_EXTRA_FLAGS", "").split()  # diagnostic builds, e.g. -DRHIP_PHASES
    force = force or bool(extra)
    # RHIP_BUILD_VARIANT=name: the diagnostic build goes to libroaring_hip_<name>.so (loaded with RHIP_LIB_VARIANT=name),
    # beside the product library instead of over it -- A/B measurements of a compile-time switch in one GPU call
    variant = os.environ.get("RHIP_BUILD_VARIANT", "")
    out = os.path.join(PKG, f"libroaring_hip_{variant}.so") if variant else OUT
    objdir = os.path.join(PKG, "build", variant) if variant else os.path.join(PKG, "build")
    os.makedirs(objdir, exist_ok=True)
    objs, rebuilt = [], False
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        if not os.path.exists(sp):
            continue
        obj = os.path.join(objdir, os.path.splitext(src)[0] + ".o")
        if force or _stale(obj, sp):
            cmd = [hipcc, *FLAGS, *extra, "-x", "hip", "-c", sp, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
            rebuilt = True
        objs.append(obj)
    if rebuilt or not os.path.exists(out):
        cmd = [hipcc, "-shared", "-fPIC", "-pthread", f"--offload-arch={ARCH}", *objs, "-o", out]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))

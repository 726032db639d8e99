"""Per-kernel register / LDS / scratch / occupancy table of the gfx950 build (no GPU needed):
    python scripts/kernel_resources.py [substring ...]
runs hipcc -Rpass-analysis=kernel-resource-usage on rhip_engine.hip and prints one line per kernel."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def table(extra=()):
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only",
           "-Rpass-analysis=kernel-resource-usage", *extra, "-x", "hip", "-c",
           os.path.join(ROOT, "croaring_amd", "csrc", "rhip_engine.hip"), "-o", "/dev/null"]
    txt = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows = []
    for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
        def g(k):
            m = re.search(k + r": (\d+)", b)
            return int(m.group(1)) if m else -1
        rows.append((b.split()[0], g("VGPRs"), g("AGPRs"), g("SGPRs"), g(r"ScratchSize \[bytes/lane\]"),
                     g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")))
    return rows


if __name__ == "__main__":
    pats = sys.argv[1:]
    for name, v, a, s, sc, occ, lds in table():
        try:
            dn = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.split("(")[0]
        except Exception:
            dn = name
        if pats and not any(p in dn for p in pats):
            continue
        print(f"{dn[:44]:44s} vgpr {v:4d} agpr {a:3d} sgpr {s:4d} scratch {sc:5d} occ {occ:2d} lds {lds:6d}")

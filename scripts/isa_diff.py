"""Per-kernel comparison of two gfx950 assembly listings: which kernels changed (instruction stream, VGPRs,
LDS, scratch, occupancy).  Used to show that a source refactor left the hot kernels' code untouched when no
GPU is at hand to re-measure them.

    hipcc -O3 --offload-arch=gfx950 -std=c++17 --cuda-device-only -S -o before.s croaring_amd/csrc/rhip_engine.hip
    ... edit ...
    hipcc ... -S -o after.s croaring_amd/csrc/rhip_engine.hip
    python scripts/isa_diff.py before.s after.s        # prints nothing for identical kernels
"""
import re
import subprocess
import sys


def parse(path):
    out, cur = {}, None
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1)
            out[cur] = {"ins": [], "meta": {}}
            continue
        if cur is None:
            continue
        s = line.strip()
        if re.match(r"^(v_|s_|ds_|global_|buffer_|flat_|scratch_)", s):
            out[cur]["ins"].append(re.sub(r"\.LBB\d+_", ".LBB_", s.split(";")[0].strip()))
        m = re.match(r"^; (NumVgprs|ScratchSize|Occupancy|LDSByteSize|NumSgprs): (\d+)", s)
        if m:
            out[cur]["meta"][m.group(1)] = int(m.group(2))
    return out


def demangle(n):
    return subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()[:70]


def main():
    a, b = parse(sys.argv[1]), parse(sys.argv[2])
    empty = {"ins": [], "meta": {}}
    changed = 0
    for k in sorted(set(a) | set(b)):
        x, y = a.get(k, empty), b.get(k, empty)
        if x["ins"] != y["ins"] or x["meta"] != y["meta"]:
            changed += 1
            print(f"{demangle(k)}: {len(x['ins'])} -> {len(y['ins'])} instructions, {x['meta']} -> {y['meta']}")
    print(f"{changed} of {len(set(a) | set(b))} kernels differ")


if __name__ == "__main__":
    main()

#!/bin/bash
# The kernel sources through the CPU emulator (tests/emu) under AddressSanitizer + alignment traps, lanes resumed in
# shuffled order: out-of-bounds global / LDS / stack accesses, misaligned typed accesses and unordered cross-lane
# LDS traffic abort the run.  No GPU needed.   bash scripts/emu_asan.sh [pytest -k expression]
cd "$(dirname "$0")/.."
RT=$(python - <<'PY'
import sys; sys.path.insert(0, "tests")
from emu import asan_runtime
print(asan_runtime())
PY
)
HIPEMU_SHUFFLE=${HIPEMU_SHUFFLE:-11} HIPEMU_ASAN=1 LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0 \
  python -m pytest tests/test_emu_kernels.py -q -x ${1:+-k "$1"}

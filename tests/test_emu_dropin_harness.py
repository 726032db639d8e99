"""The reference's own UNMODIFIED unit tests driving the kernels WITHOUT a GPU: tests/toplevel_unit.c (169 registered
tests) and tests/cpp_random_unit.cpp (randomised, double-checked against std::set) are linked against the symbol-renamed
reference library (everything that is not on the hot path) and the CPU-emulator build of the engine
(tests/emu/_build/librhip_emu.so, test infrastructure) for the 20 hot-path symbols.  Same harness as
tests/test_gpu_dropin_harness.py, which links the real libroaring_hip.so and needs an MI355X.

Built by `make -C oracle dropin_emu` (only where /root/reference exists); skipped when the binaries are absent."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref")


def _build():
    if not os.path.isdir("/root/reference/src"):
        return
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from emu import build_emu
    if not os.path.exists(build_emu.CXX):
        return
    lib = build_emu.build()
    for b in ("toplevel_unit_emu", "cpp_random_unit_emu", "roaring64_unit_emu"):
        exe = os.path.join(REF, b)
        if not os.path.exists(exe) or os.path.getmtime(exe) < os.path.getmtime(lib):
            subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "dropin_emu", "dropin64_emu"], check=False,
                           capture_output=True)
            break


@pytest.fixture(scope="module", autouse=True)
def built():
    _build()


def _run(name, min_tests):
    exe = os.path.join(REF, name)
    if not os.path.exists(exe):
        pytest.skip(f"oracle/_ref/{name} not built (needs /root/reference and the emulator build)")
    env = dict(os.environ, RHIP_COMPAT_STATS="1")
    p = subprocess.run([exe], capture_output=True, text=True, timeout=900, env=env, cwd=REF)
    tail = (p.stdout + p.stderr)[-3000:]
    m = re.search(r"(\d+) tests, (\d+) failed", p.stdout)
    assert m, tail
    assert int(m.group(1)) >= min_tests and int(m.group(2)) == 0 and p.returncode == 0, tail
    c = re.search(r"pairwise (\d+), in-place (\d+), cardinality (\d+), many-way (\d+)", p.stderr)
    assert c, tail
    return tuple(int(x) for x in c.groups())


def test_reference_toplevel_unit_through_the_emulator():
    pw, ip, card, many = _run("toplevel_unit_emu", 160)
    assert pw > 100 and ip > 20 and many > 10, (pw, ip, card, many)   # the calls really went through the engine


def test_reference_cpp_random_unit_through_the_emulator():
    pw, ip, card, many = _run("cpp_random_unit_emu", 5)
    assert pw + ip > 100, (pw, ip, card, many)


def test_reference_realdata_unit_through_the_emulator():
    """tests/realdata_unit.c, unmodified: every realdata directory, with and without copy-on-write (213 000 pairwise,
    28 400 in-place, 71 000 cardinality and 108 many-way calls through the emulated kernels).  9 minutes: opt-in;
    the last run is recorded in profiles/r01_emu_realdata_unit.txt."""
    if not os.environ.get("RHIP_SLOW_HARNESS"):
        pytest.skip("slow (9 min): set RHIP_SLOW_HARNESS=1")
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "dropin_emu_realdata"], check=False, capture_output=True)
    exe = os.path.join(REF, "realdata_unit_emu")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/realdata_unit_emu not built (needs /root/reference/benchmarks/realdata)")
    p = subprocess.run([exe], capture_output=True, text=True, timeout=3000, env=dict(os.environ, RHIP_COMPAT_STATS="1"))
    assert p.returncode == 0 and "failure" not in p.stdout, (p.stdout + p.stderr)[-2000:]
    c = re.search(r"pairwise (\d+), in-place (\d+), cardinality (\d+), many-way (\d+)", p.stderr)
    assert c and int(c.group(1)) > 100000, (p.stdout + p.stderr)[-500:]


def test_reference_roaring64_unit_on_emulated_kernels():
    """tests/roaring64_unit.cpp, unmodified (75 tests): every roaring64_bitmap_{and,or,xor,andnot}(_inplace,
    _cardinality) and roaring64_bitmap_flip it issues goes through the 64-bit drop-ins (portable format in, device
    pipeline, portable format out) on the emulated kernels."""
    exe = os.path.join(REF, "roaring64_unit_emu")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/roaring64_unit_emu not built (needs /root/reference and the emulator build)")
    p = subprocess.run([exe], capture_output=True, text=True, timeout=900, env=dict(os.environ, RHIP_COMPAT_STATS="1"), cwd=REF)
    tail = (p.stdout + p.stderr)[-3000:]
    m = re.search(r"(\d+) tests, (\d+) failed", p.stdout)
    assert m, tail
    assert int(m.group(1)) >= 70 and int(m.group(2)) == 0 and p.returncode == 0, tail
    c = re.search(r"compat64\] device-executed calls: pairwise (\d+), in-place (\d+), cardinality (\d+), flip (\d+)", p.stderr)
    assert c, tail
    assert int(c.group(1)) >= 8 and int(c.group(2)) >= 4 and int(c.group(3)) >= 4 and int(c.group(4)) >= 1, c.group(0)

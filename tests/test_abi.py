"""The C-ABI library loads on a CPU-only host and exports every symbol include/*.h declares; the
product path refuses to run without a HIP device (no CPU fallback)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    names = set()
    for m in re.finditer(r"\b((?:rhip|roaring64?)_[A-Za-z0-9_]+)\s*\(", src):
        names.add(m.group(1))
    return sorted(names)


def test_library_exports_every_declared_symbol():
    import croaring_amd
    from croaring_amd import _lib
    lib = croaring_amd.load()
    hdrs = [h for h in os.listdir(os.path.join(ROOT, "include")) if h.endswith(".h")]
    assert "roaring_hip.h" in hdrs
    total = 0
    for h in hdrs:
        for name in declared_symbols(h):
            assert hasattr(lib, name), f"{h}: {name} is declared but not exported"
            total += 1
    assert total >= 25
    bound = {s[0] for s in _lib.SYMBOLS}
    assert set(declared_symbols("roaring_hip.h")) <= bound, "ctypes table is missing a declared entry point"


def test_no_cpu_fallback():
    """Without a HIP device the engine must fail loudly, never compute on the host."""
    import croaring_amd
    lib = croaring_amd.load()
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        pytest.skip("a GPU is present")
    with pytest.raises(croaring_amd.RoaringHipError):
        croaring_amd.Engine()
    assert b"no HIP device" in lib.rhip_last_error()


def test_product_does_not_reference_oracle():
    """croaring_amd/ must not import, link or name anything under oracle/."""
    pkg = os.path.join(ROOT, "croaring_amd")
    for dp, _, fs in os.walk(pkg):
        if os.path.basename(dp) == "build":
            continue
        for f in fs:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".inc")):
                txt = open(os.path.join(dp, f)).read()
                assert "pyoracle" not in txt and "liboracle" not in txt and "roaring_oracle" not in txt, f
                assert "hipemu" not in txt and "rhip_emu" not in txt, f  # the CPU emulator is test-only too
    out = os.popen(f"readelf -d {os.path.join(pkg, 'libroaring_hip.so')}").read()
    assert "oracle" not in out and "croaring_ref" not in out


def test_headers_compile_as_c_and_cpp(tmp_path):
    """include/*.h are what a binding compiles against: plain C11 and C++17, no HIP or torch types, every round-4 entry
    point (prepared pair lists, the allocator hook) callable as INTEGRATION.md shows."""
    import shutil
    import subprocess
    src = tmp_path / "hdr.c"
    src.write_text('''
#include "roaring_hip.h"
#include "roaring_hip_compat.h"
int main(void) {
    rhip_ctx_t *ctx = 0; rhip_pool_t *pool = 0; rhip_op ops[4] = {RHIP_AND, RHIP_OR, RHIP_XOR, RHIP_ANDNOT};
    rhip_pairlist_t *all = rhip_pairlist_all_pairs(ctx, pool);
    rhip_pool_t *four = rhip_pairwise_list(ctx, 4, ops, all, 0);
    rhip_batch_t *b = rhip_pairwise_list_begin(ctx, 1, ops, all, 0);
    unsigned long long cards[1]; float rates[4];
    (void)rhip_pairwise_list_cardinality(ctx, RHIP_AND, all, (uint64_t *)cards);
    (void)rhip_debug_last_placement(ctx, rates, 4);
    (void)rhip_install_pinned_allocator(0);
    (void)four; (void)b;
    rhip_pairlist_free(all);
    return 0;
}
''')
    inc = os.path.join(ROOT, "include")
    for cc, std, lang in (("gcc", "-std=c11", "c"), ("g++", "-std=c++17", "c++")):
        if not shutil.which(cc):
            continue
        p = subprocess.run([cc, std, "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-x", lang, "-I", inc, str(src)],
                           capture_output=True, text=True)
        assert p.returncode == 0, p.stderr

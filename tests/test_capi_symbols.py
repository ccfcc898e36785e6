"""The C-ABI library loads on a machine without a GPU, exports every symbol include/zoe_sw.h declares, and its
GPU entry points fail loudly (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "zoe_sw.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(zsw_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_lists_agree():
    from zoe_amd import _lib

    assert header_symbols() == sorted(_lib.SYMBOLS)


def test_library_exports_every_declared_symbol():
    from zoe_amd import _lib, build

    build.build()
    lib = _lib.load()
    for sym in header_symbols():
        assert getattr(lib, sym) is not None, sym


def test_no_gpu_fails_loudly():
    import torch

    from zoe_amd import SwContext, _lib

    if torch.cuda.is_available():
        return
    lib = _lib.load()
    assert lib.zsw_device_count() == 0
    h = C.c_void_p()
    assert lib.zsw_create(0, C.byref(h)) == -3  # ZSW_ERR_NO_DEVICE
    assert b"hipGetDeviceCount" in lib.zsw_last_error_string(None)
    try:
        SwContext(0)
    except _lib.ZswError as e:
        assert e.code == -3
    else:
        raise AssertionError("SwContext must raise without a GPU")


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under zoe_amd/ or include/ may reference it."""
    bad = []
    for base in ("zoe_amd", "include"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                    txt = open(os.path.join(dp, f), errors="ignore").read()
                    if re.search(r"(from|import)\s+oracle|zoe_oracle|libzoe_oracle|oracle/", txt):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_host_synth_twins():
    from zoe_amd import synth

    ref = synth.reference_host(2000)
    assert len(ref) == 2000 and set(ref) <= set(b"ACGT")
    a = synth.reads_host(ref, 100, 50, 150)
    b = synth.reads_host(ref, 0, 150, 150)
    assert np.array_equal(a, b[100:150])  # counter based: any shard regenerates its slice
    assert set(np.unique(a)) <= set(b"ACGTN")
    hb, off = synth.reads_ragged_host(ref, 5, 40, 75, 400)
    lens = np.diff(off)
    assert lens.min() >= 75 and lens.max() <= 400 and off[-1] == len(hb)
    assert np.array_equal(lens, synth.ragged_lengths(5, 40, 75, 400))


def test_header_is_c_and_links_from_c(tmp_path):
    """include/zoe_sw.h is a C header: examples/zsw_c_abi.c (C99, -pedantic) names every entry point, links against the library
    and runs on a machine without a GPU (zsw_create reports ZSW_ERR_NO_DEVICE)."""
    import subprocess

    from zoe_amd import build

    build.build()
    src = os.path.join(ROOT, "examples", "zsw_c_abi.c")
    txt = open(src).read()
    for sym in header_symbols():
        assert sym in txt, f"{sym} missing from examples/zsw_c_abi.c"
    exe = str(tmp_path / "zsw_c_abi")
    libdir = os.path.join(ROOT, "zoe_amd")
    subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), src, "-o", exe, "-L" + libdir,
                    "-lzoe_sw_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath-link,/opt/rocm/lib"], check=True)
    import torch

    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = os.path.join(os.path.dirname(torch.__file__), "lib") + ":/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    out = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert f"{len(header_symbols())} entry points, sizeof(zsw_alignment) = 40" in out.stdout


def test_the_library_exports_the_c_abi_and_nothing_else():
    """nm -D: every defined dynamic symbol is a zsw_* entry point of the header (linker version script zsw_exports.map); no C++
    internals (_ZN3zsw...) leak out of the library."""
    import subprocess

    from zoe_amd import _lib, build

    out = subprocess.run(["nm", "-D", "--defined-only", build.LIB], capture_output=True, text=True, check=True).stdout
    names = [line.split()[-1] for line in out.splitlines() if line.strip()]
    exported = sorted(n for n in names if not n.startswith(("__", "_init", "_fini", "_edata", "_end")))
    assert exported, "no exported symbols"
    assert all(n.startswith("zsw_") for n in exported), [n for n in exported if not n.startswith("zsw_")][:5]
    assert set(exported) == set(_lib.SYMBOLS)

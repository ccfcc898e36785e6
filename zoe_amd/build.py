"""Builds libzoe_sw_hip.so (hand-written HIP for gfx950) in-tree with hipcc.

hipcc cross-compiles without a GPU; the built .so travels to the GPU box with the repo snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libzoe_sw_hip.so")
SOURCES = ["zsw_capi.hip", "zsw_capi_shared.hip", "zsw_shared.hip", "zsw_score.hip", "zsw_align.hip", "zsw_align_pk8.hip", "zsw_align_pk16.hip", "zsw_align_pk32.hip", "zsw_align_pk64.hip", "zsw_group.hip", "zsw_threepass.hip", "zsw_filter.hip", "zsw_score_wide.hip", "zsw_score_w32.hip", "zsw_score_prune.hip", "zsw_score_seed.hip", "zsw_score_seed_m0.hip", "zsw_score_seed_m1.hip", "zsw_score_seed_m2.hip", "zsw_score_band.hip", "zsw_multi.hip"]
HEADERS = ["zsw_internal.hpp", "zsw_align_dev.hpp", "zsw_align_pk.hpp", "zsw_align_pk_kernel.hpp", "zsw_score_v1.hpp", "zsw_score_v2.hpp", "zsw_score_prune.hpp", "zsw_seed.hpp", "zsw_context.hpp", "zsw_shared.hpp", "zsw_score_seed.hpp", "zsw_score_seed_kernel.hpp", "zsw_align.hpp", "zsw_timer.hpp", "zsw_synth.h", "zsw_exports.map", os.path.join("..", "..", "include", "zoe_sw.h")]
ARCH = "gfx950"
# per-file code generation flags. Measured and rejected for zsw_align_pk*.hip: -mllvm -amdgpu-sched-strategy=max-ilp
# (s_nop between dependent packed instructions 206 -> 92 in the <16,10> kernel, but 190 VGPRs = two waves per SIMD: 46 instead of 41 ms)
EXTRA_FLAGS = {}


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP extension cannot be built")


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB
    objs = []
    procs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        if not os.path.exists(src):
            continue
        obj = os.path.join(CSRC, s.replace(".hip", ".o"))
        cmd = [_hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-c", src, "-o", obj,
               "-Wall", "-Wno-unused-function", "-Wno-unused-result"] + EXTRA_FLAGS.get(s, [])
        if verbose:
            cmd.append("-Rpass-analysis=kernel-resource-usage")
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(obj)
    failed = False
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed = True
            sys.stderr.write(f"--- hipcc failed for {s}\n{out}\n")
        elif verbose or out.strip():
            sys.stderr.write(out)
    if failed:
        raise RuntimeError("hipcc compilation failed")
    # only the zsw_* entry points of include/zoe_sw.h are exported (the C++ internals stay local to the library)
    cmd = [_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB, "-Wl,--version-script=" + os.path.join(CSRC, "zsw_exports.map")] + objs
    subprocess.run(cmd, check=True)
    return LIB


def fatbin_sha256(path: str = LIB) -> str:
    """sha256 of the library's .hip_fatbin section (the gfx950 code objects): what a rocprofv3 summary under profiles/ was taken
    from, and what bench.py compares with the library it runs (a kernel change that was not re-profiled shows as "stale")."""
    import hashlib
    import struct

    with open(path, "rb") as f:
        data = f.read()
    if data[:4] != b"\x7fELF" or data[4] != 2:
        raise RuntimeError("not a 64-bit ELF file: " + path)
    shoff, = struct.unpack_from("<Q", data, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", data, 0x3A)

    def section(i):
        name, _type, _flags, _addr, off, size = struct.unpack_from("<IIQQQQ", data, shoff + i * shentsize)
        return name, off, size

    _, stroff, strsize = section(shstrndx)
    names = data[stroff:stroff + strsize]
    for i in range(shnum):
        name, off, size = section(i)
        if names[name:names.index(b"\0", name)] == b".hip_fatbin":
            return hashlib.sha256(data[off:off + size]).hexdigest()
    raise RuntimeError("no .hip_fatbin section in " + path)


def build_driver(name: str = "zsw_driver") -> str:
    """A C++ host program over include/zoe_sw.hpp (examples/zsw_driver.cpp: FASTQ -> SAM; examples/zsw_selftest.cpp: the
    reference's known-answer vectors), linked against the in-tree library."""
    root = os.path.dirname(HERE)
    src = os.path.join(root, "examples", name + ".cpp")
    out = os.path.join(root, "examples", name)
    deps = [src, os.path.join(root, "include", "zoe_sw.hpp"), os.path.join(root, "include", "zoe_sw.h"), LIB]
    if os.path.exists(out) and all(os.path.getmtime(d) <= os.path.getmtime(out) for d in deps):
        return out
    subprocess.run(["g++", "-O2", "-std=c++17", "-Wall", "-I" + os.path.join(root, "include"), src, "-o", out, "-L" + HERE,
                    "-lzoe_sw_hip", "-Wl,-rpath,$ORIGIN/../zoe_amd", "-Wl,-rpath-link,/opt/rocm/lib"], check=True)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))

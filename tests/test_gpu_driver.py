"""The C++ host driver (examples/zsw_driver.cpp over include/zoe_sw.hpp): FASTQ in, SAM out, checked field by field
against the oracle (POS = ref_range.start + 1, CIGAR, AS — src/data/records/sam/mod.rs:223-245)."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_driver_fastq_to_sam(tmp_path, oracle):
    from zoe_amd import build, synth

    exe = build.build_driver()
    ref = synth.reference_host(1200)
    reads = synth.reads_host(ref, 4242, 200, 100)
    reads[3] = ord("N")
    (tmp_path / "ref.fa").write_bytes(b">synthref test\n" + ref[:600] + b"\n" + ref[600:] + b"\n")
    with open(tmp_path / "reads.fq", "wb") as f:
        for i, r in enumerate(reads):
            f.write(b"@read%d extra\n" % i + r.tobytes() + b"\n+\n" + b"I" * len(r) + b"\n")
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = os.path.join(os.path.dirname(__import__("torch").__file__), "lib") + ":" + env.get("LD_LIBRARY_PATH", "")
    out = subprocess.run([exe, str(tmp_path / "ref.fa"), str(tmp_path / "reads.fq")], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr
    lines = [l for l in out.stdout.splitlines() if not l.startswith("@")]
    assert len(lines) == 200
    sc = oracle.dna_scoring(2, -5, b"N", -10, -1)
    for i, line in enumerate(lines):
        f = line.split("\t")
        want, _ = oracle.cascade_align(8, 256, sc, reads[i], ref)
        assert f[0] == f"read{i}"
        if want.status == 0:
            assert (f[1], f[2], int(f[3]), f[5], f[11]) == ("0", "synthref", want.ref_range[0] + 1, want.cigar, f"AS:i:{want.score}"), i
        else:
            assert f[1] == "4" and f[5] == "*"
    out = subprocess.run([exe, str(tmp_path / "ref.fa"), str(tmp_path / "reads.fq"), "--3pass"], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr
    lines3 = [l for l in out.stdout.splitlines() if not l.startswith("@")]
    for i, line in enumerate(lines3):
        f = line.split("\t")
        want, _, _ = oracle.cascade_align_3pass(8, 256, sc, reads[i], ref)
        if want.status == 0:
            assert (f[1], int(f[3]), f[5], f[11]) == ("0", want.ref_range[0] + 1, want.cigar, f"AS:i:{want.score}"), i
        else:
            assert f[1] == "4"
    out = subprocess.run([exe, str(tmp_path / "ref.fa"), str(tmp_path / "reads.fq"), "--score-only"], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr
    for i, line in enumerate(out.stdout.splitlines()):
        st, s, _ = oracle.cascade_score(8, 256, sc, reads[i], ref)
        assert line.split("\t")[1] == (str(s) if st == 0 else "*")


def test_cpp_mirror_known_answer_vectors():
    """examples/zsw_selftest.cpp: the reference's known-answer tests written against the C++ mirror (include/zoe_sw.hpp)."""
    from zoe_amd import build

    exe = build.build_driver("zsw_selftest")
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = os.path.join(os.path.dirname(__import__("torch").__file__), "lib") + ":" + env.get("LD_LIBRARY_PATH", "")
    out = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "ALL PASSED" in out.stdout and "FAIL" not in out.stdout.replace("FAILED", "")
    assert out.stdout.count("ok ") == 23

"""Oracle restatement of `sneaky_snake` (reference: src/alignment/sneaky_snake.rs:78-131) against the reference's own example
and against the properties its documentation states (the filter's edit estimate never exceeds the true edit distance)."""
import numpy as np
import pytest


def edit_distance(a: bytes, b: bytes) -> int:
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i] + [0] * len(b)
        for j, cb in enumerate(b, 1):
            cur[j] = min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb))
        prev = cur
    return prev[-1]


def test_doc_example(oracle):
    # sneaky_snake.rs:55-60 and examples/demo_sneaky_snake.rs:4-6
    assert oracle.sneaky_snake(b"GGTGCAGAGCTC", b"GGTGAGAGTTGT", 0.25) is True
    assert oracle.sneaky_snake(b"GGTGCAGAGCTC", b"GGTGAGAGTTGT", 3.0 / 12.0) is True


def test_invalid_inputs_are_none(oracle):
    # :79-88: threshold outside [0,1]; length difference above the allowed edits
    assert oracle.sneaky_snake(b"ACGT", b"ACGT", -0.1) is None
    assert oracle.sneaky_snake(b"ACGT", b"ACGT", 1.01) is None
    assert oracle.sneaky_snake(b"ACGT", b"ACGT", float("nan")) is None
    assert oracle.sneaky_snake(b"ACGTACGTAC", b"ACGTACGT", 0.125) is None  # |10-8| = 2 > floor(8*0.125) = 1
    assert oracle.sneaky_snake(b"ACGTACGTAC", b"ACGTACGT", 0.25) is True   # 2 unpenalised residues
    # :89-91: everything may be edited
    assert oracle.sneaky_snake(b"AAAA", b"CCCC", 1.0) is True
    assert oracle.sneaky_snake(b"", b"", 0.3) is True


def test_identical_and_hopeless(oracle):
    s = b"ACGTTGCAAGGCTTAACCGGTTAACG"
    assert oracle.sneaky_snake(s, s, 0.0) is True
    assert oracle.sneaky_snake(s, s[:-1] + b"T", 0.0) is False
    assert oracle.sneaky_snake(b"A" * 40, b"C" * 40, 0.5) is False


def test_never_rejects_within_true_edit_distance(oracle):
    # "Its calculated edit distance is always less than or equal to the actual edit distance" (:30-33): for equal lengths a pair
    # whose global edit distance is within the threshold must pass.
    rng = np.random.default_rng(5)
    checked = rejected = 0
    for _ in range(600):
        n = int(rng.integers(8, 60))
        a = rng.choice(list(b"ACGT"), n).astype(np.uint8)
        b = a.copy()
        for _ in range(int(rng.integers(0, 6))):
            k = int(rng.integers(0, 3))
            p = int(rng.integers(0, len(b)))
            if k == 0:
                b[p] = rng.choice(list(b"ACGT"))
            elif k == 1 and len(b) > 2:
                b = np.delete(b, p)
                b = np.append(b, rng.choice(list(b"ACGT"))).astype(np.uint8)
            else:
                b = np.insert(b, p, rng.choice(list(b"ACGT")))[:n].astype(np.uint8)
        thr = float(rng.choice([0.05, 0.1, 0.2, 0.3]))
        et = int(np.floor(np.float32(n) * np.float32(thr)))
        got = oracle.sneaky_snake(a.tobytes(), b.tobytes(), thr)
        assert got is not None
        if edit_distance(a.tobytes(), b.tobytes()) <= et:
            assert got is True
            checked += 1
        elif got is False:
            rejected += 1
    assert checked > 100 and rejected > 20

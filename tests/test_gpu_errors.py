"""Error behaviour of the C ABI on a GPU box: every misuse returns an integer code and a message, never aborts, and leaves the
context usable ("integer error code return, never abort"). ProfileError codes follow profile.rs:32-44."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: the -m gpu tests need an MI355X")
    import zoe_amd
    from zoe_amd import _lib

    lib = _lib.load()
    h = C.c_void_p()
    assert lib.zsw_create(0, C.byref(h)) == 0
    yield zoe_amd, _lib, lib, h
    lib.zsw_destroy(h)


def host_batch(_lib, seqs):
    cat = np.frombuffer(b"".join(seqs), dtype=np.uint8).copy()
    offs = np.zeros(len(seqs) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([len(s) for s in seqs])
    b = _lib.ZswBatch()
    b.bases, b.offsets, b.fixed_len, b.n_reads, b.mem = cat.ctypes.data, offs.ctypes.data, 0, len(seqs), _lib.MEM_HOST
    return b, cat, offs


def test_error_codes_and_recovery(env):
    za, _lib, lib, h = env
    m = za.WeightMatrix.new_dna_matrix(2, -5, b"N")
    w = np.ascontiguousarray(m.signed_weights(), dtype=np.int8)
    im = m.mapping.index_map
    reads = [b"ACGTACGTAC", b"GGGTTTAAACCC"]
    b, cat, offs = host_batch(_lib, reads)
    score = np.zeros(2, dtype=np.uint32)
    status = np.zeros(2, dtype=np.uint8)
    msg = lambda: lib.zsw_last_error_string(h).decode()
    # not configured yet
    assert lib.zsw_score_batch(h, C.byref(b), 1, 16, score.ctypes.data, status.ctypes.data, None) == -5 and "not set" in msg()
    # ProfileError codes of validate_profile_args (profile.rs:32-44)
    assert lib.zsw_set_scoring(h, w.ctypes.data, 5, im.ctypes.data, 1, -1) == 2      # GapOpenOutOfRange
    assert lib.zsw_set_scoring(h, w.ctypes.data, 5, im.ctypes.data, -10, 1) == 3     # GapExtendOutOfRange
    assert lib.zsw_set_scoring(h, w.ctypes.data, 5, im.ctypes.data, -1, -10) == 4    # BadGapWeights
    assert lib.zsw_set_scoring(h, w.ctypes.data, 0, im.ctypes.data, -10, -1) == -1   # S out of range
    assert lib.zsw_set_scoring(h, w.ctypes.data, 3, im.ctypes.data, -10, -1) == -1 and "index_map" in msg()  # map entry >= S
    assert lib.zsw_set_scoring(h, w.ctypes.data, 5, im.ctypes.data, -10, -1) == 0
    ref = np.frombuffer(b"TTACGTACGTACTTGGGTTTAAACCCTT", dtype=np.uint8)
    assert lib.zsw_set_reference(h, ref.ctypes.data, len(ref), _lib.MEM_HOST) == 0
    # bad instantiation parameters
    assert lib.zsw_score_batch(h, C.byref(b), 1, 3, score.ctypes.data, status.ctypes.data, None) == -1 and "lanes" in msg()
    assert lib.zsw_score_batch(h, C.byref(b), 9, 16, score.ctypes.data, status.ctypes.data, None) == -1
    assert lib.zsw_score_batch_from(h, C.byref(b), 8, 200, score.ctypes.data, status.ctypes.data, None, None) == -1 and "preset" in msg()
    assert lib.zsw_score_batch_from(h, C.byref(b), 12, 256, score.ctypes.data, status.ctypes.data, None, None) == -1 and "from_width" in msg()
    assert lib.zsw_score_batch(h, C.byref(b), 1, 16, None, status.ctypes.data, None) == -1 and "null" in msg()
    assert lib.zsw_score_batch(None, C.byref(b), 1, 16, score.ctypes.data, status.ctypes.data, None) == -1
    # offsets that go backwards
    bad = offs.copy()
    bad[1] = 50
    bb = _lib.ZswBatch()
    bb.bases, bb.offsets, bb.fixed_len, bb.n_reads, bb.mem = cat.ctypes.data, bad.ctypes.data, 0, 2, _lib.MEM_HOST
    assert lib.zsw_score_batch(h, C.byref(bb), 1, 16, score.ctypes.data, status.ctypes.data, None) == -1 and "monotone" in msg()
    # neither offsets nor a fixed length: every read would be empty
    be = _lib.ZswBatch()
    be.bases, be.offsets, be.fixed_len, be.n_reads, be.mem = cat.ctypes.data, None, 0, 2, _lib.MEM_HOST
    assert lib.zsw_score_batch(h, C.byref(be), 1, 16, score.ctypes.data, status.ctypes.data, None) == 1  # EmptySequence
    # ciglet capacity too small: INVALID_ARGUMENT and the required size comes back
    aln = np.zeros(2 * 40, dtype=np.uint8)
    inc = np.zeros(1, dtype=np.uint32)
    op = np.zeros(1, dtype=np.uint8)
    total = C.c_uint64(0)
    rc = lib.zsw_align_batch(h, C.byref(b), 1, 16, 0, aln.ctypes.data, status.ctypes.data, inc.ctypes.data, op.ctypes.data, 1, C.byref(total), None)
    assert rc == -1 and total.value >= 2
    inc = np.zeros(total.value, dtype=np.uint32)
    op = np.zeros(total.value, dtype=np.uint8)
    rc = lib.zsw_align_batch(h, C.byref(b), 1, 16, 0, aln.ctypes.data, status.ctypes.data, inc.ctypes.data, op.ctypes.data, total.value, C.byref(total), None)
    assert rc == 0 and list(status) == [0, 0]
    # the context still works after all of the above
    assert lib.zsw_score_batch(h, C.byref(b), 1, 16, score.ctypes.data, status.ctypes.data, None) == 0
    assert list(score) == [20, 24] and list(status) == [0, 0]
    # zero reads is a valid call
    bz = _lib.ZswBatch()
    bz.bases, bz.offsets, bz.fixed_len, bz.n_reads, bz.mem = cat.ctypes.data, offs.ctypes.data, 0, 0, _lib.MEM_HOST
    assert lib.zsw_score_batch(h, C.byref(bz), 1, 16, score.ctypes.data, status.ctypes.data, None) == 0


def test_two_contexts_on_two_host_threads(env):
    """"Thread-safe per context": two host threads, each with its own context on the same GPU and its own stream, score ragged
    batches concurrently (ctypes releases the GIL during the calls); every result equals the single-threaded one."""
    import threading

    import torch

    za, _lib, lib, _ = env
    from zoe_amd import synth

    m = za.WeightMatrix.new_dna_matrix(2, -5, b"N")
    w = np.ascontiguousarray(m.signed_weights(), dtype=np.int8)
    im = m.mapping.index_map
    refs = [synth.reference_host(3000, seed=s) for s in (1, 2)]
    batches = [synth.reads_ragged_host(refs[k], 0, 20000, 60, 400, seed=10 + k) for k in range(2)]
    results = [[], []]
    errors = []

    def worker(k):
        try:
            h = C.c_void_p()
            assert lib.zsw_create(0, C.byref(h)) == 0
            assert lib.zsw_set_scoring(h, w.ctypes.data, 5, im.ctypes.data, -10, -1) == 0
            ref = np.frombuffer(refs[k], dtype=np.uint8)
            assert lib.zsw_set_reference(h, ref.ctypes.data, len(ref), _lib.MEM_HOST) == 0
            bases, offs = batches[k]
            offs = np.ascontiguousarray(offs, dtype=np.uint64)
            b = _lib.ZswBatch()
            b.bases, b.offsets, b.fixed_len, b.n_reads, b.mem = bases.ctypes.data, offs.ctypes.data, 0, len(offs) - 1, _lib.MEM_HOST
            stream = torch.cuda.Stream()
            for _ in range(6):
                score = np.zeros(len(offs) - 1, dtype=np.uint32)
                status = np.zeros(len(offs) - 1, dtype=np.uint8)
                tier = np.zeros(len(offs) - 1, dtype=np.uint8)
                rc = lib.zsw_score_batch_from(h, C.byref(b), 8, 256, score.ctypes.data, status.ctypes.data, tier.ctypes.data, C.c_void_p(stream.cuda_stream))
                assert rc == 0
                results[k].append((score, status, tier))
            lib.zsw_destroy(h)
        except Exception as e:  # noqa: BLE001
            errors.append((k, repr(e)))

    ts = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors
    for k in range(2):
        first = results[k][0]
        for other in results[k][1:]:
            for a, b_ in zip(first, other):
                assert np.array_equal(a, b_)
        # and equal to a quiet, single-threaded run through the Python mirror
        bases, offs = batches[k]
        seqs = [bases[int(offs[i]):int(offs[i + 1])].tobytes() for i in range(0, 400)]
        quiet = za.LocalProfilesBatch.new_with_w256(seqs, m, -10, -1).sw_score_from_i8(refs[k])
        assert np.array_equal(quiet.score.cpu().numpy().view(np.uint32), first[0][:400]) and np.array_equal(quiet.status.cpu().numpy(), first[1][:400])

"""GPU tests of the multi-device entry points (zsw_group_*, include/zoe_sw.h): a group shards a batch over its contexts and
must return exactly what one context returns. A one-GPU box exercises the host-memory path with two contexts on device 0
(two host threads, two streams of work on one GPU) and the RCCL leg at world size 1; ragged and fixed-length batches."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def za():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: the -m gpu tests need an MI355X")
    import zoe_amd

    return zoe_amd


def _reference_and_reads(n, fixed):
    from zoe_amd import synth

    ref = synth.reference_host(2000)
    if fixed:
        host = synth.reads_host(ref, 5000, n, 150)
        return ref, host.reshape(-1), None, 150
    rng = np.random.default_rng(11)
    lens = rng.integers(1, 400, size=n)
    lens[::97] = 0  # empty reads: ZSW_STATUS_EMPTY
    off = np.zeros(n + 1, dtype=np.uint64)
    np.cumsum(lens, out=off[1:])
    full = synth.reads_host(ref, 0, n, 400)
    bases = np.concatenate([full[i, : lens[i]] for i in range(n)]) if n else np.zeros(0, np.uint8)
    return ref, bases, off, 0


@pytest.mark.parametrize("fixed", [True, False])
@pytest.mark.parametrize("n", [0, 1, 7, 20001])
def test_two_contexts_on_one_gpu_equal_one_context(za, fixed, n):
    import torch

    ref, bases, off, L = _reference_and_reads(n, fixed)
    dna = za.WeightMatrix.new_dna_matrix(2, -5, b"N")
    g = za.SwGroup([0, 0])
    try:
        assert len(g) == 2
        g.configure(dna, -10, -1, ref)
        s, st, t = g.sw_score_from_host(bases, n, fixed_len=L, offsets=off)
    finally:
        g.close()
    if n == 0:
        return
    if fixed:
        rb = za.ReadBatch.from_fixed(torch.from_numpy(bases).cuda(), 150)
    else:
        rb = za.ReadBatch(torch.from_numpy(bases if bases.size else np.zeros(1, np.uint8)).cuda(), n, offsets=torch.from_numpy(off.astype(np.int64)).cuda())
    one = za.LocalProfilesBatch.new_with_w256(rb, dna, -10, -1).sw_score_from_i8(ref)
    assert np.array_equal(st, one.status.cpu().numpy())
    some = st == 0
    assert np.array_equal(s[some], one.score.cpu().numpy().view(np.uint32)[some])
    assert np.array_equal(t[some], one.tier.cpu().numpy()[some])


@pytest.mark.parametrize("three_pass", [False, True])
@pytest.mark.parametrize("fixed", [True, False])
@pytest.mark.parametrize("n", [0, 1, 5003])
def test_group_alignments_equal_one_context(za, fixed, n, three_pass):
    """zsw_group_align_batch_from / _3pass_batch_from: records in read order, the shards' ciglets back to back."""
    import torch

    ref, bases, off, L = _reference_and_reads(n, fixed)
    dna = za.WeightMatrix.new_dna_matrix(2, -5, b"N")
    g = za.SwGroup([0, 0, 0])
    try:
        g.configure(dna, -10, -1, ref)
        got = g.sw_align_from_host(bases, n, fixed_len=L, offsets=off, three_pass=three_pass)
        if n:
            tiny = g.lib.zsw_group_align_3pass_batch_from if three_pass else g.lib.zsw_group_align_batch_from  # capacity too small: the required size comes back, nothing is written
            b = za._lib.ZswBatch()
            b.bases, b.fixed_len, b.n_reads, b.mem = bases.ctypes.data, L, n, za._lib.MEM_HOST
            o64 = None
            if off is not None:
                o64 = np.ascontiguousarray(off, dtype=np.uint64)
                b.offsets = o64.ctypes.data
            import ctypes as C

            rec = np.zeros(n, dtype=za.alignment.ALN_DTYPE)
            st = np.zeros(n, dtype=np.uint8)
            total = C.c_uint64(0)
            one_inc, one_op = np.zeros(1, np.uint32), np.zeros(1, np.uint8)
            rc = tiny(g.h, C.byref(b), 8, 256, 0, rec.ctypes.data, st.ctypes.data, None, one_inc.ctypes.data, one_op.ctypes.data,
                      1 if len(got.inc) > 1 else 0, C.byref(total))
            if len(got.inc) > 1:
                assert rc == -1 and total.value == len(got.inc)
                assert b"capacity" in g.lib.zsw_group_last_error_string(g.h)
    finally:
        g.close()
    if n == 0:
        assert len(got.status) == 0 and len(got.inc) == 0
        return
    if fixed:
        rb = za.ReadBatch.from_fixed(torch.from_numpy(bases).cuda(), 150)
    else:
        rb = za.ReadBatch(torch.from_numpy(bases if bases.size else np.zeros(1, np.uint8)).cuda(), n, offsets=torch.from_numpy(off.astype(np.int64)).cuda())
    prof = za.LocalProfilesBatch.new_with_w256(rb, dna, -10, -1)
    seq = za.SeqSrc.Reference(ref)
    one = prof.sw_align_from_i8_3pass(seq) if three_pass else prof.sw_align_from_i8(seq)
    assert np.array_equal(got.status, one.status)
    assert np.array_equal(got.tier, one.tier)
    for i in range(n):
        assert got.key(i) == one.key(i), i


@pytest.mark.parametrize("fixed", [True, False])
def test_eight_contexts_on_one_gpu_equal_one_context(za, fixed):
    """The shard arithmetic of a whole node (G = 8: [i*n/8, (i+1)*n/8), eight worker threads that live with the group) through the
    host path, on device 0: scores, tiers and alignments of a batch whose size is not a multiple of 8 equal one context's; two
    calls in a row reuse the workers."""
    import torch

    n = 40_003
    ref, bases, off, L = _reference_and_reads(n, fixed)
    dna = za.WeightMatrix.new_dna_matrix(2, -5, b"N")
    g = za.SwGroup([0] * 8)
    try:
        assert len(g) == 8
        g.configure(dna, -10, -1, ref)
        s, st, t = g.sw_score_from_host(bases, n, fixed_len=L, offsets=off)
        s2, st2, t2 = g.sw_score_from_host(bases, n, fixed_len=L, offsets=off)
        aln = g.sw_align_from_host(bases, 2001, fixed_len=L, offsets=None if off is None else off[:2002])
    finally:
        g.close()
    assert np.array_equal(s, s2) and np.array_equal(st, st2) and np.array_equal(t, t2)
    if fixed:
        rb = za.ReadBatch.from_fixed(torch.from_numpy(bases).cuda(), 150)
        rb_a = za.ReadBatch.from_fixed(torch.from_numpy(bases[: 2001 * 150]).cuda(), 150)
    else:
        o64 = torch.from_numpy(off.astype(np.int64)).cuda()
        rb = za.ReadBatch(torch.from_numpy(bases).cuda(), n, offsets=o64)
        rb_a = za.ReadBatch(torch.from_numpy(bases).cuda(), 2001, offsets=o64[:2002].contiguous())
    one = za.LocalProfilesBatch.new_with_w256(rb, dna, -10, -1).sw_score_from_i8(ref)
    assert np.array_equal(st, one.status.cpu().numpy())
    some = st == 0
    assert np.array_equal(s[some], one.score.cpu().numpy().view(np.uint32)[some])
    assert np.array_equal(t[some], one.tier.cpu().numpy()[some])
    one_a = za.LocalProfilesBatch.new_with_w256(rb_a, dna, -10, -1).sw_align_from_i8(za.SeqSrc.Reference(ref))
    assert np.array_equal(aln.status, one_a.status)
    for i in range(0, 2001, 7):
        assert aln.key(i) == one_a.key(i), i


def test_the_device_entry_point_rejects_duplicate_devices(za):
    """RCCL wants one rank per GPU: two contexts on device 0 are fine for the host path and ZSW_ERR_INVALID_ARGUMENT here (before
    ncclCommInitAll is reached)."""
    import torch

    from zoe_amd import _lib, synth

    ref = synth.reference_host(2000)
    dna = za.WeightMatrix.new_dna_matrix(2, -5, b"N")
    g = za.SwGroup([0, 0])
    try:
        g.configure(dna, -10, -1, ref)
        rb = za.ReadBatch.from_fixed(torch.from_numpy(synth.reads_host(ref, 1, 64, 150).reshape(-1)).cuda(), 150)
        with pytest.raises(_lib.ZswError) as ei:
            g.sw_score_from_device([rb, rb])
        assert ei.value.code == -1 and "distinct devices" in str(ei.value)
    finally:
        g.close()


def test_device_shards_and_the_rccl_gather_at_world_size_one(za, oracle):
    """Device-memory form: the context writes its slice of the device's result arrays and the grouped RCCL broadcast completes
    them (world size 1 here: the collective is issued and must leave the results intact); checked against the oracle."""
    import torch

    from zoe_amd import synth

    ref = synth.reference_host(2000)
    n = 3000
    host = synth.reads_host(ref, 123, n, 150)
    dna = za.WeightMatrix.new_dna_matrix(2, -5, b"N")
    g = za.SwGroup([0])
    try:
        g.configure(dna, -10, -1, ref)
        rb = za.ReadBatch.from_fixed(torch.from_numpy(host.reshape(-1)).cuda(), 150)
        (score, status), = g.sw_score_from_device([rb])
    finally:
        g.close()
    sc = oracle.dna_scoring(2, -5, b"N", -10, -1)
    ws, wst, _ = oracle.batch_score_w256(8, sc, host, ref, fixed_len=150, threads=8)
    assert np.array_equal(status.cpu().numpy(), wst)
    assert np.array_equal(score.cpu().numpy().view(np.uint32), ws)


def test_group_errors_are_reported_not_fatal(za):
    from zoe_amd import _lib

    with pytest.raises(_lib.ZswError):
        za.SwGroup([99])  # no such device
    g = za.SwGroup([0, 0])
    try:
        with pytest.raises(_lib.ZswError):  # not configured
            g.sw_score_from_host(np.zeros(150, np.uint8), 1, fixed_len=150)
    finally:
        g.close()

"""Host-side mirror (zoe_amd.alignment) of the reference's types: same tables, same validation, same errors."""
import numpy as np
import pytest


def test_dna_profile_map_matches_oracle(oracle):
    import zoe_amd as za

    assert np.array_equal(za.DNA_PROFILE_MAP.index_map, oracle.dna_profile_map())
    m = za.ByteIndexMap.new(b"ABCD", b"A")
    assert np.array_equal(m.index_map, oracle.byte_index_map(b"ABCD", b"A"))
    with pytest.raises(ValueError):
        za.ByteIndexMap.new(b"ABCD", b"Z")


def test_weight_matrix_matches_oracle(oracle):
    import zoe_amd as za

    for ma, mi, ign in ((2, -5, b"N"), (4, -2, None), (127, 0, b"N"), (1, -1, b"A")):
        m = za.WeightMatrix.new_dna_matrix(ma, mi, ign)
        want = oracle.weight_matrix_new(oracle.dna_profile_map(), 5, ma, mi, ign)
        assert np.array_equal(m.weights, want)
        b = m.to_biased_matrix()
        wb, bias = oracle.to_biased_matrix(want)
        assert np.array_equal(b.weights, wb) and b.bias == bias
        assert np.array_equal(b.signed_weights(), want)
    # src/data/matrices/mod.rs:601-631
    m = za.WeightMatrix.new(za.DNA_PROFILE_MAP, 2, -5, b"N")
    assert m.weights.tolist() == [[2, -5, -5, -5, 0], [-5, 2, -5, -5, 0], [-5, -5, 2, -5, 0], [-5, -5, -5, 2, 0], [0, 0, 0, 0, 0]]
    assert m.get_weight(ord("a"), ord("A")) == 2 and m.get_weight(ord("U"), ord("t")) == 2 and m.get_weight(ord("N"), ord("A")) == 0
    with pytest.raises(ValueError):
        za.WeightMatrix.new(za.DNA_PROFILE_MAP, 2, -5, b"Z")


def test_validate_profile_args(oracle):
    import zoe_amd as za

    for args in ((0, -10, -1), (5, 1, 0), (5, -128, -1), (5, -10, 1), (5, -10, -128), (5, -1, -2), (5, -10, -1), (5, 0, 0), (5, -127, -127)):
        code = oracle.validate_profile_args(*args)
        if code == 0:
            za.validate_profile_args(*args)
        else:
            with pytest.raises(za.ProfileError) as ei:
                za.validate_profile_args(*args)
            assert ei.value.code == code


def test_shard_range():
    from zoe_amd.dist import shard_range

    for n in (0, 1, 7, 10_000_000, 500_000_000):
        for world in (1, 2, 3, 8):
            parts = [shard_range(n, r, world) for r in range(world)]
            assert parts[0][0] == 0 and sum(c for _, c in parts) == n
            for (f0, c0), (f1, _) in zip(parts, parts[1:]):
                assert f0 + c0 == f1
            assert max(c for _, c in parts) - min(c for _, c in parts) <= 1


def test_shared_profiles_mirror_has_the_profile_set_methods():
    # profile_set.rs:552-700: SharedProfiles carries the ProfileSets methods; here it is built from ONE sequence and its methods
    # take the batch of reads (the one-profile-many-sequences role, zsw_*_shared_batch)
    import zoe_amd

    for name in ("sw_score_from_i8", "sw_score_from_i32", "sw_align_from_i16", "sw_score_ranges_from_i8", "new_with_w128", "new_with_w512"):
        assert hasattr(zoe_amd.SharedProfilesBatch, name)
    for name in ("sw_score", "sw_score_ends", "sw_score_ranges", "sw_align"):
        assert hasattr(zoe_amd.SharedStripedProfile, name)
    import inspect

    assert list(inspect.signature(zoe_amd.into_shared_profile).parameters)[:4] == ["sequence", "matrix", "gap_open", "gap_extend"]

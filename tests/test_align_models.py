"""CPU checks of the alignment kernel's arithmetic (no GPU): the closed form of Zoe's lazy-F loop (striped.rs:528-553)
and the packed row update the gfx950 kernel is compiled from (zoe_amd/csrc/zsw_align_pk.hpp), both compared cell by
cell with the oracle's literal restatement of sw_simd_align; and the bound checks of the column-pruned first pass
(zoe_amd/csrc/zsw_score_prune.hip) against the full DP matrix."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


_built = {}


def _build(tmp_path, name):
    """Compiles tests/models/<name>.cpp once per session (the twin instantiates 6 lane counts x 12 vector counts)."""
    if name not in _built:
        import tempfile

        exe = os.path.join(tempfile.mkdtemp(prefix="zsw_models_"), name)
        subprocess.run(["g++", "-O1", "-std=c++17", "-Wall", "-Wno-unknown-pragmas", "-o", exe, os.path.join(ROOT, "tests", "models", name + ".cpp")], check=True)
        _built[name] = exe
    return _built[name]


@pytest.mark.parametrize("seed", [20261004, 7])
def test_closed_form_lazy_f_equals_the_literal_loop(tmp_path, seed):
    """T = (round, vector) of the loop's break from per-round bit strings, flags and H from M(v, lane): identical striped
    backtrack matrices for N = 2..64, ten scoring schemes (gap_open = 0, gap_extend = 0 and equal gaps among them)."""
    out = subprocess.run([_build(tmp_path, "align_closed_form"), "120", str(seed)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr


@pytest.mark.parametrize("seed", [20261004, 99])
def test_packed_row_update_twin_equals_the_oracle(tmp_path, seed):
    """zsw_align_pk.hpp compiled for the host (64 explicit lanes, the plain-C meaning of each gfx950 instruction): 2*64/N reads
    per wavefront with different lengths and last rows, the first rows through the flag-less scan path."""
    out = subprocess.run([_build(tmp_path, "align_pk_twin"), "5", str(seed)], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout + out.stderr


@pytest.mark.parametrize("seed", [20261004, 5, 77])
def test_pruning_bounds_model(tmp_path, seed):
    """The claim the column-pruned first pass rests on (zoe_amd/csrc/zsw_score_prune.hip), with plain integers against the full
    Gotoh matrix: a read that passes the three bound checks has the true maximum (and, with the strict checks, the true first
    row and column); strips of 8-48 columns, six scoring schemes (free gaps and gap_extend 0 among them), repeats, second
    copies, long gaps, junk ends, reads hanging over the reference ends."""
    out = subprocess.run([_build(tmp_path, "prune_bounds"), "80", str(seed)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "prune_bounds OK" in out.stdout


@pytest.mark.parametrize("seed", [20261004, 11, 314])
def test_seeded_pass_bounds_model(tmp_path, seed):
    """The claims the seeded exact pass rests on (zoe_amd/csrc/zsw_seed.hpp, compiled for the host as the kernels compile it),
    with plain integers against the full Gotoh matrix: no path that starts above the window, starts below it, or leaves it through
    the last row beats its bound — checked for every read against DPs restricted to each class of paths — and a read that passes
    has the true maximum (with the strict checks: the true first row and column). Eight scoring schemes (free extension, N scored
    -1 / +1), k = 3..6 on references of 60-420 bases so that chance k-mer hits abound, repeats, tandem repeats, N runs, chimeras,
    long gaps, reads hanging over the ends."""
    out = subprocess.run([_build(tmp_path, "seed_bounds"), "400", str(seed)], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "seed_bounds OK" in out.stdout


@pytest.mark.parametrize("seed", [20261004, 23])
def test_late_start_certificate_model(tmp_path, seed):
    """The certificate that lets the alignment's second pass start a few rows above the alignment instead of warmup_rows above
    it (zsw_seed.hpp: seed_safe_start): Zoe's own striped alignment (the oracle's sw_simd_align at <i16, 4 / 8 / 16>) of a read
    against reference[r0..] equals its alignment against the whole reference, shifted by r0 — score, ranges and CIGAR — for every
    read that gets a certificate; seven scoring schemes, repeats, tandem repeats, N, junk prefixes, long deletions."""
    out = subprocess.run([_build(tmp_path, "seed_warmup"), "400", str(seed)], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "seed_warmup OK" in out.stdout


@pytest.mark.parametrize("seed", [20261004, 7, 41])
def test_banded_pass_bounds_model(tmp_path, seed):
    """The banded form of the seeded pass (zoe_amd/csrc/zsw_score_band.hip) in its injected form (zsw_seed.hpp, "banded pass"):
    what enters the band from outside is an upper bound of the outside cell's value — from two bound programmes along the query
    columns, one per side (seed_col_step / seed_col_join / seed_strip_events / seed_exit_is_free) —, doubled and made odd, so that
    an even maximum is the score of a real path inside the band that no path through an outside cell reaches. The model walks
    a read as the kernel does and checks against the full Gotoh matrix, for every read whether accepted or not: every cell of the
    band holds a bound >= its true H; every cell above / below the band is <= a(c) / b(c) of its column and <= oa / ob; what the
    next strip's first column receives covers the cells left of it; per class of paths (wholly outside; leaving strip k through
    its right edge / its last row) the class's best path stays under the bound programme of that class alone, so that a weak
    bound cannot hide behind a larger one; an accepted read has the true score (ends tag: the true first row and column, and the
    true number of cells holding the maximum). Copies with 0-12 % substitutions + indels, chimeras, long deletions / insertions,
    overhanging and random reads, structured cases (residues without potential between the sampled k-mers, tests/models/
    adversarial_reads.hpp); eleven schemes (free gap extension, mismatch loss >= gap_open); strips of 5-32 columns, bands of a few
    diagonals, lane partners that widen the band or add strips of padding."""
    out = subprocess.run([_build(tmp_path, "seed_band"), "250", str(seed)], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "seed_band OK" in out.stdout


@pytest.mark.parametrize("seed", [3, 20261005])
def test_reversed_pass_over_whole_sequences_model(tmp_path, seed):
    """The shared role's sw_score_ranges runs its second pass over the whole reversed sequences (a seeded pass) instead of the
    reversed prefixes of striped.rs:355-388 and accepts a read only if the forward maximum and the reversed maximum each sit in
    one cell (zsw_capi_shared.hip, settle_reverse_kernel). Claim, against plain Gotoh matrices: with the forward maximum in one
    cell, the cells holding the score in the reversed matrix of the prefixes and of the whole sequences are the same positions —
    random schemes (asymmetric matrices, N scoring -1/0/+1, free gap extension), repeats, low-complexity pairs."""
    out = subprocess.run([_build(tmp_path, "reverse_unique"), "4000", str(seed)], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "reverse_unique OK" in out.stdout


@pytest.mark.parametrize("seed", [20261005, 8])
def test_gapless_alignment_certificate_model(tmp_path, seed):
    """The certificate that lets sw_simd_align's second pass be skipped (zsw_capi.hip run_align, zsw_threepass.hip classify pass in
    certificate mode): both maxima in one cell each, ranges of equal length whose diagonal adds up to the score, and the score beyond
    maxw * (n - 1) - 2 * gap_open. The oracle's literal sw_simd_align (striped.rs:449-598 restated) must then return the gapless
    diagonal at N = 2 .. 64 in 16-bit lanes and N = 16, 32 in 8-bit lanes; ten scoring schemes, repeats, homopolymer runs, N, junk
    ends, reads with indels (never certified). (Dropping the score condition produces a counter-example within 3,000 iterations.)"""
    out = subprocess.run([_build(tmp_path, "align_gapless_cert"), "500", str(seed)], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "align_gapless_cert OK" in out.stdout


@pytest.mark.parametrize("seed", [20261005, 6])
def test_one_gap_alignment_certificate_model(tmp_path, seed):
    """The second certificate of run_align (zsw_threepass.hip, classify pass in certificate mode): both maxima in one cell each,
    ranges that differ by g, one placement of ONE gap run of g between the corners — or a run of adjacent placements, of which the
    walk from the end takes the last — reaches the score (one sweep over the prefix sums of the two diagonals), and the score lies beyond maxw * min(rlen, qlen) - 2 * gap_open - max(g - 2, 0) * gap_extend,
    which no alignment with two gap runs reaches. The oracle's literal sw_simd_align must then return [p M][g D|I][m - p M] at
    N = 2 .. 64 in 16-bit lanes and N = 16, 32 in 8-bit lanes; ten schemes, gaps of 1-5 inside repeats and homopolymer runs (tied
    placements), reads with a second gap. (Taking any tied placement but the last, or dropping the two-run bound, produces a
    counter-example within 3,000 iterations.)"""
    out = subprocess.run([_build(tmp_path, "align_onegap_cert"), "500", str(seed)], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "align_onegap_cert OK" in out.stdout


@pytest.mark.parametrize("seed", [20261005, 4])
def test_row_chunked_full_pass_model(tmp_path, seed):
    """Reads the seeded pass hands back against a long reference are scored in chunks of rows, each from a zero state with an
    overlap of L + L * maxw / gap_extend + 2 rows (zsw_score_v2.hpp, ScoreArgsV2::chunk_rows): the largest (score, earliest row,
    earliest column) over the chunks is the whole matrix's, with repeats in the reference and long deletions in the reads."""
    out = subprocess.run([_build(tmp_path, "chunk_rows"), "2500", str(seed)], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "chunk_rows OK" in out.stdout

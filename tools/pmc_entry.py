"""One secondary entry point of bench.py, alone, for a rocprofv3 --pmc pass: CALLS identical calls (no warm-up distinction), so that
the counters of the whole run divided by CALLS are the counters of one call.
usage: rocprofv3 --pmc SQ_INSTS_VALU --output-format csv -d <dir> -- python3 tools/pmc_entry.py <ranges|3pass|mixed|align> [n_reads]
       then: python tools/summarize_pmc_entry.py <tag> <dir>=<entry>:<n_reads> ..."""
import sys

import torch

sys.path.insert(0, ".")
import zoe_amd
from zoe_amd import synth

CALLS = 3
entry = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
ctx = zoe_amd.SwContext.get(0)
dna = zoe_amd.WeightMatrix.new_dna_matrix(2, -5, b"N")
if entry == "mixed":
    ref = synth.reference_host(30000)
    rb = synth.reads_ragged_device(ctx, ref, 0, n, 75, 400)
    prof = zoe_amd.into_local_profile(rb, dna, -10, -1, device=ctx.device)
    fn = lambda: prof.sw_score_from_i8(ref)
else:
    ref = synth.reference_host(2000)
    rb = synth.reads_device(ctx, ref, 0, n, 150)
    if entry == "ranges":
        sp = zoe_amd.StripedProfileBatch(rb, dna, -10, -1, T="i16", N=16, device=ctx.device)
        fn = lambda: sp.sw_score_ranges(zoe_amd.SeqSrc.Reference(ref))
    else:
        prof = zoe_amd.into_local_profile(rb, dna, -10, -1, device=ctx.device)
        if entry == "3pass":
            fn = lambda: prof.sw_align_from_i8_3pass(zoe_amd.SeqSrc.Reference(ref))
        elif entry == "align":
            fn = lambda: prof.sw_align_from_i8(zoe_amd.SeqSrc.Reference(ref))
        else:
            raise SystemExit("entry: ranges | 3pass | mixed | align")
for _ in range(CALLS):
    r = fn()
    torch.cuda.synchronize()
    del r
print(f"{entry}: {CALLS} calls of {n} reads")

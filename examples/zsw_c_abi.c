/* zsw_c_abi.c — the boundary is a C ABI: this file is plain C99, includes include/zoe_sw.h and takes the address of every entry
 * point it declares (tests/test_capi_symbols.py compiles it with gcc -std=c99 -pedantic and links it against the library). */
#include <stdio.h>

#include "zoe_sw.h"

int main(void) {
    typedef void (*fn)(void);
    const fn entry_points[] = {
        (fn)zsw_create, (fn)zsw_destroy, (fn)zsw_last_error_string, (fn)zsw_device_count,
        (fn)zsw_set_scoring, (fn)zsw_set_reference, (fn)zsw_score_batch, (fn)zsw_score_batch_from,
        (fn)zsw_score_ends_batch, (fn)zsw_score_ranges_batch, (fn)zsw_score_ranges_batch_from,
        (fn)zsw_align_batch, (fn)zsw_align_batch_from, (fn)zsw_align_3pass_batch,
        (fn)zsw_align_3pass_batch_from, (fn)zsw_sneaky_snake_batch, (fn)zsw_set_profile_sequence,
        (fn)zsw_score_shared_batch, (fn)zsw_score_shared_batch_from, (fn)zsw_score_ends_shared_batch, (fn)zsw_score_ranges_shared_batch,
        (fn)zsw_score_ranges_shared_batch_from, (fn)zsw_align_shared_batch, (fn)zsw_align_shared_batch_from, (fn)zsw_align_3pass_shared_batch, (fn)zsw_align_3pass_shared_batch_from, (fn)zsw_synth_reads,
        (fn)zsw_synth_reads_ragged, (fn)zsw_synth_length, (fn)zsw_synth_reference_host,
        (fn)zsw_synth_reads_host, (fn)zsw_synth_reads_ragged_host, (fn)zsw_selftest,
        (fn)zsw_timing_enable, (fn)zsw_timing_read, (fn)zsw_timing_read_window, (fn)zsw_debug_set, (fn)zsw_debug_band_records, (fn)zsw_pack4_host, (fn)zsw_prune_rescored, (fn)zsw_set_option,
        (fn)zsw_group_create, (fn)zsw_group_destroy, (fn)zsw_group_size, (fn)zsw_group_context,
        (fn)zsw_group_last_error_string, (fn)zsw_group_set_scoring, (fn)zsw_group_set_reference,
        (fn)zsw_group_score_batch_from, (fn)zsw_group_score_batch_from_device,
        (fn)zsw_group_align_batch_from, (fn)zsw_group_align_3pass_batch_from,
    };
    zsw_context* ctx = NULL;
    zsw_batch b;
    zsw_alignment a;
    int rc;
    b.bases = NULL, b.offsets = NULL, b.fixed_len = 0, b.n_reads = 0, b.mem = ZSW_MEM_HOST, b.encoding = ZSW_ENCODING_BYTES;
    a.score = 0;
    rc = (int)zsw_create(0, &ctx); /* ZSW_OK on an MI355X, ZSW_ERR_NO_DEVICE elsewhere: never a crash */
    printf("%u entry points, sizeof(zsw_alignment) = %u, zsw_create -> %d (%s)\n", (unsigned)(sizeof(entry_points) / sizeof(entry_points[0])),
           (unsigned)sizeof(a), rc, rc == ZSW_OK ? "ok" : zsw_last_error_string(NULL));
    if (ctx) zsw_destroy(ctx);
    (void)b;
    return 0;
}

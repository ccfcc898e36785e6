// zsw_align.hip — full alignment (traceback) entry points. Placeholder until the striped-emulation
// kernel lands: the calls fail loudly rather than fall back to anything else.
#include "zsw_internal.hpp"

extern "C" {

zsw_error zsw_align_batch(zsw_context* ctx, const zsw_batch* reads, zsw_int_type int_type, int lanes, int invert,
                          zsw_alignment* out_aln, uint8_t* out_status, uint32_t* out_inc, uint8_t* out_op,
                          uint64_t ciglet_cap, uint64_t* out_n_ciglets, void* stream) {
    return ZSW_ERR_UNSUPPORTED;
}

zsw_error zsw_align_batch_from(zsw_context* ctx, const zsw_batch* reads, int from_width, int preset_bits, int invert,
                               zsw_alignment* out_aln, uint8_t* out_status, uint8_t* out_tier, uint32_t* out_inc,
                               uint8_t* out_op, uint64_t ciglet_cap, uint64_t* out_n_ciglets, void* stream) {
    return ZSW_ERR_UNSUPPORTED;
}
}

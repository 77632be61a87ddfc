// gs_kernels.h -- host-side launchers of the gfx950 kernels (definitions next to each kernel).
#pragma once
#include "gs_device.h"

void gs_launch_repack(const void* d_aos, uint32_t n, const GsScene& s, hipStream_t st);
void gs_launch_ply_chunk(const void* d_raw, uint32_t m, uint32_t first, const GsPlyTable& t, const GsScene& s, hipStream_t st);
struct GsPreprocessLaunch { // the projection's launch as data (its uniforms are the only kernel arguments that change per frame)
    const void* func;
    uint32_t blocks;
    GsScene s; GsUniforms u; GsFrame f;
    void* gdata; uint32_t* counts; GsTightOut to;
    void* args[6];
};
void gs_preprocess_prepare(GsPreprocessLaunch& L, const GsScene& s, const GsUniforms& u, const GsFrame& f, void* gdata, uint32_t* counts,
                           bool tight, uint32_t* arena, uint32_t* rowptr, GsControl* ctl, uint32_t tight_nb = 0);
void gs_launch_preprocess(GsPreprocessLaunch& L, hipStream_t st);
void gs_launch_zero(void* p, uint64_t bytes, hipStream_t st); // bytes: a multiple of 16
uint32_t gs_scan_blocks(uint32_t n);
void gs_launch_scan(const uint32_t* counts, uint32_t n, uint32_t* offsets, unsigned long long* status, uint32_t* ticket, GsControl* ctl,
                    hipStream_t st);
uint64_t gs_emit_chunks(uint64_t capacity);
// grec: the gaussian-level sort's records {id, count word, first output slot, -} in (bucket, index) order
void gs_launch_emit_balanced(const void* gdata, const void* grec, const uint32_t* chunk_table, const GsFrame& f, uint32_t* keys, uint32_t* values,
                             GsControl* ctl, uint32_t grid, uint32_t hist_bits, uint32_t hist_passes, bool keys16, hipStream_t st);
void gs_launch_emit(const void* gdata, const uint32_t* counts, const uint32_t* offsets, const uint32_t* perm, const uint32_t* n_dev,
                    const GsFrame& f, uint32_t* keys, uint32_t* values, GsControl* ctl, hipStream_t st);
void gs_launch_ranges16(const uint16_t* tiles, const GsControl* ctl, uint32_t capacity, uint32_t T, uint32_t* ranges, uint32_t grid,
                        uint32_t* sticky, GsReport* rep, hipStream_t st);
void gs_launch_rebuild_keys(const uint16_t* tiles, const uint32_t* vals, const uint32_t* counts, uint32_t count, uint32_t n, uint32_t id_mask,
                            uint32_t* keys, hipStream_t st);
void gs_launch_ranges(const uint32_t* keys, const GsControl* ctl, uint32_t capacity, uint32_t T, uint32_t* ranges, uint32_t grid,
                      uint32_t* sticky, GsReport* rep, hipStream_t st);
uint32_t gs_sort_tiles(uint64_t capacity);
void gs_launch_sort(uint32_t* keysA, uint32_t* valsA, uint32_t* keysB, uint32_t* valsB, GsControl* ctl, uint32_t* tickets, uint32_t* hist,
                    const uint32_t* n_ptr, uint32_t capacity, uint32_t passes, uint32_t bits, uint32_t by_tile, uint32_t* status,
                    uint32_t grid, bool have_hist, const uint32_t* aux_table, uint32_t* aux_out, hipStream_t st, uint32_t** out_keys,
                    uint32_t** out_vals, bool keys16 = false);
int gs_launch_blend(const void* gdata, const uint32_t* values, const uint32_t* ranges, const GsFrame& f, uint32_t* rgba8, float* rgbf,
                    GsControl* ctl, uint32_t* tile_depth, bool exact, uint32_t ablation, bool masked, hipStream_t st, uint32_t* prof = nullptr,
                    uint32_t* prof_blocks = nullptr);
void gs_launch_debug_view(const uint32_t* ranges, const GsFrame& f, uint32_t view, uint32_t* rgba8, hipStream_t st);
void gs_launch_assemble(const void* slabs, void* image, uint32_t width, uint32_t height, const uint32_t* d_px_bounds, uint32_t n_slabs,
                        uint64_t slab_stride_px, hipStream_t st);
// k_gsort.hip: the visible gaussians sorted by depth bucket (stable) with their quantity (tile count / row-item slots) scanned in that order
uint32_t gs_gsort_tiles(uint32_t n);
uint64_t gs_gsort_scratch_bytes(uint32_t n);
void gs_launch_gsort(const uint32_t* words, const uint32_t* aux_in, uint32_t n, void* scratch, void* grec, uint32_t* chunk_table, uint32_t chunk_cap,
                     uint32_t* tot_visible, uint32_t* tot_quantity, hipStream_t st);
// k_rows.hip: the tight row pipeline (row sort, per-chunk counts, scan, expansion into the final per-tile lists + ranges)
uint32_t gs_rows_sort_tiles(uint64_t row_cap);
uint32_t gs_rows_chunks(uint64_t row_cap);
void gs_launch_rows(const uint32_t* arena, const void* grec, const uint32_t* chunk_table, uint32_t* rows_sorted, GsControl* ctl, uint32_t* sort_status, uint32_t row_cap,
                    uint32_t* M3, uint32_t* tileoff, uint32_t* rowtot, const GsFrame& f, uint32_t* values, uint32_t* ranges, uint32_t cus,
                    uint32_t* sticky, GsReport* rep, hipStream_t st, void (*mark)(void*, int), void* mark_arg);
void gs_launch_rows_rebuild_keys(const uint32_t* ranges, uint32_t T, const uint32_t* vals, const uint32_t* counts, uint32_t count, uint32_t n,
                                 uint32_t* keys, hipStream_t st);

// brotlig_encoder.h -- C ABI of the functional Brotli-G encoder (test/bench input generator).
// Mirrors the role of BrotliG::Encode (inc/BrotligEncoder.h:34-37) with a plain-C options
// struct in place of BrotligDataconditionParams (inc/common/BrotligDataConditioner.h:27-61).
#ifndef BROTLIG_ENCODER_H
#define BROTLIG_ENCODER_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

enum {
    BROTLIG_ENC_OK = 0,
    BROTLIG_ENC_ERROR_PAGE_SIZE = 1,
    BROTLIG_ENC_ERROR_EMPTY = 2,
    BROTLIG_ENC_ERROR_TOO_MANY_PAGES = 3,      /* > 65535 pages, inc/DataStream.h:32 */
    BROTLIG_ENC_ERROR_PRECON_PARAMS = 4,
    BROTLIG_ENC_ERROR_OUTPUT_TOO_SMALL = 5
};

/* flags: knobs that steer which legal bitstream features get exercised */
enum {
    BROTLIG_ENC_NO_CODELEN_RLE      = 1u << 0,  /* code lengths as literals only (no 16/17 tokens) */
    BROTLIG_ENC_FORCE_STORED        = 1u << 1,  /* every page stored raw */
    BROTLIG_ENC_NO_RING_CODES       = 1u << 2,  /* never use distance codes 0..15 */
    BROTLIG_ENC_NO_LAZY             = 1u << 3,  /* greedy parse */
    BROTLIG_ENC_LITERALS_ONLY       = 1u << 4,  /* no matches: one insert-only command */
    BROTLIG_ENC_FORCE_COMPLEX_TABLES= 1u << 5,  /* complex description even for 2..4 symbols */
    BROTLIG_ENC_SEARCH_DIST_PARAMS  = 1u << 6,  /* per page: pick NPOSTFIX / NDIRECT by estimated distance cost */
    BROTLIG_ENC_OPTIMAL_PARSE       = 1u << 7,  /* shortest-path parse under the symbol costs of a first (lazy) parse */
    BROTLIG_ENC_SMOOTH_HISTOGRAMS   = 1u << 8,  /* smooth symbol counts so that the code lengths run-length encode better (kept per code only when smaller) */
    BROTLIG_ENC_DECODER_CORNERS     = 1u << 9   /* what the reference's decoder accepts and its encoder never writes: code-length token 16 straight after a 17-run, zero runs
                                                   as 0 + 16s; reserved / skipped header bits set; IS_DELTA on a page of a plain stream; simple codes listed in descending order */
};

typedef struct BrotligEncodeOptions {
    uint32_t page_size;      /* 32768 / 65536 / 131072 / 262144 (header index 3: beyond the reference encoder's maximum, within its decoders'); 0 -> 65536 */
    uint32_t npostfix;       /* 0..3 */
    uint32_t ndirect_m;      /* 0..15; NDIRECT = ndirect_m << npostfix */
    uint32_t flags;
    uint32_t max_chain;      /* hash-chain depth; 0 -> 24 */
    /* pre-conditioning (BC1..BC5 block textures) */
    uint32_t precondition, swizzle, delta, format;
    uint32_t width_blocks, height_blocks, num_mips, pitch_bytes, pitch_d3d12_aligned;
    uint32_t num_threads;    /* page-parallel workers; 0 -> one per hardware thread, at most 64 */
} BrotligEncodeOptions;

uint32_t BrotligEncMaxCompressedSize(uint32_t input_size, uint32_t page_size);
int BrotligEncode(uint32_t input_size, const uint8_t* src, uint32_t* output_size, uint8_t* output,
                  const BrotligEncodeOptions* opt);

#ifdef __cplusplus
}
#endif
#endif

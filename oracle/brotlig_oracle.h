/*
 * brotlig_oracle.h -- CPU restatement of the reference Brotli-G decoder.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under brotli_g_sdk_amd/ (the product)
 * may include, link or call this.  Allowed users: tests/, __graft_entry__.smoke()
 * and the cpu_baseline leg of bench.py.
 *
 * PARITY STATUS: "parity unpinned".  The reference (GPUOpen brotli_g_sdk 1.1)
 * ships no tests, golden vectors or sample streams, and its decode path cannot
 * be compiled in this image without writing stand-ins for <Windows.h>,
 * <d3d12.h> and google/brotli v1.0.9 internal headers
 * (inc/common/BrotligCommon.h:42-44, inc/common/BrotligBitReader.h:25-27),
 * which the build rules for this repo forbid.  What IS pinned: the constant
 * tables (insert/copy base+extra, bit-reversal) are checked against the
 * reference's own sBrotligCmdLut / sBrotligReverseBits data by
 * tests/test_reference_kats.py whenever /root/reference is present.
 *
 * Every function cites the reference file:line it restates (paths relative
 * to the reference tree).
 */
#ifndef BROTLIG_ORACLE_H
#define BROTLIG_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* inc/common/BrotligCommon.h:50-68 (same numeric values) */
enum {
    ORC_BROTLIG_OK = 0,
    ORC_BROTLIG_ABORTED = 1,
    ORC_BROTLIG_ERROR_CORRUPT_STREAM = 14,
    ORC_BROTLIG_ERROR_INCORRECT_STREAM_FORMAT = 15,
    ORC_BROTLIG_ERROR_GENERIC = 16
};

/* src/BrotligDecoder.cpp:35-39 -- no validation, exactly like the reference */
uint32_t DecompressedSize(uint8_t* src);

/* src/BrotligDecoder.cpp:495-519 / :426-493.  feedbackProc is accepted for
 * prototype compatibility and must be NULL (the reference's callback takes a
 * std::string, which has no C ABI). */
int DecodeCPU(uint32_t input_size, const uint8_t* src, uint32_t* output_size,
              uint8_t* output, void* feedbackProc);

/* Same as DecodeCPU but with an explicit worker count:
 *   workers == 0 : the reference policy (min(128, hw threads) workers when
 *                  numPages > 2*workers, else 1; src/BrotligDecoder.cpp:404-415)
 *   workers >= 1 : exactly that many page-parallel workers.
 * *workers_used receives the number of threads that actually ran (may be NULL). */
int brotlig_oracle_decode(uint32_t input_size, const uint8_t* src,
                          uint32_t* output_size, uint8_t* output,
                          int workers, int* workers_used);

/* One page, no container: restates PageDecoder::Run for a non-preconditioned
 * stream (src/decoder/PageDecoder.cpp:65-236).  `in` must hold in_size bytes;
 * reads past in+in_size return zero bits.  Returns 0 on success. */
int brotlig_oracle_decode_page(const uint8_t* in, uint32_t in_size,
                               uint8_t* out, uint32_t out_size,
                               uint32_t page_size);

/* inc/common/BrotligCommandLut.h:41-747 regenerated from the two 24-entry
 * base/extra tables (google/brotli v1.0.9 c/enc/command.h, RFC 7932 s.5).
 * sym in [0,704]: fills insert/copy extra bits and base values; 704 = sentinel
 * (all zero).  Used by the KAT test against the reference's table. */
void brotlig_oracle_cmd_lut(uint32_t sym, uint32_t* ins_extra, uint32_t* copy_extra,
                            uint32_t* ins_base, uint32_t* copy_base, int* implicit_dist);

/* Conditioned offset -> texture byte address (PageDecoder.cpp:406-444) for the
 * stream whose precondition header words are (w0,w1) and output size out_size.
 * Returns 0xFFFFFFFF when p is not covered by any block. Test helper. */
uint32_t brotlig_oracle_decondition_addr(uint32_t w0, uint32_t w1, uint32_t out_size, uint32_t p);

/* Per-format layout derived by the restatement of BrotligDataconditionParams::Initialize
 * (inc/common/BrotligDataConditioner.h:96-183): out[0] block bytes, [1] block pixels, [2] sub-blocks,
 * [3..8] sub-block sizes, [9] colour sub-block count, [10..13] colour sub-blocks, [14] total blocks.
 * Returns 0 when (w0, w1) does not describe a texture of out_size bytes.  Test helper. */
int brotlig_oracle_dc_layout(uint32_t w0, uint32_t w1, uint32_t out_size, uint32_t out[15]);

#ifdef __cplusplus
}
#endif
#endif

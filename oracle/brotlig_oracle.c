/*
 * brotlig_oracle.c -- CPU restatement of BrotliG::DecodeCPU (Brotli-G SDK 1.1).
 *
 * TEST INFRASTRUCTURE ONLY (see brotlig_oracle.h).  PARITY STATUS: parity
 * unpinned -- the reference has no golden vectors and is unbuildable in this
 * image without header stand-ins; see the header of brotlig_oracle.h.
 *
 * The structure deliberately follows the reference's algorithm (flat 2^15
 * decode tables filled per page, page-parallel workers pulling from one atomic
 * counter, full-output zero fill) so that timing it is a fair stand-in for
 * timing the reference's CPU path.  It is a restatement, not a copy: the bit
 * readers are positional, the tables are built by our own code, and all reads
 * are bounds-safe (bytes past the input read as zero, where the reference
 * over-reads by up to 8 bytes: inc/common/BrotligDeswizzler.h:74-81,:149-191).
 */
#include "brotlig_oracle.h"

#include <pthread.h>
#include <stdatomic.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#define NUM_BITSTREAMS 32u          /* inc/common/BrotligConstants.h:80-81 */
#define ICP_ALPHABET 728u           /* 704 + sentinel + 23 insert-only; Constants.h:32-39 */
#define DIST_ALPHABET 544u          /* Constants.h:42 */
#define LIT_ALPHABET 256u
#define TABLE_BITS 15u              /* Constants.h:104-106 */
#define TABLE_SIZE (1u << TABLE_BITS)
#define CL_TABLE_BITS 9u            /* Constants.h:99-101 */
#define CL_TABLE_SIZE (1u << CL_TABLE_BITS)
#define MAX_WORKERS 128             /* Constants.h:90 */
#define MAX_SUB_BLOCKS 6            /* Constants.h:241 */
#define MAX_MIPS 32                 /* Constants.h:153 */

/* google/brotli v1.0.9 c/enc/command.h kInsBase/kInsExtra/kCopyBase/kCopyExtra
 * (RFC 7932 section 5).  The reference consumes them through sBrotligCmdLut
 * (inc/common/BrotligCommandLut.h:41-747) and GetInsertBase/GetInsertExtra
 * (src/decoder/PageDecoder.cpp:311-312). */
static const uint32_t kInsBase[24] = {0, 1, 2, 3, 4, 5, 6, 8, 10, 14, 18, 26, 34, 50, 66, 98,
                                      130, 194, 322, 578, 1090, 2114, 6210, 22594};
static const uint32_t kInsExtra[24] = {0, 0, 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5,
                                       6, 7, 8, 9, 10, 12, 14, 24};
static const uint32_t kCopyBase[24] = {2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 14, 18, 22, 30, 38, 54,
                                       70, 102, 134, 198, 326, 582, 1094, 2118};
static const uint32_t kCopyExtra[24] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4,
                                        5, 5, 6, 7, 8, 9, 10, 24};

/* BrotligUtils.cpp:49-56 -- "Log2Floor" is really the bit width of x. */
static uint32_t bit_width(uint32_t x) { uint32_t r = 0; while (x) { x >>= 1; ++r; } return r; }

void brotlig_oracle_cmd_lut(uint32_t sym, uint32_t* ins_extra, uint32_t* copy_extra,
                            uint32_t* ins_base, uint32_t* copy_base, int* implicit_dist)
{
    /* RFC 7932 section 5 cell layout; reproduces every row of sBrotligCmdLut. */
    static const uint32_t insHi[11] = {0, 0, 0, 0, 1, 1, 0, 2, 1, 2, 2};
    static const uint32_t cpHi[11]  = {0, 1, 0, 1, 0, 1, 2, 0, 2, 1, 2};
    if (sym >= 704) { *ins_extra = *copy_extra = *ins_base = *copy_base = 0; *implicit_dist = 0; return; }
    uint32_t cell = sym >> 6;
    uint32_t ic = insHi[cell] * 8 + ((sym >> 3) & 7), cc = cpHi[cell] * 8 + (sym & 7);
    *ins_extra = kInsExtra[ic]; *ins_base = kInsBase[ic];
    *copy_extra = kCopyExtra[cc]; *copy_base = kCopyBase[cc];
    *implicit_dist = sym < 128;
}

/* ------------------------------------------------------------------------ */
/* Bit readers                                                               */

typedef struct {
    const uint8_t* base;   /* start of the whole input buffer */
    size_t         limit;  /* bytes readable from base; beyond -> zero */
} SafeBuf;

static inline uint64_t load64(const SafeBuf* b, size_t off)
{
    uint64_t v = 0;
    if (off + 8 <= b->limit) { memcpy(&v, b->base + off, 8); return v; }
    if (off < b->limit) memcpy(&v, b->base + off, b->limit - off);
    return v;
}

/* 32 LSB-first sub-bitstreams with a 5-bit wrapping cursor
 * (inc/common/BrotligDeswizzler.h:43-206; m_curindex is a 5-bit field, :205). */
typedef struct {
    SafeBuf  buf;
    uint64_t bitpos[NUM_BITSTREAMS];   /* absolute bit offset from buf.base */
    uint32_t cur;
} Deswizzler;

static inline uint32_t ds_peek(const Deswizzler* d, uint32_t n)      /* n <= 32 */
{
    if (n == 0) return 0;
    uint64_t p = d->bitpos[d->cur];
    uint64_t w = load64(&d->buf, (size_t)(p >> 3)) >> (p & 7);
    return (uint32_t)(w & (n == 32 ? 0xFFFFFFFFu : ((1u << n) - 1u)));
}
static inline void ds_consume(Deswizzler* d, uint32_t n) { d->bitpos[d->cur] += n; }
static inline uint32_t ds_read(Deswizzler* d, uint32_t n) { uint32_t v = ds_peek(d, n); ds_consume(d, n); return v; }
static inline void ds_switch(Deswizzler* d) { d->cur = (d->cur + 1) & (NUM_BITSTREAMS - 1); }
static inline void ds_reset(Deswizzler* d) { d->cur = 0; }

/* 15-bit / 9-bit reversal.  Like the reference (inc/common/BrotligReverseBits.h:35,:106) the
 * hot path goes through a 32768-entry LUT; ours is generated once instead of being a literal. */
static inline uint32_t rev_bits(uint32_t v, uint32_t n)
{
    uint32_t r = 0;
    for (uint32_t i = 0; i < n; ++i) { r = (r << 1) | (v & 1); v >>= 1; }
    return r;
}
typedef struct { uint32_t ins_extra, copy_extra, ins_base, copy_base; } CmdLutRow;
static uint16_t  g_rev15[TABLE_SIZE];
static CmdLutRow g_cmd_lut[705];
static pthread_once_t g_once = PTHREAD_ONCE_INIT;
static void init_luts(void)
{
    for (uint32_t i = 0; i < TABLE_SIZE; ++i) g_rev15[i] = (uint16_t)rev_bits(i, 15);
    for (uint32_t s = 0; s <= 704; ++s) {
        int implicit;
        brotlig_oracle_cmd_lut(s, &g_cmd_lut[s].ins_extra, &g_cmd_lut[s].copy_extra,
                               &g_cmd_lut[s].ins_base, &g_cmd_lut[s].copy_base, &implicit);
    }
}

/* ------------------------------------------------------------------------ */
/* Prefix-code tables (src/decoder/BrotligHuffmanTable.cpp)                  */

/* BrotligHuffmanTable.cpp:44-71: canonical codes, ascending length then
 * ascending symbol, expanded to a flat MSB-first table of 2^maxlen entries. */
static void generate_table(const uint16_t* lens, size_t size, uint16_t* counts, uint32_t numlens,
                           uint16_t* symbols, uint16_t* codelens)
{
    uint16_t next_code[16] = {0};
    counts[0] = 0;
    for (uint32_t i = 1; i < numlens; ++i) next_code[i] = (uint16_t)((next_code[i - 1] + counts[i - 1]) << 1);
    uint32_t maxlen = numlens - 1;
    for (size_t s = 0; s < size; ++s) {
        uint32_t len = lens[s];
        if (!len) continue;
        uint32_t left = maxlen - len;
        uint32_t start = (uint32_t)next_code[len]++ << left, n = 1u << left;
        for (uint32_t k = 0; k < n; ++k) { symbols[start + k] = (uint16_t)s; codelens[start + k] = (uint16_t)len; }
    }
}

/* BrotligHuffmanTable.cpp:26-38 */
static const uint16_t kFixedLens[4][4] = {{1, 1, 0, 0}, {1, 2, 2, 0}, {2, 2, 2, 2}, {1, 2, 3, 3}};
/* BrotligHuffmanTable.cpp:40-42 */
static const uint16_t kCodeLenOrder[18] = {1, 2, 3, 4, 0, 5, 17, 6, 16, 7, 8, 9, 10, 11, 12, 13, 14, 15};

/* BrotligHuffmanTable.cpp:73-205.  Returns 0, or -1 for tree type 3 (the
 * reference throws there, :202-203). */
static int load_table(Deswizzler* d, uint32_t alphabet, uint16_t* symbols, uint16_t* codelens)
{
    uint32_t max_bits = bit_width(alphabet - 1);
    uint32_t ttype = ds_read(d, 2);
    if (ttype == 0) {                                              /* trivial, :84-97 */
        ds_consume(d, 4);
        uint16_t sym = (uint16_t)ds_read(d, max_bits);
        for (uint32_t i = 0; i < TABLE_SIZE; ++i) { symbols[i] = sym; codelens[i] = 0; }
        ds_reset(d);
        return 0;
    }
    if (ttype == 1) {                                              /* simple, :98-120 */
        uint32_t nsym = ds_read(d, 2) + 1;
        uint32_t tree_select = ds_read(d, 1);
        ds_consume(d, 1);
        uint32_t idx = nsym < 4 ? nsym - 2 : (tree_select ? 3 : 2);
        uint32_t at = 0;
        for (uint32_t i = 0; i < nsym; ++i) {
            uint32_t len = kFixedLens[idx][i];
            uint16_t sym = (uint16_t)ds_read(d, max_bits);
            uint32_t n = 1u << (TABLE_BITS - len);
            for (uint32_t k = 0; k < n && at < TABLE_SIZE; ++k, ++at) { symbols[at] = sym; codelens[at] = (uint16_t)len; }
            ds_switch(d);
        }
        ds_reset(d);
        return 0;
    }
    if (ttype == 2) {                                              /* complex, :121-199 */
        uint32_t ncl = ds_read(d, 4) + 4;
        uint16_t cl_lens[32] = {0};
        uint16_t cl_counts[10] = {0};
        for (uint32_t i = 0; i < ncl; ++i) {
            uint32_t len = ds_read(d, 5);
            cl_lens[kCodeLenOrder[i < 18 ? i : 17]] = (uint16_t)len;
            if (len < 10) ++cl_counts[len];
            ds_switch(d);
        }
        static _Thread_local uint16_t cl_syms[CL_TABLE_SIZE], cl_codelens[CL_TABLE_SIZE];
        memset(cl_syms, 0, sizeof cl_syms); memset(cl_codelens, 0, sizeof cl_codelens);
        generate_table(cl_lens, ncl < 18 ? ncl : 18, cl_counts, 10, cl_syms, cl_codelens);
        ds_reset(d);

        uint16_t data[ICP_ALPHABET];
        uint16_t counts[16] = {0};
        uint32_t prev = 8;                                         /* BROTLI_INITIAL_REPEATED_CODE_LENGTH, :149 */
        uint32_t produced = 0;
        while (produced < alphabet) {
            uint32_t code = rev_bits(ds_peek(d, CL_TABLE_BITS), CL_TABLE_BITS);
            ds_consume(d, cl_codelens[code]);
            uint32_t sym = cl_syms[code];
            if (sym == 16) {                                       /* repeat previous literal length, :170-177 */
                uint32_t reps = ds_read(d, 2) + 3;
                for (uint32_t k = 0; k < reps && produced < alphabet; ++k) { data[produced++] = (uint16_t)prev; ++counts[prev]; }
            } else if (sym == 17) {                                /* zero run, :178-185 */
                uint32_t reps = ds_read(d, 3) + 3;
                for (uint32_t k = 0; k < reps && produced < alphabet; ++k) { data[produced++] = 0; ++counts[0]; }
            } else {                                               /* literal length 0..15, :186-192 */
                prev = sym; data[produced++] = (uint16_t)sym; ++counts[sym];
            }
            ds_switch(d);
        }
        generate_table(data, alphabet, counts, 16, symbols, codelens);
        ds_reset(d);
        return 0;
    }
    return -1;
}

/* ------------------------------------------------------------------------ */
/* Pre-conditioning parameters (inc/common/BrotligDataConditioner.h:92-237)  */

typedef struct {
    int      precondition, swizzle, pitch_aligned;
    uint32_t format, num_mips;
    uint32_t block_bytes, block_px, num_sub;
    uint32_t sub_size[MAX_SUB_BLOCKS], sub_off[MAX_SUB_BLOCKS];
    uint32_t color_sub[MAX_SUB_BLOCKS], num_color;
    uint32_t w[MAX_MIPS + 1], h[MAX_MIPS + 1], pitch[MAX_MIPS + 1], nblk[MAX_MIPS + 1];
    uint32_t sub_stream_off[MAX_SUB_BLOCKS + 1];
    uint32_t mip_off_bytes[MAX_MIPS + 1], mip_off_blocks[MAX_MIPS + 1];
    uint32_t total_blocks;
    int      initialized;
} DcParams;

static uint32_t round_up(uint32_t v, uint32_t a) { return (v + a - 1) / a * a; }

static int dc_init(DcParams* p, uint32_t in_size)
{
    static const struct { uint32_t bytes, nsub, sizes[6], ncol, col[4]; } fmt[6] = {
        {1, 1, {1, 0, 0, 0, 0, 0}, 0, {0, 0, 0, 0}},          /* unknown: identity, :176-182 */
        {8, 3, {2, 2, 4, 0, 0, 0}, 2, {0, 1, 0, 0}},          /* BC1 :98-111 */
        {16, 4, {8, 2, 2, 4, 0, 0}, 2, {1, 2, 0, 0}},         /* BC2 :113-127 */
        {16, 6, {1, 1, 6, 2, 2, 4}, 2, {3, 4, 0, 0}},         /* BC3 :129-145 */
        {8, 3, {1, 1, 6, 0, 0, 0}, 2, {0, 1, 0, 0}},          /* BC4 :147-160 */
        {16, 6, {1, 1, 6, 1, 1, 6}, 4, {0, 1, 3, 4}},         /* BC5 :162-181 */
    };
    uint32_t f = p->format <= 5 ? p->format : 0;
    p->block_bytes = fmt[f].bytes; p->block_px = f ? 4 : 1; p->num_sub = fmt[f].nsub;
    for (uint32_t i = 0; i < MAX_SUB_BLOCKS; ++i) p->sub_size[i] = fmt[f].sizes[i];
    p->num_color = fmt[f].ncol;
    for (uint32_t i = 0; i < 4; ++i) p->color_sub[i] = fmt[f].col[i];
    if (p->num_mips == 0) p->num_mips = 1;
    p->total_blocks = p->nblk[0] = p->w[0] * p->h[0];                               /* :193 */
    uint32_t mw = (p->w[0] * p->block_px) / 2, mh = (p->h[0] * p->block_px) / 2;   /* :201 */
    p->mip_off_bytes[0] = 0; p->mip_off_blocks[0] = 0;
    for (uint32_t mip = 1; mip <= p->num_mips; ++mip) {                             /* :202-217 */
        if (mip < p->num_mips) {
            p->w[mip] = (mw + p->block_px - 1) / p->block_px;
            p->h[mip] = (mh + p->block_px - 1) / p->block_px;
            p->nblk[mip] = p->w[mip] * p->h[mip];
            p->pitch[mip] = p->pitch_aligned ? round_up(p->w[mip] * p->block_bytes, 256) : p->w[mip] * p->block_bytes;
            p->total_blocks += p->nblk[mip];
        }
        p->mip_off_bytes[mip] = p->mip_off_bytes[mip - 1] + p->pitch[mip - 1] * p->h[mip - 1];
        p->mip_off_blocks[mip] = p->mip_off_blocks[mip - 1] + p->nblk[mip - 1];
        mw /= 2; mh /= 2;
    }
    if (p->mip_off_bytes[p->num_mips] != in_size) return 0;                         /* :219 */
    p->sub_off[0] = 0; p->sub_stream_off[0] = 0;
    for (uint32_t sub = 1; sub <= p->num_sub; ++sub) {                              /* :221-231 */
        if (sub < p->num_sub) p->sub_off[sub] = p->sub_off[sub - 1] + p->sub_size[sub - 1];
        p->sub_stream_off[sub] = p->sub_stream_off[sub - 1];
        for (uint32_t mip = 0; mip < p->num_mips; ++mip) p->sub_stream_off[sub] += p->nblk[mip] * p->sub_size[sub - 1];
    }
    if (p->sub_stream_off[p->num_sub] != p->total_blocks * p->block_bytes) return 0; /* :233 */
    p->initialized = 1;
    return 1;
}

/* src/decoder/PageDecoder.cpp:406-444 */
static uint32_t decondition_addr(const DcParams* p, uint32_t off, uint32_t sub)
{
    uint32_t adj = off, mip = 0, ss = p->sub_size[sub];
    while (adj >= p->mip_off_blocks[mip + 1] * ss) ++mip;
    adj -= p->mip_off_blocks[mip] * ss;
    uint32_t block = adj / ss, W = p->w[mip], H = p->h[mip];
    uint32_t row = block / W, col = block % W;
    int swz = p->swizzle && W >= 2 && H >= 2;
    uint32_t remW = W % 2, effW = W - remW, effH = H - H % 2;
    if (swz && row < effH && col < effW) {
        uint32_t eff = block - row * remW, gpr = effW / 2;
        uint32_t grp = eff / 4, in = eff % 4;
        row = 2 * (grp / gpr) + in / 2;
        col = 2 * (grp % gpr) + in % 2;
    }
    return p->mip_off_bytes[mip] + row * p->pitch[mip] + col * p->block_bytes + p->sub_off[sub] + adj % ss;
}

static void dc_from_header(DcParams* dc, uint32_t w0, uint32_t w1)
{
    /* inc/DataStream.h:89-98, src/BrotligDecoder.cpp:466-476 */
    memset(dc, 0, sizeof *dc);
    dc->precondition = 1;
    dc->swizzle = w0 & 1;
    dc->pitch_aligned = (w0 >> 1) & 1;
    dc->w[0] = ((w0 >> 2) & 0x7FFF) + 1;
    dc->h[0] = ((w0 >> 17) & 0x7FFF) + 1;
    dc->format = w1 & 0xFF;
    dc->num_mips = ((w1 >> 8) & 0x1F) + 1;
    dc->pitch[0] = ((w1 >> 13) & 0x7FFFF) + 1;
}

uint32_t brotlig_oracle_decondition_addr(uint32_t w0, uint32_t w1, uint32_t out_size, uint32_t p)
{
    DcParams dc; dc_from_header(&dc, w0, w1);
    if (!dc_init(&dc, out_size)) return 0xFFFFFFFFu;
    if (p >= dc.total_blocks * dc.block_bytes) return 0xFFFFFFFFu;
    uint32_t sub = 0;
    while (p >= dc.sub_stream_off[sub + 1]) ++sub;
    return decondition_addr(&dc, p - dc.sub_stream_off[sub], sub);
}

/* Test helper: the per-format layout dc_init derives (inc/common/BrotligDataConditioner.h:96-183), for the
 * KAT against the reference's BROTLIG_BC*_ macros.  out[0] block bytes, [1] block pixels, [2] sub-blocks,
 * [3..8] sub-block sizes, [9] colour sub-block count, [10..13] colour sub-blocks, [14] total blocks.
 * Returns 0 when the header does not describe out_size bytes. */
int brotlig_oracle_dc_layout(uint32_t w0, uint32_t w1, uint32_t out_size, uint32_t out[15])
{
    DcParams dc; dc_from_header(&dc, w0, w1);
    if (!dc_init(&dc, out_size)) return 0;
    out[0] = dc.block_bytes; out[1] = dc.block_px; out[2] = dc.num_sub;
    for (uint32_t i = 0; i < 6; ++i) out[3 + i] = dc.sub_size[i];
    out[9] = dc.num_color;
    for (uint32_t i = 0; i < 4; ++i) out[10 + i] = dc.color_sub[i];
    out[14] = dc.total_blocks;
    return 1;
}

/* ------------------------------------------------------------------------ */
/* Page decoder (src/decoder/PageDecoder.cpp)                                */

typedef struct {
    uint16_t*  symbols[3];     /* ICP, DIST, LIT; PageDecoder.cpp:56-60 (six 64 KiB tables) */
    uint16_t*  codelens[3];
    uint8_t*   lit_queue;      /* page_size + slack, PageDecoder.cpp:164-166 */
    uint8_t*   temp;           /* conditioned-space page for preconditioned streams */
    uint32_t   page_size;
    DcParams   dc;
} PageDecoder;

static int pd_setup(PageDecoder* pd, uint32_t page_size, const DcParams* dc)
{
    memset(pd, 0, sizeof *pd);
    pd->page_size = page_size;
    pd->dc = *dc;
    for (int i = 0; i < 3; ++i) {
        pd->symbols[i] = (uint16_t*)malloc(TABLE_SIZE * sizeof(uint16_t));
        pd->codelens[i] = (uint16_t*)malloc(TABLE_SIZE * sizeof(uint16_t));
        if (!pd->symbols[i] || !pd->codelens[i]) return -1;
    }
    pd->lit_queue = (uint8_t*)malloc((size_t)page_size + 64);
    pd->temp = (uint8_t*)malloc((size_t)page_size + 64);
    return (pd->lit_queue && pd->temp) ? 0 : -1;
}

static void pd_cleanup(PageDecoder* pd)
{
    for (int i = 0; i < 3; ++i) { free(pd->symbols[i]); free(pd->codelens[i]); }
    free(pd->lit_queue); free(pd->temp);
}

/* PageDecoder.cpp:446-471 */
static void delta_decode(const DcParams* dc, size_t page_start, size_t page_end, uint8_t* data)
{
    for (uint32_t i = 0; i < dc->num_color; ++i) {
        uint32_t sub = dc->color_sub[i];
        size_t cs = dc->sub_stream_off[sub], ce = dc->sub_stream_off[sub + 1];
        if (cs < page_end && page_start < ce) {
            size_t s = cs > page_start ? cs - page_start : 0;
            size_t e = ce < page_end ? ce - page_start : page_end - page_start;
            for (size_t el = s + 1; el < e; ++el) data[el] = (uint8_t)(data[el] + data[el - 1]);
        }
    }
}

typedef struct { uint32_t insert_len, copy_len, dist; } Cmd;
#ifdef BROTLIG_ORACLE_TRACE
static void brotlig_oracle_trace_round(const Cmd* q, uint32_t n, uint32_t out_pos, uint32_t litcount, uint32_t rlit);
static void brotlig_oracle_trace_literal(uint32_t j, uint32_t code_len);
static void brotlig_oracle_trace_symbol(int table, uint32_t v15, const uint16_t* codelens);   /* a symbol is about to be decoded from the 15-bit window v15 (MSB first) */    /* literal j of the round (sub-stream j mod 32) and the length of its code */
#endif

/* PageDecoder.cpp:65-268.  `in` is the SafeBuf of the whole compressed input,
 * the page occupies [in_off, in_off + in_size). */
static int pd_run(PageDecoder* pd, const SafeBuf* in, size_t in_size, size_t in_off,
                  uint8_t* output, size_t out_size, size_t out_off)
{
    const DcParams* dc = &pd->dc;
    uint8_t* dst = output + out_off;
    if (dc->precondition) dst = pd->temp;                          /* :72-73, :93-96 */

    if (out_size == in_size) {                                     /* stored page, :70-76 */
        for (size_t i = 0; i < out_size; ++i) dst[i] = in_off + i < in->limit ? in->base[in_off + i] : 0;
    } else {
        /* page header + sub-stream size table, :79-121 (BrotligBitReaderLSB is a
         * plain LSB-first reader: inc/common/BrotligBitReader.h:55-89) */
        Deswizzler ds; memset(&ds, 0, sizeof ds);
        ds.buf = *in;
        uint64_t hp = (uint64_t)in_off * 8;
#define HDR_READ(n) ({ uint32_t _n = (n); uint64_t _w = load64(in, (size_t)(hp >> 3)) >> (hp & 7); hp += _n; \
                       (uint32_t)(_n ? (_w & ((_n == 32) ? 0xFFFFFFFFull : ((1ull << _n) - 1))) : 0); })
        uint32_t npostfix = HDR_READ(2);
        uint32_t ndirect = HDR_READ(4) << npostfix;
        int is_delta = (int)HDR_READ(1) && dc->precondition;
        (void)HDR_READ(1);
        uint32_t avg = (uint32_t)((in_size + NUM_BITSTREAMS - 1) / NUM_BITSTREAMS);
        uint32_t base_bits = bit_width(avg);                       /* Log2FloorNonZero(x)+1, :101 */
        uint32_t log_size = bit_width((uint32_t)(in_size - 1));    /* :104 */
        uint32_t dsize_bits = bit_width(log_size);                 /* :105 */
        uint32_t base_size = HDR_READ(base_bits);
        uint32_t delta_bits = HDR_READ(dsize_bits);
        size_t table_bits = 8 + base_bits + dsize_bits + (size_t)NUM_BITSTREAMS * delta_bits;
        size_t idx = ((table_bits + 31) / 32) * 4;                 /* :110-111 */
        for (uint32_t i = 0; i < NUM_BITSTREAMS; ++i) {            /* :114-120 */
            uint32_t delta = HDR_READ(delta_bits);
            ds.bitpos[i] = (uint64_t)(in_off + idx) * 8;
            idx += base_size + delta;
        }
#undef HDR_READ
        ds_reset(&ds);

        if (load_table(&ds, ICP_ALPHABET, pd->symbols[0], pd->codelens[0])) return -1;   /* :125-131 */
        if (load_table(&ds, DIST_ALPHABET, pd->symbols[1], pd->codelens[1])) return -1;  /* :133-139 */
        if (load_table(&ds, LIT_ALPHABET, pd->symbols[2], pd->codelens[2])) return -1;   /* :141-147 */

        uint32_t ring[4] = {4, 11, 15, 16};                        /* :150-153 */
        memset(dst, 0, out_size);                                  /* :156 */

        Cmd queue[NUM_BITSTREAMS];
        uint8_t* lq_front = pd->lit_queue;
        uint8_t* lq_back = pd->lit_queue;
        uint8_t* lq_end = pd->lit_queue + pd->page_size + 64;
        uint8_t* w = dst;
        uint8_t* w_end = dst + out_size;
        uint32_t prev_tail = 0;
        int sentinel = 0;

        while (!sentinel) {                                        /* :174-236 */
            uint32_t litcount = 0, n = 0;
            while (n != NUM_BITSTREAMS) {                          /* :180-193 */
                Cmd c; c.dist = 0;
                /* DecodeCommand, :290-320 */
                uint32_t bits = g_rev15[ds_peek(&ds, 15)];
#ifdef BROTLIG_ORACLE_TRACE
                brotlig_oracle_trace_symbol(0, bits, pd->codelens[0]);
#endif
                ds_consume(&ds, pd->codelens[0][bits]);
                uint32_t sym = pd->symbols[0][bits];
                if (sym <= 704) {
                    const CmdLutRow* row = &g_cmd_lut[sym];                   /* sBrotligCmdLut[sym], :298 */
                    uint32_t ie = row->ins_extra, ce = row->copy_extra, ib = row->ins_base, cb = row->copy_base;
                    if (ib == 0 && cb == 0) { sentinel = 1; break; }          /* :302-303 */
                    c.insert_len = ib + ds_read(&ds, ie);
                    c.copy_len = cb + ds_read(&ds, ce);
                    uint32_t dcode = 0;
                    if (sym >= 128) {                              /* DecodeDistance, :338-343 */
                        uint32_t db = g_rev15[ds_peek(&ds, 15)];
#ifdef BROTLIG_ORACLE_TRACE
                        brotlig_oracle_trace_symbol(1, db, pd->codelens[1]);
#endif
                        ds_consume(&ds, pd->codelens[1][db]);
                        dcode = pd->symbols[1][db];
                    }
                    /* TranslateDistance, :345-404 */
                    if (dcode < 16) {
                        static const int8_t idx[16] = {0, 1, 2, 3, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1};
                        static const int8_t add[16] = {0, 0, 0, 0, -1, 1, -2, 2, -3, 3, -1, 1, -2, 2, -3, 3};
                        c.dist = ring[idx[dcode]] + (uint32_t)(int32_t)add[dcode];
                    } else if (ndirect > 0 && dcode < 16 + ndirect) {
                        c.dist = dcode - 15;
                    } else {
                        uint32_t x = dcode - ndirect - 16;
                        uint32_t nbits = 1 + (x >> (npostfix + 1));
                        uint32_t extra = ds_read(&ds, nbits > 32 ? 32 : nbits);
                        uint32_t hcode = x >> npostfix, lcode = x & ((1u << npostfix) - 1);
                        uint32_t offset = ((2 + (hcode & 1)) << (nbits & 31)) - 4;
                        c.dist = ((offset + extra) << npostfix) + lcode + ndirect + 1;
                    }
                    if (dcode > 0) { ring[3] = ring[2]; ring[2] = ring[1]; ring[1] = ring[0]; ring[0] = c.dist; }
                } else {                                           /* insert-only, :308-317 */
                    uint32_t ic = sym - 704;
                    if (ic > 23) return -1;
                    c.insert_len = kInsBase[ic] + ds_read(&ds, kInsExtra[ic]);
                    c.copy_len = 0;
                }
                litcount += c.insert_len;
                queue[n++] = c;
                ds_switch(&ds);
            }
            ds_reset(&ds);

            uint32_t ac = litcount > prev_tail ? litcount - prev_tail : 0;          /* :196 */
            uint32_t mult = n ? (ac + n - 1) / n : 0;                               /* :197 */
            uint32_t rlit = n * mult;                                               /* :198 */
            prev_tail = rlit + prev_tail - litcount;                                /* :199 */

            for (uint32_t j = 0; j < rlit; ++j) {                                   /* :202-206 */
                uint32_t bits = g_rev15[ds_peek(&ds, 15)];                          /* DecodeLiteral, :322-327 */
#ifdef BROTLIG_ORACLE_TRACE
                brotlig_oracle_trace_literal(j, pd->codelens[2][bits]);
                brotlig_oracle_trace_symbol(2, bits, pd->codelens[2]);
#endif
                ds_consume(&ds, pd->codelens[2][bits]);
                if (lq_back < lq_end) *lq_back++ = (uint8_t)pd->symbols[2][bits];
                ds_switch(&ds);
            }

#ifdef BROTLIG_ORACLE_TRACE      /* diagnostics builds only (profiles/tools/cmd_stats.c): the round's commands, before they are applied */
            brotlig_oracle_trace_round(queue, n, (uint32_t)(w - dst), litcount, rlit);
#endif
            for (uint32_t k = 0; k < n; ++k) {                                      /* :209-233 */
                Cmd c = queue[k];
                if ((size_t)(lq_back - lq_front) < c.insert_len || (size_t)(w_end - w) < c.insert_len) return -2;
                memcpy(w, lq_front, c.insert_len); w += c.insert_len; lq_front += c.insert_len;
                if (c.copy_len) {
                    if ((size_t)(w_end - w) < c.copy_len || c.dist == 0 || (size_t)(w - dst) < c.dist) return -2;
                    const uint8_t* s = w - c.dist;
                    if (c.dist >= c.copy_len) { memcpy(w, s, c.copy_len); w += c.copy_len; }   /* :222-230 */
                    else for (uint32_t j = 0; j < c.copy_len; ++j) *w++ = *s++;      /* byte-exact LZ77 overlap, :232 */
                }
            }
        }
        if (is_delta) delta_decode(dc, out_off, out_off + out_size, dst);           /* :240 */
    }

    /* de-conditioning scatter, :243-265 */
    if (dc->precondition && out_off < (size_t)dc->total_blocks * dc->block_bytes) {
        uint32_t tex = dc->total_blocks * dc->block_bytes, sub = 0;
        while (out_off >= dc->sub_stream_off[sub + 1]) ++sub;
        size_t index = 0;
        while (index < out_size) {
            size_t off = index + out_off - dc->sub_stream_off[sub];
            uint32_t a = decondition_addr(dc, (uint32_t)off, sub);
            output[a] = dst[index++];
            if (out_off + index >= tex) break;
            if (out_off + index >= dc->sub_stream_off[sub + 1]) sub++;
        }
    }
    return 0;
}

int brotlig_oracle_decode_page(const uint8_t* in, uint32_t in_size, uint8_t* out, uint32_t out_size, uint32_t page_size)
{
    pthread_once(&g_once, init_luts);
    PageDecoder pd; DcParams dc; memset(&dc, 0, sizeof dc);
    if (pd_setup(&pd, page_size, &dc)) { pd_cleanup(&pd); return -1; }
    SafeBuf sb = {in, in_size};
    int rc = pd_run(&pd, &sb, in_size, 0, out, out_size, 0);
    pd_cleanup(&pd);
    return rc;
}

/* ------------------------------------------------------------------------ */
/* Stream driver (src/BrotligDecoder.cpp)                                    */

typedef struct {
    SafeBuf        in;          /* page data (after the page table) */
    const uint8_t* table;       /* page table bytes (little-endian uint32[]) */
    uint8_t*       out;
    uint32_t       num_pages, last_page_size, page_size;
    const DcParams* dc;
    atomic_uint    next;
    atomic_int     error;
} JobCtx;

static uint32_t rd32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }

/* src/BrotligDecoder.cpp:296-329 */
static void* page_job(void* arg)
{
    JobCtx* ctx = (JobCtx*)arg;
    PageDecoder pd;
    if (pd_setup(&pd, ctx->page_size, ctx->dc)) { atomic_store(&ctx->error, 1); pd_cleanup(&pd); return NULL; }
    for (;;) {
        uint32_t i = atomic_fetch_add_explicit(&ctx->next, 1, memory_order_relaxed);     /* :305 */
        if (i >= ctx->num_pages) break;
        uint32_t in_off = i ? rd32(ctx->table + 4 * (size_t)i) : 0;                      /* :310 */
        uint32_t in_size = i < ctx->num_pages - 1 ? rd32(ctx->table + 4 * ((size_t)i + 1)) - in_off
                                                  : rd32(ctx->table);                    /* :311 */
        uint32_t out_off = i * ctx->page_size;                                           /* :313 */
        uint32_t out_size = (i == ctx->num_pages - 1 && ctx->last_page_size) ? ctx->last_page_size : ctx->page_size;
        if (pd_run(&pd, &ctx->in, in_size, in_off, ctx->out, out_size, out_off)) atomic_store(&ctx->error, 1);
    }
    pd_cleanup(&pd);
    return NULL;
}

uint32_t DecompressedSize(uint8_t* src)
{
    /* inc/DataStream.h:60-64 */
    uint32_t num_pages = (uint32_t)src[2] | ((uint32_t)src[3] << 8);
    uint32_t w1 = rd32(src + 4);
    uint32_t page_size = 32768u << (w1 & 3), last = (w1 >> 2) & 0x3FFFF;
    return num_pages * page_size - (last == 0 ? 0 : page_size - last);
}

int brotlig_oracle_decode(uint32_t input_size, const uint8_t* src, uint32_t* output_size,
                          uint8_t* output, int workers, int* workers_used)
{
    pthread_once(&g_once, init_luts);
    if (workers_used) *workers_used = 0;
    if (input_size < 8) return ORC_BROTLIG_ERROR_CORRUPT_STREAM;
    /* src/BrotligDecoder.cpp:437-446 */
    if (src[0] != (uint8_t)(src[1] ^ 0xFF)) return ORC_BROTLIG_ERROR_CORRUPT_STREAM;
    if (src[0] != 5) return ORC_BROTLIG_ERROR_INCORRECT_STREAM_FORMAT;

    memset(output, 0, *output_size);                               /* :448 */

    uint32_t num_pages = (uint32_t)src[2] | ((uint32_t)src[3] << 8);
    uint32_t w1 = rd32(src + 4);
    uint32_t page_size = 32768u << (w1 & 3), last = (w1 >> 2) & 0x3FFFF;
    int precon = (w1 >> 20) & 1;
    uint32_t out_size = num_pages * page_size - (last == 0 ? 0 : page_size - last);

    size_t pos = 8;
    DcParams dc; memset(&dc, 0, sizeof dc);
    if (precon) {                                                  /* :466-481 */
        if (input_size < 16) return ORC_BROTLIG_ERROR_CORRUPT_STREAM;
        dc_from_header(&dc, rd32(src + 8), rd32(src + 12));
        if (!dc_init(&dc, *output_size)) return ORC_BROTLIG_ERROR_GENERIC;  /* reference ignores the failure (UB) */
        pos += 8;
    }
    if (pos + 4 * (size_t)num_pages > input_size) return ORC_BROTLIG_ERROR_CORRUPT_STREAM;

    JobCtx ctx;
    ctx.table = src + pos;
    pos += 4 * (size_t)num_pages;
    ctx.in.base = src + pos; ctx.in.limit = input_size - pos;
    ctx.out = output; ctx.num_pages = num_pages; ctx.last_page_size = last; ctx.page_size = page_size;
    ctx.dc = &dc;
    atomic_init(&ctx.next, 0); atomic_init(&ctx.error, 0);

    /* :404-415 */
    long hw = sysconf(_SC_NPROCESSORS_ONLN); if (hw < 1) hw = 1;
    uint32_t maxw = hw > MAX_WORKERS ? MAX_WORKERS : (uint32_t)hw;
    uint32_t nw = workers > 0 ? (uint32_t)workers : (num_pages > 2 * maxw ? maxw : 1);
    if (nw > MAX_WORKERS) nw = MAX_WORKERS;
    pthread_t th[MAX_WORKERS]; uint32_t started = 0;
    for (uint32_t t = 1; t < nw; ++t) if (pthread_create(&th[started], NULL, page_job, &ctx) == 0) ++started;
    page_job(&ctx);
    for (uint32_t t = 0; t < started; ++t) pthread_join(th[t], NULL);
    if (workers_used) *workers_used = (int)started + 1;

    *output_size = out_size;                                       /* :490 */
    return atomic_load(&ctx.error) ? ORC_BROTLIG_ERROR_GENERIC : ORC_BROTLIG_OK;
}

int DecodeCPU(uint32_t input_size, const uint8_t* src, uint32_t* output_size, uint8_t* output, void* feedbackProc)
{
    (void)feedbackProc;
    return brotlig_oracle_decode(input_size, src, output_size, output, 0, NULL);
}

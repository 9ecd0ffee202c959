// brotlig_kernel_common.h -- kernel ABI (StreamDesc, DcTable, DecodeArgs), phase timers, tunables, window geometry, the LDS record of a page, the per-lane bit reader and the small helpers every stage uses.
// Part of the gfx950 Brotli-G decode kernels; brotlig_kernels.h includes the parts in order and says what the whole replaces.
#pragma once
#include <brotlig_wave_ops.h>

#include "brotlig_format.h"

namespace brotlig {

// ---- kernel ABI -------------------------------------------------------------------------
struct StreamDesc {
    uint64_t in_offset;     // byte offset of the stream (its StreamHeader) in the input buffer
    uint64_t out_offset;    // byte offset of its decompressed bytes in the output buffer
    uint64_t in_size;       // bytes of the stream (0: up to the end of the input buffer)
    uint64_t out_capacity;  // bytes the stream may write at out_offset (0: up to the end of the output buffer)
};
// end of the stream's readable bytes / of its writable region, as offsets into the batch buffers
__device__ __forceinline__ uint64_t stream_in_end(const StreamDesc& d, uint64_t in_bytes)
{
    const uint64_t e = d.in_offset + d.in_size;
    return (d.in_size != 0u && e < in_bytes) ? e : in_bytes;
}
__device__ __forceinline__ uint64_t stream_out_end(const StreamDesc& d, uint64_t out_bytes)
{
    const uint64_t e = d.out_offset + d.out_capacity;
    return (d.out_capacity != 0u && e < out_bytes) ? e : out_bytes;
}

// Per-stream pre-conditioning parameters, derived once per launch by the schedule kernel (its prepare phase) from the
// 8-byte PreconditionHeader (inc/DataStream.h:89-98) the way
// BrotligDataconditionParams::Initialize does (inc/common/BrotligDataConditioner.h:92-237).
struct DcTable {
    uint32_t precon, swizzle, block_bytes, num_sub, num_mips, total_blocks, tex_bytes, color_mask;
    uint32_t sub_size[kMaxSubBlocks], sub_off[kMaxSubBlocks], sub_stream_off[kMaxSubBlocks + 1];
    uint32_t w[kMaxMips], h[kMaxMips], pitch[kMaxMips];
    uint32_t mip_off_bytes[kMaxMips + 1], mip_off_blocks[kMaxMips + 1];
    uint32_t item_prefix[kMaxMips + 1];     // de-conditioning work items before each mip: 64 per tile of 2 rows x 32 row chunks, every tile row
                                            // padded to whole super-tiles of 4 tiles (brotlig_decondition_kernel): a multiple of 256
    uint32_t format;                        // 1..5 = BC1..BC5, 0 = unknown (one byte per block)
    uint32_t pad0;
    // ---- the record's last 128 bytes: the words that cross workgroups INSIDE the schedule kernel (brotlig_schedule.h), written and read there
    // with write-through stores / loads that look beyond the own L2.  A cache line of their own: everything above is written with ordinary
    // stores by one workgroup and read by later kernels only.
    uint32_t status;                        // kStatus* bits of THIS stream (every stream has a record, pre-conditioned or not): which asset of a
                                            // batch was damaged (BrotligDecodeBatchStreamStatus); the batch-wide OR stays in DecodeArgs::status[0]
    uint32_t super_base;                    // de-conditioning super-tiles of all streams before this one (every stream has the word; a stream that is
                                            // not pre-conditioned has none of its own): the batch's super-tiles are one list, cut evenly over the
                                            // wavefronts of brotlig_decondition_kernel
    uint32_t chunk_pages;                   // in the record of every 64th stream: the pages of the 64 streams from it on (schedule kernel, phase
                                            // "prepare": one item per 64 streams; summed by phase "scan")
    uint32_t chunk_supers;                  // like chunk_pages
    uint32_t chunk_precon;                  // like chunk_pages: pre-conditioned streams among the 64
    uint32_t chunk_pages_before, chunk_supers_before;   // in the same records: what lies before the 64 streams (phase "scan" to phase "finalize")
    uint32_t pad[25];
};
static_assert(__builtin_offsetof(DcTable, status) == 896, "the shared words begin the record's last 128-byte line");
static_assert(sizeof(DcTable) == 1024, "DcTable is addressed as 1 KiB records");

// One scheduled page as the page kernels take it: DecodeArgs::jobs[k] answers the k-th request of the page counter.  Written by the schedule
// kernel (brotlig_schedule.h) from the page tables -- stream lookup, table walk (src/BrotligDecoder.cpp:310-314), bounds against the caller's
// buffers, all done once per page THERE -- so that a page start is one 32-byte read instead of a chain of eight dependent loads (round 6;
// rounds 1-5 kept one page INDEX per request and every page start walked from it: schedule word -> log2(streams) probes of the page prefix
// -> descriptor -> stream header -> two page-table words).
struct __attribute__((aligned(16))) JobRecord {
    uint64_t in_off;        // byte offset of the compressed page in DecodeArgs::in
    uint32_t in_size;       // its bytes (== out size: a stored page)
    uint32_t shape;         // bytes it decodes to (bits 0..18) | page size index of its stream << 20 | stream pre-conditioned << 24 | valid << 25
    uint64_t out_off;       // where they go: byte offset in DecodeArgs::out (DecodeArgs::scratch for a pre-conditioned stream)
    uint32_t stream;        // index of its stream in the batch
    uint32_t page;          // index of the page in its stream
};
static_assert(sizeof(JobRecord) == 32, "two 16-byte loads");
enum : uint32_t { kJobPrecon = 1u << 24, kJobValid = 1u << 25 };

// The schedule kernel's own words in the workspace header (DecodeArgs::sync): see brotlig_schedule.h.
enum : uint32_t { kSyncCookie = 0u /* 64 bits: "the words below are clean" */, kSyncTicket = 2u /* tickets taken by the launch in progress */,
                  kSyncDone = 4u /* [kSchedPhases] items finished per phase */, kSyncStatus = 12u /* kStatus* bits found by the schedule kernel */,
                  kSyncWords = 16u };

struct DecodeArgs {
    const uint8_t* in;  uint64_t in_bytes;
    uint8_t* out;       uint64_t out_bytes;
    uint8_t* scratch;   // conditioned-space staging for preconditioned streams (same layout as out)
    const StreamDesc* streams; uint32_t num_streams;
    uint16_t decode_waves;  // the schedule kernel's business (schedule_mode): the wavefronts of brotlig_decode_kernel for this batch (0: unknown),
    uint16_t order_from_k;  // and from how many pages on (in units of 1 024) a batch gets the schedule proper
    uint32_t* page_base;    // [num_streams + 1] exclusive prefix of page counts; [num_streams] = pages of the batch, published LAST by the schedule kernel
    uint32_t* work_counter; // [1] next request of the page kernels
    uint32_t* status;       // [0] OR of kStatus*, [2] number of preconditioned streams, [3] pairing policy, [5] de-conditioning super-tiles of the
                            // batch (the end of the DcTable::super_base prefix), [6] pages of the batch (the schedule kernel's own copy), [8..8+B) pages
                            // per scheduling bucket, [8+B..8+2B) bucket fill cursors (B = kBuckets <= 64; kStatusWords in all)
    uint32_t* sync;         // [kSyncWords] tickets and phase counters of the schedule kernel
    JobRecord* jobs;        // [jobs_cap] the page schedule: one record per request of the page counter (null: no room in the workspace -- the page
    uint32_t  jobs_cap;     // kernels then take pages in stream order and walk the page tables themselves)
    uint32_t  duo_limit;    // batches of up to this many pages belong to brotlig_decode_duo_kernel (two wavefronts per page), larger ones to
                            // brotlig_decode_kernel: the host launches both when it cannot tell (it knows the output size, not the page
                            // count) and the one the batch does not belong to leaves at once.  0: never the former, ~0: always
    DcTable*  dc;           // [num_streams]
    uint16_t* far_syms;     // [workgroups of the decode grid][2][kFarSymStride] per 32-lane half: the ICP and distance symbols
                            // (canonical-code order) that do not fit the LDS arrays -- ranks kIcpSymCap.. and kDistSymCap..
    unsigned long long* prof;   // [kNumPhases] cycle sums, only written by the phase-timer instantiation
    uint64_t launch_tag;    // different for every launch of a process (never 0): marks a workspace that THIS launch is initialising, so that the
                            // schedule kernel needs no memset in front of it (brotlig_schedule.h)
    uint32_t may_pair;      // the batch may hold more pages than the page kernel has wavefronts: the pairing policy matters
};

// Phase timers (diagnostics build of the kernel only).
enum : int { kPhSetup, kPhTables, kPhCommands, kPhRing, kPhPositions, kPhLiterals, kPhCopyFence, kPhCopyLevels,
             kPhDelta, kPhTotal, kPhRounds, kPhLevels, kPhLvShort, kPhLvBytes, kPhLvLong, kPhSlow,
             kPhCmdSym, kPhCmdExtra, kPhSlide, kPhPieces, kPhBitmaps, kPhGroups, kPhLitSteps, kPhLvOverlap, kPhTeamLevels,
             kPhLevelHalves, kPhGroupHalves, kNumPhases };   // *Halves: halves (1 or 2) that had work in an iteration
template <bool kOn> struct PhaseClock;
template <> struct PhaseClock<false> {
    __device__ __forceinline__ void start(unsigned long long*) {}
    __device__ __forceinline__ void lap(int) {}
    __device__ __forceinline__ void count(int, uint32_t) {}
    __device__ __forceinline__ void halves(int, bool) {}
    __device__ __forceinline__ void flush(unsigned long long*, uint32_t) {}
};
// The sums live in LDS (lane 0 adds to them): fifty registers of accumulators would push the kernel's own state
// into scratch memory and time that instead.
template <> struct PhaseClock<true> {
    unsigned long long t0, last;
    unsigned long long* acc;        // [kNumPhases] in LDS
    __device__ __forceinline__ void start(unsigned long long* lds)
    {
        acc = lds;
        if (wave::lane_id() < (uint32_t)kNumPhases) acc[wave::lane_id()] = 0;
        wave::sync();
        t0 = last = wave::clock();
    }
    __device__ __forceinline__ void lap(int ph)
    {
        const unsigned long long t = wave::clock();
        if (wave::lane_id() == 0u) acc[ph] += t - last;
        last = t;
    }
    __device__ __forceinline__ void count(int ph, uint32_t n) { if (wave::lane_id() == 0u) acc[ph] += n; }
    // how many of the two halves take part in an iteration of a loop that runs for both (lock-step cost)
    __device__ __forceinline__ void halves(int ph, bool mine)
    {
        const uint64_t m = wave::ballot64(mine);
        count(ph, ((uint32_t)m != 0u ? 1u : 0u) + ((uint32_t)(m >> 32) != 0u ? 1u : 0u));
    }
    __device__ __forceinline__ void flush(unsigned long long* out, uint32_t lane)
    {
        if (lane == 0u) acc[kPhTotal] = wave::clock() - t0;
        wave::sync();
        if (lane < (uint32_t)kNumPhases && out) atomicAdd(out + lane, acc[lane]);
    }
};

// Ablation switches for profiling builds (profiles/tools/ablate.sh): parts of the LZ77 assembly are skipped --
// the output is wrong, the entropy decode and its control flow are unchanged -- to see what each part costs.
// The product is built without BROTLIG_ABLATE (mask 0: every `if` below folds away).
#ifndef BROTLIG_ABLATE
#define BROTLIG_ABLATE 0
#endif
enum : uint32_t { kAblLevels = 1u, kAblTeams = 2u, kAblOverlap = 4u, kAblFar = 8u, kAblSlide = 16u, kAblDeps = 32u, kAblLitStore = 64u,
                  kAblOwnLane = 128u, kAblRounds = 256u /* page starts only: job fetch, bit readers, the three table builds -- no round at all */,
                  // parts of the table build left out (with kAblRounds: what each costs): the primary LUT, the canonical build (counts, scans,
                  // symbols in code order), the RLE pass over the code lengths, the code-length code
                  kAblTabLut = 512u, kAblTabCanon = 1024u, kAblTabRle = 2048u, kAblTabCl = 4096u };
constexpr uint32_t kAblate = BROTLIG_ABLATE;

// ---- tunables ---------------------------------------------------------------------------
// (overridable for A/B builds of the kernel: profiles/tools/ab_variants.sh)
#ifndef BROTLIG_TUNE_SHORT_COPY
#define BROTLIG_TUNE_SHORT_COPY 32
#define BROTLIG_TUNE_OWN_COPY 128
#define BROTLIG_TUNE_HIST 656
#endif
#ifndef BROTLIG_TUNE_ROUND_MAX
#define BROTLIG_TUNE_ROUND_MAX 640      // round 4: groups of 640 bytes (history 656, window 1344): mixed +1.7 %, records +6.6 %, text -0.6 %, samples16 +0.2 %
#define BROTLIG_TUNE_WIN 1344
#define BROTLIG_TUNE_DIST_LUT_BITS 8
#endif
#ifndef BROTLIG_TUNE_EARLY_NEAR
#define BROTLIG_TUNE_EARLY_NEAR 0   // 1: short near copies whose source is final before the group starts are read ahead, like far ones
                                    // (round 4, measured: 2.7 % fewer instructions and as many more wait cycles -- mixed +-0, samples16 +1..2 %, text -2 %)
#endif
constexpr int kLutBitsIcp = 8;
constexpr int kLutBitsDist = BROTLIG_TUNE_DIST_LUT_BITS;
constexpr int kLutBitsLit = 8;
// Symbols in canonical-code order ("sorted" arrays, read for codes longer than the LUT index): LDS holds the first
// kIcpSymCap / kDistSymCap of them, global memory (DecodeArgs::far_syms) the rest.  Pages of the benchmark's data
// classes use at most 146 ICP symbols (mean 81) and 40 distance symbols (115 under the encoder's distance-parameter
// search), so the overflow is for odd pages only (tests/cases.py: many_command_shapes, many_distances).
#ifndef BROTLIG_ICP_SYM_CAP
#define BROTLIG_ICP_SYM_CAP 255
#define BROTLIG_DIST_SYM_CAP 96
#endif
constexpr uint32_t kIcpSymCap = BROTLIG_ICP_SYM_CAP;      // multiples of 3 fill whole words (three 10-bit fields each)
constexpr uint32_t kDistSymCap = BROTLIG_DIST_SYM_CAP;
constexpr uint32_t kFarIcp = kIcpAlphabet - kIcpSymCap, kFarDist = kDistAlphabet - kDistSymCap;
constexpr uint32_t kFarSymStride = (kFarIcp + kFarDist + 63u) & ~63u;   // uint16 per half: ICP overflow, then distance overflow
constexpr uint32_t kLongCode = 0xFFFFu;     // LUT marker: code longer than the LUT index, lengths differ under the prefix
constexpr uint32_t kLutSubtree = 0x8000u;   // LUT flag: longer code, one length under the prefix: {index in code order, length}
constexpr uint32_t kShortCopy = BROTLIG_TUNE_SHORT_COPY;         // far pieces up to this length are fetched by their own lane (four 8-byte loads)
// Pieces that are not simple (they overlap themselves with a distance below 32, or their pattern straddles the window boundary) run in their
// own lane up to kShortCopy bytes, a longer one makes its level a team level -- except the pieces with a period of 1, 2 or 4 bytes (one word,
// stored by its lane: no read back): those stay with their lane up to kOverlapOwn bytes.
#ifndef BROTLIG_TUNE_OVERLAP_OWN
#define BROTLIG_TUNE_OVERLAP_OWN 48
#endif
constexpr uint32_t kOverlapOwn = BROTLIG_TUNE_OVERLAP_OWN;
static_assert(kOverlapOwn >= 32u && kOverlapOwn <= 64u && kOverlapOwn % 8u == 0u, "the period-1/2/4 path stores up to kOverlapOwn / 8 words");
constexpr uint32_t kOwnCopy = BROTLIG_TUNE_OWN_COPY;             // simple copies up to this length run one-lane-per-command (batches of four 8-byte chunks)
// Output window: the last kWin bytes of the page under construction live in LDS.  A round whose
// output fits in kWin - kHist bytes is assembled there (literals, copies, the dependency levels
// between copies); bytes older than the window are read back from global memory.  The window is
// flushed to global memory in aligned 16-byte stores when it slides.
constexpr uint32_t kWin = BROTLIG_TUNE_WIN;
constexpr uint32_t kHist = BROTLIG_TUNE_HIST;             // history kept across a slide (>= kRoundMax + 16: see the slide below)
constexpr uint32_t kRoundMax = BROTLIG_TUNE_ROUND_MAX;     // bytes assembled per group (a multiple of 32; the flush and the slide move up to 1024 bytes)
static_assert(kHist >= kRoundMax + 16u && kWin >= kHist + 16u + kRoundMax, "window: history + one group");
constexpr uint32_t kStageBytes = kRoundMax + 8 * 32;    // far-copy staging: every copy rounded up to 8 bytes
static_assert(kStageBytes >= kRoundMax + 64u, "the staging area also holds a group's literals, with slack for 8-byte reads");
// The same four numbers as a type: the page loop, the LDS record and the stages that depend on them are templates over it.
// GeoPair (the constants above) is the layout of a wavefront that decodes two pages, one record per half.  GeoSolo (round 4) is
// the layout of a wavefront that decodes ONE page at a time (small batches: no more pages than wavefronts): it has the LDS of
// both halves for one record, so its groups are 1 024 bytes -- the per-group work (flush, slide, piece classification, bitmaps,
// level bookkeeping) is 30 % of a run-length page's time at 640 -- and its window keeps 4 KiB of history on chip.
template <uint32_t kRM, uint32_t kH, uint32_t kW> struct Geometry {
    static constexpr uint32_t kRoundMax = kRM, kHist = kH, kWin = kW, kStageBytes = kRM + 8 * 32;
    static constexpr uint32_t kFlushPieces = (kRM + 16u + 511u) / 512u;     // 16-byte pieces per lane that a group's flush can need
    static constexpr uint32_t kSlidePieces = (kH + 16u + 511u) / 512u;      // ... and the slide of the history
    static_assert(kH >= kRM + 16u && kW >= kH + 16u + kRM && kRM % 32u == 0u && kRM <= 1024u, "window: history + one group; 32 bitmap words at most");
    static_assert(kFlushPieces <= 3u && kSlidePieces <= 3u, "flush_and_slide moves up to three pieces per lane");
};
typedef Geometry<kRoundMax, kHist, kWin> GeoPair;
#ifndef BROTLIG_TUNE_SOLO_ROUND_MAX
#define BROTLIG_TUNE_SOLO_ROUND_MAX 1024
#define BROTLIG_TUNE_SOLO_HIST 1040
#define BROTLIG_TUNE_SOLO_WIN 5120
#endif
typedef Geometry<BROTLIG_TUNE_SOLO_ROUND_MAX, BROTLIG_TUNE_SOLO_HIST, BROTLIG_TUNE_SOLO_WIN> GeoSolo;

// insert / copy length codes: base | extra_bits << 16   (RFC 7932 section 5; the reference carries
// them as sBrotligCmdLut, inc/common/BrotligCommandLut.h:41-747, and the shader regenerates them
// by prefix sums, BrotliGCompute.hlsl:1061-1075)
__device__ static const uint32_t kLenCodeTab[48] = {
    // insert
    0u | 0u << 16, 1u | 0u << 16, 2u | 0u << 16, 3u | 0u << 16, 4u | 0u << 16, 5u | 0u << 16,
    6u | 1u << 16, 8u | 1u << 16, 10u | 2u << 16, 14u | 2u << 16, 18u | 3u << 16, 26u | 3u << 16,
    34u | 4u << 16, 50u | 4u << 16, 66u | 5u << 16, 98u | 5u << 16, 130u | 6u << 16, 194u | 7u << 16,
    322u | 8u << 16, 578u | 9u << 16, 1090u | 10u << 16, 2114u | 12u << 16, 6210u | 14u << 16, 22594u | 24u << 16,
    // copy
    2u | 0u << 16, 3u | 0u << 16, 4u | 0u << 16, 5u | 0u << 16, 6u | 0u << 16, 7u | 0u << 16,
    8u | 0u << 16, 9u | 0u << 16, 10u | 1u << 16, 12u | 1u << 16, 14u | 2u << 16, 18u | 2u << 16,
    22u | 3u << 16, 30u | 3u << 16, 38u | 4u << 16, 54u | 4u << 16, 70u | 5u << 16, 102u | 5u << 16,
    134u | 6u << 16, 198u | 7u << 16, 326u | 8u << 16, 582u | 9u << 16, 1094u | 10u << 16, 2118u | 24u << 16};

// order in which the code-length-code lengths are stored (BrotligHuffmanTable.cpp:40-42)
__device__ static const uint8_t kCodeLenOrder[18] = {1, 2, 3, 4, 0, 5, 17, 6, 16, 7, 8, 9, 10, 11, 12, 13, 14, 15};

// ---- LDS layout: one of these per 32-lane half ---------------------------------------------
template <class G>
struct __attribute__((aligned(16))) PageLdsT {
    // decode LUTs, then the staging area: while a table is being built its LUT and the 1 KiB behind it
    // serve as scratch (code-length LUT, counting-sort counters), so the order of these four matters --
    // ICP borrows the distance LUT, distance borrows the literal LUT, literal borrows the staging area,
    // each of which is still (or again) free at that point.
    uint16_t lut_icp[1 << kLutBitsIcp];
    uint16_t lut_dist[1 << kLutBitsDist];
    uint16_t lut_lit[1 << kLutBitsLit];
    uint64_t stage[G::kStageBytes / 8];     // per group: first the group's literals in consumption order (they move to
                                            // the window before the far sources arrive), then the source bytes of far
                                            // copies (older than the window)
    uint32_t sorted_icp[(kIcpSymCap + 2) / 3];         // symbols in canonical-code order, three 10-bit fields per word
    uint32_t sorted_dist[(kDistSymCap + 2) / 3];
    uint32_t sorted_lit[kLitAlphabet / 4];             // literals fit a byte each: plain byte array
    uint16_t limit[3][16] __attribute__((aligned(16)));     // per code length: exclusive upper bound, left-justified to 15 bits
    uint32_t first_offs[3][16]; // per code length: first code (left-justified) | index of its first symbol in sorted_* << 16
    uint32_t start_bits[G::kRoundMax / 32]; // per group: bit p set <=> a command's piece starts at group byte p
    uint8_t  start_cum[G::kRoundMax / 32];  // per group: piece starts in earlier words of start_bits
    uint8_t  carry[64];             // ring of literals decoded ahead of their command (< 32 live)
    uint32_t page_params;           // NPOSTFIX | (NDIRECT << NPOSTFIX) << 8 | delta-coded flag << 16 of the page being decoded
    uint32_t page_stream;           // index of its stream in the batch (read only when the page turns out damaged; lives in what was padding)
    uint32_t ring[8] __attribute__((aligned(16)));  // the distance ring, circular: the t-th distance pushed in the page lives in word t & 7 (DistanceRing)
    uint8_t  win[G::kWin + 16] __attribute__((aligned(16)));    // output window; doubles as the code-length
                                                                 // scratch (728 B) while tables are built
};
typedef PageLdsT<GeoPair> PageLds;
typedef PageLdsT<GeoSolo> PageLdsSolo;
constexpr uint32_t kTableScratchBytes = 1024;   // 512-entry code-length LUT, or 16 x 32 counters, as uint16
static_assert(sizeof(uint16_t) * ((1 << kLutBitsIcp) + (1 << kLutBitsDist) + (1 << kLutBitsLit)) >= kTableScratchBytes, "ICP build scratch");
static_assert(sizeof(uint16_t) * ((1 << kLutBitsDist) + (1 << kLutBitsLit)) + kStageBytes >= kTableScratchBytes, "distance build scratch");
static_assert(sizeof(uint16_t) * (1 << kLutBitsLit) + kStageBytes >= kTableScratchBytes, "literal build scratch");
static_assert(__builtin_offsetof(PageLds, lut_dist) == sizeof(uint16_t) * (1 << kLutBitsIcp), "LUTs must be contiguous");
static_assert(__builtin_offsetof(PageLds, stage) == sizeof(uint16_t) * ((1 << kLutBitsIcp) + (1 << kLutBitsDist) + (1 << kLutBitsLit)), "staging area must follow the LUTs");
static_assert(kWin + 16 >= kIcpAlphabet, "the window holds the code lengths during the table build");

struct __attribute__((aligned(16))) WaveLds {
    PageLds  page[2];
    uint32_t len_code_tab[48];
#if BROTLIG_EXP_GLDS
    uint8_t  glds[64 * 16] __attribute__((aligned(16)));   // (experiment builds only: a 16-byte slot per lane for the bit readers)
#endif
};
// the one-page layout lives in the same storage (decode_kernel_body); len_code_tab stays where it is
static_assert(sizeof(PageLdsSolo) <= 2 * sizeof(PageLds), "the one-page record must fit the LDS of the two halves");
static_assert(__builtin_offsetof(PageLdsSolo, stage) == __builtin_offsetof(PageLds, stage), "same table-build scratch order in both layouts");

__device__ __forceinline__ uint64_t load_u64u_g(const uint8_t* p) { uint64_t v; __builtin_memcpy(&v, p, 8); return v; }
// sixteen bytes at a 16-byte aligned address, kept in registers (one b128 access)
typedef uint32_t Bytes16 __attribute__((vector_size(16)));
__device__ __forceinline__ Bytes16 load16(const uint8_t* p) { return *reinterpret_cast<const Bytes16*>(__builtin_assume_aligned(p, 16)); }
__device__ __forceinline__ void store16(uint8_t* p, Bytes16 v) { *reinterpret_cast<Bytes16*>(__builtin_assume_aligned(p, 16)) = v; }

// ---- per-lane bit reader over one sub-bitstream ---------------------------------------------
// LSB-first.  `buf` holds `avail` valid bits.  Behind it sit 64 queued bits (`queue`, `queued` of them
// still unread) and 64 bits in flight from global memory (`flight`): a refill takes 32 queued bits,
// and only every second refill touches the in-flight pair -- loaded at least two refills earlier --
// and issues the next 8-byte load.  Loads are 8 bytes at 4-byte aligned offsets.
// Bounds: the input allocation extends 16 bytes past in_bytes (include/brotlig_amd.h), so no load a valid
// stream needs is ever cut short; a reader that has run away on a corrupt stream is held at the last 8
// readable bytes (`limit8`) and decodes whatever is there (the reference over-reads unchecked,
// inc/common/BrotligDeswizzler.h:74-81).
// (BROTLIG_EXP_GLDS, an experiment of round 6 -- profiles/r06_traffic.md -- never the product: 1 = the reader's 8 bytes "in flight" in two
// registers become 16 bytes in an LDS slot of the lane's own, filled by CDNA4's direct global -> LDS load (global_load_lds_dwordx4: no
// register, and half as many trips to a 128-byte line of the input); 2 = the control: the register reader with the same 1 KiB more LDS per
// wavefront, i.e. the same 14 workgroups per compute unit.)
#ifndef BROTLIG_EXP_GLDS
#define BROTLIG_EXP_GLDS 0
#endif
template <bool kGlds>
struct BitReaderT {
    const uint8_t* base;    // page start in the input buffer
    uint32_t limit8;        // last byte offset from base at which 8 bytes may be loaded
    uint64_t buf;
    uint32_t avail;
    uint32_t next;          // byte offset of the next 8-byte load, dword aligned relative to base
    uint64_t queue;
    uint32_t queued;        // 0, 32 or 64
    uint64_t flight;        // the 8 bytes loaded last, not waited for until they are needed (kGlds: bit 0 = which half of the slot is next)
    uint32_t zero;          // wave::opaque_zero()
    uint8_t* slot;          // kGlds: this lane's 16 bytes of LDS; the wavefront's 64 slots lie back to back from slot0
    uint32_t slot0;         // kGlds: LDS byte address of lane 0's slot (wave-uniform: the direct load's M0)

    // Issues the 8-byte load for byte offset `rel` without touching its result.  Branch-free on purpose: a
    // conditional load would reach `flight` through a register copy, and the copy would wait for the load
    // just issued.
    __device__ __forceinline__ uint64_t load8(uint32_t rel) const { return load_u64u_g(base + min_rel(rel)); }
    __device__ __forceinline__ uint32_t min_rel(uint32_t rel) const { return rel < limit8 ? rel : limit8; }
    // kGlds: 16 bytes from byte offset `rel` (held at the last 16 readable bytes) straight into this lane's slot; not waited for
    __device__ __forceinline__ void dma16(uint32_t rel) const
    {
#if defined(__HIP_DEVICE_COMPILE__) && BROTLIG_EXP_GLDS == 1
        const uint32_t lim16 = limit8 >= 8u ? limit8 - 8u : 0u;
        const uint8_t* src = base + (rel < lim16 ? rel : lim16);
        uint32_t keep;
        // (every LDS read of the slot has returned before the load that overwrites it is issued: lgkmcnt(0))
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(slot0) : "memory");
#else
        (void)rel;
#endif
    }
    __device__ __forceinline__ void init(const uint8_t* b, uint32_t lim, uint32_t start)
    {
        base = b; limit8 = lim >= 8u ? lim - 8u : 0u; zero = wave::opaque_zero();
        const uint32_t a = start & ~3u, skip = (start & 3u) * 8u;
        const uint64_t first = load8(a);
        next = a + 8u;
        if constexpr (kGlds) { dma16(next); next += 16u; flight = 0; }
        else { flight = load8(next); next += 8u; }
        buf = (uint64_t)((uint32_t)first >> skip);
        avail = 32u - skip;
        queue = first >> 32; queued = 32u;
        if (avail < 32u) refill();
    }
    __device__ __forceinline__ void refill()
    {
        // `flight >> zero` rather than a copy: with a plain copy the compiler keeps the old pair where it is, loads
        // the new one into a scratch pair and copies it over -- and that copy waits for the load just issued
        if (queued == 0u) {
            if constexpr (kGlds) {
#if defined(__HIP_DEVICE_COMPILE__) && BROTLIG_EXP_GLDS == 1
                if ((uint32_t)flight == 0u) {           // the slot's first half: the direct load has to have landed
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    queue = *reinterpret_cast<const volatile uint64_t*>(slot);
                    flight = 1;
                } else {                                // its second half, then the next 16 bytes on their way
                    queue = *reinterpret_cast<const volatile uint64_t*>(slot + 8);
                    flight = 0;
                    dma16(next); next += 16u;
                }
#endif
                queued = 64u;
            } else { queue = flight >> zero; queued = 64u; flight = load8(next); next += 8u; }
        }
        buf |= (uint64_t)(uint32_t)queue << avail;
        queue >>= 32; queued -= 32u;
        avail += 32u;
    }
    __device__ __forceinline__ void ensure(uint32_t n) { if (avail < n) refill(); }          // n <= 32
    __device__ __forceinline__ uint32_t peek(uint32_t n) const                               // n <= 32
    {
        return n >= 32u ? (uint32_t)buf : ((uint32_t)buf & ((1u << n) - 1u));
    }
    __device__ __forceinline__ void consume(uint32_t n) { buf >>= n; avail -= n; }
    __device__ __forceinline__ uint32_t read(uint32_t n)
    {
        if (n == 0u) return 0u;
        ensure(n);
        const uint32_t v = peek(n);
        consume(n);
        return v;
    }
};
typedef BitReaderT<false> BitReader;

__device__ __forceinline__ uint32_t min_u32(uint32_t a, uint32_t b) { return a < b ? a : b; }
__device__ __forceinline__ uint32_t bit_width_u32(uint32_t x) { return x ? 32u - (uint32_t)__clz((int)x) : 0u; }
__device__ __forceinline__ uint32_t ctz_u32(uint32_t x) { return (uint32_t)__ffs((int)x) - 1u; }      // x != 0
__device__ __forceinline__ uint32_t msb_u32(uint32_t x) { return 31u - (uint32_t)__clz((int)x); }     // x != 0
__device__ __forceinline__ uint32_t load_u32(const uint8_t* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }     // (any byte address: a damaged page table can place a page anywhere)

// Unaligned 8-byte access (gfx950 global memory takes any byte address; hipcc emits dwordx2).
__device__ __forceinline__ uint64_t load_u64u(const uint8_t* p) { uint64_t v; __builtin_memcpy(&v, p, 8); return v; }
// Store the low n (1..8) bytes of v at p: at most two stores, the second overlapping the first.
__device__ __forceinline__ void store_bytes(uint8_t* p, uint64_t v, uint32_t n)
{
    if (n >= 8u) { __builtin_memcpy(p, &v, 8); return; }
    if (n >= 4u) {
        const uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> (8u * (n - 4u)));
        __builtin_memcpy(p, &lo, 4);
        __builtin_memcpy(p + (n - 4u), &hi, 4);
    } else if (n >= 2u) {
        const uint16_t lo = (uint16_t)v, hi = (uint16_t)(v >> (8u * (n - 2u)));
        __builtin_memcpy(p, &lo, 2);
        __builtin_memcpy(p + (n - 2u), &hi, 2);
    } else {
        *p = (uint8_t)v;
    }
}
// Where a page's bytes are while it is being decoded: positions >= win_base are in the LDS window
// (win[pos - win_base]); everything below `flushed` (tracked by the caller) is in global memory.
struct OutView {
    uint8_t* win;           // the LDS window
    uint32_t win_base;      // page position of win[0]
};

// Eight bytes of an LZ77 copy's source pattern (which lies entirely in LDS at `s`), starting at offset
// r (< d) of its period: byte k is s[(r + k) mod d].  For d >= copy length this is a plain read; for
// overlapping copies it replays the first d bytes, so no byte written by the copy itself is ever read
// back (out[t + j] = out[t - d + (j mod d)], PageDecoder.cpp:219-232 / BrotliGCompute.hlsl:1414-1418).
__device__ __forceinline__ uint64_t pattern_source8(const uint8_t* s, uint32_t d, uint32_t r)
{
    if (r + 8u <= d) return load_u64u(s + r);
    if (d >= 8u) {
        const uint32_t n = d - r;
        const uint64_t lo = load_u64u(s + r), hi = load_u64u(s);
        return (lo & ((1ull << (8u * n)) - 1ull)) | (hi << (8u * n));
    }
    // d < 8: rotate the d-byte period so that it starts at offset r, then double it up to 8 bytes
    const uint32_t db = 8u * d;
    const uint64_t p = load_u64u(s) & ((1ull << db) - 1ull);
    uint64_t q = r ? ((p >> (8u * r)) | (p << (8u * (d - r)))) & ((1ull << db) - 1ull) : p;
    q |= q << db;                                                   // 2 periods
    if (2u * db < 64u) q |= q << (2u * db);                          // 4 periods
    if (4u * db < 64u) q |= q << (4u * db);                          // 8 periods
    return q;
}
// Position of the q-th (0-based) set bit of m; q < popcount(m).
__device__ __forceinline__ uint32_t select_bit(uint32_t m, uint32_t q)
{
    uint32_t pos = 0, c;
    c = (uint32_t)__popc(m & 0xFFFFu); if (q >= c) { q -= c; pos += 16u; m >>= 16; }
    c = (uint32_t)__popc(m & 0xFFu);   if (q >= c) { q -= c; pos += 8u;  m >>= 8; }
    c = (uint32_t)__popc(m & 0xFu);    if (q >= c) { q -= c; pos += 4u;  m >>= 4; }
    c = (uint32_t)__popc(m & 0x3u);    if (q >= c) { q -= c; pos += 2u;  m >>= 2; }
    c = m & 1u;                        if (q >= c) { pos += 1u; }
    return pos;
}
// j mod d for j < 2^16, d >= 1: reciprocal estimate plus one correction either way.
__device__ __forceinline__ uint32_t mod_u16(uint32_t j, uint32_t d)
{
    const uint32_t q = (uint32_t)((float)j * __builtin_amdgcn_rcpf((float)d));
    int32_t rem = (int32_t)(j - q * d);
    if (rem < 0) rem += (int32_t)d;
    if ((uint32_t)rem >= d) rem -= (int32_t)d;
    return (uint32_t)rem;
}
// Up to 32 bytes of a piece as 8-byte chunks at offsets 0, 8, 16, 24 clipped to len - 8 (len >= 8): the last chunk ends at the
// piece's end and overlaps its predecessor, so there are no tail cases -- and a chunk beyond the piece's length, clipped onto
// the last one, is harmless (same bytes to the same place).  BROTLIG_TUNE_CHUNKS says how many of the four are issued without
// asking whether the piece is that long: each question is an exec-mask branch, each unconditional chunk an LDS access.
#ifndef BROTLIG_TUNE_CHUNKS
#define BROTLIG_TUNE_CHUNKS 1     // measured (round 3, 4 GiB): 0 / 1 / 2 -> mixed 434 / 442 / 444, text 439 / 451 / 449, records 449 / 468 / 473 GB/s; round 4, once the
                                  // questions had moved to the scalar unit, 1 against 2: mixed +0.9 %, text +1.0 %, records -0.3 %, samples16 +0.3 %
#endif
struct Chunks32 { uint64_t v0, v1, v2, v3; };
#ifndef BROTLIG_TUNE_LIT_CHUNKS
#define BROTLIG_TUNE_LIT_CHUNKS BROTLIG_TUNE_CHUNKS
#endif
template <int kUncond = BROTLIG_TUNE_CHUNKS>
__device__ __forceinline__ Chunks32 load_chunks32(const uint8_t* sp, uint32_t len, uint32_t clip8)
{
    Chunks32 c{0, 0, 0, 0};
    const uint32_t c1 = min_u32(8u, clip8), c2 = min_u32(16u, clip8), c3 = min_u32(24u, clip8);
    c.v0 = load_u64u(sp);
    if (kUncond >= 1 || len > 8u) c.v1 = load_u64u(sp + c1);
    if (kUncond == 1) { if (len > 16u) { c.v2 = load_u64u(sp + c2); c.v3 = load_u64u(sp + c3); } }
    else {
        if (kUncond >= 2 || len > 16u) c.v2 = load_u64u(sp + c2);
        if (kUncond >= 2 || len > 24u) c.v3 = load_u64u(sp + c3);
    }
    return c;
}
template <int kUncond = BROTLIG_TUNE_CHUNKS>
__device__ __forceinline__ void store_chunks32(uint8_t* dp, const Chunks32& c, uint32_t len, uint32_t clip8)
{
    const uint32_t c1 = min_u32(8u, clip8), c2 = min_u32(16u, clip8), c3 = min_u32(24u, clip8);
    __builtin_memcpy(dp, &c.v0, 8);
    if (kUncond >= 1 || len > 8u) __builtin_memcpy(dp + c1, &c.v1, 8);
    if (kUncond == 1) { if (len > 16u) { __builtin_memcpy(dp + c2, &c.v2, 8); __builtin_memcpy(dp + c3, &c.v3, 8); } }
    else {
        if (kUncond >= 2 || len > 16u) __builtin_memcpy(dp + c2, &c.v2, 8);
        if (kUncond >= 2 || len > 24u) __builtin_memcpy(dp + c3, &c.v3, 8);
    }
}
// Copy of `len` bytes by the lane itself when no chunk of a 32-byte batch reads what an earlier chunk of the batch
// wrote (no overlap, or distance >= 32): 8-byte chunks at offsets clipped to len - 8 (the last chunk ends at the
// piece's end and overlaps its predecessor), the loads of a batch before its stores.
__device__ __forceinline__ void own_copy_simple(const uint8_t* sp, uint8_t* dp, uint32_t len, uint64_t on_w)
{
    const uint32_t clip8 = len >= 8u ? len - 8u : 0u;
    const uint64_t ge8_w = wave::ballot_gt_k<7u>(len);
    if (wave::from_mask(on_w & ge8_w)) {
        const Chunks32 c = load_chunks32<BROTLIG_TUNE_LIT_CHUNKS>(sp, len, clip8);
        store_chunks32<BROTLIG_TUNE_LIT_CHUNKS>(dp, c, len, clip8);
    }
    if (wave::from_mask(on_w & ~ge8_w)) store_bytes(dp, load_u64u(sp), len);
    uint64_t more_w = on_w & wave::ballot_gt_k<32u>(len);
    for (uint32_t o = 32u; more_w != 0ull; o += 32u, more_w &= wave::ballot_gt(len, o)) {
        if (wave::from_mask(more_w)) {
            const uint32_t c0 = min_u32(o, clip8), c1 = min_u32(o + 8u, clip8), c2 = min_u32(o + 16u, clip8), c3 = min_u32(o + 24u, clip8);
            uint64_t v0, v1 = 0, v2 = 0, v3 = 0;
            v0 = load_u64u(sp + c0);
            if (len > o + 8u) v1 = load_u64u(sp + c1);
            if (len > o + 16u) v2 = load_u64u(sp + c2);
            if (len > o + 24u) v3 = load_u64u(sp + c3);
            __builtin_memcpy(dp + c0, &v0, 8);
            if (len > o + 8u) __builtin_memcpy(dp + c1, &v1, 8);
            if (len > o + 16u) __builtin_memcpy(dp + c2, &v2, 8);
            if (len > o + 24u) __builtin_memcpy(dp + c3, &v3, 8);
        }
    }
}
// j / d for j < 2^22, 1 <= d <= 64: reciprocal estimate plus one correction either way.
__device__ __forceinline__ uint32_t div_small(uint32_t j, uint32_t d)
{
    uint32_t q = (uint32_t)((float)j * __builtin_amdgcn_rcpf((float)d));
    const int32_t rem = (int32_t)(j - q * d);
    if (rem < 0) --q;
    if (rem >= (int32_t)d) ++q;
    return q;
}
// Teams: `count` jobs share the 32 lanes of a half; each job gets 32 >> ceil_log2(count) lanes.
struct Team { uint32_t log2_size; uint32_t job; uint32_t member; bool serves; };
__device__ __forceinline__ Team make_team(uint32_t job_mask, uint32_t sl)
{
    const uint32_t count = (uint32_t)__popc(job_mask);
    const uint32_t need = count <= 1u ? 0u : 32u - (uint32_t)__clz((int)(count - 1u));     // ceil_log2(count)
    Team t;
    t.log2_size = 5u - need;
    const uint32_t q = sl >> t.log2_size;
    t.member = sl & ((1u << t.log2_size) - 1u);
    t.serves = q < count;
    t.job = select_bit(job_mask, t.serves ? q : 0u);               // lane (0..31) of the piece this team serves
    return t;
}

// The same over all 64 lanes of a wavefront that decodes ONE page (small batches: the upper half has no page of its own, see
// decode_pages): `job_mask` = the ready pieces of the lower half, `lane` = 0..63; each job gets 64 >> ceil_log2(count) lanes.
__device__ __forceinline__ Team make_team64(uint32_t job_mask, uint32_t lane)
{
    const uint32_t count = (uint32_t)__popc(job_mask);
    const uint32_t need = count <= 1u ? 0u : 32u - (uint32_t)__clz((int)(count - 1u));
    Team t;
    t.log2_size = 6u - need;
    const uint32_t q = lane >> t.log2_size;
    t.member = lane & ((1u << t.log2_size) - 1u);
    t.serves = q < count;
    t.job = select_bit(job_mask, t.serves ? q : 0u);
    return t;
}

// Store window bytes [from, to) of the page to global memory: up to 15 head bytes, then aligned
// 16-byte pieces (one per lane per step), then -- only when `exact` -- the tail bytes.  Without
// `exact` the range is cut at the last 16-byte boundary.  Returns the new flushed position.
__device__ __forceinline__ uint32_t flush_window(uint8_t* out, const OutView& o, uint32_t from, uint32_t to, bool exact, uint32_t sl)
{
    const uint32_t end = exact ? to : (to & ~15u);
    if (end <= from) return from;
    const uint32_t a = min_u32(end, (from + 15u) & ~15u);
    for (uint32_t p = from + sl; p < a; p += 32u) out[p] = o.win[p - o.win_base];
    const uint32_t e16 = end & ~15u;
    for (uint32_t p = a + 16u * sl; p < e16; p += 512u) {
        uint64_t v[2];
        __builtin_memcpy(v, o.win + (p - o.win_base), 16);
        __builtin_memcpy(out + p, v, 16);
    }
    for (uint32_t p = (e16 > a ? e16 : a) + sl; p < end; p += 32u) out[p] = o.win[p - o.win_base];
    return end;
}
// r <- (r + step) mod d, for r < d
__device__ __forceinline__ uint32_t advance_mod(uint32_t r, uint32_t step, uint32_t d)
{
    r += step;
    if (r >= d) r = d >= step ? r - d : r % d;
    return r;
}

}  // namespace brotlig

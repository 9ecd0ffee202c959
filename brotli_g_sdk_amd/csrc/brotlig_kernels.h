// brotlig_kernels.h -- Brotli-G page decode for gfx950 (CDNA4), written from the format
//  rather than from the reference shader.
//
// What it replaces: the reference's single D3D12 compute kernel CSMain
// (src/decoder/BrotliGCompute.hlsl:1753-1882) and its CPU twin PageDecoder::Run
// (src/decoder/PageDecoder.cpp:65-268).  Differences in design:
//   * wave64 hosts TWO pages at a time, one per 32-lane half (the format fixes 32 sub-streams per
//     page), and each half takes its next page on its own (decode_pages);
//   * symbols are decoded through LSB-first primary LUTs in LDS (one ds_read per symbol) with a
//     canonical-code fallback for long codes, instead of the shader's <=16-step length search
//     (BrotliGCompute.hlsl:500-526) or the CPU's three 64 KiB tables (BrotligHuffmanTable.cpp:44-71);
//   * each lane streams its own sub-bitstream from global memory through a 64-bit window, 64 queued
//     bits and 64 bits in flight;
//   * the round's literals are decoded into a small queue in LDS (consumption order, one assembly group at a
//     time) and every command moves its own literal run to the window like a short copy;
//   * the page is assembled in an LDS window (1.5 KiB, flushed group by group in 16-byte stores); a copy
//     whose source has left the window is fetched from global memory by its own lane while the literals are
//     being decoded and goes from registers straight to its place; the remaining LZ77 copies of a round run
//     in dependency levels computed exactly (which earlier pieces own bytes of my source range), one lane
//     per command in 8-byte chunks, or in teams of lanes for long copies (the shader walks the 32 commands
//     serially, BrotliGCompute.hlsl:1401-1419);
//   * the distance ring travels from round to round through LDS (the last four pushes of a round, written by
//     their lanes) instead of being rebuilt from lane broadcasts;
//   * work is pulled from one device-side counter by persistent waves, through a page schedule
//     that puts similar pages side by side (order kernels below).
//
// Style rule: every wave::* call sits in wave-uniform control flow.  Per-lane loops and
// branches contain only memory and ALU work.  (tests/sim runs this same source on the CPU with
// one fiber per lane and checks the rule.)
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

// Per-stream pre-conditioning parameters, derived once per launch by the prepare kernel from the
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
    uint32_t status;                        // kStatus* bits of THIS stream (every stream has a record, pre-conditioned or not): which asset of a
                                            // batch was damaged (BrotligDecodeBatchStreamStatus); the batch-wide OR stays in DecodeArgs::status[0]
    uint32_t chunk_pages;                   // in the record of every 64th stream: the pages of the 64 streams from it on (brotlig_prepare_kernel,
                                            // one workgroup per 64 streams, to brotlig_prepare_finish_kernel)
    uint32_t super_base;                    // de-conditioning super-tiles of all streams before this one (every stream has the word; a stream that is
                                            // not pre-conditioned has none of its own): the batch's super-tiles are one list, cut evenly over the
                                            // wavefronts of brotlig_decondition_kernel
    uint32_t chunk_supers;                  // like chunk_pages
    uint32_t pad[29];
};
static_assert(sizeof(DcTable) == 1024, "DcTable is addressed as 1 KiB records");

struct DecodeArgs {
    const uint8_t* in;  uint64_t in_bytes;
    uint8_t* out;       uint64_t out_bytes;
    uint8_t* scratch;   // conditioned-space staging for preconditioned streams (same layout as out)
    const StreamDesc* streams; uint32_t num_streams;
    uint16_t decode_waves;  // the order kernels' business (schedule_mode below; in what was padding: the page kernels' code does not move): the wavefronts
    uint16_t order_from_k;  // of brotlig_decode_kernel for this batch (0: unknown), and from how many pages on (in units of 1 024) a batch gets the schedule proper
    uint32_t* page_base;    // [num_streams + 1] exclusive prefix of page counts
    uint32_t* work_counter; // [1] next global page index
    uint32_t* status;       // [0] OR of kStatus*, [2] number of preconditioned streams, [3] pairing policy, [5] de-conditioning super-tiles of the
                            // batch (the end of the DcTable::super_base prefix), [8..8+B) pages per scheduling bucket, [8+B..8+2B) bucket fill
                            // cursors (B = kBuckets <= 64; kStatusWords in all)
    uint32_t* order;        // [order_cap] page schedule: global page indices grouped by bucket (null: page order)
    uint32_t  order_cap;
    uint32_t  duo_limit;    // batches of up to this many pages belong to brotlig_decode_duo_kernel (two wavefronts per page), larger ones to
                            // brotlig_decode_kernel: the host launches both when it cannot tell (it knows the output size, not the page
                            // count) and the one the batch does not belong to leaves at once.  0: never the former, ~0: always
    DcTable*  dc;           // [num_streams]
    uint16_t* far_syms;     // [workgroups of the decode grid][2][kFarSymStride] per 32-lane half: the ICP and distance symbols
                            // (canonical-code order) that do not fit the LDS arrays -- ranks kIcpSymCap.. and kDistSymCap..
    unsigned long long* prof;   // [kNumPhases] cycle sums, only written by the phase-timer instantiation
#ifdef BROTLIG_WITH_SPLIT
    // split path (profiles/experiments/split_path/brotlig_split_kernels.h): the entropy kernel leaves every compressed page as a command array and a
    // literal array in global memory, the assembly kernel builds the page from them.  Slots are indexed by global page index.
    uint64_t* cmds;             // [pages][cmd_cap + 1] packed commands, then one terminal entry
    uint8_t*  lits;             // [pages][lit_stride] literals in consumption order
    uint32_t* slot_hdr;         // [pages][2]: number of commands, flags (kSlot*)
    uint32_t  cmd_cap, lit_stride;
    uint32_t* work_counter2;    // [1] page counter of the assembly kernel
#endif
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
struct BitReader {
    const uint8_t* base;    // page start in the input buffer
    uint32_t limit8;        // last byte offset from base at which 8 bytes may be loaded
    uint64_t buf;
    uint32_t avail;
    uint32_t next;          // byte offset of the next 8-byte load, dword aligned relative to base
    uint64_t queue;
    uint32_t queued;        // 0, 32 or 64
    uint64_t flight;        // the 8 bytes loaded last, not waited for until they are needed
    uint32_t zero;          // wave::opaque_zero()

    // Issues the 8-byte load for byte offset `rel` without touching its result.  Branch-free on purpose: a
    // conditional load would reach `flight` through a register copy, and the copy would wait for the load
    // just issued.
    __device__ __forceinline__ uint64_t load8(uint32_t rel) const { return load_u64u_g(base + min_rel(rel)); }
    __device__ __forceinline__ uint32_t min_rel(uint32_t rel) const { return rel < limit8 ? rel : limit8; }
    __device__ __forceinline__ void init(const uint8_t* b, uint32_t lim, uint32_t start)
    {
        base = b; limit8 = lim >= 8u ? lim - 8u : 0u; zero = wave::opaque_zero();
        const uint32_t a = start & ~3u, skip = (start & 3u) * 8u;
        const uint64_t first = load8(a);
        next = a + 8u;
        flight = load8(next); next += 8u;
        buf = (uint64_t)((uint32_t)first >> skip);
        avail = 32u - skip;
        queue = first >> 32; queued = 32u;
        if (avail < 32u) refill();
    }
    __device__ __forceinline__ void refill()
    {
        // `flight >> zero` rather than a copy: with a plain copy the compiler keeps the old pair where it is, loads
        // the new one into a scratch pair and copies it over -- and that copy waits for the load just issued
        if (queued == 0u) { queue = flight >> zero; queued = 64u; flight = load8(next); next += 8u; }
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

// Lanes (0..17) of a half whose code-length symbol (kCodeLenOrder[lane]) is smaller than lane sl's: the ties of the canonical order.
// kClSmaller[k] = sum over j of (kCodeLenOrder[j] < kCodeLenOrder[k]) << j; checked against the order at compile time below.  (Written out
// and selected by a chain of compares: an indexed read would be a global load per lane, in front of the table build's first LDS access.)
constexpr uint32_t kClSmaller[18] = {0x00010u, 0x00011u, 0x00013u, 0x00017u, 0x00000u, 0x0001Fu, 0x3FFBFu, 0x0003Fu, 0x3FEBFu, 0x000BFu, 0x002BFu, 0x006BFu, 0x00EBFu, 0x01EBFu, 0x03EBFu, 0x07EBFu, 0x0FEBFu, 0x1FEBFu};
constexpr bool cl_smaller_matches_the_order()
{
    constexpr uint8_t order[18] = {1, 2, 3, 4, 0, 5, 17, 6, 16, 7, 8, 9, 10, 11, 12, 13, 14, 15};      // = kCodeLenOrder
    for (int k = 0; k < 18; ++k) {
        uint32_t mk = 0;
        for (int j = 0; j < 18; ++j) mk |= (order[j] < order[k] ? 1u : 0u) << j;
        if (mk != kClSmaller[k]) return false;
    }
    return true;
}
static_assert(cl_smaller_matches_the_order(), "kClSmaller must follow kCodeLenOrder");
template <uint32_t K> __device__ __forceinline__ uint32_t cl_smaller_select(uint32_t sl, uint32_t m)
{
    if constexpr (K < 18u) return cl_smaller_select<K + 1u>(sl, sl == K ? kClSmaller[K] : m);
    else return m;
}
__device__ __forceinline__ uint32_t cl_smaller_lanes(uint32_t sl) { return cl_smaller_select<0u>(sl, 0u); }

// One prefix-code table: which LDS arrays it lives in.
struct TableRef {
    uint16_t* lut; uint32_t* sorted; uint16_t* limit; uint32_t* first_offs;
    uint32_t alphabet; int lut_bits;
    uint16_t* far_syms;         // global overflow of `sorted` (ICP and distance tables): slot of the workgroup's first half; the
                                // second half's follows (kFarSymStride)
};
// symbols of canonical rank >= cap live in global memory, at far_syms[far_slot(alphabet) + rank - cap]
__device__ __forceinline__ uint32_t sym_cap(uint32_t alphabet)
{
    return alphabet == kIcpAlphabet ? kIcpSymCap : alphabet == kDistAlphabet ? kDistSymCap : kLitAlphabet;
}
// Offset of this half's slot behind TableRef::far_syms.  Formed where it is used (a few instructions) rather than
// carried in a register through the whole kernel: the reads are rare.
__device__ __forceinline__ uint32_t far_slot(uint32_t alphabet)
{
    return (wave::lane_id_fresh() >> 5) * kFarSymStride + (alphabet == kDistAlphabet ? kFarIcp : 0u);
}

// sorted-symbol arrays: element i lives in bits [10 * (i % 3), +10) of word i / 3
__device__ __forceinline__ uint32_t sorted_get(const uint32_t* words, uint32_t i)
{
    const uint32_t w = (i * 43691u) >> 17;                      // i / 3 for i < 98304
    return (words[w] >> (10u * (i - 3u * w))) & 0x3FFu;
}
__device__ __forceinline__ void sorted_put(uint32_t* words, uint32_t i, uint32_t sym)    // words pre-zeroed
{
    const uint32_t w = (i * 43691u) >> 17;
    atomicOr(&words[w], sym << (10u * (i - 3u * w)));
}
// the literal table (256 symbols) keeps its symbols as bytes instead
__device__ __forceinline__ uint32_t table_sym(const TableRef& t, uint32_t i)
{
    if (t.alphabet == kLitAlphabet) return (uint32_t)reinterpret_cast<const uint8_t*>(t.sorted)[i];
    const uint32_t cap = sym_cap(t.alphabet);
    if (i < cap) return sorted_get(t.sorted, i);
    // a rank the page's code never assigned (incomplete or damaged code) reads whatever an earlier page left in the slot:
    // held inside the alphabet, so that what a damaged page decodes to does not depend on the workspace's history
    return min_u32((uint32_t)t.far_syms[far_slot(t.alphabet) + (i - cap)] & 0x3FFu, t.alphabet - 1u);
}
__device__ __forceinline__ void table_set_sym(const TableRef& t, uint32_t i, uint32_t sym)   // packed words pre-zeroed
{
    if (t.alphabet == kLitAlphabet) { reinterpret_cast<uint8_t*>(t.sorted)[i] = (uint8_t)sym; return; }
    const uint32_t cap = sym_cap(t.alphabet);
    if (i < cap) sorted_put(t.sorted, i, sym);
    else t.far_syms[far_slot(t.alphabet) + (i - cap)] = (uint16_t)sym;
}

// Decode one symbol from `br` (needs avail >= 15 on entry).  Returns symbol, sets len.
// kBits = index width of the table's primary LUT.  Codes longer than that take the canonical route:
// the limits of lengths 8..15 arrive in one aligned 16-byte LDS read (same address for the whole
// half), the length is a count of compares, then one read for {first code, offset} and one for the
// symbol -- two dependent reads instead of a search loop.
template <int kBits, class Reader>
__device__ __forceinline__ uint32_t decode_symbol(const TableRef& t, const Reader& br, uint32_t& len)
{
    static_assert(kBits >= 7 && kBits <= 14, "limit words 8..15 must cover every long length");
    const uint32_t bits = (uint32_t)br.buf;
    const uint32_t e = t.lut[bits & ((1u << kBits) - 1u)];
    if (e < kLutSubtree) { len = e & 15u; return e >> 4; }
    const uint32_t rb = __brev(bits);                           // stream bits, first bit on top
    if (e != kLongCode) {                                       // all codes under this prefix share one length
        const uint32_t l = e & 15u;
        const uint32_t idx = ((e >> 4) & 0x3FFu) + ((rb >> (32u - l)) & ((1u << (l - (uint32_t)kBits)) - 1u));
        len = l;
        return table_sym(t, idx);
    }
    const uint32_t v = rb >> 17;                                // next 15 bits, MSB-first
    uint32_t lim[4];
    __builtin_memcpy(lim, t.limit + 8, 16);                     // limits of lengths 8..15, two per word
    uint32_t l = (uint32_t)kBits + 1u;
#pragma unroll
    for (int k = kBits + 1; k < 15; ++k) {
        const uint32_t w = lim[(k - 8) >> 1];
        const uint32_t lk = (k & 1) ? (w >> 16) : (w & 0xFFFFu);
        l += v >= lk ? 1u : 0u;
    }
    const uint32_t fo = t.first_offs[l];
    uint32_t idx = (fo >> 16) + ((v - (fo & 0xFFFFu)) >> (15u - l));
    idx = min_u32(idx, t.alphabet - 1u);
    len = l;
    return table_sym(t, idx);
}

// -------------------------------------------------------------------------------------------
// Prefix-code description -> decode tables (format: SURVEY.md A.5; reference reader:
// src/decoder/BrotligHuffmanTable.cpp:73-205).  Runs for both halves at once; `live` says
// whether this half has a compressed page; `codelens` = alphabet bytes of LDS for the code lengths.  Returns false for a description the format does not define (the page
// is then rejected): a `simple` code of one symbol, for which the reference indexes FixedCodelengths[-1]
// (BrotligHuffmanTable.cpp:103); DecodeCPU (csrc/brotlig_cpu.cpp) rejects the same.
template <class Reader>
__device__ inline bool build_table(const TableRef& t, uint8_t* codelens, Reader& br, bool live, uint32_t sl)
{
    const uint32_t A = t.alphabet;
    const uint32_t maxbits = bit_width_u32(A - 1u);
    const uint32_t lut_size = 1u << t.lut_bits;
    uint16_t* scratch16 = t.lut;            // LUT area doubles as scratch until the LUT itself is written

    // -- header: lane 0 of the half reads 6 bits from sub-stream 0
    uint32_t hdr = 0;
    if (live && sl == 0u) hdr = br.read(6);
    hdr = wave::half_bcast(hdr, 0);
    const uint32_t type = hdr & 3u;
    const bool is_trivial = live && type == 0u;
    const bool is_simple = live && type == 1u;
    const bool is_complex = live && type >= 2u;    // type 3 is invalid; treated as complex, fails bounds later

    // -- trivial / simple: up to 4 symbols, symbol k from sub-stream k
    const uint32_t nsym = is_trivial ? 1u : ((hdr >> 2) & 3u) + 1u;
    const bool defined = !(is_simple && nsym < 2u);
    const uint32_t tree_select = (hdr >> 4) & 1u;
    uint32_t mysym = 0;
    if ((is_trivial || is_simple) && sl < nsym) mysym = br.read(maxbits);
    const uint32_t s0 = wave::half_bcast(mysym, 0), s1 = wave::half_bcast(mysym, 1);
    const uint32_t s2 = wave::half_bcast(mysym, 2), s3 = wave::half_bcast(mysym, 3);
    // -- complex: code-length code, then RLE-coded code lengths
    if (wave::any(is_complex)) {
        // 18 code-length-code lengths, the k-th from sub-stream k, for symbols in a fixed order.  Fewer than 18
        // (header field < 14) is undefined in the reference: it builds the code-length table over the first ncl
        // symbol INDICES of an array whose other entries were never written (uninitialised stack,
        // BrotligHuffmanTable.cpp:125,:141), and its encoder always writes 18 (src/encoder/BrotligHuffman.cpp:358).
        // Here every length that was read gets its code.
        const uint32_t ncl = min_u32(((hdr >> 2) & 15u) + 4u, 18u);
        uint32_t cl_len = 0;
        const uint32_t cl_sym = sl < 18u ? kCodeLenOrder[sl] : 31u;
        if (is_complex && sl < ncl) cl_len = br.read(5);
        if (cl_len > 9u) cl_len = 0u;                              // > 9 is invalid (2^9 table in the reference)
        // canonical code of my code-length symbol: the symbols that precede it in (length, symbol) order each take 2^(my length - theirs) of
        // its code space.  Round 5: counted from nine ballots (which lanes hold a code of length l?) -- the shorter ones by popcount, the
        // ones of my own length by popcount under "lanes whose symbol is smaller than mine" (the symbol order is fixed: kCodeLenOrder) --
        // instead of eighteen broadcasts of every lane's (length, symbol) to every lane.
        uint32_t cl_code = 0;
        if (!(kAblate & kAblTabCl)) {
            const uint32_t smaller = cl_smaller_lanes(sl);
#pragma unroll
            for (uint32_t l = 1; l <= 9u; ++l) {
                const uint32_t m = wave::half_of(wave::ballot_eq(cl_len, l));
                if (l < cl_len) cl_code += (uint32_t)__popc(m) << (cl_len - l);
                else if (l == cl_len) cl_code += (uint32_t)__popc(m & smaller);
            }
        }
        // LUT over the next `tb` stream bits (LSB-first), tb = the longest code-length code of the page (<= 9; typically 4 .. 6): entry =
        // sym << 4 | len.  (Rounds 1-4 always built the reference's 2^9 table: a lane with a 1- or 2-bit code wrote 256 or 128 entries.)
        const uint32_t cl_longest = wave::half_max(cl_len);
        const uint32_t tb = cl_longest > 0u ? cl_longest : 1u;
        if (is_complex) for (uint32_t e = sl; e < (1u << tb); e += 32u) scratch16[e] = 0;
        wave::sync();
        {
            const uint32_t reps = (is_complex && sl < 18u && cl_len) ? (1u << (tb - cl_len)) : 0u;
            const uint32_t rcode = cl_len ? (__brev(cl_code) >> (32u - cl_len)) : 0u;
            for (uint32_t m = 0; m < reps; ++m) scratch16[rcode + (m << cl_len)] = (uint16_t)((cl_sym << 4) | cl_len);
        }
        // the code lengths start out as zeros: the RLE pass below only stores the non-zero ones (most of an alphabet is unused)
        if (is_complex) for (uint32_t o = 16u * sl; o < ((A + 15u) & ~15u); o += 512u) store16(codelens + o, Bytes16{0u, 0u, 0u, 0u});
        wave::sync();

        // RLE symbols: one (plus its extra bits) per sub-stream, round-robin, until A lengths exist
        uint32_t produced = (is_complex && !(kAblate & kAblTabRle)) ? 0u : A;
        uint32_t prev_len = 8;                                     // BROTLI_INITIAL_REPEATED_CODE_LENGTH
        while (wave::any(produced < A)) {
            const bool act = produced < A;
            uint32_t sym = 0, clen = 0, run = 0, extra = 0, nextra = 0;
            if (act) {
                br.ensure(16);                                     // <= 9-bit code + up to 3 extra bits
                const uint32_t e = scratch16[br.peek(tb)];
                sym = e >> 4; clen = e & 15u;
                nextra = sym == 16u ? 2u : (sym == 17u ? 3u : 0u);
                extra = ((uint32_t)(br.buf >> clen)) & ((1u << nextra) - 1u);
                run = sym >= 16u ? 3u + extra : 1u;
            }
            const uint32_t incl = wave::half_scan_incl(run);
            const uint32_t start = produced + incl - run;
            const bool valid = act && start < A;
            if (valid) br.consume(clen + nextra);
            const uint32_t lit_mask = wave::half_ballot(valid && sym < 16u);
            const uint32_t before = lit_mask & ((1u << sl) - 1u);
            const uint32_t from_lane = wave::half_shfl(sym, before ? msb_u32(before) : 0u);
            const uint32_t last_lit = wave::half_bcast(sym, lit_mask ? msb_u32(lit_mask) : 0u);
            uint32_t value = sym;                                  // literal length
            if (sym == 17u) value = 0u;
            else if (sym == 16u) value = before ? from_lane : prev_len;    // repeat previous *literal* length
            if (valid && value != 0u) {                             // (zeros are there already)
                const uint32_t end = min_u32(start + run, A);
                for (uint32_t s = start; s < end; ++s) codelens[s] = (uint8_t)value;
            }
            // (the valid lanes are a prefix of the half, and when a lane is not valid the lengths are complete: the sum over the valid lanes
            // and the sum over all lanes give the same `produced` after the clamp -- one broadcast instead of a second scan)
            produced = min_u32(A, produced + wave::half_bcast(incl, 31u));
            if (lit_mask) prev_len = last_lit;
        }
        wave::sync();

        // canonical build.  Each lane owns a contiguous block of symbols; per-(length, lane)
        // counters give every symbol its rank without atomics.
        if (!(kAblate & kAblTabCanon)) {
        uint16_t* cnt = scratch16;                                 // [16][32]
        const uint32_t blk = (A + 31u) / 32u;
        const uint32_t b0 = sl * blk, b1 = min_u32(A, b0 + blk);
        if (is_complex) for (uint32_t l = 0; l < 16u; ++l) cnt[l * 32u + sl] = 0;
        if (is_complex && A != kLitAlphabet) for (uint32_t w = sl; w < (sym_cap(A) + 2u) / 3u; w += 32u) t.sorted[w] = 0u;
        wave::sync();
        if (is_complex)
            for (uint32_t s = b0; s < b1; ++s) { const uint32_t l = codelens[s] & 15u; if (l) cnt[l * 32u + sl]++; }
        wave::sync();
        uint32_t code = 0, off = 0, prev_count = 0;
        for (uint32_t l = 1; l < 16u; ++l) {
            const uint32_t c = is_complex ? cnt[l * 32u + sl] : 0u;
            const uint32_t incl = wave::half_scan_incl(c);
            const uint32_t total = wave::half_bcast(incl, 31);
            if (is_complex) cnt[l * 32u + sl] = (uint16_t)(off + incl - c);
            code = (code + prev_count) << 1;
            if (is_complex && sl == 0u) {
                t.limit[l] = (uint16_t)min_u32((code + total) << (15u - l), 32768u);
                t.first_offs[l] = min_u32(code << (15u - l), 32768u) | (off << 16);
            }
            off += total; prev_count = total;
        }
        wave::sync();
        if (is_complex)
            for (uint32_t s = b0; s < b1; ++s) {
                const uint32_t l = codelens[s] & 15u;
                if (l) { const uint32_t p = cnt[l * 32u + sl]++; table_set_sym(t, min_u32(p, A - 1u), s); }
            }
        // symbols beyond the LDS arrays went to global memory: stores first, then the reads below and in the rounds
        // (same CU, same L1: workgroup scope is enough)
        }
        if (A != kLitAlphabet) wave::global_fence(); else wave::sync();
        // primary LUT.  Round 5: filled in CODE order -- lane sl owns the lut_size / 32 consecutive code prefixes from (lut_size / 32) * sl on,
        // entry index = the prefix bit-reversed.  The length of a code is monotone in its left-justified value (the limits are: each is the
        // previous one plus the codes of its length, clamped), so only a lane's first prefix takes the search over all fifteen limits; from one
        // prefix to the next the length is walked up against limit[l].  (Rounds 1-4: index order, two fifteen-compare searches per entry.)
        if (is_complex && !(kAblate & kAblTabLut)) {
            // length of the code whose left-justified 15-bit value range contains v (16: none): the limits are monotone, so a binary search
            // over limit[1..15] (four reads) finds the first one above v
            auto length_of = [&t](uint32_t v) {
                uint32_t l = 0;                                     // invariant: limit[l] <= v (limit[0] taken as 0), answer in (l, l + span]
                l += v >= (uint32_t)t.limit[l + 8u] ? 8u : 0u;
                l += v >= (uint32_t)t.limit[l + 4u] ? 4u : 0u;
                l += v >= (uint32_t)t.limit[l + 2u] ? 2u : 0u;
                l += v >= (uint32_t)t.limit[l + 1u] ? 1u : 0u;
                return l + 1u;
            };
            const uint32_t lut_bits = (uint32_t)t.lut_bits, per = lut_size >> 5, step = 1u << (15u - lut_bits);
            uint32_t v = (per * sl) << (15u - lut_bits);           // left-justified 15-bit value of my first prefix
            uint32_t l = length_of(v);
            uint32_t lim_l = l <= 15u ? (uint32_t)t.limit[l] : 0xFFFFFFFFu;
            uint32_t fo = l <= 15u ? t.first_offs[l] : 0u;
            for (uint32_t i = 0; i < per; ++i, v += step) {
                if (v >= lim_l) {
                    do { ++l; lim_l = l <= 15u ? (uint32_t)t.limit[l] : 0xFFFFFFFFu; } while (v >= lim_l);
                    fo = l <= 15u ? t.first_offs[l] : 0u;
                }
                uint32_t entry = kLongCode;
                if (l <= 15u) {
                    const uint32_t idx = (fo >> 16) + ((v - (fo & 0xFFFFu)) >> (15u - l));
                    if (l <= lut_bits) {
                        entry = (table_sym(t, min_u32(idx, A - 1u)) << 4) | l;
                    } else if (v + step - 1u < lim_l && idx + (1u << (l - lut_bits)) <= A) {
                        // every code under this prefix has length l: they are consecutive in code order, so the
                        // symbol is sorted[idx + the next l - lut_bits code bits] -- no length search at decode time
                        entry = kLutSubtree | (idx << 4) | l;
                    }
                }
                t.lut[__brev(per * sl + i) >> (32u - lut_bits)] = (uint16_t)entry;
            }
        }
    }
    // -- trivial / simple LUTs (written last: the complex path uses LUT areas as scratch)
    if (is_trivial) {
        for (uint32_t e = sl; e < lut_size; e += 32u) t.lut[e] = (uint16_t)(s0 << 4);
    } else if (is_simple) {
        const uint32_t shape = nsym < 4u ? nsym - 2u : (tree_select ? 3u : 2u);   // BrotligHuffmanTable.cpp:26-38
        for (uint32_t e = sl; e < lut_size; e += 32u) {
            const uint32_t b0 = e & 1u, b1 = (e >> 1) & 1u, b2 = (e >> 2) & 1u;
            uint32_t k, len;
            if (shape == 0u) { k = b0; len = 1u; }
            else if (shape == 1u) { k = b0 ? 1u + b1 : 0u; len = b0 ? 2u : 1u; }
            else if (shape == 2u) { k = b0 * 2u + b1; len = 2u; }
            else { k = !b0 ? 0u : (!b1 ? 1u : 2u + b2); len = !b0 ? 1u : (!b1 ? 2u : 3u); }
            const uint32_t sym = k == 0u ? s0 : k == 1u ? s1 : k == 2u ? s2 : s3;
            t.lut[e] = (uint16_t)((sym << 4) | len);
        }
    }
    wave::sync();
    return defined;
}

// -------------------------------------------------------------------------------------------
// Decode the pages `page_a` (lanes 0-31) and `page_b` (lanes 32-63); either may be absent.
struct PageJob {
    const uint8_t* in;      // compressed page
    uint32_t in_size;       // bytes
    uint32_t in_limit;      // bytes readable from `in` without leaving the input buffer
    uint8_t* out;           // where the page's bytes go (final output, or conditioned-space scratch)
    uint32_t out_size;
    uint32_t page_size;
    uint32_t page_off;      // offset of the page in its stream's (conditioned) byte space
    const DcTable* dc;      // non-null for preconditioned streams
    uint32_t index;         // global page index (position in stream order, before the schedule)
    uint32_t stream;        // index of the page's stream in the batch
    bool     valid;
};

// dword of the compressed page at byte offset `rel`, zero beyond the readable input
__device__ __forceinline__ uint32_t br_load(const PageJob& job, uint32_t rel)
{
    return rel + 4u <= job.in_limit ? load_u32(job.in + rel) : 0u;
}

// byte k of the result = b0 + ... + bk (mod 256) of the dword's bytes
__device__ __forceinline__ uint32_t byte_prefix(uint32_t x)
{
    const uint32_t lo = (x & 0x00FF00FFu) * 0x00010001u;              // 16-bit fields (b0, b0+b2)
    const uint32_t hi = ((x >> 8) & 0x00FF00FFu) * 0x00010001u;       //               (b1, b1+b3)
    const uint32_t even = lo + (hi << 16), odd = lo + hi;             // (b0, b0+b1+b2), (b0+b1, b0+..+b3)
    return (even & 0x00FF00FFu) | ((odd & 0x00FF00FFu) << 8);
}
// adds the byte `c` to each byte of `x` (mod 256, no carries between bytes)
__device__ __forceinline__ uint32_t byte_add(uint32_t x, uint32_t c)
{
    const uint32_t cc = (c & 0xFFu) * 0x01010101u;
    return ((x & 0x7F7F7F7Fu) + (cc & 0x7F7F7F7Fu)) ^ ((x ^ cc) & 0x80808080u);
}

// A page of stream `s` failed: the batch-wide status word (the shader's meta[0], BrotliGCompute.hlsl:1757-1881) and the stream's own
// (round 5: a batch of up to 4 096 assets names the damaged ones).  Rare path, one lane.
__device__ __forceinline__ void flag_bad_page(const DecodeArgs& a, uint32_t s)
{
    atomicOr(a.status, kStatusBadPage);
    atomicOr(&a.dc[s].status, kStatusBadPage);
}

// What the order kernels write into the page schedule (DecodeArgs::order, there whenever the workspace has room for it) for a batch of
// `total` pages -- the page kernel takes order[k] for its k-th request whatever it holds:
//   0  page order (order[k] = k): the batch is too small for anything else to pay;
//   1  the schedule proper: bucket by bucket, dense pages first, similar pages side by side (large batches: every half-wave decodes many
//      pages, and two pages that share a wavefront cost the slower one's time in every phase of a round);
//   2  the schedule FOLDED (late round 5): the batch has more pages than the launch has wavefronts and at most twice as many -- every
//      half-wave gets one page at most, all at the start, and what the launch takes is its most loaded wavefront.  Even requests are answered
//      from the front of the schedule and odd ones from its back: the two halves of a wavefront ask together, so the densest page meets the
//      lightest, the second densest the second lightest ...  (4 096 textures with mip chains -- 6 827 pages of 64, 44 and 23 KiB -- in page
//      order: wavefronts with two full pages while others hold none; profiles/experiments/r05_many_textures.md.)
#ifndef BROTLIG_TUNE_FOLD
#define BROTLIG_TUNE_FOLD 1
#endif
__device__ __forceinline__ uint32_t schedule_mode(const DecodeArgs& a, uint32_t total)
{
    const uint32_t waves = a.decode_waves;
    if (BROTLIG_TUNE_FOLD && waves != 0u && total > waves && total - waves <= waves) return 2u;
    return total >= 1024u * a.order_from_k ? 1u : 0u;
}

// The job of global page index `g` (meaningful when `ok`): stream lookup, page table walk
// (src/BrotligDecoder.cpp:310-314), bounds against the caller's buffers.
__device__ inline PageJob fetch_job(const DecodeArgs& a, const uint32_t* order, uint32_t g, bool ok)
{
    PageJob job;
    job.valid = ok;
    job.in = nullptr; job.out = nullptr; job.in_size = job.out_size = 0; job.in_limit = 0; job.page_size = kMinPageSize;
    job.page_off = 0; job.dc = nullptr; job.index = 0; job.stream = 0;
    if (job.valid) {
        if (order != nullptr) g = order[g];                             // the schedule built by the order kernels
        job.index = g;
        const uint32_t* const page_base = a.page_base;
        const StreamDesc* const streams = a.streams;
        const uint8_t* const in = a.in;
        const uint64_t in_bytes = a.in_bytes, out_bytes = a.out_bytes;
        // stream lookup: largest s with page_base[s] <= g
        uint32_t lo = 0, hi = a.num_streams;
        while (hi - lo > 1u) { const uint32_t mid = (lo + hi) >> 1; if (page_base[mid] <= g) lo = mid; else hi = mid; }
        const uint32_t i = g - page_base[lo];
        job.stream = lo;
        const uint64_t s_in = streams[lo].in_offset, s_out = streams[lo].out_offset;
        const uint8_t* sp = in + s_in;
        StreamInfo si;
        parse_stream_header(load_u32(sp), load_u32(sp + 4), si);
        const uint8_t* table = sp + si.header_bytes;
        const uint8_t* pages = table + 4u * si.num_pages;
        const uint32_t off = i ? load_u32(table + 4u * i) : 0u;                      // src/BrotligDecoder.cpp:310
        job.in_size = i + 1u < si.num_pages ? load_u32(table + 4u * (i + 1u)) - off : load_u32(table);   // :311
        job.out_size = (i + 1u == si.num_pages && si.last_page_size) ? si.last_page_size : si.page_size;  // :314
        job.page_size = si.page_size;
        job.in = pages + off;
        const uint64_t abs_in = (uint64_t)(job.in - in);
        const uint64_t in_end = stream_in_end(streams[lo], in_bytes);
        const uint64_t room = abs_in < in_end ? in_end - abs_in : 0;
        // bytes readable from the page start: up to the end of the input buffer plus its 16 bytes of padding
        // (reads may run into the next stream: harmless, a valid page never consumes those bits)
        const uint64_t readable = abs_in < in_bytes ? in_bytes - abs_in + 16ull : 0ull;
        job.in_limit = (uint32_t)(readable > 0xFFFFFFF0ull ? 0xFFFFFFF0ull : readable);
        const uint64_t abs_out = s_out + (uint64_t)i * si.page_size;
        uint8_t* dst_base = si.preconditioned ? a.scratch : a.out;
        job.page_off = i * si.page_size;
        job.dc = si.preconditioned ? a.dc + lo : nullptr;
        job.out = dst_base + abs_out;
        // (an empty page is not a page: with a damaged table entry it can lie anywhere -- `room` is 0 beyond the stream and 0 > 0 let it through,
        // the bit readers then started at an address outside the input; found by the device soak of round 4)
        if (abs_out + job.out_size > stream_out_end(streams[lo], out_bytes) || job.in_size > room || job.in_size == 0u || dst_base == nullptr) {
            job.valid = false;
            flag_bad_page(a, lo);
        }
    }
    return job;
}

// ---- stage: flush and slide of the output window, at the start of a group whose first byte is page position `gpos` and
// whose last is `gend - 1`.  Every group first stores the finished bytes below it (aligned 16-byte pieces; `flushed` is
// 16-byte aligned until the page's last flush and at most kRoundMax + 15 bytes behind), so that a far copy -- source
// below the window, i.e. more than kHist >= kRoundMax + 16 bytes back -- only ever reads global memory written by an
// EARLIER group's flush.  When the group does not fit behind what the window holds, the window slides: kHist .. kHist + 15
// bytes of history are kept and brought down in one step, all reads before the writes.
template <class G = GeoPair>
__device__ __forceinline__ void flush_and_slide(OutView& view, uint32_t& flushed, uint8_t* out, bool on, uint64_t on_w, uint32_t gpos, uint32_t gend, uint32_t sl)
{
    const uint64_t slide_w = (kAblate & kAblSlide) ? 0ull : on_w & wave::ballot_gt(gend, view.win_base + G::kWin);       // (on_w: `on` of every lane)
    const bool slide = wave::from_mask(slide_w);
    wave::sync();
    // (the two-piece and the three-piece forms are written out separately: with the third piece as a folded-away branch inside
    // the two-piece code the compiler dropped the skip branches around the second store and issued it with an empty mask --
    // one global store and one load more per round, 4.7 % on the mixed data; round 4)
    if constexpr (G::kFlushPieces <= 2u) {
        const uint32_t e16 = gpos & ~15u;
        const uint32_t p0 = flushed + 16u * sl, p1 = p0 + 512u;
        const bool f0 = on && p0 < e16, f1 = on && p1 < e16;
        Bytes16 a0, a1;         // (each stored under the condition it is loaded under: zeroing them is four v_mov apiece)
        if (f0) a0 = load16(view.win + (p0 - view.win_base));
        if (f1) a1 = load16(view.win + (p1 - view.win_base));
        if (f0) store16(out + p0, a0);
        if (f1) store16(out + p1, a1);
        if (on && e16 > flushed) flushed = e16;
    } else {
        const uint32_t e16 = gpos & ~15u;
        const uint32_t p0 = flushed + 16u * sl, p1 = p0 + 512u, p2 = p0 + 1024u;
        const bool f0 = on && p0 < e16, f1 = on && p1 < e16, f2 = on && p2 < e16;
        Bytes16 a0, a1, a2;
        if (f0) a0 = load16(view.win + (p0 - view.win_base));
        if (f1) a1 = load16(view.win + (p1 - view.win_base));
        if (f2) a2 = load16(view.win + (p2 - view.win_base));
        if (f0) store16(out + p0, a0);
        if (f1) store16(out + p1, a1);
        if (f2) store16(out + p2, a2);
        if (on && e16 > flushed) flushed = e16;
    }
    if (slide_w != 0ull) {
        const uint32_t nb = slide ? (gpos - G::kHist) & ~15u : view.win_base;
        const uint32_t shift = nb - view.win_base, count = shift ? gpos - nb : 0u;
        if constexpr (G::kSlidePieces <= 2u) {
            const uint32_t i0 = 16u * sl, i1 = 512u + 16u * sl;
            Bytes16 m0, m1;
            if (i0 < count) m0 = load16(view.win + shift + i0);
            if (i1 < count) m1 = load16(view.win + shift + i1);
            wave::sync();
            if (i0 < count) store16(view.win + i0, m0);
            if (i1 < count) store16(view.win + i1, m1);
        } else {
            const uint32_t i0 = 16u * sl, i1 = 512u + 16u * sl, i2 = 1024u + 16u * sl;
            Bytes16 m0, m1, m2;
            if (i0 < count) m0 = load16(view.win + shift + i0);
            if (i1 < count) m1 = load16(view.win + shift + i1);
            if (i2 < count) m2 = load16(view.win + shift + i2);
            wave::sync();
            if (i0 < count) store16(view.win + i0, m0);
            if (i1 < count) store16(view.win + i1, m1);
            if (i2 < count) store16(view.win + i2, m2);
        }
        view.win_base = nb;
    }
    wave::sync();
}
template <class G = GeoPair>
__device__ __forceinline__ void flush_and_slide(OutView& view, uint32_t& flushed, uint8_t* out, bool on, uint32_t gpos, uint32_t gend, uint32_t sl)
{
    flush_and_slide<G>(view, flushed, out, on, wave::ballot64(on), gpos, gend, sl);
}

// ---- stage: the LZ77 copies of a group in dependency levels (PageDecoder.cpp:219-232 / BrotliGCompute.hlsl:1401-1419).
// One piece per lane: `plen` bytes to window index dst_idx from `dist` bytes back; the first far_len bytes of its pattern
// come from the staging area at stage_off (far sources, stored there before the call), the rest from the window at src_idx.
// A piece runs as soon as none of the pieces its source overlaps is still unfinished (dep_mask: lanes of the half).  A
// level without long pieces runs one lane per piece; otherwise the ready pieces share the 32 lanes as teams, 8 bytes per
// lane per step.  Overlapping copies replay their pattern modulo the distance, so a copy never waits for itself.
// Pieces with `far_direct` went from registers straight to their place and take no part.
#ifndef BROTLIG_TUNE_POW2_OVERLAP
#define BROTLIG_TUNE_POW2_OVERLAP 1
#endif
#ifndef BROTLIG_TUNE_PLAIN_LEVELS
#define BROTLIG_TUNE_PLAIN_LEVELS 1
#endif
// The same for a group in which every piece that takes part is simple and at most 32 bytes long (decided once per group by the
// caller, for both halves): a level is then one batch of own-lane chunk copies and nothing else -- no question about teams, about
// further batches or about the overlap path in any iteration.  (Round 4: those three questions are ~19 of a level's ~90 issued
// instructions, 3.7 levels a round; text pages take this path in nearly every group.)
template <class Clock>
__device__ __forceinline__ void copy_levels_plain(uint8_t* win, const uint64_t* stage, uint32_t plen, uint32_t far_len, uint32_t stage_off,
                                                  uint32_t src_idx, uint32_t dst_idx, uint64_t direct_w, uint32_t dep_mask, uint32_t sl, Clock& clk)
{
    const uint32_t clip8 = plen >= 8u ? plen - 8u : 0u;
    const uint8_t* const sp = far_len ? reinterpret_cast<const uint8_t*>(stage) + stage_off : win + (int32_t)src_idx;
    uint8_t* const dp = win + dst_idx;
    // The level loop's questions as wave-wide lane masks in scalar registers (wave::from_mask turns a mask back into a lane predicate
    // without an instruction): the ballot of a COMPOUND predicate goes through a 0 / 1 register and a second compare.
    uint64_t todo_w = wave::ballot_ne0(plen) & ~direct_w;
    uint32_t todo = wave::half_of(todo_w);
    const uint64_t ge8_w = wave::ballot_gt_k<7u>(plen);
    while (todo_w != 0ull) {
        clk.count(kPhLevels, 1);
        clk.halves(kPhLevelHalves, todo != 0u);
        const uint64_t ready_w = todo_w & wave::ballot_eq0(todo & dep_mask);
        if (wave::from_mask(ready_w & ge8_w)) {
            const Chunks32 c = load_chunks32(sp, plen, clip8);
            store_chunks32(dp, c, plen, clip8);
        }
        if (wave::from_mask(ready_w & ~ge8_w)) store_bytes(dp, load_u64u(sp), plen);
        clk.lap(kPhLvShort);
        todo &= ~wave::half_of(ready_w);
        todo_w &= ~ready_w;
        wave::sync();
    }
    (void)sl;
}

template <class Clock>
__device__ __forceinline__ void copy_levels(uint8_t* win, const uint64_t* stage, uint32_t plen, uint32_t dist, uint32_t far_len, uint32_t stage_off,
                                            uint32_t src_idx, uint32_t dst_idx, uint64_t direct_w, uint32_t dep_mask, uint32_t sl, bool solo, Clock& clk)
{
    const uint32_t pattern = min_u32(plen, dist);
    const uint32_t clip8 = plen >= 8u ? plen - 8u : 0u;
    const uint32_t packed = plen | (far_len << 11) | ((stage_off >> 3) << 22);
    // simple piece: pattern in one place (window or staging area) and no chunk of a 32-byte batch reads
    // what an earlier chunk of the batch wrote
    // (the questions of the level loop as wave-wide lane masks in scalar registers, see copy_levels_plain)
    uint64_t todo_w = (kAblate & kAblLevels) ? 0ull : wave::ballot_ne0(plen) & ~direct_w;
    uint32_t todo = wave::half_of(todo_w);
    const uint64_t whole_w = wave::ballot_eq0(far_len) | wave::ballot_eq(far_len, pattern);         // pattern in one place
    const uint64_t simple_w = whole_w & (wave::ballot_gt_k<31u>(dist) | ~wave::ballot_lt(dist, plen));
    const bool simple = wave::from_mask(simple_w);
    const uint64_t long_w = (simple_w & wave::ballot_gt_k<kOwnCopy>(plen)) | (~simple_w & wave::ballot_gt_k<kShortCopy>(plen));
    const uint64_t ge8_w = wave::ballot_gt_k<7u>(plen), gt32_w = wave::ballot_gt_k<32u>(plen);
#if BROTLIG_TUNE_POW2_OVERLAP
    // self-overlapping pieces with a period of 1, 2 or 4 bytes whose pattern lies in one place (asked once per group, and only when there is a
    // piece that overlaps itself at all)
    const uint64_t pow2_dist_w = (~simple_w & todo_w) != 0ull ? whole_w & ~simple_w & wave::ballot_lt_k<5u>(dist) & ~wave::ballot_eq_k<3u>(dist) : 0ull;
#endif
    while (todo_w != 0ull) {
        clk.count(kPhLevels, 1);
        clk.halves(kPhLevelHalves, todo != 0u);
        const uint64_t ready_w = todo_w & wave::ballot_eq0(todo & dep_mask);
        const bool ready = wave::from_mask(ready_w);
        const uint32_t ready_mask = wave::half_of(ready_w);
        if ((kAblate & kAblTeams) || (ready_w & long_w) == 0ull) {
            // Own-lane copies.  The usual piece (pattern in one place; distance >= 32 or no overlap
            // with itself) moves in batches of four 8-byte chunks, loads before stores, at offsets
            // clipped to plen - 8: within a batch no chunk reads what an earlier chunk of the batch
            // wrote, and every byte loaded belongs to the source (a piece ready in this level never
            // has another ready piece inside its source).
            const uint8_t* sp = far_len ? reinterpret_cast<const uint8_t*>(stage) + stage_off : win + (int32_t)src_idx;
            uint8_t* dp = win + dst_idx;
            const bool whole = wave::from_mask(whole_w);
            const uint64_t a_w = (kAblate & kAblOwnLane) ? 0ull : ready_w & simple_w, b_w = (kAblate & kAblOverlap) ? 0ull : ready_w & ~simple_w;
            if (wave::from_mask(a_w & ge8_w)) {
                const Chunks32 c = load_chunks32(sp, plen, clip8);
                store_chunks32(dp, c, plen, clip8);
            }
            if (wave::from_mask(a_w & ~ge8_w)) store_bytes(dp, load_u64u(sp), plen);
            uint64_t more_w = a_w & gt32_w;
            for (uint32_t o = 32u; more_w != 0ull; o += 32u, more_w &= wave::ballot_gt(plen, o)) {      // further batches: bytes o .. min(o + 32, plen) - 1
                if (wave::from_mask(more_w)) {
                    const uint32_t c0 = min_u32(o, clip8), c1 = min_u32(o + 8u, clip8), c2 = min_u32(o + 16u, clip8), c3 = min_u32(o + 24u, clip8);
                    uint64_t v0, v1 = 0, v2 = 0, v3 = 0;
                    v0 = load_u64u(sp + c0);
                    if (plen > o + 8u) v1 = load_u64u(sp + c1);
                    if (plen > o + 16u) v2 = load_u64u(sp + c2);
                    if (plen > o + 24u) v3 = load_u64u(sp + c3);
                    __builtin_memcpy(dp + c0, &v0, 8);
                    if (plen > o + 8u) __builtin_memcpy(dp + c1, &v1, 8);
                    if (plen > o + 16u) __builtin_memcpy(dp + c2, &v2, 8);
                    if (plen > o + 24u) __builtin_memcpy(dp + c3, &v3, 8);
                }
            }
            clk.lap(kPhLvShort);
#if BROTLIG_TUNE_POW2_OVERLAP
            // Periods of 1, 2 and 4 bytes (a repeated byte, 16-bit sample, 32-bit word: most self-overlapping pieces of sampled data; here
            // the piece is at most 32 bytes, longer ones run in teams): the pattern as ONE 8-byte word whose halves are alike, stored at 0 / 8 /
            // 16 and, rotated to its phase, at plen - 8.  Nothing the piece wrote is read back: no LDS round trip per chunk.
            uint64_t rest_w = b_w;
            {
                const uint64_t pow2_w = b_w & pow2_dist_w;
                if (pow2_w != 0ull) {
                    if (wave::from_mask(pow2_w)) {
                        // (everything here depends on an opaque zero: otherwise the loop-invariant part -- phase, addresses, comparisons -- is
                        // hoisted in front of the level loop, paid by every group and kept in registers across the levels: text -2 %)
                        const uint32_t z = wave::opaque_zero();
                        const uint32_t d = dist | z, n = plen | z, c8 = n - 8u;
                        uint8_t* const q = dp + z;
                        uint32_t x;
                        __builtin_memcpy(&x, sp + z, 4);
                        const uint32_t w = d == 4u ? x : d == 2u ? (x & 0xFFFFu) * 0x00010001u : (x & 0xFFu) * 0x01010101u;
                        const uint32_t ph = 8u * (c8 & (d - 1u));                  // phase of the chunk that ends the piece
                        const uint32_t wt = ph ? (w >> ph) | (w << (32u - ph)) : w;
                        const uint64_t v = (uint64_t)w | ((uint64_t)w << 32), vt = (uint64_t)wt | ((uint64_t)wt << 32);
                        if (n >= 8u) {
                            __builtin_memcpy(q, &v, 8);
                            if (n > 8u) __builtin_memcpy(q + c8, &vt, 8);
                            if (n >= 16u) __builtin_memcpy(q + 8u, &v, 8);
                            if (n >= 24u) __builtin_memcpy(q + 16u, &v, 8);
                        } else store_bytes(q, v, n);
                    }
                    rest_w &= ~pow2_w;
                }
            }
            const bool lane_b = wave::from_mask(rest_w);
            if (rest_w != 0ull) {
#else
            const bool lane_b = wave::from_mask(b_w);
            if (b_w != 0ull) {
#endif
                // The rest.  Self-overlapping pieces with a distance below 32 are copied forward in
                // 8-byte chunks from `dd` bytes back, each chunk reading what its predecessors wrote
                // (LDS accesses of a wave execute in order); a distance below 8 first lays down eight
                // bytes of its pattern and then continues from the smallest multiple of itself that is
                // >= 8 (8 - dd >= -dist: the read never reaches below the pattern).  Patterns that
                // straddle the window boundary go byte by byte.
                const uint8_t* own_stage = reinterpret_cast<const uint8_t*>(stage) + stage_off;
                const uint8_t* own_win = win + (int32_t)src_idx;
                uint32_t dd = dist, o0 = 0u, r = 0u;
                if (lane_b && whole && dist < 8u) {
                    store_bytes(dp, pattern_source8(sp, dist, 0u), plen);
                    dd = (uint32_t)(0x0E0C0A0809080800ull >> (8u * dist)) & 0xFFu;     // 8, 8, 9, 8, 10, 12, 14 for 1..7
                    o0 = 8u;
                }
                for (uint32_t o = o0; wave::any(lane_b && o < plen); o += 8u) {
                    if (lane_b && o < plen) {
                        uint64_t v;
                        if (whole) v = load_u64u(dp + o - dd);
                        else {
                            v = 0;
                            uint32_t rr = r;
                            for (uint32_t b = 0; b < 8u; ++b) {
                                const uint64_t x = rr < far_len ? own_stage[rr] : own_win[rr];
                                v |= x << (8u * b);
                                rr = rr + 1u == dist ? 0u : rr + 1u;
                            }
                            r = advance_mod(r, 8u, dist);
                        }
                        store_bytes(dp + o, v, plen - o);
                    }
                }
                clk.lap(kPhLvOverlap);
            }
        } else {
        clk.count(kPhTeamLevels, 1);
        // Small batches (round 4): a wavefront that decodes one page lends the idle upper half to the teams -- twice the lanes
        // per long piece.  All 64 lanes work in the one record of the wavefront (PageRecord<true>); the upper lanes take the
        // pieces' fields from the lower half's lanes.
        Team t;
        uint32_t t_pk, t_dist, t_src, t_dst, team_mask = ready_mask;
        uint8_t* t_lds = win;
        const uint8_t* t_stg = reinterpret_cast<const uint8_t*>(stage);
        if (solo) {
            const uint32_t lane = wave::lane_id();
            team_mask = wave::bcast(ready_mask, 0u);
            t = make_team64(team_mask, lane);
            t_pk = wave::bcast(packed, t.job); t_dist = wave::bcast(dist, t.job);
            t_src = wave::bcast(src_idx, t.job); t_dst = wave::bcast(dst_idx, t.job);
        } else {
            t = make_team(ready_mask, sl);
            t_pk = wave::half_shfl(packed, t.job); t_dist = wave::half_shfl(dist, t.job);
            t_src = wave::half_shfl(src_idx, t.job); t_dst = wave::half_shfl(dst_idx, t.job);
        }
        const uint32_t t_len = t_pk & 0x7FFu, t_far = (t_pk >> 11) & 0x7FFu;
        const uint8_t* t_stage = t_stg + ((t_pk >> 22) << 3);
        const uint8_t* t_win = t_lds + (int32_t)t_src;
        uint8_t* t_out = t_lds + t_dst;
        const bool act = t.serves && team_mask != 0u;
        const uint32_t t_pat = t_dist < t_len ? t_dist : t_len;
        const bool whole = t_far == 0u || t_far == t_pat;    // pattern in one place (window or staging area)
        const uint8_t* t_base = t_far ? t_stage : t_win;
        const bool overlap = t_dist < t_len;
        clk.lap(kPhLvShort);
#if BROTLIG_TUNE_POW2_OVERLAP
        // Every served piece a run with a period that divides 8 (a repeated byte, 16-bit sample, 32 / 64-bit word; pattern in one place): every
        // 8-byte chunk of it is the same word -- read once, stored by the team, no remainder per chunk (byte runs: config 2).
        const uint64_t act_w = wave::ballot64(act);
        const uint64_t p8_w = act_w & wave::ballot_lt(t_dist, t_len) & wave::ballot_lt_k<9u>(t_dist) & wave::ballot_eq0(t_dist & (t_dist - 1u)) &
                              (wave::ballot_eq0(t_far) | wave::ballot_eq(t_far, t_pat));
        if (act_w != 0ull && p8_w == act_w) {
            uint64_t v = 0;
            if (act) v = pattern_source8(t_base, t_dist, 0u);
            for (uint32_t c = t.member; wave::any(act && 8u * c < t_len); c += 1u << t.log2_size) {
                const uint32_t j = 8u * c;
                if (act && j < t_len) store_bytes(t_out + j, v, t_len - j);
            }
            wave::sync();
        } else
#endif
        for (uint32_t c = t.member; wave::any(act && 8u * c < t_len); c += 1u << t.log2_size) {
            const uint32_t j = 8u * c;
            if (act && j < t_len) {
                uint32_t r = j;
                if (overlap) r = mod_u16(j, t_dist);
                uint64_t v;
                if (whole) v = pattern_source8(t_base, t_dist, r);
                else {                                      // pattern straddles the window boundary: byte by byte
                    v = 0;
                    uint32_t rr = r;
                    for (uint32_t b = 0; b < 8u; ++b) {
                        const uint32_t x = rr < t_far ? t_stage[rr] : t_win[rr];
                        v |= (uint64_t)x << (8u * b);
                        rr = rr + 1u == t_dist ? 0u : rr + 1u;
                    }
                }
                store_bytes(t_out + j, v, t_len - j);
            }
            wave::sync();
        }
        clk.lap(kPhLvBytes);
        }
        todo &= ~ready_mask;
        todo_w &= ~ready_w;
        wave::sync();
    }
}

// ---- stage: per-page delta decode of the colour sub-streams (PageDecoder.cpp:446-471): a running byte sum over each
// colour range inside the page, in place in global memory, for the halves with `do_delta`.  16 bytes per lane and step,
// 512 contiguous bytes per half-wave: byte prefix inside the lane's chunk, half-wave scan of the chunk totals, running
// carry from step to step.
__device__ __forceinline__ void delta_decode_page(const PageJob& job, bool do_delta, uint32_t sl)
{
    if (!wave::any(do_delta)) return;
    wave::global_fence();                       // the page's own stores first
    for (uint32_t c = 0; c < kMaxSubBlocks; ++c) {
        uint32_t lo = 0, hi = 0;
        if (do_delta && ((job.dc->color_mask >> c) & 1u)) {
            const uint32_t cs = job.dc->sub_stream_off[c], ce = job.dc->sub_stream_off[c + 1];
            const uint32_t ps = job.page_off, pe = job.page_off + job.out_size;
            if (cs < pe && ps < ce) { lo = (cs > ps ? cs : ps) - ps; hi = (ce < pe ? ce : pe) - ps; }
        }
        uint32_t carry = 0;
        for (uint32_t base = lo & ~15u; wave::any(base < hi); base += 512u) {
            const uint32_t pos = base + sl * 16u;
            const bool full = pos >= lo && pos + 16u <= hi;
            uint32_t w[4] = {0u, 0u, 0u, 0u};
            if (full) {
                __builtin_memcpy(w, __builtin_assume_aligned(job.out + pos, 16), 16);
            } else {
                for (uint32_t i = 0; i < 16u; ++i)
                    if (pos + i >= lo && pos + i < hi) w[i >> 2] |= (uint32_t)job.out[pos + i] << (8u * (i & 3u));
            }
            w[0] = byte_prefix(w[0]);
            w[1] = byte_add(byte_prefix(w[1]), w[0] >> 24);
            w[2] = byte_add(byte_prefix(w[2]), w[1] >> 24);
            w[3] = byte_add(byte_prefix(w[3]), w[2] >> 24);
            const uint32_t total = w[3] >> 24;
            const uint32_t incl = wave::half_scan_incl(total) & 0xFFu;
            const uint32_t add = (carry + incl - total) & 0xFFu;
            for (uint32_t k = 0; k < 4u; ++k) w[k] = byte_add(w[k], add);
            if (full) {
                __builtin_memcpy(__builtin_assume_aligned(job.out + pos, 16), w, 16);
            } else {
                for (uint32_t i = 0; i < 16u; ++i)
                    if (pos + i >= lo && pos + i < hi) job.out[pos + i] = (uint8_t)(w[i >> 2] >> (8u * (i & 3u)));
            }
            carry = (carry + wave::half_shfl(incl, 31u)) & 0xFFu;
        }
    }
}

// ---- stage: sources of copies that lie below the output window ("far": in global memory, flushed by an earlier group).
// The first far_len bytes of a piece's pattern are far.  A piece that lies below the window as a whole, does not overlap
// itself and is at most kShortCopy bytes long (far_len == plen) never touches the staging area: its own lane fetches it
// and its bytes go from registers straight to their place in the window (`direct`).  Pieces of 8 bytes and more are
// covered by 8-byte chunks at offsets 0, 8, 16, 24 clipped to plen - 8 (the last chunk ends exactly at the piece's end
// and overlaps its predecessor); shorter ones by one load and a split store.  Everything else that reaches below the
// window is staged (8-byte aligned slots of the staging area, offsets by a half-wave scan): longer pieces, and patterns
// that straddle the window boundary.  Staged pieces of up to kShortCopy bytes are fetched by their own lane too; as soon
// as one is longer, all staged pieces get teams of lanes (two chunks per lane at once, the rest at store time).
// fetch_far_sources issues the loads and returns without waiting; store_far_sources puts the bytes where they belong.
struct FarSources {
    uint64_t fe0, fe1, fe2, fe3;    // own-lane chunks
    uint64_t te0, te1;              // team chunks
    Team     team;
    uint32_t t_src, t_len, t_stage; // the team's piece: page position of its source, far bytes, staging offset
    uint32_t stage_off;             // this lane's piece: 8-byte aligned offset into the staging area
    bool     direct, staged, any_staged, teams;
    uint64_t direct_w;              // `direct` of every lane (wave-wide mask)
};
__device__ __forceinline__ FarSources fetch_far_sources(const uint8_t* out, const uint8_t* win, uint32_t src_idx, bool near_direct,
                                                        uint32_t plen, uint32_t psrc, uint32_t far_len, uint32_t sl)
{
    FarSources f;
    // (the questions as lane masks: see wave::ballot_gt)
    const uint64_t far_w = wave::ballot_ne0(far_len);
    f.direct_w = far_w & wave::ballot_eq(far_len, plen) & wave::ballot_lt_k<kShortCopy + 1u>(plen);
    if (BROTLIG_TUNE_EARLY_NEAR) f.direct_w |= wave::ballot64(near_direct);
    const uint64_t staged_w = far_w & ~f.direct_w;
    f.direct = wave::from_mask(f.direct_w);
    f.staged = wave::from_mask(staged_w);
    const uint32_t stage_len = f.staged ? (far_len + 7u) & ~7u : 0u;
    f.any_staged = staged_w != 0ull;
    f.stage_off = 0;
    if (f.any_staged) f.stage_off = wave::half_scan_incl(stage_len) - stage_len;
    f.teams = f.any_staged && (staged_w & wave::ballot_gt_k<kShortCopy>(far_len)) != 0ull;
    f.fe0 = f.fe1 = f.fe2 = f.fe3 = f.te0 = f.te1 = 0;       // (leaving these unset saves six moves a group and costs 21 spilled values: measured, not done)
    f.team = Team{5u, 0u, 0u, false};
    f.t_src = f.t_len = f.t_stage = 0;
    const uint32_t clip8 = plen >= 8u ? plen - 8u : 0u;
    if (far_len != 0u && (f.direct || !f.teams)) {        // (conditional on purpose: unconditional clipped chunks cost 1.7 % -- more lane-loads on the memory path)
        const uint8_t* s8 = out + psrc;
        const uint32_t lim = f.direct ? clip8 : 24u;            // a staged piece keeps plain offsets
        f.fe0 = load_u64u(s8);
        if (far_len > 8u) f.fe1 = load_u64u(s8 + min_u32(8u, lim));
        if (far_len > 16u) f.fe2 = load_u64u(s8 + min_u32(16u, lim));
        if (far_len > 24u) f.fe3 = load_u64u(s8 + min_u32(24u, lim));
    }
    if (near_direct) {      // the same from the window: the source lies below the group, so it is final, and whole in LDS
        const uint8_t* s8 = win + src_idx;
        f.fe0 = load_u64u(s8);
        if (plen > 8u) f.fe1 = load_u64u(s8 + min_u32(8u, clip8));
        if (plen > 16u) { f.fe2 = load_u64u(s8 + min_u32(16u, clip8)); f.fe3 = load_u64u(s8 + clip8); }
    }
    if (f.teams) {
        const uint32_t staged_mask = wave::half_of(staged_w);
        f.team = make_team(staged_mask, sl);
        f.t_src = wave::half_shfl(psrc, f.team.job); f.t_len = wave::half_shfl(far_len, f.team.job);
        f.t_stage = wave::half_shfl(f.stage_off, f.team.job);
        f.team.serves = f.team.serves && staged_mask != 0u;
        const uint32_t tsz = 1u << f.team.log2_size;
        if (f.team.serves && 8u * f.team.member < f.t_len) f.te0 = load_u64u(out + f.t_src + 8u * f.team.member);
        if (f.team.serves && 8u * (f.team.member + tsz) < f.t_len) f.te1 = load_u64u(out + f.t_src + 8u * (f.team.member + tsz));
    }
    return f;
}
__device__ __forceinline__ void store_far_sources(uint8_t* win, uint64_t* stage, const uint8_t* out, const FarSources& f, uint32_t plen, uint32_t far_len, uint32_t dst_idx)
{
    const uint32_t clip8 = plen >= 8u ? plen - 8u : 0u;
    if (f.direct) {
        uint8_t* d = win + dst_idx;
        if (plen >= 8u) {                        // (conditional on purpose: many lanes take part, and a byte-misaligned LDS store costs a cycle per lane)
            __builtin_memcpy(d, &f.fe0, 8);
            if (plen > 8u) __builtin_memcpy(d + min_u32(8u, clip8), &f.fe1, 8);
            if (plen > 16u) __builtin_memcpy(d + min_u32(16u, clip8), &f.fe2, 8);
            if (plen > 24u) __builtin_memcpy(d + clip8, &f.fe3, 8);
        } else store_bytes(d, f.fe0, plen);
    }
    if (f.any_staged) {
        if (!f.teams) {
            if (f.staged) {
                uint64_t* st = &stage[f.stage_off >> 3];
                st[0] = f.fe0;
                if (far_len > 8u) st[1] = f.fe1;
                if (far_len > 16u) st[2] = f.fe2;
                if (far_len > 24u) st[3] = f.fe3;
            }
        } else {
            const uint32_t tsz = 1u << f.team.log2_size;
            if (f.team.serves && 8u * f.team.member < f.t_len) stage[(f.t_stage >> 3) + f.team.member] = f.te0;
            if (f.team.serves && 8u * (f.team.member + tsz) < f.t_len) stage[(f.t_stage >> 3) + f.team.member + tsz] = f.te1;
            for (uint32_t c = f.team.member + 2u * tsz; wave::any(f.team.serves && 8u * c < f.t_len); c += tsz) {
                if (f.team.serves && 8u * c < f.t_len) stage[(f.t_stage >> 3) + c] = load_u64u(out + f.t_src + 8u * c);
            }
        }
    }
}

// ---- stage: which earlier pieces of the group does my copy read?  The pieces of a group are consecutive commands (every
// command has at least one byte), each starting at byte `first_rel` of the group: a bitmap of the starts (bit p <=> a
// piece starts at group byte p) and the number of starts before each of its words answer "which piece owns byte x" with
// one popcount.  Returns the lanes (of the half) whose pieces own bytes of [psrc, src_end) inside the group and come
// before me; everything below the group (page position gpos) is final.
template <class Clock, class G = GeoPair>
__device__ __forceinline__ uint32_t piece_dependencies(uint32_t* start_bits, uint8_t* start_cum, bool on, uint64_t in_group_w, uint32_t first_rel, uint32_t gpos,
                                                        uint32_t psrc, uint32_t src_end, bool has_piece, uint32_t sl, Clock& clk)
{
    const uint32_t piece_mask = wave::half_of(in_group_w);              // (in_group_w: which lanes have a piece in the group)
    const bool in_group = wave::from_mask(in_group_w);
    if (on && sl < G::kRoundMax / 32u) start_bits[sl] = 0u;
    wave::sync();
    if (in_group) atomicOr(&start_bits[first_rel >> 5], 1u << (first_rel & 31u));
    wave::sync();
    {
        const bool rd = on && sl < G::kRoundMax / 32u;
        const uint32_t w = rd ? start_bits[sl] : 0u;
        const uint32_t cw = wave::half_scan_incl((uint32_t)__popc(w));
        if (rd) start_cum[sl] = (uint8_t)(cw - (uint32_t)__popc(w));
    }
    wave::sync();
    clk.lap(kPhBitmaps);
    uint32_t m = 0;
    if (has_piece && src_end > gpos) {
        const uint32_t first_piece = ctz_u32(piece_mask);
        const uint32_t hi_rel = src_end - 1u - gpos;
        const uint32_t hi = start_cum[hi_rel >> 5] + (uint32_t)__popc(start_bits[hi_rel >> 5] & (0xFFFFFFFFu >> (31u - (hi_rel & 31u))));
        uint32_t lo = 0;
        if (psrc > gpos) {
            const uint32_t lo_rel = psrc - gpos;
            lo = start_cum[lo_rel >> 5] + (uint32_t)__popc(start_bits[lo_rel >> 5] & (0xFFFFFFFFu >> (31u - (lo_rel & 31u)))) - 1u;
        }
        // ranks lo .. hi-1 among the group's pieces: rank r is lane first_piece + r; only pieces before me can be unfinished
        const uint32_t lo_l = first_piece + lo, hi_l = min_u32(first_piece + hi, sl);
        if (hi_l > lo_l) m = ((1u << hi_l) - 1u) & ~((1u << lo_l) - 1u);
    }
    return m;
}

// ===========================================================================================
// Stages of a page decode shared by the fused kernel (decode_pages) and the entropy kernel of the
// split experiment (profiles/experiments/split_path/brotlig_split_kernels.h).  `Lds` is the per-half LDS record: both kinds carry
// lut_icp / lut_dist / lut_lit, sorted_*, limit, first_offs, page_params and ring_push under
// these names; where a table's code lengths live while it is built differs (build_lens).
// All of them run in wave-uniform control flow, with per-half predicates as operands.

template <class Lds>
__device__ __forceinline__ TableRef table_of(Lds& L, uint32_t k, uint16_t* far_syms)
{
    return TableRef{k == 0u ? L.lut_icp : k == 1u ? L.lut_dist : L.lut_lit,
                    k == 0u ? L.sorted_icp : k == 1u ? L.sorted_dist : L.sorted_lit,
                    L.limit[k], L.first_offs[k],
                    k == 0u ? kIcpAlphabet : k == 1u ? kDistAlphabet : kLitAlphabet,
                    k == 0u ? kLutBitsIcp : k == 1u ? kLutBitsDist : kLutBitsLit, far_syms};
}
// fused kernel: the output window holds the code lengths of whichever table is being built
template <class G> __device__ __forceinline__ uint8_t* build_lens(PageLdsT<G>& L, uint32_t) { return L.win; }

// ---- stage: page start.  The halves with `want` take pages from the work counter until each holds a compressed one
// (stored pages, PageDecoder.cpp:70-76, are copied on the spot; rejected ones skipped), then read the page header and
// the sub-stream size table (:79-121), start their bit readers and build the three prefix-code tables (:125-147).
// `on_pull(job)` is called by every lane of a half for every page the half takes.  Returns whether this half starts a
// page; `tables_ok` = all three descriptions were defined.
template <class Lds, class Reader, class OnPull, class Clock>
__device__ __forceinline__ bool start_pages(const DecodeArgs& a, Lds& L, PageJob& job, Reader& br, bool want, bool& finished,
                                            uint32_t sl, uint16_t* far_syms, bool& tables_ok, OnPull on_pull, Clock& clk)
{
    const uint32_t total = a.page_base[a.num_streams];
    const uint32_t* const order = (a.order != nullptr && total <= a.order_cap) ? a.order : nullptr;
    uint32_t* const work_counter = a.work_counter;
    bool need = want, start = false;
    while (wave::any(need)) {
        uint32_t g = 0;
        if (need && sl == 0u) g = atomicAdd(work_counter, 1u);
        g = wave::half_bcast(g, 0u);
        const bool got = need && g < total;
        if (need && !got) { finished = true; need = false; }
        {
            const PageJob nj = fetch_job(a, order, g, got);
            if (got) job = nj;
        }
        if (got) on_pull(job);
        const bool fresh = got && job.valid;
        const bool stored = fresh && job.in_size == job.out_size;
        if (stored) {                                       // plain copy: 16 bytes per lane, four loads in flight per step (round 5; 4 bytes per step
                                                            // until then -- 0.06 ms for a page whose neighbour half waits for it)
            const uint32_t vecs = job.out_size >> 4;        // (the page's output is 16-byte aligned; its input lies where the page table says)
            for (uint32_t i = sl; i < vecs; i += 128u) {
                Bytes16 v0, v1, v2, v3;
                __builtin_memcpy(&v0, job.in + 16u * i, 16);
                if (i + 32u < vecs) __builtin_memcpy(&v1, job.in + 16u * (i + 32u), 16);
                if (i + 64u < vecs) __builtin_memcpy(&v2, job.in + 16u * (i + 64u), 16);
                if (i + 96u < vecs) __builtin_memcpy(&v3, job.in + 16u * (i + 96u), 16);
                store16(job.out + 16u * i, v0);
                if (i + 32u < vecs) store16(job.out + 16u * (i + 32u), v1);
                if (i + 64u < vecs) store16(job.out + 16u * (i + 64u), v2);
                if (i + 96u < vecs) store16(job.out + 16u * (i + 96u), v3);
            }
            for (uint32_t i = (vecs << 4) + sl; i < job.out_size; i += 32u) job.out[i] = job.in[i];
        }
        if (fresh && !stored) { start = true; need = false; }
    }
    {
        uint32_t my_len = 0, hdr_bytes = 0;
        if (start) {
            const uint32_t w0 = br_load(job, 0u), w1 = br_load(job, 4u);
            const uint64_t h = (uint64_t)w0 | ((uint64_t)w1 << 32);
            const uint32_t npostfix = (uint32_t)h & 3u;
            const uint32_t is_delta = ((((uint32_t)h >> 6) & 1u) != 0u && job.dc != nullptr) ? 1u : 0u;   // PageDecoder.cpp:87-88
            // kept in LDS rather than in a register for the whole page: read once per round at most
            if (sl == 0u) { L.page_params = npostfix | ((((uint32_t)h >> 2) & 15u) << (npostfix + 8u)) | (is_delta << 16); L.page_stream = job.stream; }
            const uint32_t base_bits = bit_width_u32((job.in_size + 31u) / 32u);
            const uint32_t dsize_bits = bit_width_u32(bit_width_u32(job.in_size - 1u));
            const uint32_t base_size = (uint32_t)(h >> 8) & ((1u << base_bits) - 1u);
            const uint32_t delta_bits = (uint32_t)(h >> (8u + base_bits)) & ((1u << dsize_bits) - 1u);
            const uint32_t table_at = 8u + base_bits + dsize_bits;
            const uint32_t bit = table_at + sl * delta_bits;
            const uint32_t wi = (bit >> 5) * 4u;
            const uint64_t d = (uint64_t)br_load(job, wi) | ((uint64_t)br_load(job, wi + 4u) << 32);
            const uint32_t delta = (uint32_t)(d >> (bit & 31u)) & ((1u << delta_bits) - 1u);
            my_len = base_size + delta;
            hdr_bytes = ((table_at + 32u * delta_bits + 31u) / 32u) * 4u;
        }
        const uint32_t incl = wave::half_scan_incl(my_len);
        if (start) br.init(job.in, job.in_limit, hdr_bytes + incl - my_len);
    }
    clk.lap(kPhSetup);
    // one copy of the table builder in the instruction stream, run three times (ICP, distance, literal):
    // inlined three times it was most of the kernel's code size, beyond what the instruction cache holds
    tables_ok = true;
#pragma nounroll
    for (uint32_t k = 0; k < 3u; ++k) {
        const bool ok = build_table(table_of(L, k, far_syms), build_lens(L, k), br, start, sl);
        tables_ok = tables_ok && ok;
    }
    return start;
}

// ---- stage: the commands of a round (PageDecoder.cpp:290-320, :338-404; format A.6 step 1, A.7, A.8).
struct RoundCommands {
    uint32_t sent_mask;     // lanes of the half that decoded the sentinel (704): the page's last round
    uint32_t n;             // real commands of the round (0..32)
    bool     is_cmd;        // this lane holds one
    uint32_t ins, copy;     // insert and copy length (copy 0: insert-only command)
    uint32_t dcode;         // distance code (0 = implicit "last distance")
    uint32_t dist;          // distance for explicit codes >= 16; ring codes are resolved by resolve_distance_ring
};
// One command per lane.  Two refill points per command: with >= 32 bits in the window the command symbol (<= 15 bits)
// leaves >= 17 for the insert/copy extra bits, and likewise the distance symbol for its extra bits; longer fields
// (rare) take the general read.
template <class Lds, class Reader, class Clock>
__device__ __forceinline__ RoundCommands decode_round_commands(const Lds& L, const uint32_t* len_code_tab, const TableRef& t_icp, const TableRef& t_dist,
                                                                Reader& br, bool live, uint32_t sl, Clock& clk)
{
    RoundCommands c;
    uint32_t sym = 0, len = 0;
    if (live) { br.ensure(32); sym = decode_symbol<kLutBitsIcp>(t_icp, br, len); }
    clk.lap(kPhCmdSym);
    c.sent_mask = wave::half_of(wave::ballot_eq_k<kSentinel>(sym));          // (sym stays 0 in a half without a page)
    c.n = c.sent_mask ? ctz_u32(c.sent_mask) : 32u;
    c.is_cmd = live && sl < c.n;
    if (live && sl <= c.n) br.consume(len);                           // the sentinel's own bits are consumed too
    c.ins = 0; c.copy = 0; c.dist = 0; c.dcode = 0;
    if (c.is_cmd) {
        // insert and copy length codes (for insert-only symbols 705..727 the copy length stays 0)
        const bool has_copy = sym < kSentinel;
        const uint32_t cell = sym >> 6;
        const uint32_t ic = has_copy ? ((0x298500u >> (2u * cell)) & 3u) * 8u + ((sym >> 3) & 7u) : min_u32(sym - kSentinel, 23u);
        const uint32_t cc = ((0x262444u >> (2u * cell)) & 3u) * 8u + (sym & 7u);
        const uint32_t it = len_code_tab[ic], ct = has_copy ? len_code_tab[24u + cc] : 0u;
        const uint32_t ie = it >> 16, ce = ct >> 16;
        uint32_t xi, xc;
        if (ie + ce <= 17u) {                                       // both fields are already in the window
            const uint32_t x = br.peek(ie + ce);
            br.consume(ie + ce);
            xi = x & ((1u << ie) - 1u); xc = x >> ie;
        } else { xi = br.read(ie); xc = br.read(ce); }
        c.ins = (it & 0xFFFFu) + xi;
        c.copy = has_copy ? (ct & 0xFFFFu) + xc : 0u;
        clk.lap(kPhCmdExtra);
        if (has_copy && sym >= 128u) {                              // explicit distance symbol
            uint32_t dl;
            br.ensure(32);
            c.dcode = decode_symbol<kLutBitsDist>(t_dist, br, dl);
            br.consume(dl);
            if (c.dcode >= 16u) {                                   // PageDecoder.cpp:365-390
                const uint32_t pp = L.page_params;
                const uint32_t npostfix = pp & 3u, ndirect = (pp >> 8) & 0xFFu;
                if (c.dcode < 16u + ndirect) c.dist = c.dcode - 15u;
                else {
                    const uint32_t x = c.dcode - ndirect - 16u;
                    const uint32_t nbits = min_u32(1u + (x >> (npostfix + 1u)), 24u);
                    uint32_t extra;
                    if (nbits <= 17u) { extra = br.peek(nbits); br.consume(nbits); } else extra = br.read(nbits);
                    const uint32_t hcode = x >> npostfix, lcode = x & ((1u << npostfix) - 1u);
                    c.dist = ((((2u + (hcode & 1u)) << nbits) - 4u + extra) << npostfix) + lcode + ndirect + 1u;
                }
            }
        }
    }
    return c;
}

// ---- stage: the distance ring (PageDecoder.cpp:345-364, :396-403): the last four distances pushed, most recent first.
// It lives in LDS as a circular buffer of eight words: the t-th distance pushed in the page (t counts from 4: the four initial
// entries 16, 15, 11, 4 are pushes 0..3) sits in word t & 7, and all a lane keeps is the page's push count so far.  The q-th most
// recent push before a round is word (T - 1 - q) & 7 -- ONE LDS read per lane, for the lanes that need a carried entry at all --
// and a round stores its last four pushes in words T .. T + cnt - 1 (& 7): they cannot meet the four words below T that the same
// round still reads (eight consecutive push numbers at most).  Rounds 1-3 kept the four entries in registers and folded the previous
// round's pushes in with a chain of selects on the push count (sixteen v_cndmask a round, on a kernel bound by the vector ALU).
struct DistanceRing {
    uint32_t total = 4;                             // pushes of the page so far, the four initial entries included
    template <class Lds> __device__ __forceinline__ void reset(Lds& L, bool starting, uint32_t sl)
    {
        // 4, 11, 15, 16 most recent first (PageDecoder.cpp:150-153) = pushes 3, 2, 1, 0; one word per lane out of a packed constant
        // (four constants become a constant vector that is kept in registers for the whole kernel, spilled, and reloaded every round)
        if (starting && sl < 4u) L.ring[sl] = (0x040B0F10u >> (8u * sl)) & 0xFFu;
        if (starting) total = 4u;
    }
};
struct RingWords {};                                // (rounds 1-3: the ring words, loaded at the top of a round)
template <class Lds>
__device__ __forceinline__ RingWords load_ring_pushes(const Lds&, const DistanceRing&) { return RingWords{}; }
// Codes 1..15 are resolved in command order; explicit distances and code 0 need no serial step.  On return c.dist
// is final for every copy command of the round.
template <class Lds>
__device__ __forceinline__ void resolve_distance_ring(Lds& L, DistanceRing& ring, const RingWords&, RoundCommands& c, uint32_t sl)
{
    const uint32_t T = ring.total;
    const uint32_t dcode = c.dcode;
    uint32_t dist = c.dist;
    const bool is_copy = c.is_cmd && c.copy > 0u;
    const uint32_t push_mask = wave::half_of(wave::ballot_ne0(dcode));        // (a distance code is only decoded for a command with a copy)
    // A code 1..15 refers to the r-th most recent push before the command (r from the code): either
    // a command of this round (lane `src`) or the ring carried in from earlier rounds.  All lanes
    // whose source is already known resolve together; a chain of ring codes takes one pass per link
    // (the lowest unresolved lane is always resolvable).  Code 0 ("the last distance") is r = 0 without a push: it waits
    // until the chains are done.
    uint32_t pend = wave::half_of(wave::ballot_lt_k<15u>(dcode - 1u));        // codes 1 .. 15
    const uint32_t r = dcode < 4u ? dcode : (dcode < 10u ? 0u : 1u);
    const uint32_t below0 = push_mask & ((1u << sl) - 1u);
    uint32_t below = below0;
    const uint32_t cnt = (uint32_t)__popc(below);
    // the carried entry r - cnt (when the round has fewer than r + 1 pushes before me): requested now, used in the loop
    const uint32_t carried = L.ring[(T - 1u - (r - cnt)) & 7u];
    {
        if (r >= 1u && below) below &= ~(1u << msb_u32(below));
        if (r >= 2u && below) below &= ~(1u << msb_u32(below));
        if (r >= 3u && below) below &= ~(1u << msb_u32(below));
        const bool from_round = r < cnt;
        const uint32_t src = from_round ? msb_u32(below) : 0u;
        const uint32_t j = dcode >= 4u ? (dcode - 4u) % 6u : 0u, mag = dcode >= 4u ? (j >> 1) + 1u : 0u;
        while (wave::any(pend != 0u)) {
            const bool mine = ((pend >> sl) & 1u) != 0u;
            const bool ready = mine && (!from_round || ((pend >> src) & 1u) == 0u);
            const uint32_t from = wave::half_shfl(dist, src);
            if (ready) {
                const uint32_t val = from_round ? from : carried;
                dist = (j & 1u) ? val + mag : val - mag;
            }
            pend &= ~wave::half_ballot(ready);
        }
    }
    {
        const uint32_t from = wave::half_shfl(dist, below0 ? msb_u32(below0) : 0u);
        if (is_copy && dcode == 0u) dist = below0 ? from : carried;     // (r = 0, cnt = 0: `carried` is the most recent push of earlier rounds)
        // the round's last four pushes go to the ring
        const bool pusher = is_copy && dcode != 0u;
        const uint32_t above = (uint32_t)__popc((push_mask >> sl) >> 1);    // pushes after mine
        const uint32_t pushes = (uint32_t)__popc(push_mask);
        if (pusher && above < 4u) L.ring[(T + pushes - 1u - above) & 7u] = dist;
        ring.total = T + pushes;
    }
    c.dist = dist;
}

// The persistent page loop of one wavefront.  Each 32-lane half decodes its own page and takes the
// next page from the work counter as soon as it is done, independently of the other half: pages
// differ a lot in their number of rounds (stored, run-length and text pages side by side), and a
// half that waited for its neighbour would idle for the difference.  The wavefront's control flow
// stays uniform: one iteration = (page start for the halves that need one) + (one round for the
// halves inside a page) + (page end for the halves whose page just finished), each under per-half
// predicates.
// which LDS record a lane works in, and with which geometry: a record per half, or (one page per wavefront) one record for the whole
// wavefront in the storage of both -- the upper half has no page of its own and never writes to it except as a member of a copy team
template <bool kSolo> struct PageRecord;
template <> struct PageRecord<false> {
    typedef GeoPair G;
    static __device__ __forceinline__ PageLds& of(WaveLds& W, uint32_t lane) { return W.page[lane >> 5]; }
};
template <> struct PageRecord<true> {
    typedef GeoSolo G;
    static __device__ __forceinline__ PageLdsSolo& of(WaveLds& W, uint32_t) { return *reinterpret_cast<PageLdsSolo*>(&W.page[0]); }
};

template <bool kProf, bool kSolo>
__device__ inline void decode_pages(WaveLds& W, const DecodeArgs& a, unsigned long long* prof_lds)
{
    typedef typename PageRecord<kSolo>::G G;
    PhaseClock<kProf> clk;
    clk.start(prof_lds);
    const uint32_t lane = wave::lane_id();
    const uint32_t sl = lane & 31u;
    PageLdsT<G>& L = PageRecord<kSolo>::of(W, lane);

    // the three prefix codes of a page: ICP, distance, literal (PageDecoder.cpp:125-147)
    uint16_t* const far_syms = a.far_syms + (size_t)blockIdx.x * (2u * kFarSymStride);
    const TableRef t_icp{L.lut_icp, L.sorted_icp, L.limit[0], L.first_offs[0], kIcpAlphabet, kLutBitsIcp, far_syms};
    const TableRef t_dist{L.lut_dist, L.sorted_dist, L.limit[1], L.first_offs[1], kDistAlphabet, kLutBitsDist, far_syms};
    const TableRef t_lit{L.lut_lit, L.sorted_lit, L.limit[2], L.first_offs[2], kLitAlphabet, kLutBitsLit, nullptr};

    const uint32_t resync_quarters = a.status[3];                       // pairing policy, set by the prepare kernel
    // ---- per-half state of the page under construction
    PageJob job = fetch_job(a, nullptr, 0u, false);
    bool live = false;               // inside a compressed page
    // kSolo (chosen per wavefront by decode_kernel_body): this wavefront decodes one page at a time, its upper half takes no pages
    // and helps with long copies instead.  A template parameter, not a flag: the two-page instantiation is compiled without it.
    constexpr bool solo = kSolo;
    bool finished = lane >= 32u && solo;    // the work counter ran out for this half (or it sits this launch out)
    BitReader br;
    br.base = a.in; br.limit8 = 0; br.buf = 0; br.avail = 64; br.next = 0; br.queue = 0; br.queued = 64; br.flight = 0; br.zero = wave::opaque_zero();
    DistanceRing ring;
    uint32_t out_pos = 0;            // bytes of the page produced so far
    uint32_t prev_tail = 0;          // literals decoded but not yet consumed
    uint32_t carry_head = 0;
    bool bad = false;
    OutView view{L.win, 0u};
    uint32_t flushed = 0;            // page bytes below this are in global memory

    for (;;) {
        // ---- page start.  A half without a page takes one -- unless the other half is within
        //      resync_quarters / 4 of finishing its own page: then it waits and both start together (one
        //      joint table build instead of two single ones).  The prepare kernel sets the threshold per
        //      launch: 1 when neighbouring pages differ in cost (a free half starts over at once), 4 when
        //      they are alike -- then the halves stay in step, which keeps rounds of the same shape
        //      paired (measured on the BC3 config: 7 % faster in step than out of phase).
        {
            const uint32_t near_end = (live && (job.out_size - out_pos) * 4u < job.out_size * resync_quarters) ? 1u : 0u;
            const uint32_t other_near = wave::other_half(near_end);
            const bool want = !live && !finished && other_near == 0u;
            if (wave::any(want)) {
                clk.lap(kPhDelta);
                bool tables_ok = true;
                const bool start = start_pages(a, L, job, br, want, finished, sl, far_syms, tables_ok, [](const PageJob&) {}, clk);
                ring.reset(L, start, sl);
                if (start) {
                    out_pos = 0; prev_tail = 0; carry_head = 0; flushed = 0; bad = false;
                    view.win_base = 0u;
                    live = true;
                    // an undefined code description rejects the page: with the page "full" its first round is refused
                    // (or is a bare sentinel), nothing is assembled or flushed, and the page ends with `bad` set
                    if (!tables_ok) { bad = true; out_pos = flushed = job.out_size; }
                    if (kAblate & kAblRounds) live = false;             // (timing build: what the page starts alone cost)
                }
                clk.lap(kPhTables);
            }
        }
        if ((kAblate & kAblRounds) && wave::any(!finished)) continue;
        if (!wave::any(live)) break;                                    // a half without a page has none left to take
        const bool in_page = live;

        // ---- rounds (PageDecoder.cpp:174-236; format A.6) until a page ends
        do {
        // -- 1. one command per lane (the previous round's ring pushes are requested first: they are needed in step 2)
        const RingWords pushed = load_ring_pushes(L, ring);
        RoundCommands cmd = decode_round_commands(L, W.len_code_tab, t_icp, t_dist, br, live, sl, clk);
        const uint32_t sent_mask = cmd.sent_mask, n = cmd.n;
        const bool is_cmd = cmd.is_cmd;

        clk.lap(kPhCommands);
        clk.count(kPhRounds, 1);
        if constexpr (kProf) {                                          // rounds in which one half has no page left
            const uint64_t lm = wave::ballot64(live);
            clk.count(kPhSlow, (((uint32_t)lm != 0u) != ((uint32_t)(lm >> 32) != 0u)) ? 1u : 0u);
        }
        // -- 2. distance ring
        resolve_distance_ring(L, ring, pushed, cmd, sl);
        const uint32_t ins = cmd.ins, copy = cmd.copy, dist = cmd.dist;

        clk.lap(kPhRing);
        // -- 3. output positions
        const uint32_t tot = ins + copy;
        const uint32_t incl_tot = wave::half_scan_incl(tot);
        const uint32_t incl_ins = wave::half_scan_incl(ins);
        const uint32_t round_bytes = wave::half_bcast(incl_tot, 31);
        const uint32_t litcount = wave::half_bcast(incl_ins, 31);
        const uint32_t cmd_out = out_pos + incl_tot - tot;              // first literal of my command
        const uint32_t copy_dst = cmd_out + ins;
        // (every command emits at least one byte, so a page of full rounds ends here after out_size / 32 rounds at most)
        if (live && round_bytes > job.out_size - out_pos) { bad = true; live = false; }
        const bool ok_cmd = is_cmd && live;

        // literal bookkeeping of the round (PageDecoder.cpp:196-199)
        const uint32_t lit_a = incl_ins - ins;                          // my literals are consumption indices [lit_a, lit_a + ins)
        const uint32_t rel0 = incl_tot - tot;                           // my first byte, relative to the round
        const uint32_t ac = litcount > prev_tail ? litcount - prev_tail : 0u;
        const uint32_t mult = (live && n) ? div_small(min_u32(ac, 0x200000u) + n - 1u, n) : 0u;
        const uint32_t rlit = n * mult;                                 // literals decoded this round (0 when !live)
        uint32_t next_j = sl;                                           // next literal of the round this lane decodes
        const bool dist_ok = dist != 0u && dist <= copy_dst;
        if (ok_cmd && copy > 0u && !dist_ok) bad = true;
        const bool cp = ok_cmd && copy > 0u && dist_ok;
        clk.lap(kPhPositions);

        // The round's output is assembled in the LDS window in byte ranges ("groups") of at most
        // kRoundMax bytes -- nearly always a single group.  A command that crosses a group boundary
        // contributes a piece to each group; a copy piece past the first is an ordinary copy from
        // `dist` bytes back (its earlier bytes are final by then).
        const uint32_t ngroups = live ? (round_bytes + G::kRoundMax - 1u) / G::kRoundMax : 0u;
        const bool multi_group = wave::any(ngroups > 1u);
        const uint64_t okcmd_w = wave::ballot64(ok_cmd), cp_w = wave::ballot64(cp);     // (lane masks: see wave::ballot_gt)
        for (uint32_t g = 0; ; ++g) {
            const uint64_t on_w = wave::ballot_lt(g, ngroups);
            if (on_w == 0ull) break;
            const bool on = wave::from_mask(on_w);
            const uint32_t g0 = g * G::kRoundMax, g1 = on ? min_u32(round_bytes, g0 + G::kRoundMax) : g0;
            const uint32_t gpos = out_pos + g0;                         // page position of the group's first byte

            // -- 3b. flush the finished bytes, slide the window when the group does not fit
            flush_and_slide<G>(view, flushed, job.out, on, on_w, gpos, out_pos + g1, sl);
            clk.lap(kPhSlide);
            clk.count(kPhGroups, 1);
            clk.halves(kPhGroupHalves, on);
            const uint32_t span0 = gpos - view.win_base;                // window index of the group's first byte

            // -- 3c. my pieces in this group
            const uint32_t cs = rel0 + ins;                             // my copy starts here (round-relative)
            uint64_t in_group_w;                                        // lanes with a piece in the group
            uint32_t la, nlit, lit_f, plen, pdst;                       // my first byte in the group; my literal bytes in it and the consumption
                                                                        // index of the first; my copy bytes in it and their page position
            if (multi_group) {
                in_group_w = on_w & okcmd_w & wave::ballot_lt(rel0, g1) & wave::ballot_gt(rel0 + tot, g0);
                const bool in_group = wave::from_mask(in_group_w);
                const uint32_t lb = cs < g1 ? cs : g1;
                la = rel0 > g0 ? rel0 : g0;
                nlit = (in_group && lb > la) ? lb - la : 0u;
                lit_f = lit_a + (la - rel0);
                const uint32_t ca = cs > g0 ? cs : g0, cb = rel0 + tot < g1 ? rel0 + tot : g1;
                plen = (in_group && cp && cb > ca) ? cb - ca : 0u;
                pdst = out_pos + ca;
            } else {                                                    // the round is one group (nearly always): every command lies in it whole
                in_group_w = on_w & okcmd_w;
                la = rel0;
                nlit = wave::from_mask(in_group_w) ? ins : 0u;
                lit_f = lit_a;
                plen = wave::from_mask(on_w & cp_w) ? copy : 0u;
                pdst = copy_dst;
            }
            const uint32_t psrc = pdst - dist;
            const uint32_t pattern = min_u32(plen, dist);
            const uint32_t src_end = psrc + pattern;
            // the first far_len bytes of the pattern lie below the window: fetched from global memory
            // into the staging area (loads issued now, consumed after the literal decode)
            const uint32_t far_len = (plen && psrc < view.win_base && !(kAblate & kAblFar)) ? min_u32(pattern, view.win_base - psrc) : 0u;
            // A piece that lies below the window as a whole, does not overlap itself and is at most kShortCopy bytes
            // long (far_len == plen) never touches the staging area: its own lane fetches it and its bytes go from
            // these registers straight to their place in the window once the literals are decoded.  Pieces of 8 bytes
            // and more are covered by 8-byte chunks at offsets 0, 8, 16, 24 clipped to plen - 8 (the last chunk ends
            // exactly at the piece's end and overlaps its predecessor); shorter ones by one load and a split store.
            // Everything else that reaches below the window is staged: longer pieces, and patterns that straddle
            // the window boundary.  Staged pieces of up to kShortCopy bytes are fetched by their own lane too; as
            // soon as one is longer, all staged pieces get teams of lanes (two chunks per lane now, the rest later).
            // Round 4: the same for a short piece whose source lies in the window but wholly below the group (final before the
            // group started -- 27 % of the copy pieces of the mixed data, most of the first dependency level): read ahead from LDS
            // into the same registers and stored with the far pieces, instead of a level of its own.
            const bool near_direct = BROTLIG_TUNE_EARLY_NEAR && plen != 0u && far_len == 0u && plen <= kShortCopy && dist >= plen &&
                                     src_end <= gpos && !(kAblate & kAblLevels);
            const FarSources far = fetch_far_sources(job.out, L.win, psrc - view.win_base, near_direct, plen, psrc, far_len, sl);
            const bool far_direct = far.direct;
            const uint32_t stage_off = far.stage_off;
            clk.lap(kPhPieces);
            // literals of the group: consumption indices [F0, F1)
            const uint32_t mine_before = (on && ok_cmd) ? (cs <= g0 ? ins : (rel0 < g0 ? g0 - rel0 : 0u)) : 0u;   // my literals before g0
            uint32_t F0 = 0, F1 = litcount;                             // single group: all of the round's literals
            if (multi_group) { F0 = wave::half_sum(mine_before); F1 = F0 + wave::half_sum(nlit); }
            // exact dependencies of my copy piece: the pieces (of commands before me) that own bytes of its source range
            // inside this group; everything below the group is final
            const uint32_t dep_mask = piece_dependencies<PhaseClock<kProf>, G>(L.start_bits, L.start_cum, on, in_group_w, la - g0, gpos,
                                                         psrc, src_end, plen != 0u && !far_direct && !(kAblate & kAblDeps), sl, clk);
            clk.lap(kPhCopyFence);

            // -- 4. literals of the group.  Literal j of the round comes from sub-stream j mod 32 and is
            //       consumption index prev_tail + j (PageDecoder.cpp:196-206); indices below prev_tail were
            //       decoded in earlier rounds and wait in the carry ring.  They are laid down in consumption
            //       order (the reference's literal queue, PageDecoder.cpp:164-166,:209-211, one group at a time), in
            //       the staging area, which is free until the far sources are stored; then every command moves its
            //       own run to the window like a short copy.
            uint8_t* const lits = reinterpret_cast<uint8_t*>(L.stage);
            if (on) {
                const uint32_t cf1 = F1 < prev_tail ? F1 : prev_tail;                   // carried part of [F0, F1)
                for (uint32_t f = F0 + sl; f < cf1; f += 32u) lits[f - F0] = L.carry[(carry_head + f) & 63u];
                // the last group also decodes the literals beyond what the round consumes (fewer than 32):
                // they wait in the carry ring for the next round
                const bool last_group = g + 1u == ngroups;
                const uint32_t J1 = last_group ? rlit : (F1 > prev_tail ? F1 - prev_tail : 0u);
                const uint32_t keep_at = carry_head + prev_tail;        // ring index of consumption index `litcount` (mod 64)
                auto place = [&](uint32_t j, uint32_t lit) {
                    const uint32_t f = prev_tail + j;
                    if (f < litcount) lits[f - F0] = (uint8_t)lit;
                    else L.carry[(keep_at + (f - litcount)) & 63u] = (uint8_t)lit;
                };
                // two literals per refill check while at least two are left (a literal is at most 15 bits)
                for (; next_j + 32u < J1; next_j += 64u) {
                    uint32_t l0, l1;
                    clk.count(kPhLitSteps, 1);
                    br.ensure(30);
                    const uint32_t lit0 = decode_symbol<kLutBitsLit>(t_lit, br, l0);
                    br.consume(l0);
                    const uint32_t lit1 = decode_symbol<kLutBitsLit>(t_lit, br, l1);
                    br.consume(l1);
                    place(next_j, lit0);
                    place(next_j + 32u, lit1);
                }
                if (next_j < J1) {
                    uint32_t ll;
                    br.ensure(15);
                    const uint32_t lit = decode_symbol<kLutBitsLit>(t_lit, br, ll);
                    br.consume(ll);
                    place(next_j, lit);
                    next_j += 32u;
                }
            }
            wave::sync();
            // -- 4b. literal runs: from the queue to their place in the window (own lane; long inserts in teams)
            if (!(kAblate & kAblLitStore)) {
                const uint32_t q_idx = lit_f - F0, w_idx = span0 - g0 + la;
                own_copy_simple(lits + q_idx, L.win + w_idx, nlit, wave::ballot_lt_k<kOwnCopy>(nlit - 1u));      // 1 <= nlit <= kOwnCopy
                const uint64_t long_w = wave::ballot_gt_k<kOwnCopy>(nlit);
                if (long_w != 0ull) {
                    const uint32_t lmask = wave::half_of(long_w);
                    const Team tl = make_team(lmask, sl);
                    const uint32_t l_src = wave::half_shfl(q_idx, tl.job), l_dst = wave::half_shfl(w_idx, tl.job);
                    const uint32_t l_len = wave::half_shfl(nlit, tl.job);
                    const bool act = tl.serves && lmask != 0u;
                    for (uint32_t c = tl.member; wave::any(act && 8u * c < l_len); c += 1u << tl.log2_size) {
                        const uint32_t j = 8u * c;
                        if (act && j < l_len) store_bytes(L.win + l_dst + j, load_u64u(lits + l_src + j), l_len - j);
                    }
                }
            }
            wave::sync();
            clk.lap(kPhLiterals);

            // -- 5a. far sources: short whole pieces straight into the window, everything else into the
            //        staging area (aligned 8-byte LDS writes)
            const uint32_t src_idx = psrc - view.win_base;              // window index of the pattern start (negative when far)
            const uint32_t dst_idx = pdst - view.win_base;
            store_far_sources(L.win, L.stage, job.out, far, plen, far_len, dst_idx);
            wave::sync();
            clk.lap(kPhLvLong);

            // -- 5b. LZ77 copies in dependency levels.  The levels are the longest dependent chain of a round (LDS read -> LDS
            //        write -> ballot, three to four times over): while a wave is in them it goes first at its SIMD's issue
            //        port (s_setprio; +1.2 .. 1.5 % measured, any level 1..3; raised around the command decode as well it is
            //        the same on mixed data and +0.6 % on text, around the whole group loop it loses)
            wave::set_priority(1);
#if BROTLIG_TUNE_PLAIN_LEVELS
            {   // one question per group instead of three per level: does any piece need more than the plain own-lane batch?
                const uint64_t plain_w = (wave::ballot_eq0(far_len) | wave::ballot_eq(far_len, pattern)) & ~wave::ballot_lt(dist, plen) & wave::ballot_lt_k<33u>(plen);      // simple, and one batch
                if (!kAblate && (wave::ballot_ne0(plen) & ~far.direct_w & ~plain_w) == 0ull)
                    copy_levels_plain(L.win, L.stage, plen, far_len, stage_off, src_idx, dst_idx, far.direct_w, dep_mask, sl, clk);
                else
                    copy_levels(L.win, L.stage, plen, dist, far_len, stage_off, src_idx, dst_idx, far.direct_w, dep_mask, sl, solo, clk);
            }
#else
            copy_levels(L.win, L.stage, plen, dist, far_len, stage_off, src_idx, dst_idx, far.direct_w, dep_mask, sl, solo, clk);
#endif
            wave::set_priority(0);
            clk.lap(kPhCopyLevels);
        }

        // carry ring bookkeeping: consumed entries leave at the head; when the round consumed fewer
        // literals than were waiting (rlit == 0 then), the rest stays where it is
        if (live) {
            const uint32_t new_head = carry_head + min_u32(prev_tail, litcount);
            carry_head = new_head;
            prev_tail = rlit + prev_tail - litcount;
        }

        if (live) out_pos += round_bytes;                               // (a rejected round produced nothing)
        if (sent_mask) live = false;
        } while (!wave::any(in_page && !live));

        // ---- page end for the halves whose page finished (or was rejected) in the last round
        const bool ended = in_page && !live;
        wave::sync();
        if (ended) flushed = flush_window(job.out, view, flushed, out_pos, true, sl);
        if (ended && out_pos != job.out_size) bad = true;                // a valid page fills its output exactly

    // ---- per-page delta decode of the colour sub-streams
    delta_decode_page(job, ended && (L.page_params >> 16) != 0u && !bad, sl);
    if (ended && bad && sl == 0u) flag_bad_page(a, L.page_stream);
    }
    clk.lap(kPhDelta);
    clk.flush(a.prof, lane);
}

// ===========================================================================================
// Small batches, second form: TWO wavefronts per page (brotlig_decode_duo_kernel).  With fewer pages than SIMDs a page's
// latency is the whole launch, and the latency of a page is its rounds times the dependent chain of one round.  That
// chain splits where the reference shader's phases split (BrotliGCompute.hlsl:761-1347 entropy, :1401-1419 assembly): the
// first wavefront of a workgroup decodes commands, distances and literals (everything that touches the bit streams), the
// second one assembles the output window from them (everything that touches the output), one group behind -- through a
// ring of kDuoSlots step records in LDS.  A step is one group of a round: its literals in consumption order and, for a
// round's first group, the round's 32 commands.  Page start / page end / "no more pages" travel through the same ring, so
// the two wavefronts never meet at a barrier: the producer is up to kDuoSlots - 1 steps ahead and builds the next page's
// tables while the consumer still flushes the last one.
// Hand-over: the producer fills a slot, then publishes `produced` (LDS store with release semantics, wave_ops.h); the
// consumer polls it, reads the slot, and gives it back through `consumed` as soon as the group's literals are in the
// window.
constexpr uint32_t kDuoSlots = 4;
// When the producer is this many steps ahead of the consumer (0 = never) it also does the group's dependency analysis -- which needs
// positions only -- and sends the masks along: on copy-dense pages the consumer is the longer half (samples16: 64 % of the fused time).
#ifndef BROTLIG_TUNE_DUO_DEPS_AHEAD
#define BROTLIG_TUNE_DUO_DEPS_AHEAD 1     // round 5, timed on the device: one samples16 page 0.688 -> 0.655 ms, records 0.343 -> 0.335, runs 0.277 -> 0.270, text even
#endif
enum : uint32_t { kDuoGroup = 1u, kDuoRound = 2u, kDuoPageStart = 4u, kDuoPageEnd = 8u, kDuoFinish = 16u, kDuoBad = 32u, kDuoDelta = 64u, kDuoDeps = 128u };
enum : uint32_t { kDuoOk = 1u << 31, kDuoCopies = 1u << 30 };         // flags above the distance (< 2^18)
struct __attribute__((aligned(16))) DuoStep {
    uint32_t kind, round_bytes, litcount, f0;                           // what the step is; sizes of its round; first literal of its group
    uint32_t ins[32], tot[32], dist[32], rel0[32], lit_a[32];           // the round's commands (steps with kDuoRound)
    uint32_t dep[32];                                                   // the group's dependency masks (steps with kDuoDeps)
    uint64_t lits[GeoSolo::kStageBytes / 8];                            // the group's literals, consumption order (or the PageJob of a page start)
};
static_assert(sizeof(PageJob) <= GeoSolo::kStageBytes, "a page start carries its job in the literal area");
typedef Geometry<256, 272, 720> GeoDuoEntropy;                          // the producer's record only needs the table-build scratch of stage / win
static_assert(GeoDuoEntropy::kWin + 16u >= kIcpAlphabet && sizeof(uint16_t) * (1 << kLutBitsLit) + GeoDuoEntropy::kStageBytes >= kTableScratchBytes,
              "table-build scratch of the producer's record");
struct __attribute__((aligned(16))) DuoLds {
    PageLdsT<GeoDuoEntropy> e;                                          // producer: tables, carry ring, distance ring
    uint64_t stage[GeoSolo::kStageBytes / 8];                           // consumer: far sources, piece bitmaps, output window
    uint32_t start_bits[GeoSolo::kRoundMax / 32];
    uint8_t  start_cum[GeoSolo::kRoundMax / 32];
    uint8_t  win[GeoSolo::kWin + 16] __attribute__((aligned(16)));
    DuoStep  step[kDuoSlots];
    uint32_t produced, consumed;                                        // steps handed over / given back so far
    uint32_t p_start_bits[GeoSolo::kRoundMax / 32];                     // the producer's own piece bitmaps (kDuoDeps)
    uint8_t  p_start_cum[GeoSolo::kRoundMax / 32];
    uint32_t len_code_tab[48];
};

// First wavefront: phases K4..K8 of the reference shader for one page at a time (its lower half; the upper half idles).  The
// code is decode_pages<> without steps 3b-3d, 4b and 5: see there for the comments on each stage.
__device__ inline void duo_producer(DuoLds& D, const DecodeArgs& a)
{
    typedef GeoSolo G;                                                  // group size of the consumer's window
    PhaseClock<false> clk;
    const uint32_t lane = wave::lane_id(), sl = lane & 31u;
    PageLdsT<GeoDuoEntropy>& L = D.e;
    uint16_t* const far_syms = a.far_syms + (size_t)blockIdx.x * (2u * kFarSymStride);
    const TableRef t_icp{L.lut_icp, L.sorted_icp, L.limit[0], L.first_offs[0], kIcpAlphabet, kLutBitsIcp, far_syms};
    const TableRef t_dist{L.lut_dist, L.sorted_dist, L.limit[1], L.first_offs[1], kDistAlphabet, kLutBitsDist, far_syms};
    const TableRef t_lit{L.lut_lit, L.sorted_lit, L.limit[2], L.first_offs[2], kLitAlphabet, kLutBitsLit, nullptr};

    PageJob job = fetch_job(a, nullptr, 0u, false);
    bool live = false, finished = lane >= 32u, bad = false;
    BitReader br;
    br.base = a.in; br.limit8 = 0; br.buf = 0; br.avail = 64; br.next = 0; br.queue = 0; br.queued = 64; br.flight = 0; br.zero = wave::opaque_zero();
    DistanceRing ring;
    uint32_t out_pos = 0, prev_tail = 0, carry_head = 0;
    uint32_t k = 0;                                                     // steps produced so far
    auto acquire = [&D](uint32_t step) -> DuoStep& {
        while (step - wave::lds_load_acquire(&D.consumed) >= kDuoSlots) wave::nap();
        return D.step[step % kDuoSlots];
    };
    auto publish = [&D, lane](uint32_t steps) { wave::sync(); if (lane == 0u) wave::lds_store_release(&D.produced, steps); };

    for (;;) {
        if (wave::any(!live && !finished)) {
            bool tables_ok = true;
            const bool start = start_pages(a, L, job, br, !live && !finished, finished, sl, far_syms, tables_ok, [](const PageJob&) {}, clk);
            ring.reset(L, start, sl);
            if (start) {
                out_pos = 0; prev_tail = 0; carry_head = 0; bad = false;
                live = true;
                if (!tables_ok) { bad = true; out_pos = job.out_size; }  // the first round is refused (or is a bare sentinel): nothing is assembled
            }
        }
        if (!wave::any(live)) break;
        {
            DuoStep& S = acquire(k);
            if (lane == 0u) { S.kind = kDuoPageStart; *reinterpret_cast<PageJob*>(S.lits) = job; }
            publish(++k);
        }
        const bool in_page = live;
        do {
            const RingWords pushed = load_ring_pushes(L, ring);
            RoundCommands cmd = decode_round_commands(L, D.len_code_tab, t_icp, t_dist, br, live, sl, clk);
            const uint32_t sent_mask = cmd.sent_mask, n = cmd.n;
            const bool is_cmd = cmd.is_cmd;
            resolve_distance_ring(L, ring, pushed, cmd, sl);
            const uint32_t ins = cmd.ins, copy = cmd.copy, dist = cmd.dist;
            const uint32_t tot = ins + copy;
            const uint32_t incl_tot = wave::half_scan_incl(tot);
            const uint32_t incl_ins = wave::half_scan_incl(ins);
            const uint32_t round_bytes = wave::half_bcast(incl_tot, 31);
            const uint32_t litcount = wave::half_bcast(incl_ins, 31);
            const uint32_t copy_dst = out_pos + incl_tot - tot + ins;
            if (live && round_bytes > job.out_size - out_pos) { bad = true; live = false; }
            const bool ok_cmd = is_cmd && live;
            const uint32_t lit_a = incl_ins - ins, rel0 = incl_tot - tot;
            const uint32_t ac = litcount > prev_tail ? litcount - prev_tail : 0u;
            const uint32_t mult = (live && n) ? div_small(min_u32(ac, 0x200000u) + n - 1u, n) : 0u;
            const uint32_t rlit = n * mult;
            uint32_t next_j = sl;
            const bool dist_ok = dist != 0u && dist <= copy_dst;
            if (ok_cmd && copy > 0u && !dist_ok) bad = true;
            const bool cp = ok_cmd && copy > 0u && dist_ok;

            const uint32_t ngroups = live ? (round_bytes + G::kRoundMax - 1u) / G::kRoundMax : 0u;
            const bool multi_group = wave::any(ngroups > 1u);
            for (uint32_t g = 0; wave::any(g < ngroups); ++g) {
                const bool on = g < ngroups;
                const uint32_t g0 = g * G::kRoundMax, g1 = on ? min_u32(round_bytes, g0 + G::kRoundMax) : g0;
                const uint32_t cs = rel0 + ins;
                const bool in_group = on && ok_cmd && rel0 < g1 && rel0 + tot > g0;
                const uint32_t la = rel0 > g0 ? rel0 : g0, lb = cs < g1 ? cs : g1;
                const uint32_t nlit = (in_group && lb > la) ? lb - la : 0u;
                const uint32_t mine_before = (on && ok_cmd) ? (cs <= g0 ? ins : (rel0 < g0 ? g0 - rel0 : 0u)) : 0u;
                uint32_t F0 = 0, F1 = litcount;
                if (multi_group) { F0 = wave::half_sum(mine_before); F1 = F0 + wave::half_sum(nlit); }

                DuoStep& S = acquire(k);
                if (g == 0u && lane < 32u) {
                    S.ins[sl] = ok_cmd ? ins : 0u; S.tot[sl] = ok_cmd ? tot : 0u; S.rel0[sl] = rel0; S.lit_a[sl] = lit_a;
                    // (the distance only where the copy is valid: a damaged stream's ring code can wrap below zero -- 1 - 3 -- and its high bits
                    // would read as the flags; a copy that is not valid is not made, as in decode_pages)
                    S.dist[sl] = (cp ? dist : 0u) | (ok_cmd ? kDuoOk : 0u) | (cp ? kDuoCopies : 0u);
                }
                uint32_t with_deps = 0u;
                if (BROTLIG_TUNE_DUO_DEPS_AHEAD != 0) {
                    const uint32_t ahead = wave::bcast(k - wave::lds_load_acquire(&D.consumed), 0u);
                    if (ahead >= (uint32_t)BROTLIG_TUNE_DUO_DEPS_AHEAD) {
                        const uint32_t ca = cs > g0 ? cs : g0, cb = rel0 + tot < g1 ? rel0 + tot : g1;
                        const uint32_t plen = (in_group && cp && cb > ca) ? cb - ca : 0u;
                        const uint32_t pdst = out_pos + ca, psrc = pdst - (cp ? dist : 0u);
                        const uint32_t src_end = psrc + min_u32(plen, dist);
                        const uint32_t dep = piece_dependencies<PhaseClock<false>, G>(D.p_start_bits, D.p_start_cum, on, wave::ballot64(in_group), la - g0, out_pos + g0,
                                                                                     psrc, src_end, plen != 0u, sl, clk);
                        if (lane < 32u) S.dep[sl] = dep;
                        with_deps = kDuoDeps;
                    }
                }
                if (lane == 0u) { S.kind = kDuoGroup | (g == 0u ? kDuoRound : 0u) | with_deps; S.round_bytes = round_bytes; S.litcount = litcount; S.f0 = F0; }
                uint8_t* const lits = reinterpret_cast<uint8_t*>(S.lits);
                if (on) {
                    const uint32_t cf1 = F1 < prev_tail ? F1 : prev_tail;
                    for (uint32_t f = F0 + sl; f < cf1; f += 32u) lits[f - F0] = L.carry[(carry_head + f) & 63u];
                    const bool last_group = g + 1u == ngroups;
                    const uint32_t J1 = last_group ? rlit : (F1 > prev_tail ? F1 - prev_tail : 0u);
                    const uint32_t keep_at = carry_head + prev_tail;
                    auto place = [&](uint32_t j, uint32_t lit) {
                        const uint32_t f = prev_tail + j;
                        if (f < litcount) lits[f - F0] = (uint8_t)lit;
                        else L.carry[(keep_at + (f - litcount)) & 63u] = (uint8_t)lit;
                    };
                    for (; next_j + 32u < J1; next_j += 64u) {
                        uint32_t l0, l1;
                        br.ensure(30);
                        const uint32_t lit0 = decode_symbol<kLutBitsLit>(t_lit, br, l0);
                        br.consume(l0);
                        const uint32_t lit1 = decode_symbol<kLutBitsLit>(t_lit, br, l1);
                        br.consume(l1);
                        place(next_j, lit0);
                        place(next_j + 32u, lit1);
                    }
                    if (next_j < J1) {
                        uint32_t ll;
                        br.ensure(15);
                        const uint32_t lit = decode_symbol<kLutBitsLit>(t_lit, br, ll);
                        br.consume(ll);
                        place(next_j, lit);
                        next_j += 32u;
                    }
                }
                publish(++k);
            }
            if (live) {
                carry_head += min_u32(prev_tail, litcount);
                prev_tail = rlit + prev_tail - litcount;
                out_pos += round_bytes;
            }
            if (sent_mask) live = false;
        } while (!wave::any(in_page && !live));

        const bool ended = in_page && !live;
        if (ended && out_pos != job.out_size) bad = true;
        {
            const uint64_t verdict = wave::ballot64(ended && bad);
            DuoStep& S = acquire(k);
            if (lane == 0u) S.kind = kDuoPageEnd | (verdict != 0ull ? kDuoBad : 0u) | ((L.page_params >> 16) != 0u ? kDuoDelta : 0u);
            publish(++k);
        }
        if (ended && bad && sl == 0u) flag_bad_page(a, L.page_stream);
    }
    DuoStep& S = acquire(k);
    if (lane == 0u) S.kind = kDuoFinish;
    publish(++k);
}

// Second wavefront: phase K9 (LZ77 assembly, BrotliGCompute.hlsl:1401-1419; PageDecoder.cpp:209-233) from the step records, with
// the window, staging area, dependency levels and copy teams of decode_pages<> (steps 3b-3d, 4b, 5a, 5b there).  The page
// lives in the lower half; the upper half joins the copy teams of long pieces (kSolo in copy_levels).
__device__ inline void duo_consumer(DuoLds& D, const DecodeArgs& a)
{
    typedef GeoSolo G;
    PhaseClock<false> clk;
    const uint32_t lane = wave::lane_id(), sl = lane & 31u;
    const bool lower = lane < 32u;
    PageJob job = fetch_job(a, nullptr, 0u, false);
    OutView view{D.win, 0u};
    uint32_t out_pos = 0, flushed = 0;
    uint32_t k = 0;                                                     // steps consumed so far
    auto wait_for = [&D](uint32_t step) -> DuoStep& {
        while (wave::lds_load_acquire(&D.produced) <= step) wave::nap();
        return D.step[step % kDuoSlots];
    };
    auto give_back = [&D, lane](uint32_t steps) { wave::sync(); if (lane == 0u) wave::lds_store_release(&D.consumed, steps); };

    for (;;) {
        DuoStep* S = &wait_for(k);
        const uint32_t kind = wave::uniform(S->kind);
        if (kind & kDuoFinish) break;
        if (kind & kDuoPageStart) {
            job = *reinterpret_cast<const PageJob*>(S->lits);
            out_pos = 0; flushed = 0; view.win_base = 0u;
            give_back(++k);
            continue;
        }
        if (kind & kDuoPageEnd) {
            wave::sync();
            if (lower) flushed = flush_window(job.out, view, flushed, out_pos, true, sl);
            delta_decode_page(job, lower && (kind & kDuoDelta) != 0u && (kind & kDuoBad) == 0u, sl);
            give_back(++k);
            continue;
        }
        // a round: its commands come with its first group
        const uint32_t ins = lower ? S->ins[sl] : 0u, tot = lower ? S->tot[sl] : 0u, rel0 = lower ? S->rel0[sl] : 0u, lit_a = lower ? S->lit_a[sl] : 0u;
        const uint32_t dword = lower ? S->dist[sl] : 0u;
        const uint32_t dist = dword & 0x3FFFFFFFu;
        const bool ok_cmd = (dword & kDuoOk) != 0u, cp = (dword & kDuoCopies) != 0u;
        const uint32_t round_bytes = wave::uniform(S->round_bytes), litcount = wave::uniform(S->litcount);
        const uint32_t ngroups = (round_bytes + G::kRoundMax - 1u) / G::kRoundMax;
        for (uint32_t g = 0; g < ngroups; ++g) {
            if (g != 0u) S = &wait_for(k);
            const bool on = lower;
            const uint32_t F0 = wave::uniform(S->f0);
            const uint32_t g0 = g * G::kRoundMax, g1 = min_u32(round_bytes, g0 + G::kRoundMax);
            const uint32_t gpos = out_pos + g0;

            flush_and_slide<G>(view, flushed, job.out, on, gpos, out_pos + g1, sl);
            const uint32_t span0 = gpos - view.win_base;

            const uint32_t cs = rel0 + ins;
            const bool in_group = on && ok_cmd && rel0 < g1 && rel0 + tot > g0;
            const uint32_t la = rel0 > g0 ? rel0 : g0, lb = cs < g1 ? cs : g1;
            const uint32_t nlit = (in_group && lb > la) ? lb - la : 0u;
            const uint32_t lit_f = lit_a + (la - rel0);
            const uint32_t ca = cs > g0 ? cs : g0, cb = rel0 + tot < g1 ? rel0 + tot : g1;
            const uint32_t plen = (in_group && cp && cb > ca) ? cb - ca : 0u;
            const uint32_t pdst = out_pos + ca;
            const uint32_t psrc = pdst - dist;
            const uint32_t pattern = min_u32(plen, dist);
            const uint32_t src_end = psrc + pattern;
            const uint32_t far_len = (plen && psrc < view.win_base) ? min_u32(pattern, view.win_base - psrc) : 0u;
            const FarSources far = fetch_far_sources(job.out, D.win, psrc - view.win_base, false, plen, psrc, far_len, sl);
            const bool far_direct = far.direct;
            const uint32_t stage_off = far.stage_off;
            uint32_t dep_mask;
            if (BROTLIG_TUNE_DUO_DEPS_AHEAD != 0 && (wave::uniform(S->kind) & kDuoDeps) != 0u) dep_mask = lower ? S->dep[sl] : 0u;
            else dep_mask = piece_dependencies<PhaseClock<false>, G>(D.start_bits, D.start_cum, on, wave::ballot64(in_group), (rel0 > g0 ? rel0 : g0) - g0, gpos,
                                                                              psrc, src_end, plen != 0u && !far_direct, sl, clk);
            // literal runs: from the step's queue to their place in the window; then the slot goes back to the producer
            {
                const uint8_t* const lits = reinterpret_cast<const uint8_t*>(S->lits);
                const uint32_t q_idx = lit_f - F0, w_idx = span0 - g0 + la;
                own_copy_simple(lits + q_idx, D.win + w_idx, nlit, wave::ballot_lt_k<kOwnCopy>(nlit - 1u));
                const uint64_t long_w = wave::ballot_gt_k<kOwnCopy>(nlit);
                if (long_w != 0ull) {
                    const uint32_t lmask = wave::half_of(long_w);
                    const Team tl = make_team(lmask, sl);
                    const uint32_t l_src = wave::half_shfl(q_idx, tl.job), l_dst = wave::half_shfl(w_idx, tl.job);
                    const uint32_t l_len = wave::half_shfl(nlit, tl.job);
                    const bool act = lower && tl.serves && lmask != 0u;
                    for (uint32_t c = tl.member; wave::any(act && 8u * c < l_len); c += 1u << tl.log2_size) {
                        const uint32_t j = 8u * c;
                        if (act && j < l_len) store_bytes(D.win + l_dst + j, load_u64u(lits + l_src + j), l_len - j);
                    }
                }
            }
            give_back(++k);

            const uint32_t src_idx = psrc - view.win_base, dst_idx = pdst - view.win_base;
            store_far_sources(D.win, D.stage, job.out, far, plen, far_len, dst_idx);
            wave::sync();
            {
                const uint64_t plain_w = (wave::ballot_eq0(far_len) | wave::ballot_eq(far_len, pattern)) & ~wave::ballot_lt(dist, plen) & wave::ballot_lt_k<33u>(plen);
                if ((wave::ballot_ne0(plen) & ~far.direct_w & ~plain_w) == 0ull)
                    copy_levels_plain(D.win, D.stage, plen, far_len, stage_off, src_idx, dst_idx, far.direct_w, dep_mask, sl, clk);
                else
                    copy_levels(D.win, D.stage, plen, dist, far_len, stage_off, src_idx, dst_idx, far.direct_w, dep_mask, sl, true, clk);
            }
        }
        out_pos += round_bytes;                                         // (a round without bytes sends no step)
        (void)litcount;
    }
}

__global__ void __launch_bounds__(128, 4) brotlig_decode_duo_kernel(DecodeArgs a)     // (4 wavefronts per SIMD = the 8 workgroups per compute unit its LDS allows: 128 registers -- round 5: the new table builder had taken 145)
{
    __shared__ DuoLds D;
    const uint32_t t = threadIdx.x;
    if (t < 48u) D.len_code_tab[t] = kLenCodeTab[t];
    if (t == 0u) { D.produced = 0u; D.consumed = 0u; }
    __syncthreads();
    const uint32_t total = a.page_base[a.num_streams];
    if (blockIdx.x >= total || total > a.duo_limit) return;             // more workgroups than pages, or a batch for brotlig_decode_kernel
    if (t < 64u) duo_producer(D, a); else duo_consumer(D, a);
}

// NARROW mips (late round 5).  Under the 2 x 2 swizzle (PageDecoder.cpp:416-436) a pair of texture rows is 2 W consecutive blocks of every
// conditioned sub-stream, whatever W: a mip narrower than a super-tile's 128 columns -- W a power of two, H even, rows without padding -- is
// cut into super-tiles of 128 / W row pairs instead of one, each again 256 CONSECUTIVE blocks per sub-stream (the wide path's shape; before,
// such mips went through the per-block gather, one half-empty super-tile per row pair).  Returns log2 of the row pairs per super-tile
// (0: one, the general case).  dc_init counts the super-tiles with it, dc_texture walks them with it.
__device__ __forceinline__ uint32_t dc_row_group_log2(uint32_t W, uint32_t H, uint32_t pitch, uint32_t bb, uint32_t swizzle)
{
    const bool narrow = swizzle != 0u && W >= 2u && W < 128u && (W & (W - 1u)) == 0u && (H & 1u) == 0u && pitch == W * bb;
    return narrow ? 7u - (31u - (uint32_t)__clz((int)W)) : 0u;
}

// -------------------------------------------------------------------------------------------
// Pre-conditioning tables (inc/common/BrotligDataConditioner.h:92-237).  `w0`/`w1` are the two
// dwords of the PreconditionHeader; `out_size` is the stream's decompressed size, which must equal
// the texture size (:219).
__device__ inline bool dc_init(DcTable& t, uint32_t w0, uint32_t w1, uint32_t out_size)
{
    const uint32_t fmt = w1 & 0xFFu;
    // sub-block sizes per format, 4 bits each, first sub-block in the low nibble (:98-175)
    const uint32_t sizes = fmt == 1u ? 0x422u : fmt == 2u ? 0x4228u : fmt == 3u ? 0x422611u
                         : fmt == 4u ? 0x611u : fmt == 5u ? 0x611611u : 0x1u;
    const uint32_t nsub = fmt == 1u ? 3u : fmt == 2u ? 4u : fmt == 3u ? 6u : fmt == 4u ? 3u : fmt == 5u ? 6u : 1u;
    const uint32_t color = fmt == 1u ? 0x3u : fmt == 2u ? 0x6u : fmt == 3u ? 0x18u : fmt == 4u ? 0x3u : fmt == 5u ? 0x1Bu : 0u;
    const uint32_t bb = (fmt == 1u || fmt == 4u) ? 8u : (fmt == 0u || fmt > 5u) ? 1u : 16u;
    const uint32_t px = (fmt >= 1u && fmt <= 5u) ? 4u : 1u;
    const bool aligned = ((w0 >> 1) & 1u) != 0u;
    t.format = (fmt >= 1u && fmt <= 5u) ? fmt : 0u;
    t.precon = 1; t.swizzle = w0 & 1u; t.block_bytes = bb; t.num_sub = nsub; t.color_mask = color;
    t.num_mips = ((w1 >> 8) & 0x1Fu) + 1u;
    uint32_t off = 0;
    for (uint32_t i = 0; i < kMaxSubBlocks; ++i) {
        t.sub_size[i] = i < nsub ? (sizes >> (4u * i)) & 15u : 0u;
        t.sub_off[i] = off; off += t.sub_size[i];
    }
    t.w[0] = ((w0 >> 2) & 0x7FFFu) + 1u; t.h[0] = ((w0 >> 17) & 0x7FFFu) + 1u;
    t.pitch[0] = ((w1 >> 13) & 0x7FFFFu) + 1u;
    uint32_t mw = (t.w[0] * px) / 2u, mh = (t.h[0] * px) / 2u;
    t.mip_off_bytes[0] = 0; t.mip_off_blocks[0] = 0; t.item_prefix[0] = 0;
    // All sizes are accumulated in 64 bits and must stay inside the stream's output: the header fields are
    // 15 + 15 + 19 bits wide, so pitch * h alone can pass 2^32 (the reference computes in 32 bits and would
    // wrap, inc/common/BrotligDataConditioner.h:204-219; a wrapped total that happens to equal out_size must
    // not be accepted, the de-conditioning kernel walks the real rows).
    uint64_t total = 0, bytes = 0, items = 0;
    bool fits = true;
    for (uint32_t m = 0; m < t.num_mips; ++m) {
        if (m > 0u) {                                                   // :204-210
            t.w[m] = (mw + px - 1u) / px; t.h[m] = (mh + px - 1u) / px;
            const uint32_t row = t.w[m] * bb;
            t.pitch[m] = aligned ? (row + 255u) / 256u * 256u : row;
            mw /= 2u; mh /= 2u;
        }
        if ((uint64_t)t.pitch[m] < (uint64_t)t.w[m] * bb) fits = false;
        total += (uint64_t)t.w[m] * t.h[m];
        bytes += (uint64_t)t.pitch[m] * t.h[m];
        const uint32_t rg = dc_row_group_log2(t.w[m], t.h[m], t.pitch[m], bb, t.swizzle);                       // (narrow mips: 128 / W tile rows per super-tile)
        items += (uint64_t)((((t.h[m] + 1u) / 2u) + (1u << rg) - 1u) >> rg) * ((((t.pitch[m] + bb - 1u) / bb + 31u) / 32u + 3u) / 4u) * 256u;    // tile rows of whole super-tiles (4 tiles)
        if (bytes > (uint64_t)out_size || total * bb > (uint64_t)out_size || items > 0xFFFFFFFFull) fits = false;
        t.mip_off_bytes[m + 1] = fits ? (uint32_t)bytes : 0u;
        t.mip_off_blocks[m + 1] = fits ? (uint32_t)total : 0u;
        t.item_prefix[m + 1] = fits ? (uint32_t)items : 0u;
    }
    for (uint32_t m = t.num_mips; m < kMaxMips; ++m) { t.w[m] = t.h[m] = t.pitch[m] = 0; }
    if (!fits) return false;
    t.total_blocks = (uint32_t)total; t.tex_bytes = (uint32_t)(total * bb);
    t.sub_stream_off[0] = 0;
    for (uint32_t i = 0; i < kMaxSubBlocks; ++i) t.sub_stream_off[i + 1] = t.sub_stream_off[i] + t.total_blocks * t.sub_size[i];
    return bytes == (uint64_t)out_size;                                 // :219
}

// Kernel 3 (preconditioned streams only): conditioned space -> texture space.
// The reference scatters byte by byte (PageDecoder.cpp:243-265,:406-444).  Here the work item is a SUPER-TILE of 2 texture rows x 128
// block columns (256 blocks), one per wavefront and step:
//   * wide path (round 5) -- a swizzled mip with even dimensions, the super-tile full of real blocks: under the 2x2 swizzle
//     (PageDecoder.cpp:416-436) its 256 blocks are CONSECUTIVE in every conditioned sub-stream, so each sub-stream contributes one
//     contiguous segment of 256 x sub-block-size bytes.  The wavefront reads all segments with 16-byte-per-lane loads (every load
//     instruction 1 KiB of contiguous bytes; all of them in flight together), parks them in LDS (4 KiB, the segments back to back), and
//     every lane then assembles four blocks from LDS -- one typed LDS read per sub-block -- and stores them, 64 lanes x 16 bytes = 1 KiB
//     of one texture row per store instruction.  Rounds 1-4 gathered with one 1 / 2 / 4 / 6-byte load per sub-block and lane: seven load
//     instructions for the kilobyte that now takes one, and the kernel was bound by the bytes it could keep in flight that way
//     (profiles/r04_final_kernel_trace_stats_bc3.md: 2.45 ms for 4 GiB in + 4 GiB out, 56 % of the copy rate).
//   * gather path -- everything else (no swizzle, odd dimensions, the last columns of a row, row-pitch padding, small mips, unknown
//     formats): the per-block gather of rounds 1-4, two tiles of 2 x 32 blocks at a time -- one thread owns one block-sized chunk of one
//     texture row, reads that block's sub-blocks (one typed load per sub-block) and writes the chunk with one store, or zeros for
//     row-pitch padding, which the reference leaves at the 0 of its initial memset (src/BrotligDecoder.cpp:448).
// Streams are spread over blockIdx.y, a stream's super-tiles over the wavefronts of blockIdx.x.
__device__ __forceinline__ uint64_t dc_load_sub(const uint8_t* src, uint32_t sz)
{
    uint64_t v = 0;
    switch (sz) {
    case 1: v = *src; break;
    case 2: { uint16_t t; __builtin_memcpy(&t, src, 2); v = t; break; }
    case 4: { uint32_t t; __builtin_memcpy(&t, src, 4); v = t; break; }
    case 6: { uint16_t t[3]; __builtin_memcpy(t, src, 6); v = (uint64_t)t[0] | ((uint64_t)t[1] << 16) | ((uint64_t)t[2] << 32); break; }
    case 8: __builtin_memcpy(&v, src, 8); break;
    default: for (uint32_t i = 0; i < sz; ++i) v |= (uint64_t)src[i] << (8u * i); break;
    }
    return v;
}
// the sub-blocks of one block, in their order, packed into the block's 16 (or 8) bytes
template <uint32_t kSizes, uint32_t kNumSub>
__device__ __forceinline__ void dc_pack_block(const uint64_t (&v)[kNumSub], uint64_t& lo, uint64_t& hi)
{
    uint32_t off = 0;
    lo = 0; hi = 0;
#pragma unroll
    for (uint32_t sub = 0; sub < kNumSub; ++sub) {
        const uint32_t sz = (kSizes >> (4u * sub)) & 15u;
        if (off < 8u) { lo |= v[sub] << (8u * off); if (off + sz > 8u) hi |= v[sub] >> (8u * (8u - off)); }
        else hi |= v[sub] << (8u * (off - 8u));
        off += sz;
    }
}
template <uint32_t kSizes, uint32_t kNumSub> constexpr uint32_t dc_sub_off(uint32_t sub)     // bytes of a block before sub-block `sub`
{
    uint32_t off = 0;
    for (uint32_t i = 0; i < sub && i < kNumSub; ++i) off += (kSizes >> (4u * i)) & 15u;
    return off;
}

constexpr uint32_t kDcSuperCols = 128, kDcSuperBlocks = 2u * kDcSuperCols, kDcSuperTiles = kDcSuperCols / 32u;
constexpr uint32_t kDcLdsBytes = kDcSuperBlocks * 16u;             // a super-tile of 16-byte blocks
static_assert(kDcSuperTiles == 4u, "dc_init pads every tile row to whole super-tiles of four tiles");
#ifndef BROTLIG_TUNE_DC_WIDE
#define BROTLIG_TUNE_DC_WIDE 1          // 0: every super-tile through the gather path (A/B)
#endif
#ifndef BROTLIG_TUNE_DC_ASM_UNROLL
#define BROTLIG_TUNE_DC_ASM_UNROLL 1    // blocks a lane assembles from LDS side by side (of its four per super-tile): registers against LDS latency
#endif

// Gather path: the tiles tc0 .. tc0 + kTiles - 1 (2 rows x 32 chunk columns each) of tile row `tr` of mip `m`, one chunk per lane and
// tile: ALL the loads of all tiles are issued before the first is used (the gather is bound by the bytes it has in flight).
template <uint32_t kSizes, uint32_t kNumSub, uint32_t kTiles>
__device__ __forceinline__ void dc_gather_tiles(const uint32_t (&sso)[kNumSub], const uint8_t* __restrict__ cond, uint8_t* __restrict__ mip_tex,
                                                uint32_t bb, uint32_t W, uint32_t H, uint32_t pitch, uint32_t per_row, uint32_t swizzle, uint32_t mip_block0,
                                                uint32_t tr, uint32_t tc0, uint32_t l)
{
    // (the mip's geometry arrives as wave-uniform values, read from the table once per super-tile by the caller: read here, through a
    // reference, every word was a vector load per lane)
    uint8_t* const tex = mip_tex;
    uint8_t* dst[kTiles];
    uint32_t nbytes[kTiles], gblock[kTiles];
    bool valid[kTiles], loads[kTiles];
#pragma unroll
    for (uint32_t u = 0; u < kTiles; ++u) {
        valid[u] = false; loads[u] = false; dst[u] = tex; nbytes[u] = 0; gblock[u] = 0;
        const uint32_t row = 2u * tr + ((l >> 1) & 1u), col = 32u * (tc0 + u) + 2u * (l >> 2) + (l & 1u);
        if (row >= H || col >= per_row) continue;
        valid[u] = true;
        dst[u] = tex + row * pitch + col * bb;
        nbytes[u] = min_u32(bb, pitch - col * bb);
        if (col < W) {
            // inverse of the 2x2 de-swizzle (PageDecoder.cpp:416-436): texture (row, col) -> block index
            uint32_t block = row * W + col;
            const uint32_t effW = W - (W & 1u), effH = H - (H & 1u);
            if (swizzle && W >= 2u && H >= 2u && row < effH && col < effW) {
                // eff = (row / 2) * 2 effW + x with x = 4 (col / 2) + 2 (row & 1) + (col & 1) < 2 effW,
                // so eff / effW and eff % effW need one compare, not a division
                const uint32_t x = (col >> 1) * 4u + (row & 1u) * 2u + (col & 1u);
                const uint32_t wrap = x >= effW ? 1u : 0u;
                block = (2u * (row >> 1) + wrap) * W + (x - (wrap ? effW : 0u));
            }
            gblock[u] = mip_block0 + block;
            loads[u] = true;
        }
    }
    // sub-block sizes known at compile time: every load of every tile is issued here, back to back, and waited for once
    uint64_t v[kTiles][kNumSub];
#pragma unroll
    for (uint32_t u = 0; u < kTiles; ++u) {
#pragma unroll
        for (uint32_t sub = 0; sub < kNumSub; ++sub) {
            const uint32_t sz = (kSizes >> (4u * sub)) & 15u;
            v[u][sub] = dc_load_sub(cond + sso[sub] + gblock[u] * sz, sz);     // (unconditional -- a lane without a block reads block 0 and drops
                                                                                // it: a branch per load keeps the loads from being in flight together)
        }
    }
#pragma unroll
    for (uint32_t u = 0; u < kTiles; ++u) {
        if (!valid[u]) continue;
        uint64_t lo, hi;
        dc_pack_block<kSizes, kNumSub>(v[u], lo, hi);
        if (!loads[u]) { lo = 0; hi = 0; }                                  // row-pitch padding
        uint8_t* const d = dst[u];
        const bool aligned = ((uint64_t)(uintptr_t)d & (uint64_t)(bb - 1u)) == 0u;
        if (nbytes[u] == 16u && aligned) { uint64_t q[2] = {lo, hi}; __builtin_memcpy(__builtin_assume_aligned(d, 16), q, 16); }
        else if (nbytes[u] == 8u && bb == 8u && aligned) __builtin_memcpy(__builtin_assume_aligned(d, 8), &lo, 8);
        else for (uint32_t i = 0; i < nbytes[u]; ++i) d[i] = (uint8_t)((i < 8u ? lo >> (8u * i) : hi >> (8u * (i - 8u))));
    }
}

// Super-tiles st_first, st_first + step, ... < st_end of one texture, by one wavefront; returns the first one of that progression it did not take.
// `lds`: kDcLdsBytes of this wavefront's own.
template <uint32_t kSizes, uint32_t kNumSub>
__device__ __forceinline__ uint32_t dc_texture(const DcTable* __restrict__ tp, const uint8_t* __restrict__ cond, uint8_t* __restrict__ tex,
                                               uint32_t st_first, uint32_t st_end, uint32_t step, uint8_t* lds)
{
    // (the table -- written by the prepare kernel, constant here -- is read through the constant address space: every word a scalar load.  As
    // plain global memory, even behind __restrict__, its words came as one vector load per lane each, waited for in front of the loads they
    // are the addresses of, and kept in vector registers)
    const BROTLIG_CONSTANT_AS DcTable& t = *(const BROTLIG_CONSTANT_AS DcTable*)tp;
    constexpr uint32_t bbK = dc_sub_off<kSizes, kNumSub>(kNumSub);         // block bytes of the format: 8 or 16 (1 for the unknown format)
    const uint32_t lane = wave::lane_id();
    const uint32_t bb = t.block_bytes;
    uint32_t sso[kNumSub];
#pragma unroll
    for (uint32_t sub = 0; sub < kNumSub; ++sub) sso[sub] = t.sub_stream_off[sub];
    // (item_prefix counts lanes x tiles -- 64 per tile of 2 x 32 chunks --, every tile row padded to whole super-tiles: >> 8 = super-tiles)
    uint32_t m = 0;
    uint32_t st0 = st_first;
    for (; st0 < st_end; st0 += step) {
        // the mip and super-tile coordinates are wave-uniform and go to the scalar unit
        const uint32_t st = wave::uniform(st0);
        while (m + 1u < kMaxMips && (st << 8) >= t.item_prefix[m + 1]) ++m;     // (bounded by the table whatever it holds)
        const uint32_t W = t.w[m], H = t.h[m], pitch = t.pitch[m];
        const uint32_t mip_bytes0 = t.mip_off_bytes[m], mip_block0 = t.mip_off_blocks[m], swizzle = t.swizzle;
        const uint32_t per_row = (pitch + bb - 1u) / bb, tiles_x = (per_row + 31u) / 32u, supers_x = (tiles_x + kDcSuperTiles - 1u) / kDcSuperTiles;
        const uint32_t local = st - (t.item_prefix[m] >> 8);
        const uint32_t trg = local / supers_x, q = local - trg * supers_x;
        // the super-tile's tile rows (pairs of texture rows): one, or 128 / W of a narrow mip (then supers_x is 1 and q is 0)
        const uint32_t rg = dc_row_group_log2(W, H, pitch, bb, swizzle);
        const uint32_t tile_rows = (H + 1u) >> 1, tr0 = trg << rg, tr1 = ((trg + 1u) << rg) < tile_rows ? (trg + 1u) << rg : tile_rows;
        const uint32_t lw = 7u - rg;                                            // log2 of the super-tile's columns
        const bool wide = BROTLIG_TUNE_DC_WIDE && bbK >= 8u && bb == bbK && swizzle != 0u && ((W | H) & 1u) == 0u && ((mip_bytes0 | pitch) & (bbK - 1u)) == 0u &&
                          (rg ? ((trg + 1u) << rg) <= tile_rows : (2u * tr0 + 1u < H && kDcSuperCols * (q + 1u) <= W));
        if (wide) {
            if constexpr (bbK >= 8u) {
                // first block of the super-tile in conditioned order: block(row, col) = 2 (row / 2) W + 4 (col / 2) + 2 (row & 1) + (col & 1)
                const uint32_t g0 = mip_block0 + 2u * tr0 * W + kDcSuperBlocks * q;
                constexpr uint32_t kLoads = bbK / 4u;                           // 16-byte units: 16 bbK of them, 64 per load instruction
                Bytes16 seg[kLoads];
#pragma unroll
                for (uint32_t i = 0; i < kLoads; ++i) {
                    const uint32_t u = 64u * i + lane;                          // unit u = LDS bytes [16 u, 16 u + 16): the segments back to back
                    uint32_t src = sso[0] + g0 * (kSizes & 15u) + 16u * u;
#pragma unroll
                    for (uint32_t sub = 1; sub < kNumSub; ++sub) {
                        const uint32_t first = 16u * dc_sub_off<kSizes, kNumSub>(sub);         // first unit of segment `sub`
                        const uint32_t sz = (kSizes >> (4u * sub)) & 15u;
                        if (u >= first) src = sso[sub] + g0 * sz + 16u * (u - first);
                    }
                    __builtin_memcpy(&seg[i], cond + src, 16);                  // (any byte alignment: the sub-streams start where they start)
                }
#pragma unroll
                for (uint32_t i = 0; i < kLoads; ++i) store16(lds + 16u * (64u * i + lane), seg[i]);
                wave::sync();
                uint8_t* const row0 = tex + mip_bytes0 + 2u * tr0 * pitch + kDcSuperCols * q * bbK;
                const uint32_t cmask = (1u << lw) - 1u;
#pragma unroll BROTLIG_TUNE_DC_ASM_UNROLL
                for (uint32_t i = 0; i < kDcSuperBlocks / 64u; ++i) {
                    // block idx of the super-tile in TEXTURE order: row r (of 2 .. 128), column c (of 128 .. 2)
                    const uint32_t idx = 64u * i + lane, r = idx >> lw, c = idx & cmask;
                    const uint32_t j = ((r >> 1) << (lw + 1u)) + 4u * (c >> 1) + 2u * (r & 1u) + (c & 1u);       // its place among the 256, conditioned order
                    uint64_t v[kNumSub];
#pragma unroll
                    for (uint32_t sub = 0; sub < kNumSub; ++sub) {
                        const uint32_t sz = (kSizes >> (4u * sub)) & 15u;
                        v[sub] = dc_load_sub(lds + kDcSuperBlocks * dc_sub_off<kSizes, kNumSub>(sub) + j * sz, sz);
                    }
                    uint64_t lo, hi;
                    dc_pack_block<kSizes, kNumSub>(v, lo, hi);
                    uint8_t* const d = row0 + r * pitch + c * bbK;
                    if constexpr (bbK == 16u) { uint64_t w[2] = {lo, hi}; __builtin_memcpy(__builtin_assume_aligned(d, 16), w, 16); }
                    else __builtin_memcpy(__builtin_assume_aligned(d, 8), &lo, 8);
                }
                wave::sync();                                                   // the next super-tile overwrites the segments
            }
        } else {
            // two tiles at a time (four at once cost more registers than their loads in flight bring: round 4, 79 VGPRs)
#pragma nounroll
            for (uint32_t tr = tr0; tr < tr1; ++tr) {
#pragma nounroll
                for (uint32_t tc = kDcSuperTiles * q; tc < kDcSuperTiles * (q + 1u) && tc < tiles_x; tc += 2u)
                    dc_gather_tiles<kSizes, kNumSub, 2u>(sso, cond, tex + mip_bytes0, bb, W, H, pitch, per_row, swizzle, mip_block0, tr, tc, lane);
            }
        }
    }
    return st0;
}

// log2 of the wavefronts of a gang of the de-conditioning kernel; -1: by the batch (see the kernel)
#ifndef BROTLIG_TUNE_DC_GANG_LOG2
#define BROTLIG_TUNE_DC_GANG_LOG2 -1
#endif
// Super-tiles first, first + step, ... < end of the BATCH's list (DcTable::super_base says where a stream's begin), by one wavefront.
__device__ __forceinline__ void dc_walk(const DecodeArgs& a, uint32_t first, uint32_t end, uint32_t step, uint8_t* lds)
{
    if (first >= end) return;
    // (the tables were written by the prepare kernels and are constant here: wave-uniform words through the constant address space are
    // scalar loads)
    const BROTLIG_CONSTANT_AS DcTable* const dc = (const BROTLIG_CONSTANT_AS DcTable*)a.dc;
    // the stream `first` falls into: the last one whose super-tiles begin at or before it (streams without any share their successor's base and sort before it)
    uint32_t s = 0;
    for (uint32_t hi = a.num_streams; hi - s > 1u;) { const uint32_t mid = (s + hi) >> 1; if (dc[mid].super_base <= first) s = mid; else hi = mid; }
    uint32_t i = first;
    while (i < end && s < a.num_streams) {
        const BROTLIG_CONSTANT_AS DcTable& t = dc[s];
        const uint32_t base = t.super_base;
        const uint32_t supers = t.precon ? t.item_prefix[t.num_mips] >> 8 : 0u;
        if (supers == 0u || base + supers <= i || base > i) { ++s; continue; }        // (base > i: a table that is not a prefix -- nothing is touched)
        const uint32_t st_first = i - base, st_end = end - base < supers ? end - base : supers;
        const uint64_t off = a.streams[s].out_offset;
        const uint8_t* cond = a.scratch + off;
        uint8_t* tex = a.out + off;
        const DcTable* tp = a.dc + s;
        uint32_t next;
        // per-format instantiations (sub-block sizes, four bits each, first sub-block lowest: dc_init)
        switch (t.format) {
        case 1: next = dc_texture<0x422u, 3u>(tp, cond, tex, st_first, st_end, step, lds); break;
        case 2: next = dc_texture<0x4228u, 4u>(tp, cond, tex, st_first, st_end, step, lds); break;
        case 3: next = dc_texture<0x422611u, 6u>(tp, cond, tex, st_first, st_end, step, lds); break;
        case 4: next = dc_texture<0x611u, 3u>(tp, cond, tex, st_first, st_end, step, lds); break;
        case 5: next = dc_texture<0x611611u, 6u>(tp, cond, tex, st_first, st_end, step, lds); break;
        default: next = dc_texture<0x1u, 1u>(tp, cond, tex, st_first, st_end, step, lds); break;
        }
        i = base + next;
        ++s;
    }
}

__global__ void __launch_bounds__(64) brotlig_decondition_kernel(DecodeArgs a)
{
    __shared__ __attribute__((aligned(16))) uint8_t seg_lds[kDcLdsBytes];
    // The batch's super-tiles are ONE list -- stream after stream -- cut into equal runs, one per GANG of wavefronts (consecutive
    // workgroups); the gang's wavefronts take the run's super-tiles in turn, and a run begins and ends wherever it does -- inside a texture,
    // inside a mip, across streams that are not pre-conditioned.  Large textures (1 024 super-tiles = 4 MiB of BC3 and more on average) are
    // walked by gangs of 256: at any moment a gang reads and writes one neighbourhood, and HBM sees a few dozen long sequential streams
    // instead of one per wavefront (config 4: prepare + de-conditioning 1.68 ms with gangs of 256, 1.78 with gangs of one); anything smaller
    // by gangs of one, a contiguous run per wavefront (4 096 textures of 64 KiB: 0.18 ms against 0.32).  (Until late in round 5 the streams
    // were spread over blockIdx.y and each stream's super-tiles over the 256 wavefronts of blockIdx.x: right for the benchmark's 16 MiB
    // textures -- 1.63 ms --, but a 64 KiB texture has 16 super-tiles: 4 096 of them took 0.53 ms, 1.0 TB/s instead of 5, a third of that
    // batch's whole step; profiles/experiments/r05_many_textures.md.)
    const uint32_t total = wave::uniform(a.status[5]);
    if (total == 0u) return;                                            // no preconditioned stream in this batch
    const uint32_t textures = wave::uniform(a.status[2]);
    const uint32_t gl = BROTLIG_TUNE_DC_GANG_LOG2 >= 0 ? (uint32_t)BROTLIG_TUNE_DC_GANG_LOG2 : ((total >> 10) >= textures ? 8u : 0u);
    const uint32_t G = 1u << gl;
    const uint32_t gangs = (gridDim.x + G - 1u) >> gl, gang = blockIdx.x >> gl, member = blockIdx.x - (gang << gl);
    const uint32_t members = gang + 1u < gangs ? G : gridDim.x - (gang << gl);
    const uint32_t per_gang = (total + gangs - 1u) / gangs;
    const uint32_t lo = wave::uniform(gang * per_gang);
    if (lo >= total) return;
    const uint32_t hi = per_gang < total - lo ? lo + per_gang : total;
    // Gangs that run side by side start at different places of their runs: equal textures sit at power-of-two distances in memory, and
    // walking them in step would hit the same HBM channels.  The start is a whole number of turns into the run (a multiply-high, not a
    // remainder: DESIGN 6.0), the part before it comes last.
    const uint32_t turns = (hi - lo) / members;                         // (32-bit operands: the exact division)
    const uint32_t mid = lo + members * (uint32_t)(((uint64_t)(gang * 2654435761u) * turns) >> 32);
    dc_walk(a, mid + member, hi, members, seg_lds);
    dc_walk(a, lo + member, mid, members, seg_lds);
}

// -------------------------------------------------------------------------------------------
// Kernel 1: page counts per stream -> exclusive prefix.  One workgroup per 64 streams.  A batch of up to 64 streams is done in this one
// launch; for more, every workgroup leaves the prefix inside its 64 streams and their page total (DcTable::chunk_pages of its first stream),
// and brotlig_prepare_finish_kernel adds what lies before.  (Rounds 1-4 walked all streams in ONE workgroup, 64 per step, every step a chain
// of dependent loads -- descriptor, header, table: 1.3 ms for a batch of 65 536 small streams, a quarter of its whole decode; round 5,
// profiles/experiments/r05_many_streams.md.)
__global__ void __launch_bounds__(64) brotlig_prepare_kernel(DecodeArgs a)
{
    const uint32_t lane = threadIdx.x;
    const uint32_t s = blockIdx.x * 64u + lane;
    uint32_t pages = 0, supers = 0;
    if (s < a.num_streams) {
        const uint8_t* p = a.in + a.streams[s].in_offset;
        StreamInfo si;
        const uint64_t in_off = a.streams[s].in_offset;
        const uint64_t in_end = stream_in_end(a.streams[s], a.in_bytes), out_end = stream_out_end(a.streams[s], a.out_bytes);
        const bool hdr_in = in_off + 8u <= in_end;
        // Beyond the reference's two checks (src/BrotligDecoder.cpp:437-446): the page table must lie inside the
        // stream, a short last page cannot be longer than a page, and the stream's pages must fit the region the
        // caller gave it -- a damaged header must not send pages into a neighbouring stream's output.
        bool ok = hdr_in && parse_stream_header(load_u32(p), load_u32(p + 4), si) &&
                  in_off + si.header_bytes + 4ull * si.num_pages <= in_end && si.last_page_size <= si.page_size;
        uint64_t usz = 0;
        if (ok) {
            usz = (uint64_t)si.num_pages * si.page_size - (si.last_page_size ? si.page_size - si.last_page_size : 0u);
            ok = a.streams[s].out_offset + usz <= out_end;
        }
        if (ok) pages = si.num_pages;
        else atomicOr(a.status, kStatusBadHeader);
        DcTable& t = a.dc[s];
        t.precon = 0;
        t.status = ok ? 0u : kStatusBadHeader;                      // the stream's own status word (pages add kStatusBadPage)
        if (pages && si.preconditioned) {
            // the texture described by the precondition header is the stream's output (:478): the de-conditioning
            // kernel writes all of it, whatever happened to the stream's pages
            if (a.scratch == nullptr || usz > 0xFFFFFFFFull || !dc_init(t, load_u32(p + 8), load_u32(p + 12), (uint32_t)usz)) {
                t.precon = 0; pages = 0;                            // the reference has undefined behaviour here
                t.status = kStatusBadHeader;
                atomicOr(a.status, kStatusBadHeader);
            } else { atomicAdd(a.status + 2, 1u); supers = t.item_prefix[t.num_mips] >> 8; }
        }
    }
    const uint32_t lo = wave::half_scan_incl(pages);
    const uint32_t lo_total = wave::half_bcast(lo, 31);
    const uint32_t first_half_total = wave::bcast(lo_total, 0);
    const uint32_t second_half_total = wave::bcast(lo_total, 32);
    const uint32_t incl = lane < 32u ? lo : lo + first_half_total;
    const uint32_t total = first_half_total + second_half_total;
    // the same for the de-conditioning super-tiles
    const uint32_t su = wave::half_scan_incl(supers);
    const uint32_t su_total = wave::half_bcast(su, 31);
    const uint32_t su_first = wave::bcast(su_total, 0), su_second = wave::bcast(su_total, 32);
    const uint32_t su_incl = lane < 32u ? su : su + su_first;
    if (s < a.num_streams) { a.page_base[s] = incl - pages; a.dc[s].super_base = su_incl - supers; }
    if (lane == 0u) {
        if (gridDim.x == 1u) { a.page_base[a.num_streams] = total; a.work_counter[0] = 0u; a.status[5] = su_first + su_second; }
        else { a.dc[s].chunk_pages = total; a.dc[s].chunk_supers = su_first + su_second; }
    }
}

// Kernel 1b (batches of more than 64 streams; same grid): the pages of all earlier workgroups' streams, added to this one's 64 entries.
__global__ void __launch_bounds__(64) brotlig_prepare_finish_kernel(DecodeArgs a)
{
    const uint32_t lane = threadIdx.x, c = blockIdx.x;
    uint32_t acc = 0, acc_su = 0;
    for (uint32_t j = lane; j < c; j += 64u) { acc += a.dc[j * 64u].chunk_pages; acc_su += a.dc[j * 64u].chunk_supers; }
    const uint32_t lo = wave::half_scan_incl(acc);
    const uint32_t lo_total = wave::half_bcast(lo, 31);
    const uint32_t before = wave::bcast(lo_total, 0) + wave::bcast(lo_total, 32);
    const uint32_t su = wave::half_scan_incl(acc_su);
    const uint32_t su_total = wave::half_bcast(su, 31);
    const uint32_t before_su = wave::bcast(su_total, 0) + wave::bcast(su_total, 32);
    const uint32_t s = c * 64u + lane;
    if (s < a.num_streams) { a.page_base[s] += before; a.dc[s].super_base += before_su; }
    if (c + 1u == gridDim.x && lane == 0u) {
        a.page_base[a.num_streams] = before + a.dc[c * 64u].chunk_pages; a.work_counter[0] = 0u;
        a.status[5] = before_su + a.dc[c * 64u].chunk_supers;
    }
}

// -------------------------------------------------------------------------------------------
// Page schedule.  The decode kernel runs two pages per wavefront and pays the maximum of the two in
// every phase of a round, so it matters which pages meet: the same 4 GiB of mixed pages decode 12 %
// faster when similar pages are neighbours.  Pages are therefore grouped into buckets by
// compressed size relative to the page size (an eighth of an octave per bucket since round 5, see below; stored pages last) and
// handed out bucket by bucket, dense pages first (they are the slow ones, which also shortens the
// tail of the launch).  Two passes over the page tables: count, then scatter into `order`.

// compressed and decompressed size of global page g (same walk as fetch_job)
__device__ inline void page_sizes(const DecodeArgs& a, uint32_t g, uint32_t total, uint32_t& in_size, uint32_t& out_size)
{
    uint32_t lo = 0, hi = a.num_streams;
    while (hi - lo > 1u) { const uint32_t mid = (lo + hi) >> 1; if (a.page_base[mid] <= g) lo = mid; else hi = mid; }
    const uint32_t i = g - a.page_base[lo];
    const uint32_t np = (lo + 1u < a.num_streams ? a.page_base[lo + 1u] : total) - a.page_base[lo];
    const uint8_t* sp = a.in + a.streams[lo].in_offset;
    StreamInfo si;
    parse_stream_header(load_u32(sp), load_u32(sp + 4), si);
    const uint8_t* table = sp + si.header_bytes;
    const uint32_t off = i ? load_u32(table + 4u * i) : 0u;
    in_size = i + 1u < np ? load_u32(table + 4u * (i + 1u)) - off : load_u32(table);
    out_size = (i + 1u == np && si.last_page_size) ? si.last_page_size : si.page_size;
}
// Round 5: the buckets are a quarter / an eighth of an octave wide instead of half an octave (BROTLIG_TUNE_BUCKETS_PER_OCTAVE).  On the mixed
// benchmark the pairing policy below keeps the two halves of a wavefront in step (a free half waits for its neighbour), so what a pair costs
// is its SLOWER page: the narrower the bucket, the closer two neighbours of the schedule are in compressed size -- the one cost signal a page
// table holds.  (The data classes of the benchmark already sat in buckets of their own: text 3, samples16 4, records 6-9, runs 12-13 of 16.)
#ifndef BROTLIG_TUNE_BUCKETS_PER_OCTAVE
#define BROTLIG_TUNE_BUCKETS_PER_OCTAVE 8
#endif
constexpr uint32_t kBucketsPerOctave = BROTLIG_TUNE_BUCKETS_PER_OCTAVE;
static_assert(kBucketsPerOctave == 2u || kBucketsPerOctave == 4u || kBucketsPerOctave == 8u, "half, quarter or eighth octaves");
constexpr uint32_t kBuckets = 8u * kBucketsPerOctave;                   // 7.5 .. 7.9 octaves of compression ratio, then "denser", then "stored"
constexpr uint32_t kBucketStep16 = kBucketsPerOctave == 2u ? 46341u : kBucketsPerOctave == 4u ? 55109u : 60097u;    // 2^(-1/n) in 16-bit fixed point
__device__ __forceinline__ uint32_t page_bucket(uint32_t in_size, uint32_t out_size)
{
    if (in_size >= out_size) return kBuckets - 1u;                      // stored (or nonsense): cheapest, last
    // bucket b holds in_size in (out / 2^((b+1)/n), out / 2^(b/n)]
    uint32_t b = 0, t = (uint32_t)(((uint64_t)out_size * kBucketStep16) >> 16);     // out_size <= 128 KiB
    while (b < kBuckets - 2u && in_size <= t) { ++b; t = (uint32_t)(((uint64_t)t * kBucketStep16) >> 16); }
    return b;
}
constexpr uint32_t kOrderHist = 8, kOrderCursor = 8 + kBuckets;         // status word offsets
constexpr uint32_t kStatusWords = 8u + 2u * 64u;                        // the workspace header: room for the widest setting
static_assert(kOrderCursor + kBuckets <= kStatusWords && kBuckets <= 64u, "status words; one lane per bucket in the order kernels");

__global__ void __launch_bounds__(64) brotlig_order_count_kernel(DecodeArgs a)
{
    __shared__ uint32_t hist[kBuckets];
    const uint32_t lane = threadIdx.x, total = a.page_base[a.num_streams];
    if (a.order == nullptr || total > a.order_cap) return;
    if (lane < kBuckets) hist[lane] = 0u;
    wave::sync();
    for (uint32_t g = blockIdx.x * 64u + lane; g < total; g += gridDim.x * 64u) {
        uint32_t in_size, out_size;
        page_sizes(a, g, total, in_size, out_size);
        atomicAdd(&hist[page_bucket(in_size, out_size)], 1u);
    }
    wave::sync();
    if (lane < kBuckets && hist[lane]) atomicAdd(a.status + kOrderHist + lane, hist[lane]);
}

__global__ void __launch_bounds__(64) brotlig_order_scatter_kernel(DecodeArgs a)
{
    __shared__ uint32_t cnt[kBuckets], base[kBuckets], start[kBuckets];
    const uint32_t lane = threadIdx.x, total = a.page_base[a.num_streams];
    if (a.order == nullptr || total > a.order_cap) return;
    const uint32_t mode = schedule_mode(a, total);
    if (mode == 0u) {                                                   // page order
        for (uint32_t g = blockIdx.x * 64u + lane; g < total; g += gridDim.x * 64u) a.order[g] = g;
        return;
    }
    {   // where each bucket starts in the schedule: exclusive prefix of the histogram, one bucket per lane
        const uint32_t h = lane < kBuckets ? a.status[kOrderHist + lane] : 0u;
        const uint32_t incl_half = wave::half_scan_incl(h);
        const uint32_t lower_total = wave::bcast(incl_half, 31u);
        const uint32_t incl = lane < 32u ? incl_half : incl_half + lower_total;
        if (lane < kBuckets) start[lane] = incl - h;
    }
    wave::sync();
    for (uint32_t g0 = blockIdx.x * 64u; g0 < total; g0 += gridDim.x * 64u) {      // uniform trip count
        const uint32_t g = g0 + lane;
        if (lane < kBuckets) cnt[lane] = 0u;
        wave::sync();
        uint32_t b = 0, rank = 0;
        if (g < total) {
            uint32_t in_size, out_size;
            page_sizes(a, g, total, in_size, out_size);
            b = page_bucket(in_size, out_size);
            rank = atomicAdd(&cnt[b], 1u);
        }
        wave::sync();
        if (lane < kBuckets) base[lane] = cnt[lane] ? atomicAdd(a.status + kOrderCursor + lane, cnt[lane]) : 0u;
        wave::sync();
        if (g < total) {
            const uint32_t p = start[b] + base[b] + rank;               // place in the schedule proper
            // folded: the front half of the schedule answers the even requests, the back half -- from the end -- the odd ones
            a.order[mode == 2u ? (p <= (total - 1u) >> 1 ? 2u * p : 2u * (total - 1u - p) + 1u) : p] = g;
        }
        wave::sync();
    }
}

// Pairing policy of the decode kernel (decode_pages): do neighbouring pages of the schedule differ in
// cost?  Up to 256 evenly spaced pairs (2k, 2k+1) are compared by compressed size; when more than a
// quarter of them differ by over 25 % (page kinds side by side) the two halves of a wavefront run free
// of each other, otherwise they stay in step (status word 3: the number of quarters of a page within
// which a free half waits for its neighbour -- 1 or 4).  One workgroup, after the order kernels.
__global__ void __launch_bounds__(64) brotlig_policy_kernel(DecodeArgs a)
{
    const uint32_t lane = threadIdx.x, total = a.page_base[a.num_streams];
    const bool ordered = a.order != nullptr && total <= a.order_cap;
    const uint32_t pairs = total / 2u, nsamp = min_u32(pairs, 256u);
    uint32_t differ = 0, valid = 0;
    const uint32_t stride = nsamp ? pairs / nsamp : 0u;
    for (uint32_t j = lane; j < nsamp; j += 64u) {
        const uint32_t g = 2u * (j * stride);                           // evenly spaced pairs (stride = pairs / nsamp, one exact division per wavefront)
        uint32_t sa, ua, sb, ub;
        page_sizes(a, ordered ? a.order[g] : g, total, sa, ua);
        page_sizes(a, ordered ? a.order[g + 1u] : g + 1u, total, sb, ub);
        const uint32_t big = sa > sb ? sa : sb, small = sa > sb ? sb : sa;
        ++valid;
        if ((big - small) * 4u > big) ++differ;
    }
    if (valid) atomicAdd(a.status + 3, differ | (valid << 16));
    wave::global_fence();
    wave::sync();
    if (lane == 0u) {
        const uint32_t packed = a.status[3];
        a.status[3] = (packed & 0xFFFFu) * 4u > (packed >> 16) ? 1u : 4u;
    }
}

// BROTLIG_WAVE_TIMES (diagnostics build, profiles/tools/wave_times.py): the PRODUCT kernel records the first and last 100 MHz tick of every
// wavefront (two scalar clock reads and one store per wavefront); BrotligDecodePhaseProfile then launches it instead of the phase-timer twin.
#ifndef BROTLIG_WAVE_TIMES
#define BROTLIG_WAVE_TIMES 0
#endif
// Kernel 2: persistent waves; each half pulls pages until the counter runs out (decode_pages).
template <bool kProf>
__device__ __forceinline__ void decode_kernel_body(const DecodeArgs& a)
{
    __shared__ WaveLds W;
    const uint32_t lane = wave::lane_id();
    if (lane < 48u) W.len_code_tab[lane] = kLenCodeTab[lane];
    wave::sync();
    unsigned long long* prof_lds = nullptr;
    if constexpr (kProf) {                  // 200 more bytes of LDS: the timed twin runs 14 workgroups per CU, not 16
        __shared__ unsigned long long prof_acc[kNumPhases];
        prof_lds = prof_acc;
    }
    // Small batches: with fewer pages than half-waves a page decodes fastest ALONE in its wavefront (no lock-step with a
    // neighbour: every phase of a round costs the maximum over the two halves).  So the upper halves only take part in as many
    // wavefronts as there are pages beyond one per wavefront; from two pages per wavefront on, every half works.
    const uint32_t total0 = a.page_base[a.num_streams];
    if (blockIdx.x >= total0 || total0 <= a.duo_limit) return;     // more wavefronts than pages: the surplus leaves before it takes a turn at the
                                                                    // page counter; all of them when the batch is the two-wavefront kernel's
    const uint32_t doubles = total0 > gridDim.x ? total0 - gridDim.x : 0u;     // wavefronts that need both halves
    unsigned long long t_begin = 0;
    constexpr bool kTimes = kProf || BROTLIG_WAVE_TIMES;
    if constexpr (kTimes) t_begin = wave::realtime();         // 100 MHz, the same counter on every compute unit
    if (blockIdx.x >= doubles) decode_pages<kProf, true>(W, a, prof_lds);
    else decode_pages<kProf, false>(W, a, prof_lds);
    if constexpr (kTimes) {     // when each wavefront came and went (round 5: how long the launch's tail is -- profiles/tools/wave_times.py)
        if (lane == 0u && a.prof != nullptr) {
            a.prof[kNumPhases + 2u * blockIdx.x] = t_begin;
            a.prof[kNumPhases + 2u * blockIdx.x + 1u] = wave::realtime();
        }
    }
}

// BROTLIG_TUNE_WAVES_PER_SIMD (diagnostics, profiles/r04_isa_stage_budget.md): the register budget of 5 (96 VGPRs) or 6 (80) wavefronts per
// SIMD instead of the 4 (128) the kernel is built for -- only honoured by the compiler when the LDS size allows that occupancy too
// (build with a smaller window / group / LUTs, e.g. -DBROTLIG_TUNE_ROUND_MAX=384 -DBROTLIG_TUNE_HIST=400 -DBROTLIG_TUNE_WIN=800 ...).
#ifndef BROTLIG_TUNE_WAVES_PER_SIMD
#define BROTLIG_TUNE_WAVES_PER_SIMD 4
#endif
__global__ void __launch_bounds__(64, BROTLIG_TUNE_WAVES_PER_SIMD) brotlig_decode_kernel(DecodeArgs a) { decode_kernel_body<false>(a); }
// Diagnostics twin: same code with s_memtime phase timers (BrotligDecodePhaseProfile).
__global__ void __launch_bounds__(64, 4) brotlig_decode_kernel_timed(DecodeArgs a) { decode_kernel_body<true>(a); }

// Device self-test of the cross-lane primitives (results checked on the host).
__global__ void __launch_bounds__(64) brotlig_selftest_kernel(uint32_t* out)
{
    const uint32_t lane = wave::lane_id();
    const uint32_t v = (lane * 2654435761u) >> 24;
    out[lane] = wave::half_scan_incl(v);
    out[64 + lane] = wave::half_scan_incl_ref(v);
    out[128 + lane] = wave::half_ballot((v & 1u) != 0u);
    out[192 + lane] = wave::half_shfl(v, lane * 7u + 3u);
    out[384 + lane] = wave::half_bcast(v, (lane & 32u) ? 5u : 29u);    // source lane uniform within each half
    out[256 + lane] = wave::half_max(v);
    out[448 + lane] = wave::other_half(wave::half_max(v));            // a half-uniform value, seen from the other half
    out[320 + lane] = v;
}

}  // namespace brotlig

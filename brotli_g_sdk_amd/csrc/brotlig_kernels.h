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
//     that puts similar pages side by side (brotlig_schedule.h).
//
// Style rule: every wave::* call sits in wave-uniform control flow.  Per-lane loops and
// branches contain only memory and ALU work.  (tests/sim runs this same source on the CPU with
// one fiber per lane and checks the rule.)
#pragma once
#include "brotlig_kernel_common.h"
#include "brotlig_tables.h"
#include "brotlig_jobs.h"
#include "brotlig_copy_levels.h"
#include "brotlig_round.h"
#include "brotlig_duo.h"
#include "brotlig_decondition.h"
#include "brotlig_schedule.h"

namespace brotlig {

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

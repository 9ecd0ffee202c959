// brotlig_hip.hip -- C ABI (include/brotlig_amd.h) over the gfx950 decode kernels.
//
// Host side of the drop-in boundary: replaces sample/BrotligGPUDecoder.cpp:260-748 (D3D12 device,
// queue, PSO, upload/readback heaps, timestamp queries) with HIP runtime calls, and exposes the
// shader's buffer-level contract (input / meta / output, BrotliGCompute.hlsl:93-95) as a
// device-pointer batch call.  No decoding happens on the host.
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <new>
#include <random>
#include <system_error>
#include <thread>
#include <vector>

#include "brotlig_amd.h"
#include "brotlig_kernels.h"
#include "brotlig_shard_plan.h"
#include "brotlig_internal.h"

using namespace brotlig;

namespace {

static_assert(sizeof(BrotligStreamDesc) == sizeof(StreamDesc), "descriptor layout");

#define HIP_OK(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { \
    fprintf(stderr, "brotlig_hip: %s failed: %s\n", #expr, hipGetErrorString(_e)); return BROTLIG_ERROR_GENERIC; } } while (0)

// Device workspace (the reference's `meta` buffer): word 0 status, word 1 page counter, word 2
// preconditioned-stream count, word 3 pairing policy, words 8..135 scheduling buckets, words 160..175 the schedule kernel's tickets and
// phase counters, words 192.. page_base[num_streams + 1], then (1 KiB aligned) one DcTable per stream, then the per-wavefront symbol
// slots, then -- if the caller's workspace has the room -- the page schedule (one 32-byte job record per page).
constexpr size_t kWsHeaderWords = 192;
constexpr size_t kWsSyncAt = 160;        // the schedule kernel's own words: a 128-byte line of their own (the workgroups that wait look at it; the histogram's atomics go elsewhere)
static_assert(kWsSyncAt >= kStatusWords && kWsSyncAt % 32u == 0u && kWsHeaderWords >= kWsSyncAt + kSyncWords, "the status words of the kernels, then the schedule kernel's");
size_t dc_offset(uint32_t n) { return ((kWsHeaderWords + (size_t)n + 1u) * 4u + 1023u) & ~(size_t)1023u; }
// per-half slots for the prefix-code symbols that overflow the LDS arrays, for every workgroup of the largest decode grid
// (8192 workgroups x 2 halves x kFarSymStride uint16 = 30 MiB: the size brotlig_amd.h documents for the workspace)
constexpr uint32_t kMaxDecodeGrid = 8192;
#ifndef BROTLIG_DUO_MAX_PAGES
#define BROTLIG_DUO_MAX_PAGES 2048
#endif
constexpr size_t kFarSymBytes = (size_t)kMaxDecodeGrid * 2u * kFarSymStride * sizeof(uint16_t);
size_t far_syms_offset(uint32_t n) { return (dc_offset(n) + (size_t)n * sizeof(DcTable) + 255u) & ~(size_t)255u; }
size_t workspace_bytes(uint32_t n) { return far_syms_offset(n) + kFarSymBytes; }
size_t jobs_offset(uint32_t n) { return (workspace_bytes(n) + 255u) & ~(size_t)255u; }      // the page schedule, behind everything a batch must have
// every page is at least 32 KiB of output, and every stream's output region is whole pages
uint64_t max_pages(uint32_t n, uint64_t out_bytes) { return out_bytes / kMinPageSize + n; }
// From this much output on (of 64 KiB pages: about two pages per half-wave) a batch gets the schedule proper; below it, page order or the folded schedule.
constexpr uint64_t kOrderMinOutBytes = 768ull << 20;
// every launch of the process carries another 40-bit tag (DecodeArgs::launch_tag, brotlig_schedule.h): a counter from a random start
uint64_t next_launch_tag()
{
    static std::atomic<uint64_t> counter{[] {
        std::random_device rd;
        return ((uint64_t)rd() << 32) ^ (uint64_t)rd() ^ (uint64_t)std::chrono::steady_clock::now().time_since_epoch().count();
    }()};
    uint64_t t;
    do { t = counter.fetch_add(1) & ((1ull << 40) - 1u); } while (t == 0u);
    return t;
}

// Launch geometry per device (CU count x occupancy of the decode kernel), looked up once per device;
// host threads driving different devices (or the same one) may arrive here concurrently.
struct Grids {
    int decode = 0, decond = 1024, sched = 256, duo = 512;
};
// Diagnostics switches, read ONCE per process (never in the launch path): BROTLIG_WG_PER_CU pins the decode grid per compute unit
// (profiles/tools/occ_probe.sh), BROTLIG_POLICY the pairing policy (policy_sweep.sh).  -1 = not set.
int env_int_once(const char* name) { const char* e = getenv(name); return e ? atoi(e) : -1; }
int diag_wg_per_cu() { static const int v = env_int_once("BROTLIG_WG_PER_CU"); return v; }
int diag_policy() { static const int v = env_int_once("BROTLIG_POLICY"); return v; }
// Diagnostics (tests): a fixed decode grid, so that a SMALL batch can be decoded two pages per wavefront (grid < pages / 2) as well as
// one page per wavefront (the default for it).  0 = the normal rule.  Process-wide, not thread-safe: tests only.
// Round 5: both knobs are INERT unless the process was started with BROTLIG_ENABLE_DEBUG_KNOBS=1 (read once, like the other
// diagnostics switches): in a production process no thread and no forgotten reset can change which kernel a launch uses.
bool diag_knobs_enabled() { static const bool v = env_int_once("BROTLIG_ENABLE_DEBUG_KNOBS") == 1; return v; }
std::atomic<uint32_t> g_debug_grid_value{0}, g_debug_mode_value{0};
struct DebugKnob {
    std::atomic<uint32_t>& v;
    uint32_t load() const { return diag_knobs_enabled() ? v.load() : 0u; }
    void store(uint32_t x) { v.store(x); }
};
DebugKnob g_debug_grid{g_debug_grid_value};
// Diagnostics (tests, profiles/tools/latency.py): 0 = the normal rule, 1 = never the two-wavefronts-per-page kernel, 2 = always.
DebugKnob g_debug_mode{g_debug_mode_value};
// Batches that cannot hold more pages than this (every page >= 32 KiB of the caller's output region) are decoded two wavefronts per
// page (brotlig_decode_duo_kernel): the machine has four SIMDs per compute unit and a page alone keeps one of them busy.
constexpr uint64_t kDuoMaxPages = BROTLIG_DUO_MAX_PAGES;
#ifndef BROTLIG_DUO_PER_CU
#define BROTLIG_DUO_PER_CU 8
#endif
constexpr int kDuoPerCu = BROTLIG_DUO_PER_CU;            // workgroups of two wavefronts per compute unit at most (one wavefront per SIMD at two, the kernel's four at eight;
                                        // measured with 4 / 8: 2 048 mixed pages 1.59 / 1.06 ms, against 1.33 ms one wavefront per page)
constexpr int kMaxDevices = 64;
std::mutex g_grid_mutex;
Grids g_grids[kMaxDevices];

BROTLIG_ERROR grid_sizes(Grids* out)
{
    int dev = 0;
    HIP_OK(hipGetDevice(&dev));
    if (dev < 0 || dev >= kMaxDevices) return BROTLIG_ERROR_GENERIC;
    std::lock_guard<std::mutex> lock(g_grid_mutex);
    Grids& g = g_grids[dev];
    if (g.decode == 0) {
        int cus = 0, per_cu = 0;
        HIP_OK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        HIP_OK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, brotlig_decode_kernel, 64, 0));
        // LDS is handed out in 1280-byte granules, 128 to a CU (measured), which the query does not know
        hipFuncAttributes fa{};
        HIP_OK(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(brotlig_decode_kernel)));
        const int granules = (int)((fa.sharedSizeBytes + 1279u) / 1280u);
        if (granules > 0 && per_cu > 128 / granules) per_cu = 128 / granules;
        if (diag_wg_per_cu() > 0) per_cu = diag_wg_per_cu();
        if (per_cu < 1) per_cu = 1;
        // wavefronts of the de-conditioning kernel (one per workgroup, 4 KiB of LDS each): 8 per SIMD.  (BROTLIG_DC_PER_CU: a diagnostics
        // switch like the others -- honoured only in a process started with BROTLIG_ENABLE_DEBUG_KNOBS=1)
        const int dc_per_cu = diag_knobs_enabled() ? env_int_once("BROTLIG_DC_PER_CU") : -1;
        g.decond = cus * (dc_per_cu > 0 && dc_per_cu <= 64 ? dc_per_cu : 32);
        g.duo = cus * kDuoPerCu;
        g.sched = cus;                                                  // workgroups that walk the pages in the schedule kernel (count, scatter), kSchedThreads pages per step each
        g.decode = cus * per_cu < (int)kMaxDecodeGrid ? cus * per_cu : (int)kMaxDecodeGrid;
    }
    *out = g;
    return BROTLIG_OK;
}

// hipEvent_t that is destroyed on scope exit
struct Event {
    hipEvent_t e = nullptr;
    Event() = default;
    Event(const Event&) = delete;
    Event& operator=(const Event&) = delete;
    Event(Event&& o) noexcept : e(o.e) { o.e = nullptr; }
    ~Event() { if (e) (void)hipEventDestroy(e); }
    hipError_t create() { return hipEventCreate(&e); }
};

DecodeArgs make_args(const void* d_in, uint64_t in_bytes, void* d_out, uint64_t out_bytes,
                     const BrotligStreamDesc* d_streams, uint32_t n, void* d_ws, size_t ws_bytes, void* d_scratch)
{
    uint32_t* ws = static_cast<uint32_t*>(d_ws);
    DecodeArgs a{};
    a.in = static_cast<const uint8_t*>(d_in); a.in_bytes = in_bytes;
    a.out = static_cast<uint8_t*>(d_out); a.out_bytes = out_bytes;
    a.scratch = static_cast<uint8_t*>(d_scratch);
    a.streams = reinterpret_cast<const StreamDesc*>(d_streams); a.num_streams = n;
    a.status = ws; a.work_counter = ws + 1; a.sync = ws + kWsSyncAt; a.page_base = ws + kWsHeaderWords;
    a.dc = reinterpret_cast<DcTable*>(static_cast<uint8_t*>(d_ws) + dc_offset(n));
    a.far_syms = reinterpret_cast<uint16_t*>(static_cast<uint8_t*>(d_ws) + far_syms_offset(n));
    // The page schedule -- one job record per page the batch can hold -- whenever the workspace has the room (BrotligDecodeWorkspaceSizeFor).
    // What the schedule kernel writes into it is schedule_mode's choice (brotlig_jobs.h; enqueue() fills in the wavefront count): the
    // schedule proper from kOrderMinOutBytes of 64 KiB pages on, folded for a batch of more pages than wavefronts and at most twice as
    // many, page order otherwise.  Without it (a workspace of the minimum size) the page kernels take pages in stream order and walk the
    // page tables themselves.
    const size_t base = jobs_offset(n);
    const uint64_t cap = max_pages(n, out_bytes);
    if (ws_bytes >= base && (ws_bytes - base) / sizeof(JobRecord) >= cap && cap <= 0xFFFFFFFFull) {
        a.jobs = reinterpret_cast<JobRecord*>(static_cast<uint8_t*>(d_ws) + base);
        a.jobs_cap = (uint32_t)cap;
    }
    a.order_from_k = (uint16_t)(kOrderMinOutBytes >> 26);                // 768 MiB of 64 KiB pages = 12 x 1 024 pages
    return a;
}

// The one launch in front of the page kernels (brotlig_schedule.h): stream headers -> page prefix -> job records in schedule order ->
// pairing policy.  A small batch (up to 64 streams, up to 2 048 pages) takes one workgroup that runs the phases one after the other;
// anything else one workgroup per item of every phase, up to `g.sched` of them walking the pages (kSchedThreads pages per step each).
constexpr uint64_t kSchedSmallPages = 2048;
void launch_schedule(DecodeArgs& a, const Grids& g, hipStream_t s)
{
    a.launch_tag = next_launch_tag();
    const uint64_t pages = max_pages(a.num_streams, a.out_bytes);
    unsigned grid = 1u;
    if (a.num_streams > 64u || pages > kSchedSmallPages) {
        // workgroups that walk the pages, kSchedThreads of them per step: as many as the batch has such groups of pages if its pages are the
        // usual 64 KiB (with 32 KiB pages every workgroup takes two steps) -- the device starts a workgroup of this kernel every ~50 ns, and
        // those that come late and find nothing to do are waited for all the same
        const uint64_t groups = (a.out_bytes / 65536u + a.num_streams + kSchedThreads - 1u) / kSchedThreads;
        grid = sched_grid(a.num_streams, (uint32_t)(groups < (uint64_t)g.sched ? groups : (uint64_t)g.sched));
    }
    hipLaunchKernelGGL(brotlig_schedule_kernel, dim3(grid), dim3(kSchedThreads), 0, s, a);
}

// prepare (page counts -> prefix) then the persistent page-decode kernel; k0/k1 bracket the latter
// wavefronts of brotlig_decode_kernel for this batch: a batch that cannot hold more pages than that needs no more (every page is at least
// 32 KiB of the caller's output region); the diagnostics knob pins it
unsigned classic_grid(const DecodeArgs& a, const Grids& g)
{
    const uint64_t bound = max_pages(a.num_streams, a.out_bytes);
    unsigned grid = bound < (uint64_t)g.decode ? (unsigned)(bound ? bound : 1u) : (unsigned)g.decode;
    const uint32_t forced = g_debug_grid.load();
    if (forced) grid = forced < (unsigned)g.decode ? forced : (unsigned)g.decode;
    return grid;
}

BROTLIG_ERROR enqueue(const DecodeArgs& args, hipStream_t s, hipEvent_t k0, hipEvent_t k1)
{
    Grids g;
    if (BROTLIG_ERROR e = grid_sizes(&g)) return e;
    DecodeArgs a = args;
    a.decode_waves = (uint16_t)std::min(classic_grid(a, g), 65535u);    // (schedule_mode: a batch of up to twice as many pages takes them folded)
    // the pairing policy only matters when two pages can meet in a wavefront: a batch that cannot hold more pages than the grid has
    // wavefronts (every page >= 32 KiB of the output region) decodes one page per wavefront
    a.may_pair = (max_pages(a.num_streams, a.out_bytes) > (uint64_t)g.decode || g_debug_grid.load() != 0u) ? 1u : 0u;
    launch_schedule(a, g, s);
    if (diag_policy() >= 0)                             // diagnostics: pin the pairing policy (quarters of a page a free half waits)
        HIP_OK(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(a.status + 3), diag_policy(), 1, s));
    if (k0) HIP_OK(hipEventRecord(k0, s));
    {   // a batch that cannot hold more pages than that needs no more wavefronts (every page is at least 32 KiB of the caller's output
        // region): a single asset launches a handful of workgroups instead of 4 096.  (The kernel itself sends home every wavefront beyond
        // the batch's page count before it touches the page counter -- that is what takes 45 us off a single page; round 4.)
        // Which kernel: up to kDuoMaxPages pages two wavefronts per page (brotlig_decode_duo_kernel), more than that one wavefront per
        // one or two pages.  The host knows the output size, not the page count (32 .. 128 KiB each): where the size leaves both
        // possible (up to 256 MiB of output) both kernels are launched and the one the batch does not belong to leaves at once
        // (DecodeArgs::duo_limit against the page count the schedule kernel found) -- a few microseconds, and only for such batches.
        const uint64_t bound = max_pages(a.num_streams, a.out_bytes);
        const uint32_t mode = g_debug_mode.load(), forced = g_debug_grid.load();
        DecodeArgs b = a;
        b.duo_limit = (mode == 1u || forced != 0u) ? 0u : mode == 2u ? 0xFFFFFFFFu : (uint32_t)kDuoMaxPages;
        if (mode != 2u && a.out_bytes / kMaxPageSize > kDuoMaxPages) b.duo_limit = 0u;     // an output this large is taken for more pages than that
        const bool duo = b.duo_limit != 0u;
        const bool classic = !duo || (mode != 2u && bound > kDuoMaxPages);
        if (duo) {
            const unsigned grid = bound < (uint64_t)g.duo ? (unsigned)(bound ? bound : 1u) : (unsigned)g.duo;
            hipLaunchKernelGGL(brotlig_decode_duo_kernel, dim3(grid), dim3(128), 0, s, b);
        }
        if (classic) {
            hipLaunchKernelGGL(brotlig_decode_kernel, dim3(classic_grid(a, g)), dim3(64), 0, s, b);
        }
    }
    if (k1) HIP_OK(hipEventRecord(k1, s));
    if (a.scratch != nullptr) {   // (without a scratch buffer no stream of the batch can be pre-conditioned: the schedule kernel rejects them)
        // The batch's super-tiles (2 x 128 blocks: 2 or 4 KiB of texture) are one list, cut evenly over the wavefronts of this launch, 32 per CU
        // at most; a small batch does not need them all (the count is an estimate -- the kernel divides whatever list it finds by the grid).
        const uint64_t want = a.out_bytes / 4096u + a.num_streams;
        const unsigned grid = want < (uint64_t)g.decond ? (unsigned)want : (unsigned)g.decond;
        hipLaunchKernelGGL(brotlig_decondition_kernel, dim3(grid ? grid : 1u), dim3(64), 0, s, a);
    }
    HIP_OK(hipGetLastError());
    return BROTLIG_OK;
}

BROTLIG_ERROR status_to_error(uint32_t st)
{
    if (st & kStatusBadHeader) return BROTLIG_ERROR_CORRUPT_STREAM;
    if (st & kStatusBadPage) return BROTLIG_ERROR_GENERIC;
    return BROTLIG_OK;
}

struct DevBuf {             // hipFree on scope exit
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
};

}  // namespace

extern "C" uint32_t DecompressedSize(uint8_t* src)
{
    uint32_t w0, w1;
    memcpy(&w0, src, 4); memcpy(&w1, src + 4, 4);
    StreamInfo si;
    parse_stream_header(w0, w1, si);
    return uncompressed_size(si);
}

extern "C" size_t BrotligDecodeWorkspaceSize(uint32_t num_streams) { return workspace_bytes(num_streams); }
extern "C" size_t BrotligDecodeWorkspaceSizeFor(uint32_t num_streams, uint64_t out_bytes)
{
    return jobs_offset(num_streams) + sizeof(JobRecord) * (size_t)max_pages(num_streams, out_bytes);
}

extern "C" BROTLIG_ERROR BrotligDecodeBatchDevice(const void* d_in, uint64_t in_bytes, void* d_out, uint64_t out_bytes,
                                                  const BrotligStreamDesc* d_streams, uint32_t num_streams,
                                                  void* d_workspace, size_t ws_bytes, void* d_scratch, void* hip_stream)
{
    if (!d_in || !d_out || !d_streams || !d_workspace || num_streams == 0) return BROTLIG_ERROR_GENERIC;
    if (ws_bytes < workspace_bytes(num_streams)) return BROTLIG_ERROR_GENERIC;
    const DecodeArgs a = make_args(d_in, in_bytes, d_out, out_bytes, d_streams, num_streams, d_workspace, ws_bytes, d_scratch);
    return enqueue(a, static_cast<hipStream_t>(hip_stream), nullptr, nullptr);
}

extern "C" BROTLIG_ERROR BrotligDecodeBatchStatus(const void* d_workspace, void* hip_stream)
{
    if (!d_workspace) return BROTLIG_ERROR_GENERIC;
    uint32_t st = 0;
    HIP_OK(hipMemcpyAsync(&st, d_workspace, sizeof st, hipMemcpyDeviceToHost, static_cast<hipStream_t>(hip_stream)));
    HIP_OK(hipStreamSynchronize(static_cast<hipStream_t>(hip_stream)));
    return status_to_error(st);
}

// The stream's own status word lives in its DcTable record of the workspace (every stream has one); one strided copy brings all of them back.
hipError_t brotlig::enqueue_stream_status_copy(const void* d_workspace, uint32_t num_streams, uint32_t* h_words, hipStream_t stream)
{
    const uint8_t* first = static_cast<const uint8_t*>(d_workspace) + dc_offset(num_streams) + offsetof(DcTable, status);
    return hipMemcpy2DAsync(h_words, sizeof(uint32_t), first, sizeof(DcTable), sizeof(uint32_t), num_streams, hipMemcpyDeviceToHost, stream);
}

extern "C" BROTLIG_ERROR BrotligDecodeBatchStreamStatus(const void* d_workspace, uint32_t num_streams, int32_t* results, void* hip_stream)
{
    if (!d_workspace || !results || num_streams == 0) return BROTLIG_ERROR_GENERIC;
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    std::vector<uint32_t> words;
    try { words.resize(num_streams); } catch (...) { return BROTLIG_ERROR_GENERIC; }
    HIP_OK(enqueue_stream_status_copy(d_workspace, num_streams, words.data(), s));
    HIP_OK(hipStreamSynchronize(s));
    uint32_t all = 0;
    for (uint32_t i = 0; i < num_streams; ++i) { results[i] = (int32_t)status_to_error(words[i]); all |= words[i]; }
    return status_to_error(all);
}

extern "C" BROTLIG_ERROR BrotligDecodeBatchTimed(const void* d_in, uint64_t in_bytes, void* d_out, uint64_t out_bytes,
                                                 const BrotligStreamDesc* d_streams, uint32_t num_streams,
                                                 void* d_workspace, size_t ws_bytes, void* d_scratch, void* hip_stream,
                                                 uint32_t warmup, uint32_t steps, double* total_ms, double* decode_kernel_ms)
{
    if (!d_in || !d_out || !d_streams || !d_workspace || num_streams == 0 || steps == 0) return BROTLIG_ERROR_GENERIC;
    if (ws_bytes < workspace_bytes(num_streams)) return BROTLIG_ERROR_GENERIC;
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    const DecodeArgs a = make_args(d_in, in_bytes, d_out, out_bytes, d_streams, num_streams, d_workspace, ws_bytes, d_scratch);
    for (uint32_t i = 0; i < warmup; ++i) if (BROTLIG_ERROR e = enqueue(a, s, nullptr, nullptr)) return e;
    // The events are kept from call to call (per host thread and device; grow-only): creating 2 K + 2 of them costs as much host time as
    // enqueueing a step, and a caller that brackets this call with its own clock (bench.py: K = 5) would book it to the steps.
    struct Pool { int device = -1; std::vector<Event> ev; };
    thread_local Pool pool;
    int dev = 0;
    HIP_OK(hipGetDevice(&dev));
    if (pool.device != dev) { pool.ev.clear(); pool.device = dev; }
    const size_t need = std::max<size_t>(2 * (size_t)steps + 2, 66);     // (the first call of a thread pays for up to 32 steps' worth)
    try { while (pool.ev.size() < need) { pool.ev.emplace_back(); HIP_OK(pool.ev.back().create()); } } catch (const std::bad_alloc&) { return BROTLIG_ERROR_GENERIC; }
    std::vector<Event>& ev = pool.ev;
    HIP_OK(hipEventRecord(ev[2 * steps].e, s));
    for (uint32_t i = 0; i < steps; ++i) if (BROTLIG_ERROR e = enqueue(a, s, ev[2 * i].e, ev[2 * i + 1].e)) return e;
    HIP_OK(hipEventRecord(ev[2 * steps + 1].e, s));
    HIP_OK(hipStreamSynchronize(s));
    float ms = 0.f;
    HIP_OK(hipEventElapsedTime(&ms, ev[2 * steps].e, ev[2 * steps + 1].e));
    if (total_ms) *total_ms = ms;
    double ksum = 0.0;
    for (uint32_t i = 0; i < steps; ++i) { HIP_OK(hipEventElapsedTime(&ms, ev[2 * i].e, ev[2 * i + 1].e)); ksum += ms; }
    if (decode_kernel_ms) *decode_kernel_ms = ksum / steps;
    return BROTLIG_OK;
}

// ---- reusable decode context ("a reusable context object is allowed ... but must be optional") --------
// Device buffers that only grow, one stream, two events: DecodeGPU through a context costs two copies and the
// launches, not five hipMalloc / hipFree pairs (sample/BrotligGPUDecoder.cpp:260-748 rebuilds its whole D3D12 state per call).
struct BrotligContext {
    int device = 0;
    hipStream_t stream = nullptr;
    Event e0, e1;
    DevBuf in, out, scratch, ws, desc;
    BrotligStreamDesc* h_desc = nullptr;    // pinned: the descriptor of the call in flight (its upload needs no host-side wait)
    size_t in_cap = 0, out_cap = 0, scratch_cap = 0, ws_cap = 0;
    BrotligContext() = default;
    BrotligContext(const BrotligContext&) = delete;
    BrotligContext& operator=(const BrotligContext&) = delete;
    // every exit path of every owner releases the stream and the pinned descriptor (the device buffers and events release
    // themselves); called with the context's device current
    ~BrotligContext()
    {
        if (stream) { (void)hipStreamSynchronize(stream); (void)hipStreamDestroy(stream); }
        if (h_desc) (void)hipHostFree(h_desc);
    }
};

namespace {

BROTLIG_ERROR grow(DevBuf& b, size_t& cap, size_t need)
{
    if (need <= cap) return BROTLIG_OK;
    if (b.p) { (void)hipFree(b.p); b.p = nullptr; cap = 0; }
    const size_t want = need + need / 4u;                               // headroom: assets of similar size reuse it
    HIP_OK(hipMalloc(&b.p, want));
    cap = want;
    return BROTLIG_OK;
}

// DecodeGPU proper; `c` owns every device resource it needs
BROTLIG_ERROR decode_gpu(BrotligContext& c, uint32_t input_size, const uint8_t* input, uint32_t* output_size, uint8_t* output, double* time_ms)
{
    if (!input || !output || !output_size || input_size < 8) return BROTLIG_ERROR_CORRUPT_STREAM;
    uint32_t w0, w1;
    memcpy(&w0, input, 4); memcpy(&w1, input + 4, 4);
    StreamInfo si;
    // src/BrotligDecoder.cpp:437-446: magic first, then id
    if ((w0 & 0xFF) != (((w0 >> 8) & 0xFF) ^ 0xFF)) return BROTLIG_ERROR_CORRUPT_STREAM;
    if (!parse_stream_header(w0, w1, si)) return BROTLIG_ERROR_INCORRECT_STREAM_FORMAT;
    const uint32_t out_size = uncompressed_size(si);
    if (*output_size < out_size) return BROTLIG_ERROR_GENERIC;
    if ((uint64_t)si.header_bytes + 4ull * si.num_pages > input_size) return BROTLIG_ERROR_CORRUPT_STREAM;

    const uint64_t in_alloc = ((uint64_t)input_size + 15u) & ~15ull;
    const uint64_t out_alloc = (((uint64_t)si.num_pages * si.page_size) + 15u) & ~15ull;
    const size_t ws_size = BrotligDecodeWorkspaceSizeFor(1, out_alloc);
    if (BROTLIG_ERROR e = grow(c.in, c.in_cap, in_alloc + 64)) return e;
    if (BROTLIG_ERROR e = grow(c.out, c.out_cap, out_alloc + 64)) return e;         // copies read up to 7 bytes past a page
    if (si.preconditioned) if (BROTLIG_ERROR e = grow(c.scratch, c.scratch_cap, out_alloc + 64)) return e;
    if (BROTLIG_ERROR e = grow(c.ws, c.ws_cap, ws_size)) return e;
    if (!c.desc.p) HIP_OK(hipMalloc(&c.desc.p, sizeof(BrotligStreamDesc)));
    HIP_OK(hipMemsetAsync(static_cast<uint8_t*>(c.in.p) + (in_alloc + 64 - 80), 0, 80, c.stream));
    HIP_OK(hipMemcpyAsync(c.in.p, input, input_size, hipMemcpyHostToDevice, c.stream));
    const BrotligStreamDesc desc{0, 0, input_size, *output_size};
    if (c.h_desc) {     // a kept context: pinned and owned by it, no wait between the upload and the launches
        *c.h_desc = desc;
        HIP_OK(hipMemcpyAsync(c.desc.p, c.h_desc, sizeof desc, hipMemcpyHostToDevice, c.stream));
    } else {            // the stateless entry: from the stack, the copy is complete when the call returns (no pinned allocation per call)
        HIP_OK(hipMemcpyAsync(c.desc.p, &desc, sizeof desc, hipMemcpyHostToDevice, c.stream));
        HIP_OK(hipStreamSynchronize(c.stream));
    }
    const DecodeArgs a = make_args(c.in.p, input_size, c.out.p, out_alloc, static_cast<BrotligStreamDesc*>(c.desc.p), 1,
                                   c.ws.p, ws_size, si.preconditioned ? c.scratch.p : nullptr);
    BROTLIG_ERROR err = enqueue(a, c.stream, c.e0.e, c.e1.e);
    if (err == BROTLIG_OK) err = BrotligDecodeBatchStatus(c.ws.p, c.stream);
    float ms = 0.f;
    if (err == BROTLIG_OK && hipEventElapsedTime(&ms, c.e0.e, c.e1.e) != hipSuccess) err = BROTLIG_ERROR_GENERIC;
    if (err != BROTLIG_OK) return err;
    HIP_OK(hipMemcpyAsync(output, c.out.p, out_size, hipMemcpyDeviceToHost, c.stream));
    HIP_OK(hipStreamSynchronize(c.stream));
    *output_size = out_size;                                            // src/BrotligDecoder.cpp:490
    if (time_ms) *time_ms = ms;
    return BROTLIG_OK;
}

BROTLIG_ERROR context_init(BrotligContext& c, int device, bool kept)
{
    if (device < 0) HIP_OK(hipGetDevice(&device)); else HIP_OK(hipSetDevice(device));
    c.device = device;
    HIP_OK(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
    HIP_OK(c.e0.create()); HIP_OK(c.e1.create());
    if (kept) HIP_OK(hipHostMalloc(reinterpret_cast<void**>(&c.h_desc), sizeof(BrotligStreamDesc), hipHostMallocDefault));
    return BROTLIG_OK;
}

// the calling thread's current device, put back on scope exit
struct DeviceGuard {
    int prev = -1;
    DeviceGuard() { if (hipGetDevice(&prev) != hipSuccess) prev = -1; }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

}  // namespace

extern "C" BROTLIG_ERROR BrotligContextCreate(int device, BrotligContext** out)
{
    if (!out) return BROTLIG_ERROR_GENERIC;
    *out = nullptr;
    BrotligContext* c = new (std::nothrow) BrotligContext;
    if (!c) return BROTLIG_ERROR_GENERIC;
    DeviceGuard guard;
    if (BROTLIG_ERROR e = context_init(*c, device, true)) { delete c; return e; }
    *out = c;
    return BROTLIG_OK;
}

extern "C" void BrotligContextDestroy(BrotligContext* c)
{
    if (!c) return;
    DeviceGuard guard;
    (void)hipSetDevice(c->device);
    delete c;
}

extern "C" BROTLIG_ERROR BrotligContextDecodeGPU(BrotligContext* c, uint32_t input_size, const uint8_t* input,
                                                 uint32_t* output_size, uint8_t* output, double* time_ms)
{
    if (!c) return BROTLIG_ERROR_GENERIC;
    DeviceGuard guard;                                                  // the caller's current device is restored
    HIP_OK(hipSetDevice(c->device));
    return decode_gpu(*c, input_size, input, output_size, output, time_ms);
}

// The reference's entry (sample/BrotligGPUDecoder.h:24): stateless -- a context for the length of the call.
extern "C" BROTLIG_ERROR DecodeGPU(int /*useWarpDevice*/, uint32_t input_size, const uint8_t* input,
                                   uint32_t* output_size, uint8_t* output, double* time_ms)
{
    BrotligContext c;                                                   // (its destructor releases the stream on every path)
    if (BROTLIG_ERROR e = context_init(c, -1, false)) return e;
    return decode_gpu(c, input_size, input, output_size, output, time_ms);
}

// ---- multi-device fan-out (SURVEY.md 8(b) row 4 "plus a multi-GPU wrapper", 8(e)) ------------------------------------
// The plan itself is host arithmetic, one definition for every library that exports it: brotlig_shard_plan.h.
extern "C" BROTLIG_ERROR BrotligShardPlan(const uint64_t* in_sizes, uint32_t num_streams, uint32_t num_shards, uint32_t* first)
{
    try { return shard_plan(in_sizes, num_streams, num_shards, first) ? BROTLIG_OK : BROTLIG_ERROR_GENERIC; }
    catch (...) { return BROTLIG_ERROR_GENERIC; }
}

namespace {

// rendezvous of the per-device host threads (C++17 has no std::barrier)
struct Rendezvous {
    std::mutex m; std::condition_variable cv; uint32_t waiting = 0, generation = 0; const uint32_t n;
    bool disabled = false;          // set when not every member could be started: nobody waits any more
    explicit Rendezvous(uint32_t count) : n(count) {}
    void arrive()
    {
        std::unique_lock<std::mutex> lock(m);
        if (disabled) return;
        const uint32_t gen = generation;
        if (++waiting == n) { waiting = 0; ++generation; cv.notify_all(); }
        else cv.wait(lock, [&] { return generation != gen || disabled; });
    }
    void disable() { std::lock_guard<std::mutex> lock(m); disabled = true; cv.notify_all(); }
};

void run_shard(BrotligDeviceBatch* b, uint32_t warmup, uint32_t steps, Rendezvous* rv)
{
    b->kernel_ms = 0.0; b->wall_ms = 0.0;
    BROTLIG_ERROR err = BROTLIG_OK;
    hipStream_t s = static_cast<hipStream_t>(b->hip_stream);
    std::vector<Event> ev;
    DecodeArgs a{};
    do {
        try { ev = std::vector<Event>(2 * (size_t)steps); }            // (nothing may leave a thread function: it would terminate the host)
        catch (...) { err = BROTLIG_ERROR_GENERIC; break; }
        if (!b->d_in || !b->d_out || !b->d_streams || !b->d_workspace || b->num_streams == 0u ||
            b->workspace_bytes < workspace_bytes(b->num_streams)) { err = BROTLIG_ERROR_GENERIC; break; }
        if (hipSetDevice(b->device) != hipSuccess) { err = BROTLIG_ERROR_GENERIC; break; }
        a = make_args(b->d_in, b->in_bytes, b->d_out, b->out_bytes, b->d_streams, b->num_streams, b->d_workspace,
                      (size_t)b->workspace_bytes, b->d_scratch);
        for (auto& e : ev) if (e.create() != hipSuccess) err = BROTLIG_ERROR_GENERIC;
        for (uint32_t i = 0; i < warmup && err == BROTLIG_OK; ++i) err = enqueue(a, s, nullptr, nullptr);
        if (err == BROTLIG_OK && hipStreamSynchronize(s) != hipSuccess) err = BROTLIG_ERROR_GENERIC;
    } while (false);
    rv->arrive();                                                       // every device starts its timed passes together
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t i = 0; i < steps && err == BROTLIG_OK; ++i) err = enqueue(a, s, ev[2 * i].e, ev[2 * i + 1].e);
    if (err == BROTLIG_OK) err = BrotligDecodeBatchStatus(b->d_workspace, s);      // waits for the stream
    b->wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (err == BROTLIG_OK) {
        double sum = 0.0;
        for (uint32_t i = 0; i < steps; ++i) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, ev[2 * i].e, ev[2 * i + 1].e) != hipSuccess) err = BROTLIG_ERROR_GENERIC;
            sum += ms;
        }
        b->kernel_ms = steps ? sum / steps : 0.0;
    }
    b->result = (int32_t)err;
    rv->arrive();
}

}  // namespace

extern "C" BROTLIG_ERROR BrotligDecodeBatchMultiDevice(BrotligDeviceBatch* shards, uint32_t num_shards, uint32_t struct_bytes,
                                                       uint32_t warmup, uint32_t steps, double* max_kernel_ms, double* max_wall_ms)
{
    if (!shards || num_shards == 0u || num_shards > 64u || steps == 0u || struct_bytes != sizeof(BrotligDeviceBatch)) return BROTLIG_ERROR_GENERIC;
    DeviceGuard guard;                                                  // the calling thread's current device is restored
    try {
        Rendezvous rv(num_shards);
        std::vector<std::thread> pool;
        pool.reserve(num_shards);
        uint32_t started = 0;
        try {
            for (uint32_t g = 1; g < num_shards; ++g) { pool.emplace_back(run_shard, shards + g, warmup, steps, &rv); ++started; }
        } catch (const std::system_error&) {
            rv.disable();           // a rendezvous short of members would never open: the shards without a thread of their
                                    // own run on this thread, in turn, and nobody waits for anybody
        }
        run_shard(shards, warmup, steps, &rv);                          // shard 0 on the calling thread
        for (uint32_t g = started + 1u; g < num_shards; ++g) run_shard(shards + g, warmup, steps, &rv);
        for (auto& t : pool) t.join();
    } catch (...) { return BROTLIG_ERROR_GENERIC; }
    BROTLIG_ERROR first_err = BROTLIG_OK;
    double mk = 0.0, mw = 0.0;
    for (uint32_t g = 0; g < num_shards; ++g) {
        if (shards[g].result != BROTLIG_OK && first_err == BROTLIG_OK) first_err = (BROTLIG_ERROR)shards[g].result;
        mk = shards[g].kernel_ms > mk ? shards[g].kernel_ms : mk;
        mw = shards[g].wall_ms > mw ? shards[g].wall_ms : mw;
    }
    if (max_kernel_ms) *max_kernel_ms = mk;
    if (max_wall_ms) *max_wall_ms = mw;
    return first_err;
}

// Non-blocking form: enqueue every shard on its own device and stream and return; nothing is timed, no host thread is
// created.  For a caller that already owns its threads / streams (one per device) and overlaps the decode with its own work.
extern "C" BROTLIG_ERROR BrotligDecodeBatchMultiDeviceAsync(BrotligDeviceBatch* shards, uint32_t num_shards, uint32_t struct_bytes)
{
    if (!shards || num_shards == 0u || num_shards > 64u || struct_bytes != sizeof(BrotligDeviceBatch)) return BROTLIG_ERROR_GENERIC;
    DeviceGuard guard;
    BROTLIG_ERROR first_err = BROTLIG_OK;
    for (uint32_t g = 0; g < num_shards; ++g) {
        BrotligDeviceBatch* b = shards + g;
        BROTLIG_ERROR err = BROTLIG_OK;
        b->kernel_ms = 0.0; b->wall_ms = 0.0;
        if (!b->d_in || !b->d_out || !b->d_streams || !b->d_workspace || b->num_streams == 0u ||
            b->workspace_bytes < workspace_bytes(b->num_streams) || hipSetDevice(b->device) != hipSuccess) err = BROTLIG_ERROR_GENERIC;
        else err = enqueue(make_args(b->d_in, b->in_bytes, b->d_out, b->out_bytes, b->d_streams, b->num_streams, b->d_workspace,
                                     (size_t)b->workspace_bytes, b->d_scratch), static_cast<hipStream_t>(b->hip_stream), nullptr, nullptr);
        b->result = (int32_t)err;
        if (err != BROTLIG_OK && first_err == BROTLIG_OK) first_err = err;
    }
    return first_err;
}

// Completes what BrotligDecodeBatchMultiDeviceAsync started: waits for every shard's stream and collects the batch status.
extern "C" BROTLIG_ERROR BrotligDecodeBatchMultiDeviceWait(BrotligDeviceBatch* shards, uint32_t num_shards, uint32_t struct_bytes)
{
    if (!shards || num_shards == 0u || num_shards > 64u || struct_bytes != sizeof(BrotligDeviceBatch)) return BROTLIG_ERROR_GENERIC;
    DeviceGuard guard;
    BROTLIG_ERROR first_err = BROTLIG_OK;
    for (uint32_t g = 0; g < num_shards; ++g) {
        BrotligDeviceBatch* b = shards + g;
        if (b->result == BROTLIG_OK) {
            if (hipSetDevice(b->device) != hipSuccess) b->result = BROTLIG_ERROR_GENERIC;
            else b->result = (int32_t)BrotligDecodeBatchStatus(b->d_workspace, b->hip_stream);
        }
        if (b->result != BROTLIG_OK && first_err == BROTLIG_OK) first_err = (BROTLIG_ERROR)b->result;
    }
    return first_err;
}

static uint32_t mx_other(const uint32_t* v, uint32_t base)     // maximum over the other half's 32 values
{
    uint32_t m = 0;
    for (uint32_t l = base ^ 32u; l < (base ^ 32u) + 32u; ++l) m = v[l] > m ? v[l] : m;
    return m;
}

extern "C" BROTLIG_ERROR BrotligDeviceSelfTest(void)
{
    DevBuf d;
    HIP_OK(hipMalloc(&d.p, 512 * sizeof(uint32_t)));
    hipLaunchKernelGGL(brotlig_selftest_kernel, dim3(1), dim3(64), 0, nullptr, static_cast<uint32_t*>(d.p));
    uint32_t h[512];
    HIP_OK(hipMemcpy(h, d.p, sizeof h, hipMemcpyDeviceToHost));
    const uint32_t* v = h + 320;
    for (uint32_t lane = 0; lane < 64; ++lane) {
        const uint32_t base = lane & 32u;
        uint32_t sum = 0, mx = 0, bal = 0;
        for (uint32_t l = base; l < base + 32; ++l) {
            if (l <= lane) sum += v[l];
            mx = v[l] > mx ? v[l] : mx;
            if (v[l] & 1u) bal |= 1u << (l - base);
        }
        if (h[lane] != sum || h[64 + lane] != sum || h[128 + lane] != bal ||
            h[192 + lane] != v[base | ((lane * 7u + 3u) & 31u)] || h[256 + lane] != mx ||
            h[384 + lane] != v[base | (base ? 5u : 29u)] || h[448 + lane] != mx_other(v, base)) {
            fprintf(stderr, "brotlig_hip: wave primitive self-test failed at lane %u\n", lane);
            return BROTLIG_ERROR_GENERIC;
        }
    }
    return BROTLIG_OK;
}

extern "C" BROTLIG_ERROR BrotligDecodePhaseProfile(const void* d_in, uint64_t in_bytes, void* d_out, uint64_t out_bytes,
                                                   const BrotligStreamDesc* d_streams, uint32_t num_streams,
                                                   void* d_workspace, size_t ws_bytes, void* d_scratch,
                                                   uint64_t* cycles_out, uint32_t n_out)
{
    if (!d_in || !d_out || !d_streams || !d_workspace || num_streams == 0 || !cycles_out) return BROTLIG_ERROR_GENERIC;
    if (ws_bytes < workspace_bytes(num_streams)) return BROTLIG_ERROR_GENERIC;
    Grids g;
    if (BROTLIG_ERROR e = grid_sizes(&g)) return e;
    DevBuf prof;
    // phase sums, then {first, last} 100 MHz tick of every wavefront, then {came, ticket, phase open, done} of the schedule kernel's tickets
    const size_t prof_words = (size_t)kNumPhases + 2u * (size_t)g.decode + 4u * (size_t)kSchedTimedTickets;
    HIP_OK(hipMalloc(&prof.p, prof_words * sizeof(unsigned long long)));
    HIP_OK(hipMemset(prof.p, 0, prof_words * sizeof(unsigned long long)));
    DecodeArgs a = make_args(d_in, in_bytes, d_out, out_bytes, d_streams, num_streams, d_workspace, ws_bytes, d_scratch);
    a.prof = static_cast<unsigned long long*>(prof.p);
    a.decode_waves = (uint16_t)std::min(g.decode, 65535);
    a.may_pair = 1u;
    launch_schedule(a, g, nullptr);
    if (BROTLIG_WAVE_TIMES) hipLaunchKernelGGL(brotlig_decode_kernel, dim3(g.decode), dim3(64), 0, nullptr, a);     // (diagnostics build: the product kernel with wave times)
    else hipLaunchKernelGGL(brotlig_decode_kernel_timed, dim3(g.decode), dim3(64), 0, nullptr, a);
    HIP_OK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(prof_words);
    HIP_OK(hipMemcpy(h.data(), prof.p, prof_words * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < n_out; ++i) cycles_out[i] = i < prof_words ? h[i] : 0;
    return BROTLIG_OK;
}

extern "C" void BrotligDebugSetDecodeGrid(uint32_t workgroups) { g_debug_grid.store(workgroups); }
extern "C" void BrotligDebugSetDecodeMode(uint32_t mode) { g_debug_mode.store(mode); }
extern "C" uint32_t BrotligDebugKnobsEnabled(void) { return diag_knobs_enabled() ? 1u : 0u; }
extern "C" uint32_t BrotligAbiVersion(void) { return BROTLIG_AMD_ABI_VERSION; }
extern "C" uint32_t BrotligKernelLdsBytes(void) { return (uint32_t)sizeof(WaveLds); }
extern "C" uint32_t BrotligKernelGridSize(void) { Grids g; return grid_sizes(&g) == BROTLIG_OK ? (uint32_t)g.decode : 0u; }

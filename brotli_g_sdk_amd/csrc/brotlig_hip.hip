// brotlig_hip.hip -- C ABI (include/brotlig_amd.h) over the gfx950 decode kernels.
//
// Host side of the drop-in boundary: replaces sample/BrotligGPUDecoder.cpp:260-748 (D3D12 device,
// queue, PSO, upload/readback heaps, timestamp queries) with HIP runtime calls, and exposes the
// shader's buffer-level contract (input / meta / output, BrotliGCompute.hlsl:93-95) as a
// device-pointer batch call.  No decoding happens on the host.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "brotlig_amd.h"
#include "brotlig_kernels.h"

using namespace brotlig;

namespace {

static_assert(sizeof(BrotligStreamDesc) == sizeof(StreamDesc), "descriptor layout");

#define HIP_OK(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { \
    fprintf(stderr, "brotlig_hip: %s failed: %s\n", #expr, hipGetErrorString(_e)); return BROTLIG_ERROR_GENERIC; } } while (0)

// Device workspace (the reference's `meta` buffer): word 0 status, word 1 page counter, word 2
// preconditioned-stream count, word 3 pairing policy, words 8..39 scheduling buckets, words 64..
// page_base[num_streams + 1], then (1 KiB aligned) one DcTable per stream, then -- if the caller's
// workspace has the room -- the page schedule (one word per page).
constexpr size_t kWsHeaderWords = 64;
size_t dc_offset(uint32_t n) { return ((kWsHeaderWords + (size_t)n + 1u) * 4u + 1023u) & ~(size_t)1023u; }
// per-half slots for the prefix-code symbols that overflow the LDS arrays, for every workgroup of the largest decode grid
constexpr uint32_t kMaxDecodeGrid = 4096;
constexpr size_t kFarSymBytes = (size_t)kMaxDecodeGrid * 2u * kFarSymStride * sizeof(uint16_t);
size_t far_syms_offset(uint32_t n) { return (dc_offset(n) + (size_t)n * sizeof(DcTable) + 255u) & ~(size_t)255u; }
size_t workspace_bytes(uint32_t n) { return far_syms_offset(n) + kFarSymBytes; }
// every page is at least 32 KiB of output, and every stream's output region is whole pages
uint64_t max_pages(uint32_t n, uint64_t out_bytes) { return out_bytes / kMinPageSize + n; }
// Below this many pages the schedule is not worth its two extra launches (about two pages per half-wave).
constexpr uint64_t kOrderMinOutBytes = 768ull << 20;

// Launch geometry per device (CU count x occupancy of the decode kernel), looked up once per device;
// host threads driving different devices (or the same one) may arrive here concurrently.
struct Grids { int decode = 0, decond = 1024, order = 1024; };
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
        if (const char* e = getenv("BROTLIG_WG_PER_CU")) per_cu = atoi(e);      // diagnostics (profiles/tools/occ_probe.sh)
        if (per_cu < 1) per_cu = 1;
        g.decond = cus * 8;
        g.order = cus * 4;
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
    a.status = ws; a.work_counter = ws + 1; a.page_base = ws + kWsHeaderWords;
    a.dc = reinterpret_cast<DcTable*>(static_cast<uint8_t*>(d_ws) + dc_offset(n));
    a.far_syms = reinterpret_cast<uint16_t*>(static_cast<uint8_t*>(d_ws) + far_syms_offset(n));
    const size_t base = workspace_bytes(n);
    const uint64_t room = ws_bytes > base ? (ws_bytes - base) / 4u : 0u;
    if (out_bytes >= kOrderMinOutBytes && room >= max_pages(n, out_bytes)) {
        a.order = reinterpret_cast<uint32_t*>(static_cast<uint8_t*>(d_ws) + base);
        a.order_cap = (uint32_t)(room > 0xFFFFFFFFull ? 0xFFFFFFFFull : room);
    }
    return a;
}

// prepare (page counts -> prefix) then the persistent page-decode kernel; k0/k1 bracket the latter
BROTLIG_ERROR enqueue(const DecodeArgs& a, hipStream_t s, hipEvent_t k0, hipEvent_t k1)
{
    Grids g;
    if (BROTLIG_ERROR e = grid_sizes(&g)) return e;
    HIP_OK(hipMemsetAsync(a.status, 0, kWsHeaderWords * sizeof(uint32_t), s));
    hipLaunchKernelGGL(brotlig_prepare_kernel, dim3(1), dim3(64), 0, s, a);
    if (a.order) {                                                      // page schedule: count, then scatter
        hipLaunchKernelGGL(brotlig_order_count_kernel, dim3(g.order), dim3(64), 0, s, a);
        hipLaunchKernelGGL(brotlig_order_scatter_kernel, dim3(g.order), dim3(64), 0, s, a);
    }
    hipLaunchKernelGGL(brotlig_policy_kernel, dim3(1), dim3(64), 0, s, a);
    if (const char* e = getenv("BROTLIG_POLICY"))       // diagnostics: pin the pairing policy (quarters of a page a free half waits)
        HIP_OK(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(a.status + 3), atoi(e), 1, s));
    if (k0) HIP_OK(hipEventRecord(k0, s));
    hipLaunchKernelGGL(brotlig_decode_kernel, dim3(g.decode), dim3(64), 0, s, a);
    if (k1) HIP_OK(hipEventRecord(k1, s));
    {   // streams over y, each stream's tiles over x; about 8 workgroups of 256 per CU in total
        const unsigned gy = a.num_streams < 32u ? a.num_streams : 32u;
        const unsigned gx = ((unsigned)g.decond + gy - 1u) / gy;
        hipLaunchKernelGGL(brotlig_decondition_kernel, dim3(gx, gy), dim3(256), 0, s, a);
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
    return workspace_bytes(num_streams) + (size_t)(4u * max_pages(num_streams, out_bytes));
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
    std::vector<Event> ev(2 * (size_t)steps + 2);
    for (auto& e : ev) HIP_OK(e.create());
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

extern "C" BROTLIG_ERROR DecodeGPU(int /*useWarpDevice*/, uint32_t input_size, const uint8_t* input,
                                   uint32_t* output_size, uint8_t* output, double* time_ms)
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
    DevBuf d_in, d_out, d_scratch, d_ws, d_desc;
    HIP_OK(hipMalloc(&d_in.p, in_alloc + 64));
    HIP_OK(hipMalloc(&d_out.p, out_alloc + 64));                        // copies read up to 7 bytes past a page
    if (si.preconditioned) HIP_OK(hipMalloc(&d_scratch.p, out_alloc + 64));
    const size_t ws_size = BrotligDecodeWorkspaceSizeFor(1, out_alloc);
    HIP_OK(hipMalloc(&d_ws.p, ws_size));
    HIP_OK(hipMalloc(&d_desc.p, sizeof(BrotligStreamDesc)));
    const BrotligStreamDesc desc{0, 0, input_size, *output_size};
    HIP_OK(hipMemset(static_cast<uint8_t*>(d_in.p) + (in_alloc + 64 - 80), 0, 80));
    HIP_OK(hipMemcpy(d_in.p, input, input_size, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_desc.p, &desc, sizeof desc, hipMemcpyHostToDevice));

    Event e0, e1;
    HIP_OK(e0.create()); HIP_OK(e1.create());
    const DecodeArgs a = make_args(d_in.p, input_size, d_out.p, out_alloc, static_cast<BrotligStreamDesc*>(d_desc.p), 1,
                                   d_ws.p, ws_size, d_scratch.p);
    BROTLIG_ERROR err = enqueue(a, nullptr, e0.e, e1.e);
    if (err == BROTLIG_OK) err = BrotligDecodeBatchStatus(d_ws.p, nullptr);
    float ms = 0.f;
    if (err == BROTLIG_OK && hipEventElapsedTime(&ms, e0.e, e1.e) != hipSuccess) err = BROTLIG_ERROR_GENERIC;
    if (err != BROTLIG_OK) return err;
    HIP_OK(hipMemcpy(output, d_out.p, out_size, hipMemcpyDeviceToHost));
    *output_size = out_size;                                            // src/BrotligDecoder.cpp:490
    if (time_ms) *time_ms = ms;
    return BROTLIG_OK;
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
    HIP_OK(hipMalloc(&prof.p, kNumPhases * sizeof(unsigned long long)));
    HIP_OK(hipMemset(prof.p, 0, kNumPhases * sizeof(unsigned long long)));
    DecodeArgs a = make_args(d_in, in_bytes, d_out, out_bytes, d_streams, num_streams, d_workspace, ws_bytes, d_scratch);
    a.prof = static_cast<unsigned long long*>(prof.p);
    HIP_OK(hipMemsetAsync(a.status, 0, kWsHeaderWords * sizeof(uint32_t), nullptr));
    hipLaunchKernelGGL(brotlig_prepare_kernel, dim3(1), dim3(64), 0, nullptr, a);
    if (a.order) {
        hipLaunchKernelGGL(brotlig_order_count_kernel, dim3(g.order), dim3(64), 0, nullptr, a);
        hipLaunchKernelGGL(brotlig_order_scatter_kernel, dim3(g.order), dim3(64), 0, nullptr, a);
    }
    hipLaunchKernelGGL(brotlig_policy_kernel, dim3(1), dim3(64), 0, nullptr, a);
    hipLaunchKernelGGL(brotlig_decode_kernel_timed, dim3(g.decode), dim3(64), 0, nullptr, a);
    HIP_OK(hipDeviceSynchronize());
    unsigned long long h[kNumPhases];
    HIP_OK(hipMemcpy(h, prof.p, sizeof h, hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < n_out; ++i) cycles_out[i] = i < (uint32_t)kNumPhases ? h[i] : 0;
    return BROTLIG_OK;
}

extern "C" uint32_t BrotligKernelLdsBytes(void) { return (uint32_t)sizeof(WaveLds); }
extern "C" uint32_t BrotligKernelGridSize(void) { Grids g; return grid_sizes(&g) == BROTLIG_OK ? (uint32_t)g.decode : 0u; }

// traffic_calib.hip -- what do FETCH_SIZE / WRITE_SIZE / TCC_* report for the access patterns of brotlig_decode_kernel?
// (VERDICT r2 "calibrate and attribute the 15.6x fetch"; MI355X_MICROARCH.md, HBM section: "calibrate on a known byte count
// in your own access pattern before trusting an absolute".)  Each kernel moves a KNOWN number of bytes over a buffer far
// larger than L2 (32 MiB) and the Infinity Cache (256 MiB); run under rocprofv3 --pmc in separate passes
// (profiles/tools/traffic_calib.sh) and divide.
//
//   wide16   16 B per lane, coalesced, every byte once            -- the guide's reference point (reports 1/2)
//   sub8     the bit readers: a half-wave owns a 16 KiB "compressed page", lane l streams its 512-byte sub-stream
//            with 8-byte loads (lane stride 512 B), pages pulled from a counter by 4096 persistent waves
//   flush16  the window flush: a half-wave owns a 64 KiB output page and stores it 512 B at a time (16 B per lane)
//   far      flush16 + after every flush each of 16 lanes reads 8 bytes from a random earlier place of its OWN page, more
//            than 528 bytes back (a far back-reference: a line this wave stored 1 us .. 1 ms ago); 8192 pages in flight
//   farx     the same with the 16 reads taken from a 4 KiB neighbourhood (short far distances)
// Usage: traffic_calib <kernel> [GiB=1] [reps=3]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void __launch_bounds__(256) wide16(const uint4* __restrict__ in, uint64_t n16, uint32_t* sink)
{
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * 256u) { const uint4 v = in[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) *sink = acc;
}

__global__ void __launch_bounds__(64) sub8(const uint8_t* __restrict__ in, uint32_t pages, uint32_t* counter, uint32_t* sink)
{
    const uint32_t lane = threadIdx.x, sl = lane & 31u;
    uint32_t acc = 0;
    for (;;) {
        uint32_t g = 0;
        if (sl == 0u) g = atomicAdd(counter, 1u);
        g = __shfl(g, lane & 32u);
        if (g >= pages) break;
        const uint8_t* p = in + (uint64_t)g * 16384u + sl * 512u;
        for (uint32_t o = 0; o < 512u; o += 8u) { uint64_t v; __builtin_memcpy(&v, p + o, 8); acc ^= (uint32_t)v ^ (uint32_t)(v >> 32); }
    }
    if (acc == 0x12345678u) *sink = acc;
}

template <int kFar>     // 0: stores only; 1: far reads anywhere below; 2: far reads within 4 KiB
__global__ void __launch_bounds__(64) flush_far(uint8_t* out, uint32_t pages, uint32_t* counter, uint32_t* sink)
{
    const uint32_t lane = threadIdx.x, sl = lane & 31u;
    uint32_t acc = 0, rng = lane * 2654435761u + blockIdx.x * 40503u + 1u;
    for (;;) {
        uint32_t g = 0;
        if (sl == 0u) g = atomicAdd(counter, 1u);
        g = __shfl(g, lane & 32u);
        if (g >= pages) break;
        uint8_t* page = out + (uint64_t)g * 65536u;
        for (uint32_t pos = 0; pos < 65536u; pos += 512u) {
            const uint4 v = {pos ^ rng, g, sl, acc};
            *reinterpret_cast<uint4*>(page + pos + 16u * sl) = v;
            if (kFar != 0 && pos >= 4096u && sl < 16u) {
                rng = rng * 1664525u + 1013904223u;
                const uint32_t span = kFar == 1 ? pos - 1024u : 3072u;
                const uint32_t back = 528u + (rng >> 8) % span;            // > 528 bytes back: flushed by an earlier step
                uint64_t w; __builtin_memcpy(&w, page + pos - back, 8);
                acc ^= (uint32_t)w ^ (uint32_t)(w >> 32);
            }
        }
    }
    if (acc == 0x12345678u) *sink = acc;
}

int main(int argc, char** argv)
{
    const char* which = argc > 1 ? argv[1] : "wide16";
    const uint64_t gib = argc > 2 ? (uint64_t)atoi(argv[2]) : 1u;
    const int reps = argc > 3 ? atoi(argv[3]) : 3;
    const uint64_t bytes = gib << 30;
    uint8_t* buf; uint32_t *counter, *sink;
    CK(hipMalloc(&buf, bytes + 4096)); CK(hipMalloc(&counter, 4)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(buf, 0x5A, bytes + 4096));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms_sum = 0.f;
    for (int r = 0; r < reps; ++r) {
        CK(hipMemset(counter, 0, 4));
        CK(hipEventRecord(e0));
        if (!strcmp(which, "wide16")) hipLaunchKernelGGL(wide16, dim3(256 * 8), dim3(256), 0, 0, (const uint4*)buf, bytes / 16u, sink);
        else if (!strcmp(which, "sub8")) hipLaunchKernelGGL(sub8, dim3(4096), dim3(64), 0, 0, buf, (uint32_t)(bytes / 16384u), counter, sink);
        else if (!strcmp(which, "flush16")) hipLaunchKernelGGL(flush_far<0>, dim3(4096), dim3(64), 0, 0, buf, (uint32_t)(bytes / 65536u), counter, sink);
        else if (!strcmp(which, "far")) hipLaunchKernelGGL(flush_far<1>, dim3(4096), dim3(64), 0, 0, buf, (uint32_t)(bytes / 65536u), counter, sink);
        else if (!strcmp(which, "farx")) hipLaunchKernelGGL(flush_far<2>, dim3(4096), dim3(64), 0, 0, buf, (uint32_t)(bytes / 65536u), counter, sink);
        else { fprintf(stderr, "unknown kernel %s\n", which); return 1; }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms_sum += ms;
    }
    const double ms = ms_sum / reps;
    // known bytes per launch
    const uint64_t pages64 = bytes / 65536u, far_reads = (!strcmp(which, "far") || !strcmp(which, "farx")) ? pages64 * 120u * 16u : 0u;
    const bool writes = !strcmp(which, "flush16") || far_reads;
    printf("{\"kernel\": \"%s\", \"buffer_bytes\": %llu, \"known_read_bytes\": %llu, \"known_write_bytes\": %llu, \"far_reads\": %llu, \"ms\": %.4f, \"GBps\": %.1f}\n",
           which, (unsigned long long)bytes, (unsigned long long)(writes ? far_reads * 8u : bytes), (unsigned long long)(writes ? bytes : 0u),
           (unsigned long long)far_reads, ms, (double)bytes / ms / 1e6);
    return 0;
}

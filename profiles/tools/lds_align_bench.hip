// Microbenchmark: cost of byte-misaligned LDS accesses on gfx950 (one wave, s_memtime around
// N back-to-back independent accesses).  Build: hipcc --offload-arch=gfx950 -O3 lds_align_bench.hip -o lds_align_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
template <typename T> __device__ __forceinline__ T ldu(const uint8_t* p) { T v; __builtin_memcpy(&v, p, sizeof(T)); return v; }
template <typename T> __device__ __forceinline__ void stu(uint8_t* p, T v) { __builtin_memcpy(p, &v, sizeof(T)); }
template <typename T, int N>
__global__ void k(uint64_t* out, uint32_t misalign, uint32_t stride)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[16384];
    const uint32_t lane = threadIdx.x;
    for (uint32_t i = lane; i < 16384; i += 64) lds[i] = (uint8_t)i;
    __syncthreads();
    const uint32_t base = lane * stride + misalign;
    // reads
    uint64_t t0 = __builtin_readcyclecounter();
    T acc = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) acc += ldu<T>(lds + ((base + i * 64 * stride) & 8191));
    uint64_t t1 = __builtin_readcyclecounter();
    // writes
#pragma unroll
    for (int i = 0; i < N; ++i) stu<T>(lds + 8192 + ((base + i * 64 * stride) & 4095), (T)(acc + i));
    __syncthreads();
    uint64_t t2 = __builtin_readcyclecounter();
    if (lane == 0) { out[0] = t1 - t0; out[1] = t2 - t1; out[2] = (uint64_t)acc + lds[8192 + 5]; }
}
// the alternative for a misaligned 8-byte read: three aligned dwords around it and two funnel shifts
template <int N>
__global__ void k3(uint64_t* out, uint32_t misalign, uint32_t stride)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[16384];
    const uint32_t lane = threadIdx.x;
    for (uint32_t i = lane; i < 16384; i += 64) lds[i] = (uint8_t)i;
    __syncthreads();
    const uint32_t base = lane * stride + misalign;
    uint64_t t0 = __builtin_readcyclecounter();
    uint64_t acc = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const uint32_t a = (base + i * 64 * stride) & 8191;
        const uint32_t* q = reinterpret_cast<const uint32_t*>(lds + (a & ~3u));
        const uint32_t w0 = q[0], w1 = q[1], w2 = q[2], sh = (a & 3u) * 8u;
        const uint32_t lo = __builtin_amdgcn_alignbit(w1, w0, sh), hi = __builtin_amdgcn_alignbit(w2, w1, sh);
        acc += (uint64_t)lo | ((uint64_t)hi << 32);
    }
    uint64_t t1 = __builtin_readcyclecounter();
    uint64_t ref = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) ref += ldu<uint64_t>(lds + ((base + i * 64 * stride) & 8191));
    if (lane == 0) { out[0] = t1 - t0; out[1] = acc == ref; out[2] = acc; }
}
void run3(uint32_t mis, uint32_t stride)
{
    uint64_t* d; hipMalloc(&d, 64); uint64_t h[3];
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL((k3<16>), dim3(1), dim3(64), 0, 0, d, mis, stride); hipMemcpy(h, d, 24, hipMemcpyDeviceToHost); }
    printf("3xu32  misalign=%u stride=%2u : 16 reads %5llu cyc (%.1f/op)   (three aligned dwords + 2 v_alignbit; equal to the u64 read: %s)\n", mis, stride,
           (unsigned long long)h[0], h[0] / 16.0, h[1] ? "yes" : "NO");
    hipFree(d);
}
template <typename T> void run(const char* name, uint32_t mis, uint32_t stride)
{
    uint64_t* d; hipMalloc(&d, 64); uint64_t h[3];
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL((k<T, 16>), dim3(1), dim3(64), 0, 0, d, mis, stride); hipMemcpy(h, d, 24, hipMemcpyDeviceToHost); }
    printf("%-6s misalign=%u stride=%2u : 16 reads %5llu cyc (%.1f/op)   16 writes %5llu cyc (%.1f/op)\n", name, mis, stride,
           (unsigned long long)h[0], h[0] / 16.0, (unsigned long long)h[1], h[1] / 16.0);
    hipFree(d);
}
int main()
{
    for (uint32_t stride : {8u, 9u, 17u}) {
        for (uint32_t mis : {0u, 1u, 3u}) {
            run3(mis, stride);
            run<uint8_t>("u8", mis, stride); run<uint16_t>("u16", mis, stride); run<uint32_t>("u32", mis, stride); run<uint64_t>("u64", mis, stride);
        }
    }
    return 0;
}

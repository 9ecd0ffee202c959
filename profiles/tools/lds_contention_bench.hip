// lds_contention_bench.hip -- what does a byte-misaligned 8-byte LDS access cost when FEW lanes take part and MANY waves do it
// at once?  (profiles/tools/lds_align_bench.hip measures one full wave alone.)  W waves of one workgroup each issue N accesses
// with the first k lanes active; reported: shader cycles per wave-instruction as seen by one wave, and LDS throughput of the CU
// in wave-instructions per 1000 cycles.  Round 3: the LZ77 assembly is made of exactly these.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
template <int kMode>   // 0 aligned read, 1 misaligned read, 2 aligned write, 3 misaligned write, 4 byte write, 5 byte read
__global__ void k(uint64_t* out, uint32_t active, uint32_t iters)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[32768];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (uint32_t i = threadIdx.x; i < 32768u; i += blockDim.x) lds[i] = (uint8_t)i;
    __syncthreads();
    const uint32_t mis = (kMode == 1 || kMode == 3) ? 1u + (lane % 7u) : 0u;
    uint32_t base = (wave * 2048u + lane * 24u) & 16383u;         // pieces ~24 bytes apart, like neighbouring commands
    uint64_t acc = lane;
    const bool on = lane < active;
    const uint64_t t0 = __builtin_readcyclecounter();
    for (uint32_t it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const uint32_t a = ((base + u * 1544u) & 16383u) + mis;
            if (on) {
                if (kMode <= 1) { uint64_t v; __builtin_memcpy(&v, lds + a, 8); acc += v; }
                else if (kMode <= 3) { const uint64_t v = acc + u; __builtin_memcpy(lds + 16384u + a, &v, 8); }
                else if (kMode == 4) lds[16384u + a] = (uint8_t)(acc + u);
                else acc += lds[a];
            }
        }
        base += 8u;
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    __syncthreads();
    if (lane == 0) out[wave] = t1 - t0;
    if (acc == 0x1234567u) out[63] = lds[16384 + lane];
}
template <int kMode> void run(const char* name, int waves, uint32_t active)
{
    uint64_t* d; hipMalloc(&d, 64 * 8); uint64_t h[64];
    const uint32_t iters = 64;
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL((k<kMode>), dim3(1), dim3(64 * waves), 0, 0, d, active, iters); hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost); }
    uint64_t mx = 0; for (int w = 0; w < waves; ++w) mx = h[w] > mx ? h[w] : mx;
    const double ops = 8.0 * iters;
    printf("%-16s waves=%2d active_lanes=%2u : %.1f cycles per wave-instruction per wave, CU throughput %.1f wave-instructions / 1000 cycles\n",
           name, waves, active, mx / ops, 1000.0 * ops * waves / mx);
    hipFree(d);
}
int main()
{
    for (int waves : {1, 4, 16})
        for (uint32_t act : {4u, 8u, 16u, 32u, 64u}) {
            run<0>("aligned read", waves, act); run<1>("misaligned read", waves, act);
            run<2>("aligned write", waves, act); run<3>("misaligned write", waves, act);
            run<4>("byte write", waves, act); run<5>("byte read", waves, act);
        }
    return 0;
}

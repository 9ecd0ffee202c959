// Does a VALU instruction cost fewer issue cycles on gfx950 when part of the wavefront is masked off in EXEC?
// One workgroup of `waves` wavefronts (1 = alone on a SIMD, 8 = two per SIMD); every wave runs the same chain
// of independent v_add_u32 under `if (lane in mask)`.  Reported: s_memtime ticks per wave-instruction, per wave.
//   hipcc --offload-arch=gfx950 -O3 -o exec_mask_bench exec_mask_bench.hip && ./exec_mask_bench
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define ITER 4000
__global__ void bench(uint64_t* out, uint64_t mask, uint32_t seed)
{
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t a0 = seed + lane, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 + 11u, a5 = a0 + 13u, a6 = a0 + 17u, a7 = a0 + 19u;
    const uint32_t s = (seed & 7u) + 1u;
    __syncthreads();
    const uint64_t w0 = wall_clock64();
    const uint64_t t0 = __builtin_readcyclecounter();
    if ((mask >> lane) & 1ull) {
        for (int i = 0; i < ITER; ++i)
            asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                         "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s));
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    if (lane == 0u) out[threadIdx.x >> 6] = t1 - t0;
    if (threadIdx.x == 0u) { out[32] = t1 - t0; out[33] = wall_clock64() - w0; }
    out[64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

int main()
{
    uint64_t* d; hipMalloc(&d, 8192 * 8);
    const struct { const char* name; uint64_t mask; } masks[] = {
        {"all 64 lanes", ~0ull}, {"lanes 0-31", 0xFFFFFFFFull}, {"lanes 32-63", 0xFFFFFFFF00000000ull}, {"lanes 0-15", 0xFFFFull},
        {"lanes 16-31", 0xFFFF0000ull}, {"lane 0", 1ull}, {"even lanes", 0x5555555555555555ull}, {"lanes 0-15 + 32-47", 0x0000FFFF0000FFFFull}};
    for (int waves : {1, 8, 16}) {
        printf("# %d wave(s) in one workgroup (%s)\n", waves, waves == 1 ? "alone on its SIMD" : waves == 8 ? "two per SIMD" : "four per SIMD");
        for (auto& m : masks) {
            uint64_t h[16];
            bench<<<1, 64 * waves>>>(d, m.mask, 1); hipDeviceSynchronize();
            bench<<<1, 64 * waves>>>(d, m.mask, 2); hipDeviceSynchronize();
            hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
            double worst = 0; for (int w = 0; w < waves; ++w) worst = h[w] > worst ? (double)h[w] : worst;
            printf("%-22s %8.2f ticks per wave-instruction (slowest wave)\n", m.name, worst / (ITER * 8.0));
        }
    }
    printf("# occupancy sweep, all lanes: waves per workgroup -> ticks per wave-instruction per wave; SIMD rate = waves/4 per that many ticks\n");
    for (int waves : {1, 2, 4, 8, 12, 16}) {
        uint64_t h[40];
        bench<<<1, 64 * waves>>>(d, ~0ull, 1); hipDeviceSynchronize();
        bench<<<1, 64 * waves>>>(d, ~0ull, 2); hipDeviceSynchronize();
        hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        double worst = 0; for (int w = 0; w < waves; ++w) worst = h[w] > worst ? (double)h[w] : worst;
        printf("waves %2d: %6.2f ticks per wave-instruction; wave 0: %llu s_memtime ticks in %llu wall_clock64 ticks (100 MHz) -> %.1f MHz\n", waves,
               worst / (ITER * 8.0), (unsigned long long)h[32], (unsigned long long)h[33], h[33] ? 100.0 * h[32] / h[33] : 0.0);
    }
    return 0;
}

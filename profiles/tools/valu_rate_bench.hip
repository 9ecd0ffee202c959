// VALU issue-rate microbenchmark for gfx950: cycles per wave64 instruction for the integer ops the
// decoder leans on.  One wave per workgroup, one workgroup; 8 independent chains per op so that the
// figure is issue rate, not dependent-result latency; a second figure uses one dependent chain.
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate_bench valu_rate_bench.hip && ./valu_rate_bench
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define REP8(x) x x x x x x x x
#define ITER 2000

#define BENCH_KERNEL(name, body_indep, body_dep)                                                   \
    __global__ void name(uint64_t* out, uint32_t seed)                                             \
    {                                                                                              \
        uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u;               \
        uint32_t a4 = a0 + 11u, a5 = a0 + 13u, a6 = a0 + 17u, a7 = a0 + 19u;                       \
        uint64_t q0 = a0 | ((uint64_t)a1 << 32), q1 = a2 | ((uint64_t)a3 << 32);                   \
        uint64_t q2 = a4 | ((uint64_t)a5 << 32), q3 = a6 | ((uint64_t)a7 << 32);                   \
        uint32_t s = (seed & 7u) + 1u;                                                             \
        uint64_t t0 = __builtin_readcyclecounter();                                                \
        for (int i = 0; i < ITER; ++i) { REP8(body_indep) }                                        \
        uint64_t t1 = __builtin_readcyclecounter();                                                \
        for (int i = 0; i < ITER; ++i) { REP8(body_dep) }                                          \
        uint64_t t2 = __builtin_readcyclecounter();                                                \
        if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = t2 - t1; }                              \
        out[2 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + q0 + q1 + q2 + q3 + s;      \
    }

#define A8(op) asm volatile(op " %0, %0, %8\n" op " %1, %1, %8\n" op " %2, %2, %8\n" op " %3, %3, %8\n"     \
                            op " %4, %4, %8\n" op " %5, %5, %8\n" op " %6, %6, %8\n" op " %7, %7, %8\n"     \
                            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s));
#define D8(op) asm volatile(op " %0, %0, %1\n" op " %0, %0, %1\n" op " %0, %0, %1\n" op " %0, %0, %1\n"     \
                            op " %0, %0, %1\n" op " %0, %0, %1\n" op " %0, %0, %1\n" op " %0, %0, %1\n"     \
                            : "+v"(a0) : "v"(s));
// shift-style: op dst, shift, src
#define AS8(op) asm volatile(op " %0, %8, %0\n" op " %1, %8, %1\n" op " %2, %8, %2\n" op " %3, %8, %3\n"    \
                             op " %4, %8, %4\n" op " %5, %8, %5\n" op " %6, %8, %6\n" op " %7, %8, %7\n"    \
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s));
#define DS8(op) asm volatile(op " %0, %1, %0\n" op " %0, %1, %0\n" op " %0, %1, %0\n" op " %0, %1, %0\n"    \
                             op " %0, %1, %0\n" op " %0, %1, %0\n" op " %0, %1, %0\n" op " %0, %1, %0\n"    \
                             : "+v"(a0) : "v"(s));
#define AQ8(op) asm volatile(op " %0, %4, %0\n" op " %1, %4, %1\n" op " %2, %4, %2\n" op " %3, %4, %3\n"    \
                             op " %0, %4, %0\n" op " %1, %4, %1\n" op " %2, %4, %2\n" op " %3, %4, %3\n"    \
                             : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(s));
#define DQ8(op) asm volatile(op " %0, %1, %0\n" op " %0, %1, %0\n" op " %0, %1, %0\n" op " %0, %1, %0\n"    \
                             op " %0, %1, %0\n" op " %0, %1, %0\n" op " %0, %1, %0\n" op " %0, %1, %0\n"    \
                             : "+v"(q0) : "v"(s));
// three-operand: op dst, a, b, c
#define A3_8(op) asm volatile(op " %0, %0, %8, %1\n" op " %1, %1, %8, %2\n" op " %2, %2, %8, %3\n" op " %3, %3, %8, %4\n" \
                              op " %4, %4, %8, %5\n" op " %5, %5, %8, %6\n" op " %6, %6, %8, %7\n" op " %7, %7, %8, %0\n" \
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s));
#define D3_8(op) asm volatile(op " %0, %0, %1, %0\n" op " %0, %0, %1, %0\n" op " %0, %0, %1, %0\n" op " %0, %0, %1, %0\n" \
                              op " %0, %0, %1, %0\n" op " %0, %0, %1, %0\n" op " %0, %0, %1, %0\n" op " %0, %0, %1, %0\n" \
                              : "+v"(a0) : "v"(s));
// unary: op dst, src
#define A1_8(op) asm volatile(op " %0, %0\n" op " %1, %1\n" op " %2, %2\n" op " %3, %3\n" op " %4, %4\n" op " %5, %5\n"   \
                              op " %6, %6\n" op " %7, %7\n"                                                              \
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
#define D1_8(op) asm volatile(op " %0, %0\n" op " %0, %0\n" op " %0, %0\n" op " %0, %0\n" op " %0, %0\n" op " %0, %0\n"   \
                              op " %0, %0\n" op " %0, %0\n" : "+v"(a0));
#define ADPP8 asm volatile("v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n"                              \
                           "v_add_u32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n"                              \
                           "v_add_u32_dpp %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n"                              \
                           "v_add_u32_dpp %3, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n"                              \
                           "v_add_u32_dpp %4, %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n"                              \
                           "v_add_u32_dpp %5, %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n"                              \
                           "v_add_u32_dpp %6, %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n"                              \
                           "v_add_u32_dpp %7, %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n"                              \
                           : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
#define DDPP8 asm volatile("v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1\n"                   \
                           "v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1\n"                   \
                           "v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1\n"                   \
                           "v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1\n"                   \
                           "v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1\n"                   \
                           "v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1\n"                   \
                           "v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1\n"                   \
                           "v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1\n" : "+v"(a0));
#define ABP8 asm volatile("ds_bpermute_b32 %0, %8, %0\n ds_bpermute_b32 %1, %8, %1\n ds_bpermute_b32 %2, %8, %2\n"       \
                          "ds_bpermute_b32 %3, %8, %3\n ds_bpermute_b32 %4, %8, %4\n ds_bpermute_b32 %5, %8, %5\n"       \
                          "ds_bpermute_b32 %6, %8, %6\n ds_bpermute_b32 %7, %8, %7\n s_waitcnt lgkmcnt(0)\n"             \
                          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s));
#define DBP8 asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)\n ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)\n" \
                          "ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)\n ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)\n" \
                          "ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)\n ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)\n" \
                          "ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)\n ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)\n" \
                          : "+v"(a0) : "v"(s));

BENCH_KERNEL(k_add_u32, A8("v_add_u32"), D8("v_add_u32"))
BENCH_KERNEL(k_and_b32, A8("v_and_b32"), D8("v_and_b32"))
BENCH_KERNEL(k_lshrrev_b32, AS8("v_lshrrev_b32"), DS8("v_lshrrev_b32"))
BENCH_KERNEL(k_lshrrev_b64, AQ8("v_lshrrev_b64"), DQ8("v_lshrrev_b64"))
BENCH_KERNEL(k_lshlrev_b64, AQ8("v_lshlrev_b64"), DQ8("v_lshlrev_b64"))
BENCH_KERNEL(k_mul_lo_u32, A8("v_mul_lo_u32"), D8("v_mul_lo_u32"))
BENCH_KERNEL(k_mul_hi_u32, A8("v_mul_hi_u32"), D8("v_mul_hi_u32"))
BENCH_KERNEL(k_mul_u32_u24, A8("v_mul_u32_u24"), D8("v_mul_u32_u24"))
BENCH_KERNEL(k_bcnt, A8("v_bcnt_u32_b32"), D8("v_bcnt_u32_b32"))
BENCH_KERNEL(k_min_u32, A8("v_min_u32"), D8("v_min_u32"))
BENCH_KERNEL(k_alignbit, A3_8("v_alignbit_b32"), D3_8("v_alignbit_b32"))
BENCH_KERNEL(k_bfe_u32, A3_8("v_bfe_u32"), D3_8("v_bfe_u32"))
BENCH_KERNEL(k_add3_u32, A3_8("v_add3_u32"), D3_8("v_add3_u32"))
BENCH_KERNEL(k_lshl_add_u32, A3_8("v_lshl_add_u32"), D3_8("v_lshl_add_u32"))
BENCH_KERNEL(k_and_or_b32, A3_8("v_and_or_b32"), D3_8("v_and_or_b32"))
BENCH_KERNEL(k_mad_u32_u24, A3_8("v_mad_u32_u24"), D3_8("v_mad_u32_u24"))
BENCH_KERNEL(k_bfrev, A1_8("v_bfrev_b32"), D1_8("v_bfrev_b32"))
BENCH_KERNEL(k_ffbl, A1_8("v_ffbl_b32"), D1_8("v_ffbl_b32"))
BENCH_KERNEL(k_ffbh, A1_8("v_ffbh_u32"), D1_8("v_ffbh_u32"))
BENCH_KERNEL(k_cvt_f32_u32, A1_8("v_cvt_f32_u32"), D1_8("v_cvt_f32_u32"))
BENCH_KERNEL(k_rcp_f32, A1_8("v_rcp_f32"), D1_8("v_rcp_f32"))
BENCH_KERNEL(k_add_dpp_row_shr, ADPP8, DDPP8)
BENCH_KERNEL(k_ds_bpermute, ABP8, DBP8)

struct Entry { const char* name; void (*fn)(uint64_t*, uint32_t); };
#define E(n) {#n, n}

int main()
{
    Entry es[] = {E(k_add_u32), E(k_and_b32), E(k_lshrrev_b32), E(k_lshrrev_b64), E(k_lshlrev_b64), E(k_mul_lo_u32),
                  E(k_mul_hi_u32), E(k_mul_u32_u24), E(k_bcnt), E(k_min_u32), E(k_alignbit), E(k_bfe_u32), E(k_add3_u32),
                  E(k_lshl_add_u32), E(k_and_or_b32), E(k_mad_u32_u24), E(k_bfrev), E(k_ffbl), E(k_ffbh), E(k_cvt_f32_u32),
                  E(k_rcp_f32), E(k_add_dpp_row_shr), E(k_ds_bpermute)};
    uint64_t* d; hipMalloc(&d, 66 * sizeof(uint64_t));
    printf("# cycles per wave64 instruction (s_memtime ticks; 64 instructions per loop iteration incl. loop overhead)\n");
    printf("%-22s %12s %12s\n", "op", "independent", "dependent");
    for (auto& e : es) {
        uint64_t h[2] = {0, 0};
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(e.fn, dim3(1), dim3(64), 0, 0, d, 3u);
            hipDeviceSynchronize();
        }
        hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        printf("%-22s %12.2f %12.2f\n", e.name, (double)h[0] / (ITER * 64.0), (double)h[1] / (ITER * 64.0));
    }
    return 0;
}

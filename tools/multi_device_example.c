/* multi_device_example.c -- a C host of the multi-device decode entry (include/brotlig_amd.h):
 *     BrotligShardPlan  ->  one BrotligDeviceBatch per shard  ->  BrotligDecodeBatchMultiDevice.
 * It is what INTEGRATION.md section 5 describes, kept compilable (tests/test_integration_shims.py builds it; the -m gpu
 * test runs it).  The reference's analogues: pages fanned out over host threads (src/BrotligDecoder.cpp:356-375) and
 * streams over the shader's queue (src/decoder/BrotliGCompute.hlsl:1757-1881).
 *
 *     multi_device_example [shards]      shards default to the number of visible devices; shard g runs on device
 *                                        g mod devices (so `3` on a one-GPU box exercises the fan-out on one device)
 * Inputs are synthetic (this repo's encoder, linked only to make them); the decoded bytes are compared with the source.
 * Build: hipcc -x hip --offload-arch=gfx950 -Iinclude tools/multi_device_example.c -Lbrotli_g_sdk_amd/csrc -lbrotlig_hip -lbrotlig_enc */
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "brotlig_amd.h"
#include "../brotli_g_sdk_amd/csrc/brotlig_encoder.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
#define N_STREAMS 12
#define MAX_SHARDS 16

static uint8_t* make_source(uint32_t n, uint32_t seed)
{
    uint8_t* p = (uint8_t*)malloc(n);
    uint32_t x = seed * 2654435761u + 1u;
    for (uint32_t i = 0; i < n; ++i) {
        x = x * 1664525u + 1013904223u;
        /* runs, repeats of earlier bytes and fresh bytes: something for every part of the decoder */
        if ((x >> 28) < 5u && i > 64u) p[i] = p[i - 1u - ((x >> 8) % 64u)];
        else if ((x >> 28) < 9u && i) p[i] = p[i - 1];
        else p[i] = (uint8_t)(x >> 16) & (seed & 1u ? 0x3Fu : 0xFFu);
    }
    return p;
}

int main(int argc, char** argv)
{
    int devices = 0;
    CHECK_HIP(hipGetDeviceCount(&devices));
    if (devices < 1) { fprintf(stderr, "no HIP device\n"); return 1; }
    uint32_t shards = argc > 1 ? (uint32_t)atoi(argv[1]) : (uint32_t)devices;
    if (shards < 1u || shards > MAX_SHARDS) shards = 1u;

    /* the batch: N_STREAMS streams of different sizes and ratios */
    uint8_t* src[N_STREAMS]; uint32_t src_size[N_STREAMS];
    uint8_t* enc[N_STREAMS]; uint64_t enc_size[N_STREAMS];
    for (uint32_t i = 0; i < N_STREAMS; ++i) {
        src_size[i] = 65536u * (1u + i % 5u) + 1000u * i;
        src[i] = make_source(src_size[i], i + 1u);
        uint32_t cap = BrotligEncMaxCompressedSize(src_size[i], 65536u);
        enc[i] = (uint8_t*)malloc(cap);
        BrotligEncodeOptions opt; memset(&opt, 0, sizeof opt);
        if (BrotligEncode(src_size[i], src[i], &cap, enc[i], &opt) != BROTLIG_ENC_OK) { fprintf(stderr, "encode failed\n"); return 1; }
        enc_size[i] = cap;
    }

    /* 1. cut the stream list: contiguous runs balanced by compressed bytes */
    uint32_t first[MAX_SHARDS + 1];
    if (BrotligShardPlan(enc_size, N_STREAMS, shards, first) != BROTLIG_OK) return 1;

    /* 2. place each run on its device: streams back to back (16-byte aligned), outputs at whole pages */
    BrotligDeviceBatch batch[MAX_SHARDS];
    uint8_t* h_out[MAX_SHARDS]; uint64_t out_off[N_STREAMS];
    memset(batch, 0, sizeof batch);
    for (uint32_t g = 0; g < shards; ++g) {
        const uint32_t n = first[g + 1] - first[g];
        BrotligDeviceBatch* b = &batch[g];
        b->device = (int32_t)(g % (uint32_t)devices);
        b->num_streams = n;
        h_out[g] = NULL;
        if (n == 0u) continue;
        CHECK_HIP(hipSetDevice(b->device));
        BrotligStreamDesc desc[N_STREAMS];
        uint64_t in_bytes = 0, out_bytes = 0;
        for (uint32_t k = 0; k < n; ++k) {
            const uint32_t i = first[g] + k;
            desc[k].in_offset = in_bytes; desc[k].in_size = enc_size[i];
            in_bytes += (enc_size[i] + 15u) & ~(uint64_t)15u;
            desc[k].out_offset = out_bytes; out_off[i] = out_bytes;
            desc[k].out_capacity = ((uint64_t)src_size[i] + 65535u) & ~(uint64_t)65535u;      /* NumPages * PageSize */
            out_bytes += desc[k].out_capacity;
        }
        void *d_in, *d_out, *d_desc, *d_ws; hipStream_t s;
        CHECK_HIP(hipMalloc(&d_in, in_bytes + 16u));                 /* 16 bytes past in_bytes (contract) */
        CHECK_HIP(hipMalloc(&d_out, out_bytes + 8u));                /* 8 bytes past out_bytes (contract) */
        CHECK_HIP(hipMalloc(&d_desc, sizeof(BrotligStreamDesc) * n));
        const size_t ws = BrotligDecodeWorkspaceSizeFor(n, out_bytes);
        CHECK_HIP(hipMalloc(&d_ws, ws));
        CHECK_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        for (uint32_t k = 0; k < n; ++k)
            CHECK_HIP(hipMemcpy((uint8_t*)d_in + desc[k].in_offset, enc[first[g] + k], enc_size[first[g] + k], hipMemcpyHostToDevice));
        CHECK_HIP(hipMemcpy(d_desc, desc, sizeof(BrotligStreamDesc) * n, hipMemcpyHostToDevice));
        CHECK_HIP(hipMemset(d_out, 0xCD, out_bytes));
        b->d_in = d_in; b->in_bytes = in_bytes; b->d_out = d_out; b->out_bytes = out_bytes;
        b->d_streams = (const BrotligStreamDesc*)d_desc; b->d_workspace = d_ws; b->workspace_bytes = ws;
        b->d_scratch = NULL; b->hip_stream = s;
        h_out[g] = (uint8_t*)malloc(out_bytes);
    }
    /* shards without streams cannot be passed (num_streams == 0 is an error): compact the array */
    uint32_t live = 0; BrotligDeviceBatch run[MAX_SHARDS]; uint32_t run_of[MAX_SHARDS];
    for (uint32_t g = 0; g < shards; ++g) if (batch[g].num_streams) { run[live] = batch[g]; run_of[live++] = g; }

    /* 3. one call: a host thread per shard, each on its device; the job takes as long as the slowest shard */
    double kernel_ms = 0.0, wall_ms = 0.0;
    const BROTLIG_ERROR rc = BrotligDecodeBatchMultiDevice(run, live, (uint32_t)sizeof(BrotligDeviceBatch), 1u, 3u, &kernel_ms, &wall_ms);
    if (rc != BROTLIG_OK) { fprintf(stderr, "BrotligDecodeBatchMultiDevice: %d\n", (int)rc); return 1; }

    /* 4. check every byte */
    uint64_t total = 0;
    for (uint32_t r = 0; r < live; ++r) {
        const uint32_t g = run_of[r];
        CHECK_HIP(hipSetDevice(run[r].device));
        CHECK_HIP(hipMemcpy(h_out[g], run[r].d_out, run[r].out_bytes, hipMemcpyDeviceToHost));
        for (uint32_t i = first[g]; i < first[g + 1]; ++i) {
            if (memcmp(h_out[g] + out_off[i], src[i], src_size[i]) != 0) { fprintf(stderr, "stream %u differs\n", i); return 1; }
            total += src_size[i];
        }
        printf("shard %u: device %d, streams %u..%u, kernel %.3f ms\n", g, run[r].device, first[g], first[g + 1] - 1u, run[r].kernel_ms);
    }
    printf("%u streams over %u shards on %d device(s): %llu bytes bit-exact, slowest shard kernel %.3f ms, wall %.3f ms for 3 passes\n",
           (unsigned)N_STREAMS, live, devices, (unsigned long long)total, kernel_ms, wall_ms);

    /* 5. which asset was damaged: break the magic byte of the second stream of the first shard on the device, decode that shard again,
     *    and ask for one result per stream (BrotligDecodeBatchStreamStatus, round 5) -- the batch answer alone would not say which */
    if (run[0].num_streams >= 2u) {
        CHECK_HIP(hipSetDevice(run[0].device));
        const uint32_t victim = 1u, i = first[run_of[0]] + victim;
        uint64_t at = 0;
        for (uint32_t k = 0; k < victim; ++k) at += (enc_size[first[run_of[0]] + k] + 15u) & ~(uint64_t)15u;
        const uint8_t broken = (uint8_t)(enc[i][1] ^ 0x10);
        CHECK_HIP(hipMemcpy((uint8_t*)run[0].d_in + at + 1u, &broken, 1, hipMemcpyHostToDevice));
        if (BrotligDecodeBatchDevice(run[0].d_in, run[0].in_bytes, run[0].d_out, run[0].out_bytes, run[0].d_streams, run[0].num_streams,
                                     run[0].d_workspace, (size_t)run[0].workspace_bytes, NULL, run[0].hip_stream) != BROTLIG_OK) return 1;
        int32_t per_stream[N_STREAMS];
        const BROTLIG_ERROR all = BrotligDecodeBatchStreamStatus(run[0].d_workspace, run[0].num_streams, per_stream, run[0].hip_stream);
        if (all != BROTLIG_ERROR_CORRUPT_STREAM) { fprintf(stderr, "batch status %d, expected CORRUPT_STREAM\n", (int)all); return 1; }
        for (uint32_t k = 0; k < run[0].num_streams; ++k)
            if (per_stream[k] != (k == victim ? (int32_t)BROTLIG_ERROR_CORRUPT_STREAM : (int32_t)BROTLIG_OK)) {
                fprintf(stderr, "stream %u of shard 0 reports %d\n", k, (int)per_stream[k]); return 1;
            }
        printf("per-stream status: stream %u of shard 0 named as the damaged one, the other %u are fine\n", victim, run[0].num_streams - 1u);
    }
    return 0;
}

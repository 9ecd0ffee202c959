/*
 * brotlig_amd.h -- C ABI of the MI355X-native Brotli-G decompressor (libbrotlig_hip.so).
 *
 * Drop-in boundary for the GPU decode path of GPUOpen brotli_g_sdk 1.1.  Each entry point
 * names the reference interface it replaces (paths relative to the reference tree).  Plain
 * pointers and sizes only; no C++ or torch types cross this boundary.  There is no CPU decode
 * path in this library: if no HIP device is usable every decode entry fails with
 * BROTLIG_ERROR_GENERIC.
 */
#ifndef BROTLIG_AMD_H
#define BROTLIG_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI version of the device-pointer batch interface.  The entry points whose contract changes carry the version in their SYMBOL names
 * (the macros below), so a caller built against an older header fails to link instead of passing buffers of the wrong layout;
 * BrotligAbiVersion() answers at run time (dlopen users).
 *   3 (round 3): 32-byte BrotligStreamDesc (was 16), d_in must extend 16 bytes past in_bytes, the workspace holds the per-wavefront symbol
 *     slots (+30 MiB: 8192 workgroups).  Rounds 4 and 5 only ADDED entry points; round 5 also grew the workspace header by 1 KiB without
 *     saying so -- a reason more for:
 *   4 (round 6): the workspace's page schedule is one 32-byte job record per page (was one word), so BrotligDecodeWorkspaceSizeFor
 *     returns more for the same batch, and the header holds the schedule kernel's counters.  Workspace sizes are never constants of a
 *     caller: ask the LOADED library (BrotligDecodeWorkspaceSize / ...For).  A workspace smaller than ...SizeFor's answer still decodes --
 *     in stream order, every page start walking the page tables -- as long as it has BrotligDecodeWorkspaceSize bytes; less is refused. */
#define BROTLIG_AMD_ABI_VERSION 4
#define BrotligDecodeWorkspaceSize      BrotligDecodeWorkspaceSize_v4
#define BrotligDecodeWorkspaceSizeFor   BrotligDecodeWorkspaceSizeFor_v4
#define BrotligDecodeBatchDevice        BrotligDecodeBatchDevice_v4
#define BrotligDecodeBatchStatus        BrotligDecodeBatchStatus_v4
#define BrotligDecodeBatchTimed         BrotligDecodeBatchTimed_v4
#define BrotligDecodePhaseProfile       BrotligDecodePhaseProfile_v4
#define BrotligDecodeBatchMultiDevice   BrotligDecodeBatchMultiDevice_v4
uint32_t BrotligAbiVersion(void);

/* inc/common/BrotligCommon.h:50-68 -- same enumerators, same numeric values */
typedef enum BROTLIG_ERROR {
    BROTLIG_OK = 0,
    BROTLIG_ABORTED,
    BROTLIG_ERROR_MIN_PAGE_SIZE,
    BROTLIG_ERROR_MAX_PAGE_SIZE,
    BROTLIG_ERROR_MAX_NUM_PAGES,
    BROTLIG_ERROR_PRECON_MIN_TEX_WIDTH,
    BROTLIG_ERROR_PRECON_MAX_TEX_WIDTH,
    BROTLIG_ERROR_PRECON_MIN_TEX_HEIGHT,
    BROTLIG_ERROR_PRECON_MAX_TEX_HEIGHT,
    BROTLIG_ERROR_PRECON_MIN_TEX_PITCH,
    BROTLIG_ERROR_PRECON_MAX_TEX_PITCH,
    BROTLIG_ERROR_PRECON_MIN_TEX_MIPLEVELS,
    BROTLIG_ERROR_PRECON_MAX_TEX_MIPLEVELS,
    BROTLIG_ERROR_PRECON_INCORRECT_FORMAT,
    BROTLIG_ERROR_CORRUPT_STREAM,
    BROTLIG_ERROR_INCORRECT_STREAM_FORMAT,
    BROTLIG_ERROR_GENERIC
} BROTLIG_ERROR;

/* Replaces: uint32_t BrotliG::DecompressedSize(uint8_t* src)
 *   inc/BrotligDecoder.h:32, src/BrotligDecoder.cpp:35-39.
 * `src` is a HOST pointer to at least the 8-byte stream header.  No validation, like the
 * reference: NumPages * PageSize - (LastPageSize ? PageSize - LastPageSize : 0). */
uint32_t DecompressedSize(uint8_t* src);

/* Replaces: BROTLIG_ERROR DecodeGPU(bool useWarpDevice, uint32_t input_size, const uint8_t* input,
 *                                   uint32_t* output_size, uint8_t* output, double& time)
 *   sample/BrotligGPUDecoder.h:24, sample/BrotligGPUDecoder.cpp:260-748.
 * Host pointers in and out.  *output_size must hold the capacity of `output` on entry (for a
 * preconditioned stream: exactly the texture size, as src/BrotligDecoder.cpp:478 requires) and
 * receives the decompressed size.  *time_ms (may be NULL) receives the kernel-only time in
 * milliseconds, measured with HIP events around the decode kernels, mirroring the reference's
 * timestamp pair around Dispatch (BrotligGPUDecoder.cpp:673-675).  `useWarpDevice` is ignored.
 * Errors: BROTLIG_ERROR_CORRUPT_STREAM (magic), BROTLIG_ERROR_INCORRECT_STREAM_FORMAT (id != 5)
 * as src/BrotligDecoder.cpp:437-446; BROTLIG_ERROR_GENERIC for device errors (the reference
 * throws std::exception there) and for pages that fail a bounds check. */
BROTLIG_ERROR DecodeGPU(int useWarpDevice, uint32_t input_size, const uint8_t* input,
                        uint32_t* output_size, uint8_t* output, double* time_ms);

/* Optional reusable context for DecodeGPU ("a reusable context object is allowed for HIP module / scratch
 * reuse but must be optional"): device buffers that only grow, one stream, two events.  DecodeGPU itself stays stateless
 * (it builds a context for the length of the call: five hipMalloc / hipFree pairs, which dominate small assets).
 * `device` < 0: the calling thread's current device.  One context per host thread. */
typedef struct BrotligContext BrotligContext;
BROTLIG_ERROR BrotligContextCreate(int device, BrotligContext** out);
void BrotligContextDestroy(BrotligContext* context);
BROTLIG_ERROR BrotligContextDecodeGPU(BrotligContext* context, uint32_t input_size, const uint8_t* input,
                                      uint32_t* output_size, uint8_t* output, double* time_ms);

/* ---- device-pointer batch interface -------------------------------------------------------
 * Replaces the kernel-level contract of the reference shader: three buffers (input = whole
 * streams back to back, meta = work queue, output) and one dispatch that decodes up to 4096
 * streams (src/decoder/BrotliGCompute.hlsl:93-95,:1757-1881;
 * inc/common/BrotligConstants.h:126-129).  Everything is asynchronous on `hip_stream`. */
typedef struct BrotligStreamDesc {
    uint64_t in_offset;     /* byte offset of the stream's header in d_in; multiple of 4 */
    uint64_t out_offset;    /* byte offset of its decompressed bytes in d_out; multiple of 16 */
    uint64_t in_size;       /* bytes of the stream, the `input_size` of DecodeCPU / DecodeGPU
                             * (inc/BrotligDecoder.h:33); 0 = everything up to in_bytes */
    uint64_t out_capacity;  /* bytes the stream may write at out_offset, the `*output_size` the reference's
                             * callers pass in (src/BrotligDecoder.cpp:448,:478); 0 = up to out_bytes.
                             * A stream whose header asks for more is rejected (status), so a damaged
                             * header cannot reach a neighbouring stream's output. */
} BrotligStreamDesc;

/* Bytes of device workspace needed for `num_streams` streams (the reference's `meta` buffer): status words, page
 * counts, pre-conditioning tables, and 30 MiB of per-wavefront slots (up to 8192 workgroups) for prefix-code symbols beyond
 * the kernel's LDS arrays (kept small so that 16 wavefronts fit a compute unit).  One workspace per batch in flight. */
size_t BrotligDecodeWorkspaceSize(uint32_t num_streams);
/* Workspace size that also holds the page schedule for `out_bytes` of output (one 32-byte job record per page the batch can hold: 1 KiB
 * per MiB of output): with it a page start is one read of its record instead of a walk through the page tables, batches of 768 MiB and
 * more are decoded bucket by bucket, similar pages side by side (about 12 % faster on mixed data), and batches of between one and two
 * pages per wavefront of the page kernel (256 .. 512 MiB of 64 KiB pages on an MI355X) densest page next to lightest.
 * BrotligDecodeWorkspaceSize is the minimum; a workspace in between is used like the minimum. */
size_t BrotligDecodeWorkspaceSizeFor(uint32_t num_streams, uint64_t out_bytes);

/* Enqueue the decode of `num_streams` streams.
 *   d_in / in_bytes       device buffer holding the streams; the allocation must extend 16 bytes past
 *                         in_bytes (contents irrelevant: the bit readers prefetch ahead, like the
 *                         reference's 8-byte over-read, inc/common/BrotligDeswizzler.h:74-81, and a
 *                         valid stream never consumes those bits).  d_in itself 16-byte aligned.
 *   d_out / out_bytes     device buffer receiving the decompressed bytes; the allocation must
 *                         extend 8 bytes past out_bytes (wide source reads of the last page;
 *                         nothing is written there).  Same for d_scratch.
 *   d_streams             DEVICE array of num_streams descriptors
 *   d_workspace           device memory, BrotligDecodeWorkspaceSize(num_streams) bytes
 *   d_scratch             device memory of out_bytes bytes, needed only if a stream is
 *                         preconditioned (conditioned-space staging); may be NULL otherwise
 *   hip_stream            hipStream_t (NULL = default stream)
 * Returns BROTLIG_OK when the launches were enqueued.  Stream/page level failures are
 * reported by BrotligDecodeBatchStatus. */
BROTLIG_ERROR BrotligDecodeBatchDevice(const void* d_in, uint64_t in_bytes, void* d_out, uint64_t out_bytes,
                                       const BrotligStreamDesc* d_streams, uint32_t num_streams,
                                       void* d_workspace, size_t workspace_bytes, void* d_scratch,
                                       void* hip_stream);

/* Waits for `hip_stream` and returns the status of the last batch that used `d_workspace`:
 * BROTLIG_OK, BROTLIG_ERROR_CORRUPT_STREAM (a header failed the magic/id check) or
 * BROTLIG_ERROR_GENERIC (a page failed a bounds check). */
BROTLIG_ERROR BrotligDecodeBatchStatus(const void* d_workspace, void* hip_stream);

/* Per-stream results of the last batch that used `d_workspace` (round 5).  The reference's only result is per CALL, one stream per call
 * (src/BrotligDecoder.cpp:437-446); its shader's queue of up to 4 096 streams (src/decoder/BrotliGCompute.hlsl:1757-1881) has no way to say
 * WHICH asset of a dispatch was damaged.  Waits for `hip_stream`, then results[i] = BROTLIG_OK, BROTLIG_ERROR_CORRUPT_STREAM (stream i's
 * header was refused: none of its pages was decoded) or BROTLIG_ERROR_GENERIC (a page of stream i failed a bounds check: that page's
 * bytes are undefined, the stream's other pages and every other stream are as decoded).  `results` is a HOST array of `num_streams`
 * (the count the batch was enqueued with).  Returns what BrotligDecodeBatchStatus returns for the batch. */
BROTLIG_ERROR BrotligDecodeBatchStreamStatus(const void* d_workspace, uint32_t num_streams, int32_t* results, void* hip_stream);

/* Benchmark helper: runs the batch `warmup` + `steps` times on `hip_stream` and reports
 *   *total_ms        wall time of the `steps` timed passes (HIP events on hip_stream)
 *   *decode_kernel_ms average duration of the page-decode kernel alone over the timed passes
 * (the reference's own timing convention is kernel-only: BrotligGPUDecoder.cpp:729-746). */
BROTLIG_ERROR BrotligDecodeBatchTimed(const void* d_in, uint64_t in_bytes, void* d_out, uint64_t out_bytes,
                                      const BrotligStreamDesc* d_streams, uint32_t num_streams,
                                      void* d_workspace, size_t workspace_bytes, void* d_scratch,
                                      void* hip_stream, uint32_t warmup, uint32_t steps,
                                      double* total_ms, double* decode_kernel_ms);

/* ---- multi-device fan-out ---------------------------------------------------------------
 * Streams are independent, so a batch shards over the GPUs of a node with no exchange between them.  The reference's
 * analogues: pages fanned out over host threads (src/BrotligDecoder.cpp:356-375) and streams over the shader's queue
 * (src/decoder/BrotliGCompute.hlsl:1757-1881).
 *
 * BrotligShardPlan cuts the stream list into `num_shards` contiguous runs: first[g] .. first[g+1]-1 is shard g
 * (first has num_shards + 1 entries).  The cut minimises the largest run's COMPRESSED bytes (`in_sizes`, one per
 * stream) -- decode cost follows compressed size, not the 64 KiB of output per page -- and leaves no shard empty while
 * streams remain.  Streams are never split (a pre-conditioned stream's pages scatter over its whole texture).  Pure
 * host arithmetic: needs no device. */
BROTLIG_ERROR BrotligShardPlan(const uint64_t* in_sizes, uint32_t num_streams, uint32_t num_shards, uint32_t* first);

/* One shard: the arguments of BrotligDecodeBatchDevice for the streams placed on `device`, plus results. */
typedef struct BrotligDeviceBatch {
    int32_t  device;                    /* HIP device ordinal */
    uint32_t num_streams;
    const void* d_in;   uint64_t in_bytes;
    void* d_out;        uint64_t out_bytes;
    const BrotligStreamDesc* d_streams;
    void* d_workspace;  uint64_t workspace_bytes;
    void* d_scratch;
    void* hip_stream;                   /* a stream of `device` (NULL: its default stream) */
    int32_t  result;                    /* out: BROTLIG_ERROR of this shard (enqueue + batch status) */
    uint32_t reserved;
    double   kernel_ms;                 /* out: average decode-kernel time of the timed passes (HIP events on hip_stream) */
    double   wall_ms;                   /* out: host clock from the common start to this shard's completion, all timed passes */
} BrotligDeviceBatch;

/* Decodes every shard, one host thread per shard (shard 0 on the calling thread), each thread on its shard's device:
 * `warmup` untimed passes, a rendezvous, then `steps` (>= 1) timed passes and the shard's batch status.  Shards may name
 * the same device (they then share it; use distinct hip_streams).  `struct_bytes` = sizeof(BrotligDeviceBatch).
 * *max_kernel_ms / *max_wall_ms (may be NULL): maxima over the shards -- the time of the whole job is the slowest device's.
 * Returns the first shard error, BROTLIG_OK if none.  The calling thread's current device is restored. */
BROTLIG_ERROR BrotligDecodeBatchMultiDevice(BrotligDeviceBatch* shards, uint32_t num_shards, uint32_t struct_bytes,
                                            uint32_t warmup, uint32_t steps, double* max_kernel_ms, double* max_wall_ms);

/* Non-blocking pair for a caller that owns its own threads and streams: ...Async enqueues one BrotligDecodeBatchDevice per
 * shard on the shard's device and hip_stream and returns at once (no host thread is created, nothing is timed; `result`
 * holds the enqueue result); ...Wait waits for each shard's stream and stores its batch status in `result`.  Both return
 * the first shard error and restore the calling thread's current device.  Equivalent by hand: hipSetDevice +
 * BrotligDecodeBatchDevice per shard, later BrotligDecodeBatchStatus per shard. */
BROTLIG_ERROR BrotligDecodeBatchMultiDeviceAsync(BrotligDeviceBatch* shards, uint32_t num_shards, uint32_t struct_bytes);
BROTLIG_ERROR BrotligDecodeBatchMultiDeviceWait(BrotligDeviceBatch* shards, uint32_t num_shards, uint32_t struct_bytes);

/* Device self-test of the wave primitives the kernels rely on (DPP scan vs shuffle scan,
 * half-wave ballot / shuffle / max).  Returns BROTLIG_OK when they agree. */
BROTLIG_ERROR BrotligDeviceSelfTest(void);

/* Diagnostics: decodes the batch once with the phase-timer twin of the decode kernel and returns
 * per-phase shader-clock sums over all page pairs (s_memtime deltas of lane 0 of every wave).
 * cycles_out[0..11] = setup, tables, commands, ring, positions, literals, copy-fence, copy-levels,
 * delta, total, number of rounds, number of copy levels, then the level sub-phases (wide short
 * copies, byte-wise copies, long copies) and rounds assembled in global memory.  Synchronous.
 * Round 5: entries beyond the phase sums (27 of them; api.py BatchDecoder.PHASES) are {first, last} tick of a 100 MHz counter for
 * every wavefront of the launch, 2 x BrotligKernelGridSize() values: how the launch ramps up and how long its tail is. */
BROTLIG_ERROR BrotligDecodePhaseProfile(const void* d_in, uint64_t in_bytes, void* d_out, uint64_t out_bytes,
                                        const BrotligStreamDesc* d_streams, uint32_t num_streams,
                                        void* d_workspace, size_t workspace_bytes, void* d_scratch,
                                        uint64_t* cycles_out, uint32_t n_out);

/* ---- Streaming front end -------------------------------------------------
 * Host side of the use the shader's stream queue was designed for (`meta` buffer of up to 4096 streams
 * per launch, src/decoder/BrotliGCompute.hlsl:1757-1881, inc/common/BrotligConstants.h:127-129), which the
 * reference's sample never exercises: it decodes one stream per synchronous call
 * (sample/BrotligGPUDecoder.cpp:260-748, called from sample/brotlig_cli.cpp:436-446).
 * A ring of `num_slots` slots, each with pinned host staging, device buffers and its own hipStream_t:
 * a submitted batch is uploaded, decoded and downloaded asynchronously, so consecutive batches overlap
 * (upload k+1 | decode k | download k-1).  One host thread per streamer; no global state.
 *   slot_in_bytes / slot_out_bytes: capacity of one batch (sum of stream sizes / of NumPages*PageSize).
 *   Submit copies the streams into pinned memory and returns at once with a ticket (it blocks only
 *   when every slot is in flight: then the oldest batch is completed first -- its outputs[] are filled and
 *   its result stays available to one later Wait).  A batch that Submit refuses (bad header, too large)
 *   leaves the ring untouched.  outputs[i] may be NULL (or `outputs` itself NULL): the decoded bytes then
 *   stay in the slot's pinned buffer, readable through BrotligStreamerOutput until the slot is reused
 *   (num_slots submissions later; after that Output returns NULL).
 *   Wait returns the batch's result (BROTLIG_OK / CORRUPT_STREAM / GENERIC) after copying to outputs[];
 *   for a batch displaced by a later Submit it returns the result recorded then (one generation back). */
typedef struct BrotligStreamer BrotligStreamer;
BROTLIG_ERROR BrotligStreamerCreate(uint32_t num_slots, uint64_t slot_in_bytes, uint64_t slot_out_bytes,
                                    uint32_t max_streams_per_batch, BrotligStreamer** out);
void BrotligStreamerDestroy(BrotligStreamer* streamer);
BROTLIG_ERROR BrotligStreamerSubmit(BrotligStreamer* streamer, uint32_t num_streams, const uint8_t* const* inputs,
                                    const uint32_t* input_sizes, uint8_t* const* outputs, const uint32_t* output_caps,
                                    uint64_t* ticket);
BROTLIG_ERROR BrotligStreamerWait(BrotligStreamer* streamer, uint64_t ticket);
const uint8_t* BrotligStreamerOutput(BrotligStreamer* streamer, uint64_t ticket, uint32_t index, uint32_t* size);
/* Round 5: the result of ONE stream of a batch (BrotligDecodeBatchStreamStatus for the streaming caller).  When Wait reports an error the
 * batch's undamaged streams are still delivered -- their outputs[] are filled, BrotligStreamerOutput returns their bytes -- and this names
 * the damaged ones: BROTLIG_OK / BROTLIG_ERROR_CORRUPT_STREAM / BROTLIG_ERROR_GENERIC for stream `index` of the batch; waits for the batch
 * like Wait; BROTLIG_ERROR_GENERIC for a ticket or index it does not know (valid as long as Wait is: the batch in the slot, or the one it
 * displaced). */
BROTLIG_ERROR BrotligStreamerStreamResult(BrotligStreamer* streamer, uint64_t ticket, uint32_t index);

/* Round 6: DEVICE-OUTPUT mode -- the consumer of the reference's GPU path is on the GPU (the shader's `output` UAV is what the renderer reads,
 * src/decoder/BrotliGCompute.hlsl:93-95; the sample only copies it back to compare, sample/BrotligGPUDecoder.cpp:635-675).  A streamer created
 * with BrotligStreamerCreateDeviceOutput downloads nothing but the status words and allocates no pinned staging for decoded bytes: Submit
 * (with outputs == NULL) uploads the compressed streams and enqueues the decode; BrotligStreamerDeviceOutput then hands back, WITHOUT
 * waiting, where stream `index` of the batch will be -- device pointer, size, and the hipEvent_t recorded behind the batch's kernels.  A
 * consumer enqueues hipStreamWaitEvent(its stream, event) and reads; when its work on the batch is enqueued it calls
 * BrotligStreamerConsumerDone(streamer, ticket, its stream), and the batch that reuses the slot (num_slots submissions later) waits for
 * that point on the device.  The pointers stay valid until that batch is submitted.  Results per batch and per stream as in host mode
 * (Wait / StreamResult: a damaged stream's bytes are undefined, the others are good); BrotligStreamerOutput returns NULL in this mode. */
BROTLIG_ERROR BrotligStreamerCreateDeviceOutput(uint32_t num_slots, uint64_t slot_in_bytes, uint64_t slot_out_bytes,
                                                uint32_t max_streams_per_batch, BrotligStreamer** out);
BROTLIG_ERROR BrotligStreamerDeviceOutput(BrotligStreamer* streamer, uint64_t ticket, uint32_t index,
                                          void** d_ptr, uint32_t* size, void** hip_event);
/* the same wait for a caller that does not handle HIP events itself: everything enqueued on `hip_stream` after this call sees the batch decoded */
BROTLIG_ERROR BrotligStreamerStreamWait(BrotligStreamer* streamer, uint64_t ticket, void* hip_stream);
BROTLIG_ERROR BrotligStreamerConsumerDone(BrotligStreamer* streamer, uint64_t ticket, void* hip_stream);

/* Round 6: the compressed bytes written where the upload reads them.  Submit copies every stream into pinned memory first (one host thread:
 * ~5 ms for a 54 MiB batch whose upload takes 1 ms and whose decode 0.5).  A loader that READS its streams -- from disk, from a network -- can
 * read them straight into that memory: Acquire hands out the staging area of the slot the next batch goes to (completing the batch the slot
 * still holds, exactly as Submit would), the caller places its streams there at 16-byte aligned, ascending offsets, and SubmitInPlace
 * validates the headers where they lie and enqueues upload and decode.  A refused SubmitInPlace leaves the area acquired (repair and retry);
 * between Acquire and a successful SubmitInPlace no other Submit is accepted.  Either mode (host or device output). */
BROTLIG_ERROR BrotligStreamerAcquire(BrotligStreamer* streamer, uint8_t** staging, uint64_t* capacity);
BROTLIG_ERROR BrotligStreamerSubmitInPlace(BrotligStreamer* streamer, uint32_t num_streams, const uint64_t* offsets, const uint32_t* input_sizes,
                                           uint8_t* const* outputs, const uint32_t* output_caps, uint64_t* ticket);

/* Diagnostics, for tests.  Both switches below are INERT unless the process was started with BROTLIG_ENABLE_DEBUG_KNOBS=1 in its
 * environment (read once; BrotligDebugKnobsEnabled() answers): a production process cannot have its kernel selection changed by a stray
 * call or a forgotten reset.
 * Decode with exactly `workgroups` wavefronts (0 = the normal rule: one per page while the batch has no more
 * pages than the device holds wavefronts, the full grid otherwise), so that a small batch can exercise the two-pages-per-wavefront
 * path as well as the one-page path it gets by default.  Process-wide. */
void BrotligDebugSetDecodeGrid(uint32_t workgroups);
/* Diagnostics, for tests and the latency report: which decode kernel a launch uses -- 0 = the normal rule (two wavefronts per page
 * for batches of up to 2 048 pages, one wavefront per one or two pages otherwise), 1 = never the two-wavefront
 * kernel, 2 = always.  Process-wide. */
void BrotligDebugSetDecodeMode(uint32_t mode);
uint32_t BrotligDebugKnobsEnabled(void);

/* Static properties, for reports: LDS bytes per workgroup, workgroups launched. */
uint32_t BrotligKernelLdsBytes(void);
uint32_t BrotligKernelGridSize(void);

#ifdef __cplusplus
}
#endif
#endif

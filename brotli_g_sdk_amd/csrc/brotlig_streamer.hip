// brotlig_streamer.hip -- streaming front end over the batch decode entry (include/brotlig_amd.h).
//
// The reference's shader consumes a queue of whole streams from its `meta` buffer, up to 4096 per
// launch (BrotliGCompute.hlsl:1757-1881, inc/common/BrotligConstants.h:127-129): the DirectStorage-style
// use, where compressed assets arrive from disk in batches and decoded bytes are wanted back on the
// host or left on the device.  The sample only ever decodes one stream per call, synchronously
// (sample/BrotligGPUDecoder.cpp:260-748).  This file is the asynchronous host side of that protocol
// for HIP: a ring of slots, each with pinned host staging, device buffers and its own hipStream_t.
// A submitted batch is packed into the slot's pinned input, copied host-to-device, decoded and copied
// back, all asynchronously on the slot's stream, so that slot k+1's upload overlaps slot k's decode
// and slot k-1's download (the copy engines and the compute queue run side by side).
//
// Round 6: DEVICE-OUTPUT mode (BrotligStreamerCreateDeviceOutput).  What the shader's queue was built for ends in GPU memory -- DirectStorage
// hands the decoded asset to the renderer where it lies (BrotliGCompute.hlsl:93-95 `output` UAV; the sample only reads it back to compare,
// sample/BrotligGPUDecoder.cpp:635-675).  In that mode nothing is downloaded but the status words: every stream of a batch is handed back as
// {device pointer, size, hipEvent_t} (BrotligStreamerDeviceOutput) as soon as Submit returns, a consumer kernel waits for the event on its
// own stream (hipStreamWaitEvent) and tells the streamer when it is done with the slot (BrotligStreamerConsumerDone), and the front end is
// bound by the upload of the COMPRESSED bytes alone.
//
// Layered on the public C entries (BrotligDecodeBatchDevice / BrotligDecodeBatchStatus) plus one internal helper of the same
// library (brotlig::enqueue_stream_status_copy, the asynchronous half of BrotligDecodeBatchStreamStatus): nothing here decodes, and
// nothing here knows the kernels or the workspace layout.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

#include "brotlig_amd.h"
#include "brotlig_format.h"
#include "brotlig_internal.h"

using namespace brotlig;

namespace {

#define HIP_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { \
    fprintf(stderr, "brotlig_streamer: %s failed: %s\n", #expr, hipGetErrorString(_e)); return BROTLIG_ERROR_GENERIC; } } while (0)

constexpr uint64_t kAlign = 16;
inline uint64_t align_up(uint64_t v, uint64_t a) { return (v + a - 1) / a * a; }

struct Slot {
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;
    hipEvent_t decoded = nullptr;       // device-output mode: recorded behind the batch's kernels (timing disabled) -- what a consumer's stream waits for
    hipEvent_t consumed = nullptr;      // device-output mode: recorded by BrotligStreamerConsumerDone on the consumer's stream; the slot's next batch waits for it
    bool consumer_pending = false;
    uint8_t* h_in = nullptr;            // pinned: packed streams, then the descriptor array
    uint8_t* h_out = nullptr;           // pinned: decoded bytes as laid out on the device
    uint32_t* h_status = nullptr;       // pinned: status word of the batch, then one status word per stream
    uint8_t* d_in = nullptr;
    uint8_t* d_out = nullptr;
    uint8_t* d_scratch = nullptr;       // allocated on the first pre-conditioned batch
    void* d_ws = nullptr;
    // the batch in flight
    bool busy = false;
    uint64_t ticket = 0;
    uint32_t n = 0;
    std::vector<uint64_t> out_off;
    std::vector<uint32_t> out_size;
    std::vector<uint8_t*> user_out;
    BROTLIG_ERROR result = BROTLIG_OK;
    std::vector<int32_t> stream_result;         // BROTLIG_ERROR per stream, valid once the batch is finished
    // the batch this slot held before the current one, when Submit had to complete it to make room:
    // its result stays available to Wait (its outputs[] were filled at that point)
    uint64_t evicted_ticket = 0;
    BROTLIG_ERROR evicted_result = BROTLIG_OK;
    std::vector<int32_t> evicted_stream_result;
};

}  // namespace

struct BrotligStreamer {
    uint32_t num_slots = 0;
    uint64_t slot_in = 0, slot_out = 0;
    uint32_t max_streams = 0;
    bool device_output = false;         // round 6: decoded bytes stay in device memory
    bool acquired = false;              // round 6: the next slot's staging area is in the caller's hands (BrotligStreamerAcquire)
    size_t ws_bytes = 0;
    uint64_t next_ticket = 1;
    std::vector<Slot> slots;
    uint64_t batches = 0, bytes_in = 0, bytes_out = 0;
};

namespace {

void release(Slot& s)
{
    if (s.stream) (void)hipStreamSynchronize(s.stream);
    if (s.h_in) (void)hipHostFree(s.h_in);
    if (s.h_out) (void)hipHostFree(s.h_out);
    if (s.h_status) (void)hipHostFree(s.h_status);
    if (s.d_in) (void)hipFree(s.d_in);
    if (s.d_out) (void)hipFree(s.d_out);
    if (s.d_scratch) (void)hipFree(s.d_scratch);
    if (s.d_ws) (void)hipFree(s.d_ws);
    if (s.done) (void)hipEventDestroy(s.done);
    if (s.decoded) (void)hipEventDestroy(s.decoded);
    if (s.consumed) (void)hipEventDestroy(s.consumed);
    if (s.stream) (void)hipStreamDestroy(s.stream);
    s = Slot{};
}

// Waits for the slot's batch, hands the decoded bytes to the caller's buffers, frees the slot.
BROTLIG_ERROR finish(Slot& s)
{
    if (!s.busy) return s.result;
    BROTLIG_ERROR err = BROTLIG_OK;
    if (hipEventSynchronize(s.done) != hipSuccess) err = BROTLIG_ERROR_GENERIC;
    auto to_error = [](uint32_t st) {                                     // same mapping as BrotligDecodeBatchStatus
        return (st & kStatusBadHeader) ? BROTLIG_ERROR_CORRUPT_STREAM : (st & kStatusBadPage) ? BROTLIG_ERROR_GENERIC : BROTLIG_OK;
    };
    s.stream_result.assign(s.n, (int32_t)BROTLIG_ERROR_GENERIC);
    if (err == BROTLIG_OK) {
        err = to_error(s.h_status[0]);
        // the undamaged streams of a batch are delivered whatever happened to their neighbours (round 5: a status word per stream)
        for (uint32_t i = 0; i < s.n; ++i) {
            s.stream_result[i] = (int32_t)to_error(s.h_status[1u + i]);
            if (s.stream_result[i] == BROTLIG_OK && s.user_out[i] && s.h_out) memcpy(s.user_out[i], s.h_out + s.out_off[i], s.out_size[i]);
        }
    }
    s.busy = false;
    s.result = err;
    return err;
}

}  // namespace

namespace {

BROTLIG_ERROR create_streamer(uint32_t num_slots, uint64_t slot_in_bytes, uint64_t slot_out_bytes, uint32_t max_streams_per_batch,
                              bool device_output, BrotligStreamer** out)
{
    if (!out || num_slots == 0 || num_slots > 16 || slot_in_bytes == 0 || slot_out_bytes == 0 ||
        max_streams_per_batch == 0 || max_streams_per_batch > 4096) return BROTLIG_ERROR_GENERIC;   // Constants.h:128
    BrotligStreamer* st = new (std::nothrow) BrotligStreamer;
    if (!st) return BROTLIG_ERROR_GENERIC;
    st->num_slots = num_slots;
    st->device_output = device_output;
    st->slot_in = align_up(slot_in_bytes, kAlign) + kAlign * max_streams_per_batch;     // per-stream alignment padding
    st->slot_out = align_up(slot_out_bytes, kAlign) + kAlign * max_streams_per_batch;
    st->max_streams = max_streams_per_batch;
    st->ws_bytes = BrotligDecodeWorkspaceSizeFor(max_streams_per_batch, st->slot_out);
    st->slots.resize(num_slots);
    const uint64_t desc_bytes = sizeof(BrotligStreamDesc) * (uint64_t)max_streams_per_batch;
    for (Slot& s : st->slots) {
        bool ok = hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking) == hipSuccess &&
                  hipEventCreateWithFlags(&s.done, hipEventDisableTiming) == hipSuccess &&
                  hipHostMalloc(reinterpret_cast<void**>(&s.h_in), st->slot_in + desc_bytes + 64, hipHostMallocDefault) == hipSuccess &&
                  // (device-output mode: no pinned staging for the decoded bytes -- nothing of them is downloaded)
                  (device_output || hipHostMalloc(reinterpret_cast<void**>(&s.h_out), st->slot_out + 64, hipHostMallocDefault) == hipSuccess) &&
                  (!device_output || (hipEventCreateWithFlags(&s.decoded, hipEventDisableTiming) == hipSuccess &&
                                      hipEventCreateWithFlags(&s.consumed, hipEventDisableTiming) == hipSuccess)) &&
                  hipHostMalloc(reinterpret_cast<void**>(&s.h_status), sizeof(uint32_t) * (1u + (size_t)max_streams_per_batch) + 64, hipHostMallocDefault) == hipSuccess &&
                  hipMalloc(reinterpret_cast<void**>(&s.d_in), st->slot_in + desc_bytes + 64) == hipSuccess &&
                  hipMalloc(reinterpret_cast<void**>(&s.d_out), st->slot_out + 64) == hipSuccess &&
                  hipMalloc(&s.d_ws, st->ws_bytes) == hipSuccess;
        if (!ok) {
            fprintf(stderr, "brotlig_streamer: allocation failed: %s\n", hipGetErrorString(hipGetLastError()));
            for (Slot& t : st->slots) release(t);
            delete st;
            return BROTLIG_ERROR_GENERIC;
        }
    }
    *out = st;
    return BROTLIG_OK;
}

}  // namespace

extern "C" BROTLIG_ERROR BrotligStreamerCreate(uint32_t num_slots, uint64_t slot_in_bytes, uint64_t slot_out_bytes,
                                               uint32_t max_streams_per_batch, BrotligStreamer** out)
{
    return create_streamer(num_slots, slot_in_bytes, slot_out_bytes, max_streams_per_batch, false, out);
}

extern "C" BROTLIG_ERROR BrotligStreamerCreateDeviceOutput(uint32_t num_slots, uint64_t slot_in_bytes, uint64_t slot_out_bytes,
                                                           uint32_t max_streams_per_batch, BrotligStreamer** out)
{
    return create_streamer(num_slots, slot_in_bytes, slot_out_bytes, max_streams_per_batch, true, out);
}

extern "C" void BrotligStreamerDestroy(BrotligStreamer* st)
{
    if (!st) return;
    for (Slot& s : st->slots) release(s);
    delete st;
}

namespace {

// Validates the headers (src/BrotligDecoder.cpp:437-446) and lays a batch out.  `where[i]`: the stream's bytes; for a batch that is already
// in the slot's pinned staging area (in_place) also its offset there -- 16-byte aligned, inside the area, in ascending order without overlap.
struct Layout {
    std::vector<uint64_t> in_off, out_off;
    std::vector<uint32_t> out_size;
    uint64_t in_pos = 0, out_pos = 0, sum_in = 0, sum_out = 0;
    bool precon = false;
};
BROTLIG_ERROR lay_out(const BrotligStreamer* st, uint32_t n, const uint8_t* const* where, const uint32_t* input_sizes, const uint64_t* in_place_offsets,
                      uint8_t* const* outputs, const uint32_t* output_caps, Layout& L)
{
    L.in_off.resize(n); L.out_off.resize(n); L.out_size.resize(n);
    for (uint32_t i = 0; i < n; ++i) {
        if (!where[i] || input_sizes[i] < 12) return BROTLIG_ERROR_CORRUPT_STREAM;
        uint32_t w0, w1;
        memcpy(&w0, where[i], 4); memcpy(&w1, where[i] + 4, 4);
        if ((w0 & 0xFF) != (((w0 >> 8) & 0xFF) ^ 0xFF)) return BROTLIG_ERROR_CORRUPT_STREAM;
        StreamInfo si;
        if (!parse_stream_header(w0, w1, si)) return BROTLIG_ERROR_INCORRECT_STREAM_FORMAT;
        const uint32_t usize = uncompressed_size(si);
        if (outputs && outputs[i] && (!output_caps || output_caps[i] < usize)) return BROTLIG_ERROR_GENERIC;
        const uint64_t in_need = align_up(input_sizes[i], kAlign);
        const uint64_t out_need = align_up((uint64_t)si.num_pages * si.page_size, kAlign);
        if (in_place_offsets) {
            const uint64_t o = in_place_offsets[i];
            if ((o % kAlign) != 0 || o < L.in_pos || o + in_need > st->slot_in) return BROTLIG_ERROR_GENERIC;
            L.in_off[i] = o; L.in_pos = o + in_need;
        } else {
            if (L.in_pos + in_need > st->slot_in) return BROTLIG_ERROR_GENERIC;                 // batch too big for a slot
            L.in_off[i] = L.in_pos; L.in_pos += in_need;
        }
        if (L.out_pos + out_need > st->slot_out) return BROTLIG_ERROR_GENERIC;
        L.out_off[i] = L.out_pos; L.out_size[i] = usize;
        L.out_pos += out_need;
        L.precon = L.precon || si.preconditioned;
        L.sum_in += input_sizes[i]; L.sum_out += usize;
    }
    return BROTLIG_OK;
}

// the slot the next batch goes to, free: should it still hold a batch, that batch completes now -- its outputs[] are filled -- and its
// result is kept for a later Wait on its ticket
Slot& take_next_slot(BrotligStreamer* st)
{
    Slot& s = st->slots[st->next_ticket % st->num_slots];
    if (s.busy || s.ticket != 0) {
        s.evicted_result = finish(s);
        s.evicted_ticket = s.ticket;
        s.evicted_stream_result = s.stream_result;
        s.ticket = 0; s.n = 0;
    }
    return s;
}

// upload, decode, (download): all on the slot's stream.  Should an enqueue fail half way, whatever was already queued is drained before the
// slot is handed back (the next batch reuses its pinned memory).
BROTLIG_ERROR enqueue_batch(BrotligStreamer* st, Slot& s, uint32_t n, const Layout& L, const uint32_t* input_sizes, uint8_t* const* outputs, uint64_t* ticket)
{
    if (L.precon && !s.d_scratch) HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s.d_scratch), st->slot_out + 64));
    s.out_off = L.out_off; s.out_size = L.out_size; s.user_out.assign(n, nullptr);
    BrotligStreamDesc* desc = reinterpret_cast<BrotligStreamDesc*>(s.h_in + st->slot_in);
    for (uint32_t i = 0; i < n; ++i) {
        desc[i].in_offset = L.in_off[i]; desc[i].out_offset = L.out_off[i];
        desc[i].in_size = input_sizes[i];
        desc[i].out_capacity = (i + 1 < n ? L.out_off[i + 1] : L.out_pos) - L.out_off[i];
        s.user_out[i] = outputs ? outputs[i] : nullptr;
    }
    const uint64_t desc_bytes = sizeof(BrotligStreamDesc) * (uint64_t)n;
    BROTLIG_ERROR e = BROTLIG_OK;
    auto hip_ok = [&e](hipError_t r, const char* what) {
        if (r != hipSuccess && e == BROTLIG_OK) {
            fprintf(stderr, "brotlig_streamer: %s failed: %s\n", what, hipGetErrorString(r));
            e = BROTLIG_ERROR_GENERIC;
        }
        return e == BROTLIG_OK;
    };
    // device-output mode: the slot's last batch may still be read by its consumer (BrotligStreamerConsumerDone recorded where it ends)
    if (s.consumer_pending) { hip_ok(hipStreamWaitEvent(s.stream, s.consumed, 0), "wait for the consumer"); s.consumer_pending = false; }
    if (hip_ok(hipMemcpyAsync(s.d_in, s.h_in, L.in_pos, hipMemcpyHostToDevice, s.stream), "upload") &&
        hip_ok(hipMemcpyAsync(s.d_in + st->slot_in, desc, desc_bytes, hipMemcpyHostToDevice, s.stream), "descriptor upload")) {
        e = BrotligDecodeBatchDevice(s.d_in, L.in_pos, s.d_out, L.out_pos,
                                     reinterpret_cast<const BrotligStreamDesc*>(s.d_in + st->slot_in), n,
                                     s.d_ws, st->ws_bytes, L.precon ? s.d_scratch : nullptr, s.stream);
        if (e == BROTLIG_OK) {
            (st->device_output ? hip_ok(hipEventRecord(s.decoded, s.stream), "event record (decoded)")
                               : hip_ok(hipMemcpyAsync(s.h_out, s.d_out, L.out_pos, hipMemcpyDeviceToHost, s.stream), "download")) &&
            hip_ok(hipMemcpyAsync(s.h_status, s.d_ws, sizeof(uint32_t), hipMemcpyDeviceToHost, s.stream), "status download") &&
            hip_ok(enqueue_stream_status_copy(s.d_ws, n, s.h_status + 1, s.stream), "per-stream status download") &&
            hip_ok(hipEventRecord(s.done, s.stream), "event record");
        }
    }
    if (e != BROTLIG_OK) {
        (void)hipStreamSynchronize(s.stream);
        return e;
    }
    s.busy = true; s.n = n; s.ticket = st->next_ticket; s.result = BROTLIG_OK;
    *ticket = st->next_ticket++;
    ++st->batches; st->bytes_in += L.sum_in; st->bytes_out += L.sum_out;
    return BROTLIG_OK;
}

}  // namespace

extern "C" BROTLIG_ERROR BrotligStreamerSubmit(BrotligStreamer* st, uint32_t n, const uint8_t* const* inputs,
                                               const uint32_t* input_sizes, uint8_t* const* outputs,
                                               const uint32_t* output_caps, uint64_t* ticket)
{
    if (!st || !inputs || !input_sizes || !ticket || n == 0 || n > st->max_streams) return BROTLIG_ERROR_GENERIC;
    if (st->device_output && outputs) return BROTLIG_ERROR_GENERIC;     // (nothing is downloaded in that mode: BrotligStreamerDeviceOutput)
    if (st->acquired) return BROTLIG_ERROR_GENERIC;                     // (an acquired staging area is waiting for its SubmitInPlace)
    // ---- validate and lay out without touching the slot: a refused batch must leave the ring (and the batch it would have displaced) as it was
    Layout L;
    try { if (BROTLIG_ERROR e = lay_out(st, n, inputs, input_sizes, nullptr, outputs, output_caps, L)) return e; } catch (...) { return BROTLIG_ERROR_GENERIC; }
    Slot& s = take_next_slot(st);
    for (uint32_t i = 0; i < n; ++i) {                                  // the slot's pinned bytes are overwritten from here on
        const uint64_t in_need = align_up(input_sizes[i], kAlign);
        memcpy(s.h_in + L.in_off[i], inputs[i], input_sizes[i]);
        memset(s.h_in + L.in_off[i] + input_sizes[i], 0, in_need - input_sizes[i]);
    }
    return enqueue_batch(st, s, n, L, input_sizes, outputs, ticket);
}

// Round 6: the compressed bytes written where the upload reads them.  Submit copies every stream into the slot's pinned staging area first --
// one host thread, ~10 GB/s: 5 ms for a 54 MiB batch whose upload takes 1 and whose decode takes 0.5.  A loader that reads files can read
// them straight INTO that area: Acquire hands out the next slot's staging area (completing the batch it still holds, like Submit), the
// caller places its streams there (16-byte aligned offsets, ascending), SubmitInPlace validates the headers where they lie and enqueues.
extern "C" BROTLIG_ERROR BrotligStreamerAcquire(BrotligStreamer* st, uint8_t** staging, uint64_t* capacity)
{
    if (!st || !staging || !capacity || st->acquired) return BROTLIG_ERROR_GENERIC;
    Slot& s = take_next_slot(st);
    // the upload of the slot's last batch has long finished (its batch was completed above or earlier); the area is the caller's until SubmitInPlace
    st->acquired = true;
    *staging = s.h_in; *capacity = st->slot_in;
    return BROTLIG_OK;
}

extern "C" BROTLIG_ERROR BrotligStreamerSubmitInPlace(BrotligStreamer* st, uint32_t n, const uint64_t* offsets, const uint32_t* input_sizes,
                                                      uint8_t* const* outputs, const uint32_t* output_caps, uint64_t* ticket)
{
    if (!st || !offsets || !input_sizes || !ticket || !st->acquired || n == 0 || n > st->max_streams) return BROTLIG_ERROR_GENERIC;
    if (st->device_output && outputs) return BROTLIG_ERROR_GENERIC;
    Slot& s = st->slots[st->next_ticket % st->num_slots];
    Layout L;
    try {
        std::vector<const uint8_t*> where(n);
        for (uint32_t i = 0; i < n; ++i) { if (offsets[i] > st->slot_in) return BROTLIG_ERROR_GENERIC; where[i] = s.h_in + offsets[i]; }
        if (BROTLIG_ERROR e = lay_out(st, n, where.data(), input_sizes, offsets, outputs, output_caps, L)) return e;     // (still acquired: the caller may repair and retry)
    } catch (...) { return BROTLIG_ERROR_GENERIC; }
    st->acquired = false;
    return enqueue_batch(st, s, n, L, input_sizes, outputs, ticket);
}

extern "C" BROTLIG_ERROR BrotligStreamerWait(BrotligStreamer* st, uint64_t ticket)
{
    if (!st || ticket == 0 || ticket >= st->next_ticket) return BROTLIG_ERROR_GENERIC;
    Slot& s = st->slots[ticket % st->num_slots];
    if (s.ticket == ticket) return finish(s);
    if (s.evicted_ticket == ticket) return s.evicted_result;            // completed by a later Submit; outputs[] are filled
    return BROTLIG_ERROR_GENERIC;                                       // more than one generation old: forgotten
}

extern "C" BROTLIG_ERROR BrotligStreamerStreamResult(BrotligStreamer* st, uint64_t ticket, uint32_t index)
{
    if (!st || ticket == 0 || ticket >= st->next_ticket) return BROTLIG_ERROR_GENERIC;
    Slot& s = st->slots[ticket % st->num_slots];
    if (s.ticket == ticket) {
        (void)finish(s);
        return index < s.stream_result.size() ? (BROTLIG_ERROR)s.stream_result[index] : BROTLIG_ERROR_GENERIC;
    }
    if (s.evicted_ticket == ticket)
        return index < s.evicted_stream_result.size() ? (BROTLIG_ERROR)s.evicted_stream_result[index] : BROTLIG_ERROR_GENERIC;
    return BROTLIG_ERROR_GENERIC;
}

extern "C" const uint8_t* BrotligStreamerOutput(BrotligStreamer* st, uint64_t ticket, uint32_t index, uint32_t* size)
{
    if (!st || ticket == 0 || ticket >= st->next_ticket) return nullptr;
    Slot& s = st->slots[ticket % st->num_slots];
    if (s.ticket != ticket || index >= s.n) return nullptr;
    (void)finish(s);
    if (index >= s.stream_result.size() || s.stream_result[index] != BROTLIG_OK) return nullptr;      // a damaged stream has no bytes to show
    if (!s.h_out) return nullptr;                                       // device-output mode: BrotligStreamerDeviceOutput
    if (size) *size = s.out_size[index];
    return s.h_out + s.out_off[index];
}

extern "C" BROTLIG_ERROR BrotligStreamerDeviceOutput(BrotligStreamer* st, uint64_t ticket, uint32_t index, void** d_ptr, uint32_t* size, void** hip_event)
{
    if (!st || !st->device_output || ticket == 0 || ticket >= st->next_ticket) return BROTLIG_ERROR_GENERIC;
    Slot& s = st->slots[ticket % st->num_slots];
    if (s.ticket != ticket || index >= s.n) return BROTLIG_ERROR_GENERIC;          // the slot has been reused: the bytes are gone
    if (d_ptr) *d_ptr = s.d_out + s.out_off[index];
    if (size) *size = s.out_size[index];
    if (hip_event) *hip_event = s.decoded;
    return BROTLIG_OK;
}

extern "C" BROTLIG_ERROR BrotligStreamerStreamWait(BrotligStreamer* st, uint64_t ticket, void* hip_stream)
{
    if (!st || !st->device_output || ticket == 0 || ticket >= st->next_ticket) return BROTLIG_ERROR_GENERIC;
    Slot& s = st->slots[ticket % st->num_slots];
    if (s.ticket != ticket) return BROTLIG_ERROR_GENERIC;
    HIP_TRY(hipStreamWaitEvent(static_cast<hipStream_t>(hip_stream), s.decoded, 0));
    return BROTLIG_OK;
}

extern "C" BROTLIG_ERROR BrotligStreamerConsumerDone(BrotligStreamer* st, uint64_t ticket, void* hip_stream)
{
    if (!st || !st->device_output || ticket == 0 || ticket >= st->next_ticket) return BROTLIG_ERROR_GENERIC;
    Slot& s = st->slots[ticket % st->num_slots];
    if (s.ticket != ticket) return BROTLIG_ERROR_GENERIC;
    HIP_TRY(hipEventRecord(s.consumed, static_cast<hipStream_t>(hip_stream)));
    s.consumer_pending = true;
    return BROTLIG_OK;
}

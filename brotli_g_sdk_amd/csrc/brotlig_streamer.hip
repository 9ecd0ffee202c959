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
            if (s.stream_result[i] == BROTLIG_OK && s.user_out[i]) memcpy(s.user_out[i], s.h_out + s.out_off[i], s.out_size[i]);
        }
    }
    s.busy = false;
    s.result = err;
    return err;
}

}  // namespace

extern "C" BROTLIG_ERROR BrotligStreamerCreate(uint32_t num_slots, uint64_t slot_in_bytes, uint64_t slot_out_bytes,
                                               uint32_t max_streams_per_batch, BrotligStreamer** out)
{
    if (!out || num_slots == 0 || num_slots > 16 || slot_in_bytes == 0 || slot_out_bytes == 0 ||
        max_streams_per_batch == 0 || max_streams_per_batch > 4096) return BROTLIG_ERROR_GENERIC;   // Constants.h:128
    BrotligStreamer* st = new (std::nothrow) BrotligStreamer;
    if (!st) return BROTLIG_ERROR_GENERIC;
    st->num_slots = num_slots;
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
                  hipHostMalloc(reinterpret_cast<void**>(&s.h_out), st->slot_out + 64, hipHostMallocDefault) == hipSuccess &&
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

extern "C" void BrotligStreamerDestroy(BrotligStreamer* st)
{
    if (!st) return;
    for (Slot& s : st->slots) release(s);
    delete st;
}

extern "C" BROTLIG_ERROR BrotligStreamerSubmit(BrotligStreamer* st, uint32_t n, const uint8_t* const* inputs,
                                               const uint32_t* input_sizes, uint8_t* const* outputs,
                                               const uint32_t* output_caps, uint64_t* ticket)
{
    if (!st || !inputs || !input_sizes || !ticket || n == 0 || n > st->max_streams) return BROTLIG_ERROR_GENERIC;
    // ---- validate the headers (src/BrotligDecoder.cpp:437-446) and lay the batch out, without touching the
    //      slot: a refused batch must leave the ring (and the batch it would have displaced) as it was
    std::vector<uint64_t> in_off(n), out_off(n);
    std::vector<uint32_t> out_size(n);
    uint64_t in_pos = 0, out_pos = 0, sum_in = 0, sum_out = 0;
    bool precon = false;
    for (uint32_t i = 0; i < n; ++i) {
        if (!inputs[i] || input_sizes[i] < 12) return BROTLIG_ERROR_CORRUPT_STREAM;
        uint32_t w0, w1;
        memcpy(&w0, inputs[i], 4); memcpy(&w1, inputs[i] + 4, 4);
        if ((w0 & 0xFF) != (((w0 >> 8) & 0xFF) ^ 0xFF)) return BROTLIG_ERROR_CORRUPT_STREAM;
        StreamInfo si;
        if (!parse_stream_header(w0, w1, si)) return BROTLIG_ERROR_INCORRECT_STREAM_FORMAT;
        const uint32_t usize = uncompressed_size(si);
        if (outputs && outputs[i] && (!output_caps || output_caps[i] < usize)) return BROTLIG_ERROR_GENERIC;
        const uint64_t in_need = align_up(input_sizes[i], kAlign);
        const uint64_t out_need = align_up((uint64_t)si.num_pages * si.page_size, kAlign);
        if (in_pos + in_need > st->slot_in || out_pos + out_need > st->slot_out) return BROTLIG_ERROR_GENERIC;   // batch too big for a slot
        in_off[i] = in_pos; out_off[i] = out_pos; out_size[i] = usize;
        in_pos += in_need; out_pos += out_need;
        precon = precon || si.preconditioned;
        sum_in += input_sizes[i]; sum_out += usize;
    }

    Slot& s = st->slots[st->next_ticket % st->num_slots];
    if (s.busy || s.ticket != 0) {
        // ring full (or the slot's last batch was never waited for): that batch completes now -- its outputs[]
        // are filled -- and its result is kept for a later Wait on its ticket
        s.evicted_result = finish(s);
        s.evicted_ticket = s.ticket;
        s.evicted_stream_result = s.stream_result;
    }
    if (precon && !s.d_scratch) HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s.d_scratch), st->slot_out + 64));
    s.ticket = 0; s.n = 0;                                              // the slot's pinned bytes are overwritten from here on
    s.out_off = out_off; s.out_size = out_size; s.user_out.assign(n, nullptr);
    BrotligStreamDesc* desc = reinterpret_cast<BrotligStreamDesc*>(s.h_in + st->slot_in);
    for (uint32_t i = 0; i < n; ++i) {
        const uint64_t in_need = align_up(input_sizes[i], kAlign);
        memcpy(s.h_in + in_off[i], inputs[i], input_sizes[i]);
        memset(s.h_in + in_off[i] + input_sizes[i], 0, in_need - input_sizes[i]);
        desc[i].in_offset = in_off[i]; desc[i].out_offset = out_off[i];
        desc[i].in_size = input_sizes[i];
        desc[i].out_capacity = (i + 1 < n ? out_off[i + 1] : out_pos) - out_off[i];
        s.user_out[i] = outputs ? outputs[i] : nullptr;
    }

    // ---- upload, decode, download: all on the slot's stream.  Should an enqueue fail half way, whatever was
    //      already queued is drained before the slot is handed back (the next batch reuses its pinned memory).
    const uint64_t desc_bytes = sizeof(BrotligStreamDesc) * (uint64_t)n;
    BROTLIG_ERROR e = BROTLIG_OK;
    auto hip_ok = [&e](hipError_t r, const char* what) {
        if (r != hipSuccess && e == BROTLIG_OK) {
            fprintf(stderr, "brotlig_streamer: %s failed: %s\n", what, hipGetErrorString(r));
            e = BROTLIG_ERROR_GENERIC;
        }
        return e == BROTLIG_OK;
    };
    if (hip_ok(hipMemcpyAsync(s.d_in, s.h_in, in_pos, hipMemcpyHostToDevice, s.stream), "upload") &&
        hip_ok(hipMemcpyAsync(s.d_in + st->slot_in, desc, desc_bytes, hipMemcpyHostToDevice, s.stream), "descriptor upload")) {
        e = BrotligDecodeBatchDevice(s.d_in, in_pos, s.d_out, out_pos,
                                     reinterpret_cast<const BrotligStreamDesc*>(s.d_in + st->slot_in), n,
                                     s.d_ws, st->ws_bytes, precon ? s.d_scratch : nullptr, s.stream);
        if (e == BROTLIG_OK) {
            hip_ok(hipMemcpyAsync(s.h_out, s.d_out, out_pos, hipMemcpyDeviceToHost, s.stream), "download") &&
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
    ++st->batches; st->bytes_in += sum_in; st->bytes_out += sum_out;
    return BROTLIG_OK;
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
    if (size) *size = s.out_size[index];
    return s.h_out + s.out_off[index];
}

// brotlig_internal.h -- what the two translation units of libbrotlig_hip.so share beside the public header.  Not installed, not part of the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace brotlig {

// Asynchronous half of BrotligDecodeBatchStreamStatus (include/brotlig_amd.h): enqueues the copy of the `num_streams` per-stream status
// words (kStatus* bits, brotlig_format.h) of the batch that used `d_workspace` into `h_words` -- pinned host memory that must stay valid
// until the stream reaches this point.  Defined in brotlig_hip.hip, which owns the workspace layout.
hipError_t enqueue_stream_status_copy(const void* d_workspace, uint32_t num_streams, uint32_t* h_words, hipStream_t stream);

}  // namespace brotlig

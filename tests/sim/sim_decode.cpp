// TEST INFRASTRUCTURE: runs the HIP decode kernels (brotli_g_sdk_amd/csrc/brotlig_kernels.h) on
// the CPU simulator.  Built as a shared library and driven from tests/test_sim_decode.py.
#include <cstring>
#include <vector>

#include <brotlig_wave_ops.h>

#include "brotlig_kernels.h"
#ifdef BROTLIG_WITH_SPLIT       // the two-kernel experiment of round 3 (profiles/experiments/split_path/): only tests/test_sim_split.py builds it in
#include "brotlig_split_kernels.h"
#endif

using namespace brotlig;

static uint32_t g_last_policy = 0;
static int g_use_order = 1;      // page schedule on (the GPU host code only uses it for large batches)
static uint16_t g_order_from_k = 0;   // DecodeArgs::order_from_k: 0 = every batch gets the schedule proper (the host: from 12 x 1 024 pages on)
static void prepare_body(void* p) { brotlig_prepare_kernel(*(DecodeArgs*)p); }
static void prepare_finish_body(void* p) { brotlig_prepare_finish_kernel(*(DecodeArgs*)p); }
// the host's launch_prepare (csrc/brotlig_hip.hip): one workgroup per 64 streams, the second kernel for more than 64 streams
static void run_prepare(DecodeArgs& a)
{
    const uint32_t groups = (a.num_streams + 63u) / 64u;
    sim::run_grid(groups, prepare_body, &a);
    if (groups > 1u) sim::run_grid(groups, prepare_finish_body, &a);
}
static void order_count_body(void* p) { brotlig_order_count_kernel(*(DecodeArgs*)p); }
static void order_scatter_body(void* p) { brotlig_order_scatter_kernel(*(DecodeArgs*)p); }
static void policy_body(void* p) { brotlig_policy_kernel(*(DecodeArgs*)p); }
static void decode_body(void* p) { brotlig_decode_kernel(*(DecodeArgs*)p); }
static void duo_body(void* p) { brotlig_decode_duo_kernel(*(DecodeArgs*)p); }
static int g_duo = 0;          // small-batch form: two wavefronts per page (brotlig_decode_duo_kernel)
static uint64_t g_duo_launches = 0;
extern "C" void sim_set_duo(int on) { g_duo = on; }
extern "C" uint64_t sim_duo_launches() { return g_duo_launches; }
static void decond_body(void* p) { brotlig_decondition_kernel(*(DecodeArgs*)p); }
static uint32_t g_decond_gx = 3, g_decond_gy = 0;      // workgroups of the de-conditioning kernel: gx x gy (one row; the two factors are history)
extern "C" void sim_set_decond_grid(uint32_t gx, uint32_t gy) { g_decond_gx = gx ? gx : 3u; g_decond_gy = gy; }
static void selftest_body(void* p) { brotlig_selftest_kernel((uint32_t*)p); }
// the last batch's per-stream status words (DcTable::status; BrotligDecodeBatchStreamStatus reads the same words on the device)
static std::vector<uint32_t> g_stream_status;
extern "C" uint32_t sim_stream_status(uint32_t i) { return i < g_stream_status.size() ? g_stream_status[i] : 0xFFFFFFFFu; }
// Host rule (csrc/brotlig_hip.hip enqueue()): 0 = one kernel pinned by sim_set_duo, otherwise the page limit up to which a batch belongs to the
// two-wavefront kernel -- BOTH kernels are then run back to back, each decides by DecodeArgs::duo_limit, and the policy kernel is skipped
// when no two pages can meet in a wavefront, exactly as the host does it.
static uint32_t g_host_rule_limit = 0;
static uint32_t g_pages_by_duo = 0, g_pages_by_classic = 0;     // what each kernel took from the page counter in the last batch
extern "C" void sim_set_host_rule(uint32_t duo_limit) { g_host_rule_limit = duo_limit; }
extern "C" uint32_t sim_counter_after_duo() { return g_pages_by_duo; }
extern "C" uint32_t sim_counter_after_classic() { return g_pages_by_classic; }

extern "C" int sim_decode_batch(const uint8_t* in, uint64_t in_bytes, uint8_t* out, uint64_t out_bytes,
                                uint8_t* scratch, const uint64_t* in_offsets, const uint64_t* out_offsets,
                                uint32_t num_streams, uint32_t grid, uint32_t* status_out)
{
    std::vector<StreamDesc> sd(num_streams);
    for (uint32_t i = 0; i < num_streams; ++i) {
        sd[i].in_offset = in_offsets[i]; sd[i].out_offset = out_offsets[i];
        sd[i].in_size = (i + 1 < num_streams ? in_offsets[i + 1] : in_bytes) - in_offsets[i];        // streams lie back to back
        sd[i].out_capacity = (i + 1 < num_streams ? out_offsets[i + 1] : out_bytes) - out_offsets[i];
    }
    std::vector<uint32_t> page_base(num_streams + 1, 0);
    uint32_t counter = 0;
    uint32_t status_words[kStatusWords] = {0};
    std::vector<DcTable> dc(num_streams);
    DecodeArgs a{};
    a.in = in; a.in_bytes = in_bytes; a.out = out; a.out_bytes = out_bytes; a.scratch = scratch;
    a.streams = sd.data(); a.num_streams = num_streams;
    a.page_base = page_base.data(); a.work_counter = &counter; a.status = status_words; a.dc = dc.data();
    std::vector<uint32_t> order(g_use_order ? (size_t)(out_bytes / 32768 + num_streams + 1) : 0);
    if (g_use_order) { a.order = order.data(); a.order_cap = (uint32_t)order.size(); }
    const int decode_grid = grid ? grid : 4;
    a.order_from_k = g_order_from_k;
    a.decode_waves = (uint16_t)decode_grid;     // (the schedule proper for every batch, order_from_k = 0 -- except the batches schedule_mode folds)
    std::vector<uint16_t> far_syms((size_t)decode_grid * 2u * kFarSymStride, 0xFFFFu);     // stale garbage between pages, as on the device
    a.far_syms = far_syms.data();
    run_prepare(a);
    if (a.order) { sim::run_grid(3, order_count_body, &a); sim::run_grid(3, order_scatter_body, &a); }
    if (g_host_rule_limit) {
        // as enqueue(): the policy kernel only when two pages can meet in a wavefront; both decode kernels, the device decides
        const uint64_t bound = out_bytes / kMinPageSize + num_streams;
        if (bound > (uint64_t)decode_grid) sim::run_grid(1, policy_body, &a);
        a.duo_limit = g_host_rule_limit;
        ++g_duo_launches;
        sim::run_grid(decode_grid, duo_body, &a, 2);
        g_pages_by_duo = counter;
        sim::run_grid(decode_grid, decode_body, &a);
        g_pages_by_classic = counter - g_pages_by_duo;
    } else {
        sim::run_grid(1, policy_body, &a);
        if (g_duo) { ++g_duo_launches; a.duo_limit = 0xFFFFFFFFu; sim::run_grid(decode_grid, duo_body, &a, 2); a.duo_limit = 0u; }
        else sim::run_grid(decode_grid, decode_body, &a);
    }
    // the de-conditioning launch (csrc/brotlig_hip.hip enqueue()): one row of workgroups, the batch's super-tiles cut evenly over them
    sim::run_grid(g_decond_gx * (g_decond_gy ? g_decond_gy : 1u), decond_body, &a);
    *status_out = status_words[0];
    g_last_policy = status_words[3];
    g_stream_status.resize(num_streams);
    for (uint32_t i = 0; i < num_streams; ++i) g_stream_status[i] = dc[i].status;
    return 0;
}

#ifdef BROTLIG_WITH_SPLIT
static void entropy_body(void* p) { brotlig_entropy_kernel(*(DecodeArgs*)p); }
static void assemble_body(void* p) { brotlig_assemble_kernel(*(DecodeArgs*)p); }
static void assemble_global_body(void* p) { brotlig_assemble_global_kernel(*(DecodeArgs*)p); }
static void assemble_page_body(void* p) { brotlig_assemble_page_kernel(*(DecodeArgs*)p); }
static int g_run_assemble = 0;
static uint8_t* g_scratch = nullptr;
extern "C" void sim_set_scratch(uint8_t* p) { g_scratch = p; }
extern "C" void sim_set_assemble(int on) { g_run_assemble = on; }

// Split path, first kernel only: the command / literal arrays of every page, for inspection by the tests.
// cmds: [pages][cmd_cap + 1], lits: [pages][lit_stride], hdr: [pages][2]
extern "C" int sim_entropy_batch(const uint8_t* in, uint64_t in_bytes, uint8_t* out, uint64_t out_bytes,
                                 const uint64_t* in_offsets, const uint64_t* out_offsets, uint32_t num_streams, uint32_t grid,
                                 uint64_t* cmds, uint8_t* lits, uint32_t* hdr, uint32_t cmd_cap, uint32_t lit_stride, uint32_t* status_out)
{
    std::vector<StreamDesc> sd(num_streams);
    for (uint32_t i = 0; i < num_streams; ++i) {
        sd[i].in_offset = in_offsets[i]; sd[i].out_offset = out_offsets[i];
        sd[i].in_size = (i + 1 < num_streams ? in_offsets[i + 1] : in_bytes) - in_offsets[i];
        sd[i].out_capacity = (i + 1 < num_streams ? out_offsets[i + 1] : out_bytes) - out_offsets[i];
    }
    std::vector<uint32_t> page_base(num_streams + 1, 0);
    uint32_t counter = 0, counter2 = 0;
    uint32_t status_words[kStatusWords] = {0};
    std::vector<DcTable> dc(num_streams);
    DecodeArgs a{};
    a.in = in; a.in_bytes = in_bytes; a.out = out; a.out_bytes = out_bytes; a.scratch = g_scratch;
    a.streams = sd.data(); a.num_streams = num_streams;
    a.page_base = page_base.data(); a.work_counter = &counter; a.work_counter2 = &counter2; a.status = status_words; a.dc = dc.data();
    const int decode_grid = grid ? grid : 4;
    std::vector<uint16_t> far_syms((size_t)decode_grid * 2u * kFarSymStride, 0xFFFFu);
    a.far_syms = far_syms.data();
    a.cmds = cmds; a.lits = lits; a.slot_hdr = hdr; a.cmd_cap = cmd_cap; a.lit_stride = lit_stride;
    run_prepare(a);
    sim::run_grid(1, policy_body, &a);
    sim::run_grid(decode_grid, entropy_body, &a);
    if (g_run_assemble == 3) { sim::run_grid(decode_grid, assemble_page_body, &a, (int)kPageWaves); sim::run_grid(3, decond_body, &a); }
    else if (g_run_assemble) { sim::run_grid(decode_grid, g_run_assemble == 2 ? assemble_global_body : assemble_body, &a); sim::run_grid(3, decond_body, &a); }
    *status_out = status_words[0];
    return 0;
}
#endif  // BROTLIG_WITH_SPLIT

extern "C" uint32_t sim_last_policy() { return g_last_policy; }   // status word 3: pairing policy chosen by the prepare kernel
extern "C" void sim_set_order(int on) { g_use_order = on; }
extern "C" void sim_set_order_from_k(uint32_t k) { g_order_from_k = (uint16_t)k; }
extern "C" void sim_selftest(uint32_t* out) { sim::run_grid(1, selftest_body, out); }
extern "C" uint64_t sim_collectives() { return sim::g_waves[0].n_collectives; }

// The kernels' own view of the two headers, for the constant checks of tests/test_reference_kats.py.
extern "C" int sim_dc_table(uint32_t w0, uint32_t w1, uint32_t out_size, uint32_t* words /* sizeof(DcTable) / 4 */)
{
    DcTable t{};
    const bool ok = dc_init(t, w0, w1, out_size);
    memcpy(words, &t, sizeof t);
    return ok ? 1 : 0;
}
extern "C" int sim_stream_header(uint32_t w0, uint32_t w1, uint32_t out[6])
{
    StreamInfo si;
    const bool ok = parse_stream_header(w0, w1, si);
    out[0] = si.num_pages; out[1] = si.page_size; out[2] = si.last_page_size; out[3] = si.preconditioned;
    out[4] = si.header_bytes; out[5] = uncompressed_size(si);
    return ok ? 1 : 0;
}

// TEST INFRASTRUCTURE: runs the HIP decode kernels (brotli_g_sdk_amd/csrc/brotlig_kernels.h) on
// the CPU simulator.  Built as a shared library and driven from tests/test_sim_decode.py.
#include <cstring>
#include <vector>

#include <brotlig_wave_ops.h>

#include "brotlig_kernels.h"

using namespace brotlig;

static uint32_t g_last_policy = 0;
static int g_use_order = 1;      // page schedule on (the GPU host code only uses it for large batches)
static uint16_t g_order_from_k = 0;   // DecodeArgs::order_from_k: 0 = every batch gets the schedule proper (the host: from 12 x 1 024 pages on)
// the host's launch_schedule (csrc/brotlig_hip.hip): one workgroup for a small batch, else one per item of every phase.  g_sched_workers
// pins the count / scatter workers (0: the host's rule with a small device); g_sched_force_tickets sends small batches through tickets too
static int g_fresh_workspace = 0;
extern "C" void sim_set_fresh_workspace(int on) { g_fresh_workspace = on; }
static uint32_t g_sched_workers = 0, g_sched_force_tickets = 0, g_last_sched_grid = 0;
static uint64_t g_launch_tag = 0x5EED000000ull;
extern "C" void sim_set_schedule(uint32_t workers, uint32_t force_tickets) { g_sched_workers = workers; g_sched_force_tickets = force_tickets; }
extern "C" uint32_t sim_last_schedule_grid() { return g_last_sched_grid; }
static void schedule_body(void* p) { brotlig_schedule_kernel(*(DecodeArgs*)p); }
static void run_schedule(DecodeArgs& a)
{
    a.launch_tag = ++g_launch_tag & ((1ull << 40) - 1u);
    const uint64_t pages = a.out_bytes / kMinPageSize + a.num_streams;
    uint32_t grid = 1u;
    if (a.num_streams > 64u || pages > 2048u || g_sched_force_tickets) {
        const uint64_t groups = (pages + kSchedThreads - 1u) / kSchedThreads;
        // (a pinned worker count is taken as it is: more workgroups than groups of pages -- the surplus finds no page and only counts itself done)
        grid = sched_grid(a.num_streams, g_sched_workers ? g_sched_workers : (uint32_t)(groups < 3u ? groups : 3u));
    }
    g_last_sched_grid = grid;
    sim::run_grid(grid, schedule_body, &a, (int)(kSchedThreads / 64u));
}
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
    // the workspace header as the device has it (the schedule kernel needs no memset in front of it): garbage in a fresh workspace -- the first
    // batch of a process here, and every batch after sim_set_fresh_workspace(1) --, what the last batch left otherwise
    static uint32_t stale = 0x1234567u;
    std::vector<uint32_t> page_base(num_streams + 1, stale);
    uint32_t counter = stale;
    alignas(8) static uint32_t status_words[kStatusWords + kSyncWords];
    static bool have_header = false;
    if (!have_header || g_fresh_workspace) for (uint32_t& w : status_words) w = (stale = stale * 1664525u + 1013904223u);
    have_header = true;
    std::vector<DcTable> dc(num_streams);
    DecodeArgs a{};
    a.in = in; a.in_bytes = in_bytes; a.out = out; a.out_bytes = out_bytes; a.scratch = scratch;
    a.streams = sd.data(); a.num_streams = num_streams;
    a.page_base = page_base.data(); a.work_counter = &counter; a.status = status_words; a.sync = status_words + kStatusWords; a.dc = dc.data();
    std::vector<JobRecord> jobs(g_use_order ? (size_t)(out_bytes / 32768 + num_streams + 1) : 0);
    if (g_use_order) { a.jobs = jobs.data(); a.jobs_cap = (uint32_t)jobs.size(); }
    const int decode_grid = grid ? grid : 4;
    a.order_from_k = g_order_from_k;
    a.decode_waves = (uint16_t)decode_grid;     // (the schedule proper for every batch, order_from_k = 0 -- except the batches schedule_mode folds)
    std::vector<uint16_t> far_syms((size_t)decode_grid * 2u * kFarSymStride, 0xFFFFu);     // stale garbage between pages, as on the device
    a.far_syms = far_syms.data();
    const uint64_t bound = out_bytes / kMinPageSize + num_streams;
    a.may_pair = (!g_host_rule_limit || bound > (uint64_t)decode_grid) ? 1u : 0u;       // as enqueue(): the policy only when two pages can meet in a wavefront
    run_schedule(a);
    if (g_host_rule_limit) {
        // as enqueue(): both decode kernels, the device decides
        a.duo_limit = g_host_rule_limit;
        ++g_duo_launches;
        sim::run_grid(decode_grid, duo_body, &a, 2);
        g_pages_by_duo = counter;
        sim::run_grid(decode_grid, decode_body, &a);
        g_pages_by_classic = counter - g_pages_by_duo;
    } else {
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


extern "C" uint32_t sim_last_policy() { return g_last_policy; }   // status word 3: pairing policy chosen by the schedule kernel
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

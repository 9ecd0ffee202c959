// brotlig -- portable command-line tool over this repo's two libraries.
//
// Same switches and file conventions as the reference's Windows sample (sample/brotlig_cli.cpp:174-327):
//   brotlig [Options] filename [outfilename]      compress to filename.brotlig
//   brotlig [Options] filename.brotlig [out]      decompress to filename (or `out`)
// Compression calls BrotligEncode (brotli_g_sdk_amd/csrc/brotlig_encoder.h); decompression calls
// DecodeGPU (include/brotlig_amd.h) and fails if no HIP device is usable (`-gpu` is accepted and redundant, `-warp`
// is ignored); there is no silent fallback.  `-cpu` asks for DecodeCPU (include/brotlig_amd_cpu.h) instead.
// The stock-Brotli switches of the sample (-brotli ...) are not provided.
// Reports sizes, milliseconds and throughput like the sample (sample/brotlig_cli.cpp:626-639: source
// bytes per second in GiB/s), plus the decompressed GB/s for decodes.
//
// Build: brotli_g_sdk_amd/_build.py build_cli() -- g++ against libbrotlig_enc.so, libbrotlig_hip.so and libbrotlig_cpu.so.
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "brotlig_amd.h"
#include "brotlig_amd_cpu.h"
#include "brotlig_encoder.h"

namespace {

struct Options {
    uint32_t page_size = 65536;
    bool precondition = false, swizzle = false, delta = false, pitch_aligned = false;
    uint32_t format = 0, tex_width = 0, tex_height = 0, row_pitch = 0, num_mips = 1;
    uint32_t repeat = 1;
    bool verbose = false, fast = false, cpu = false;
    std::string src, dst;
};

void usage()
{
    printf("Usage: brotlig [Options] filename [outfilename]\n"
           "  filename          -> filename.brotlig (compress)\n"
           "  filename.brotlig  -> filename (decompress on the GPU)\n"
           "Options:\n"
           " -pagesize <value>             : encode page size in bytes: 32768, 65536 (default), 131072 or 262144\n"
           " -precondition                 : apply format-based pre-conditioning before compression\n"
           " -swizzle                      : 2x2 block swizzle (pre-conditioning only)\n"
           " -delta-encode                 : delta-encode the colour endpoints (pre-conditioning only)\n"
           " -data-format <value>          : 1..5 = BC1..BC5, 0 = unknown (pre-conditioning only)\n"
           " -texture-width <value>        : width of the top mip in pixels\n"
           " -texture-height <value>       : height of the top mip in pixels\n"
           " -row-pitch <value>            : row pitch of the top mip in bytes (default: tight)\n"
           " -num-mip-levels <value>       : mip levels packed in the texture (1..16)\n"
           " -texture-pitch-d3d12-aligned  : mip pitches are 256-byte aligned\n"
           " -gpu                          : decompress on the GPU (always the case here)\n"
           " -warp                         : ignored\n"
           " -cpu                          : decompress with DecodeCPU on the host's threads instead\n"
           " -num-repeat <value>           : repeat the task (default 1)\n"
           " -fast                         : quicker compression (lazy parse, default distance parameters)\n"
           " -verbose                      : print progress\n");
}

bool ends_with(const std::string& s, const char* suffix)
{
    const size_t n = strlen(suffix);
    return s.size() >= n && s.compare(s.size() - n, n, suffix) == 0;
}

bool read_file(const std::string& path, std::vector<uint8_t>& out)
{
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    const long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    if (n < 0) { fclose(f); return false; }
    out.resize((size_t)n);
    const size_t got = n ? fread(out.data(), 1, (size_t)n, f) : 0;
    fclose(f);
    return got == (size_t)n;
}

bool write_file(const std::string& path, const uint8_t* p, size_t n)
{
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) return false;
    const size_t put = n ? fwrite(p, 1, n, f) : 0;
    return fclose(f) == 0 && put == n;
}

bool parse(int argc, char** argv, Options& o)
{
    auto value = [&](int& i, uint32_t& dst) { if (i + 1 >= argc) return false; dst = (uint32_t)strtoul(argv[++i], nullptr, 10); return true; };
    for (int i = 1; i < argc; ++i) {
        const char* a = argv[i];
        uint32_t ignored;
        if (!strcmp(a, "-pagesize")) { if (!value(i, o.page_size)) return false; }
        else if (!strcmp(a, "-precondition")) o.precondition = true;
        else if (!strcmp(a, "-swizzle")) o.swizzle = true;
        else if (!strcmp(a, "-delta-encode")) o.delta = true;
        else if (!strcmp(a, "-data-format")) { if (!value(i, o.format)) return false; }
        else if (!strcmp(a, "-texture-width")) { if (!value(i, o.tex_width)) return false; }
        else if (!strcmp(a, "-texture-height")) { if (!value(i, o.tex_height)) return false; }
        else if (!strcmp(a, "-row-pitch")) { if (!value(i, o.row_pitch)) return false; }
        else if (!strcmp(a, "-num-mip-levels")) { if (!value(i, o.num_mips)) return false; }
        else if (!strcmp(a, "-texture-pitch-d3d12-aligned")) o.pitch_aligned = true;
        else if (!strcmp(a, "-gpu") || !strcmp(a, "-warp")) {}
        else if (!strcmp(a, "-cpu")) o.cpu = true;
        else if (!strcmp(a, "-num-repeat")) { if (!value(i, o.repeat)) return false; }
        else if (!strcmp(a, "-verbose")) o.verbose = true;
        else if (!strcmp(a, "-fast")) o.fast = true;
        else if (!strcmp(a, "-brotli")) { fprintf(stderr, "brotlig: the stock-Brotli mode of the sample is not provided\n"); return false; }
        else if (!strcmp(a, "-brotli-quality") || !strcmp(a, "-brotli-windowsize") || !strcmp(a, "-brotli-decode-output-size")) { value(i, ignored); }
        else if (a[0] == '-') { fprintf(stderr, "brotlig: unknown option %s\n", a); return false; }
        else if (o.src.empty()) o.src = a;
        else if (o.dst.empty()) o.dst = a;
        else return false;
    }
    if (o.repeat == 0) o.repeat = 1;
    return !o.src.empty();
}

double now_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int compress(const Options& o, const std::vector<uint8_t>& src)
{
    BrotligEncodeOptions e{};
    e.page_size = o.page_size;
    // best ratio by default: shortest-path parse and per-page NPOSTFIX / NDIRECT search; -fast keeps the lazy parse
    e.flags = o.fast ? 0u : (BROTLIG_ENC_SEARCH_DIST_PARAMS | BROTLIG_ENC_OPTIMAL_PARSE | BROTLIG_ENC_SMOOTH_HISTOGRAMS);
    if (o.precondition) {
        if (o.format < 1 || o.format > 5 || !o.tex_width || !o.tex_height) {
            fprintf(stderr, "brotlig: -precondition needs -data-format 1..5, -texture-width and -texture-height\n");
            return 2;
        }
        e.precondition = 1; e.swizzle = o.swizzle; e.delta = o.delta; e.format = o.format;
        e.width_blocks = (o.tex_width + 3) / 4; e.height_blocks = (o.tex_height + 3) / 4;      // 4x4-pixel blocks
        e.num_mips = o.num_mips; e.pitch_bytes = o.row_pitch; e.pitch_d3d12_aligned = o.pitch_aligned;
    }
    if (src.size() > 0xFFFFFFFFull) { fprintf(stderr, "brotlig: input larger than 4 GiB\n"); return 2; }
    std::vector<uint8_t> out(BrotligEncMaxCompressedSize((uint32_t)src.size(), e.page_size ? e.page_size : 65536));
    uint32_t out_size = 0;
    double total = 0;
    for (uint32_t r = 0; r < o.repeat; ++r) {
        out_size = (uint32_t)out.size();
        const double t0 = now_ms();
        const int rc = BrotligEncode((uint32_t)src.size(), src.data(), &out_size, out.data(), &e);
        total += now_ms() - t0;
        if (rc != BROTLIG_ENC_OK) { fprintf(stderr, "brotlig: BrotligEncode failed with code %d\n", rc); return 3; }
        if (o.verbose) printf("  pass %u: %u bytes\n", r + 1, out_size);
    }
    const std::string dst = o.dst.empty() ? o.src + ".brotlig" : o.dst;
    if (!write_file(dst, out.data(), out_size)) { fprintf(stderr, "brotlig: cannot write %s\n", dst.c_str()); return 4; }
    const double ms = total / o.repeat;
    printf("Compressed %zu -> %u bytes (ratio %.3f) in %.3f ms: %.3f GiB/s  -> %s\n", src.size(), out_size,
           out_size ? (double)src.size() / out_size : 0.0, ms, src.size() / (ms * 1e-3) / (1024.0 * 1024.0 * 1024.0), dst.c_str());
    return 0;
}

int decompress(const Options& o, std::vector<uint8_t>& src)
{
    if (src.size() < 8 || src.size() > 0xFFFFFFFFull) { fprintf(stderr, "brotlig: not a .brotlig stream\n"); return 2; }
    const uint32_t size = DecompressedSize(src.data());
    std::vector<uint8_t> out((size_t)size + 16);
    double kernel_total = 0, wall_total = 0;
    uint32_t out_size = 0;
    for (uint32_t r = 0; r < o.repeat; ++r) {
        out_size = size;
        double kernel_ms = 0;
        const double t0 = now_ms();
        const BROTLIG_ERROR rc = o.cpu ? DecodeCPU((uint32_t)src.size(), src.data(), &out_size, out.data(), nullptr)
                                       : DecodeGPU(0, (uint32_t)src.size(), src.data(), &out_size, out.data(), &kernel_ms);
        const double wall = now_ms() - t0;
        wall_total += wall;
        if (rc != BROTLIG_OK) { fprintf(stderr, "brotlig: %s failed with BROTLIG_ERROR %d\n", o.cpu ? "DecodeCPU" : "DecodeGPU", (int)rc); return 3; }
        kernel_total += o.cpu ? wall : kernel_ms;
        if (o.verbose) printf("  pass %u: kernel %.3f ms\n", r + 1, kernel_ms);
    }
    std::string dst = o.dst;
    if (dst.empty()) dst = o.src.substr(0, o.src.size() - strlen(".brotlig"));
    if (!write_file(dst, out.data(), out_size)) { fprintf(stderr, "brotlig: cannot write %s\n", dst.c_str()); return 4; }
    const double ms = kernel_total / o.repeat;
    if (o.cpu) {
        printf("Decompressed %zu -> %u bytes on the CPU: %.3f ms (%.3f GiB/s of source, %.2f GB/s decompressed)  -> %s\n", src.size(), out_size,
               ms, src.size() / (ms * 1e-3) / (1024.0 * 1024.0 * 1024.0), out_size / (ms * 1e-3) / 1e9, dst.c_str());
        return 0;
    }
    printf("Decompressed %zu -> %u bytes on the GPU: kernel %.3f ms (%.3f GiB/s of source, %.2f GB/s decompressed), "
           "call %.3f ms incl. PCIe  -> %s\n", src.size(), out_size, ms, src.size() / (ms * 1e-3) / (1024.0 * 1024.0 * 1024.0),
           out_size / (ms * 1e-3) / 1e9, wall_total / o.repeat, dst.c_str());
    return 0;
}

}  // namespace

int main(int argc, char** argv)
{
    Options o;
    if (!parse(argc, argv, o)) { usage(); return 1; }
    std::vector<uint8_t> src;
    if (!read_file(o.src, src)) { fprintf(stderr, "brotlig: cannot read %s\n", o.src.c_str()); return 2; }
    return ends_with(o.src, ".brotlig") ? decompress(o, src) : compress(o, src);
}

#!/bin/bash
# Collect the per-round profile set on an MI355X box (run through gpurun from the repo root):
#   bash profiles/tools/collect.sh <tag>
# Writes raw rocprofv3 output under gpurun_out/<tag>/; profiles/tools/summarize.py turns it into the
# committed summaries.  PMC passes are separate runs with --kernel-trace only (no sys/hip traces).
set -u
tag=${1:-final}
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
python bench.py > "$out/bench.json" 2> "$out/bench.err"
python bench.py --workload bc3 --streams 256 --no-cpu-baseline > "$out/bench_bc3.json" 2>> "$out/bench.err"
python bench.py --workload runs --streams 1 --no-cpu-baseline > "$out/bench_runs.json" 2>> "$out/bench.err"
python bench.py --workload text --no-cpu-baseline > "$out/bench_text.json" 2>> "$out/bench.err"
python bench.py --workload samples16 --no-cpu-baseline > "$out/bench_samples16.json" 2>> "$out/bench.err"
python bench.py --workload records --no-cpu-baseline > "$out/bench_records.json" 2>> "$out/bench.err"
# every page distinct (about 1 GB of compressed data, beyond the 256 MiB Infinity Cache): does the tiling of 256 pages matter?
python bench.py --distinct 4096 --no-cpu-baseline > "$out/bench_distinct4096.json" 2>> "$out/bench.err"
# round 6: real bytes -- files of the image instead of a generator (both parses: the line carries "alt")
python bench.py --workload files --no-cpu-baseline --no-alt-parse > "$out/bench_files.json" 2>> "$out/bench.err"
python bench.py --workload files --encoder-flags 192 --no-cpu-baseline --no-alt-parse --steps 3 > "$out/bench_files_optimal_parse.json" 2>> "$out/bench.err"
python profiles/tools/sched_times.py mixed runs:1 mixed:16:64 > "$out/sched_times.jsonl" 2>> "$out/bench.err"
python profiles/tools/r05_many_streams.py > "$out/many_streams.jsonl" 2>> "$out/bench.err"
python profiles/tools/r05_many_textures.py > "$out/many_textures.jsonl" 2>> "$out/bench.err"
python profiles/tools/latency.py auto one_wavefront two_wavefronts > "$out/latency.json" 2>> "$out/bench.err"
for m in one_wavefront two_wavefronts; do python profiles/tools/page_latency.py $m 2>> "$out/bench.err" | grep "^{"; done > "$out/page_latency.jsonl"
python profiles/tools/streamer_bench.py > "$out/streamer_bench.json" 2>> "$out/bench.err"
python profiles/tools/cpu_decode_bench.py > "$out/cpu_decode.json" 2>> "$out/bench.err"
for k in "mixed 16" "text 16" "runs 16" "bc3 64" "samples16 16" "records 16" "files 16"; do python profiles/phase_profile.py $k; done > "$out/phase_profile.jsonl" 2>> "$out/bench.err"
BROTLIG_ENCODER_FLAGS=192 python profiles/phase_profile.py mixed 4 >> "$out/phase_profile.jsonl" 2>> "$out/bench.err"      # the optimal-parse streams (`alt`), 4 streams: the encode is slow
python profiles/tools/config5_projection.py --out "$out/config5_projection.json" > /dev/null 2>> "$out/bench.err"
# round 5: when each wavefront of a launch came and went (the launch's tail), on the PRODUCT kernel built with -DBROTLIG_WAVE_TIMES=1
if [ -f build/abv/lib_wavetimes.so ]; then
  for w in mixed text; do BROTLIG_HIP_SO=$root/build/abv/lib_wavetimes.so python profiles/tools/wave_times.py --workload $w 2>> "$out/bench.err"; done > "$out/wave_times.jsonl"
fi
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o f -- python "$root/bench.py" --steps 5 --warmup 2 --no-cpu-baseline > "$out/trace.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -o f -- python "$root/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$out/pmc_fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write" -o f -- python "$root/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$out/pmc_write.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d "$out/pmc_sq1" -o f -- python "$root/bench.py" --steps 1 --warmup 1 --no-cpu-baseline > "$out/pmc_sq1.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d "$out/pmc_sq2" -o f -- python "$root/bench.py" --steps 1 --warmup 1 --no-cpu-baseline > "$out/pmc_sq2.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM --output-format csv -d "$out/pmc_sq3" -o f -- python "$root/bench.py" --steps 1 --warmup 1 --no-cpu-baseline > "$out/pmc_sq3.log" 2>&1
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$out/pmc_tcc" -o f -- python "$root/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-alt-parse > "$out/pmc_tcc.log" 2>&1
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH --output-format csv -d "$out/pmc_icache" -o f -- python "$root/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-alt-parse > "$out/pmc_icache.log" 2>&1
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$out/pmc_tcc_distinct4096" -o f -- python "$root/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-alt-parse --distinct 4096 > "$out/pmc_tcc_distinct4096.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace_bc3" -o f -- python "$root/bench.py" --workload bc3 --streams 256 --steps 5 --warmup 2 --no-cpu-baseline > "$out/trace_bc3.log" 2>&1
# keep only what the summariser needs (the merge-back limit is 64 MiB)
find "$out" -name '*_kernel_trace.csv' -size +8M -delete
find "$out" -name '*agent_info*' -delete
ls -laR "$out" | tail -40

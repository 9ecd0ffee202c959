#!/bin/bash
# Round 6, call 6: device-output streamer + real files (tests, bench), page starts alone (ABLATE=256, round 5 against round 6), I-cache counters,
# TCC counters with every page distinct beside the default tiling.
export TMPDIR=/tmp
out=gpurun_out/r06c6; mkdir -p $out; root=$(pwd)
( timeout 600 python -m pytest tests/test_gpu_decode.py tests/test_gpu_fullsize.py -m gpu -x -q -k "streamer or real_files" ) > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest.log
timeout 300 python profiles/tools/streamer_bench.py > $out/streamer_bench.json 2>>$out/err.log; cat $out/streamer_bench.json
timeout 600 python bench.py --workload files --no-cpu-baseline > $out/bench_files.json 2>>$out/err.log; cut -c1-700 $out/bench_files.json
for v in r5 r6; do BROTLIG_HIP_SO=$root/build/abl256_$v.so timeout 300 python bench.py --no-cpu-baseline --no-alt-parse --steps 5 --warmup 2 2>>$out/err.log | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('abl256 $v kernel_ms', d['roofline']['kernel_ms'], 'step_ms', d['ms_per_step'])
"; done | tee $out/abl256.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $root/$out/pmc_icache -o f -- python $root/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-alt-parse > $root/$out/pmc_icache.log 2>&1
for d in 256 4096; do
timeout 400 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $root/$out/pmc_tcc_d$d -o f -- python $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt-parse --distinct $d > $root/$out/pmc_tcc_d$d.log 2>&1
done
cd $root
python - <<'PY'
import csv, glob, collections
for tag in ("pmc_icache", "pmc_tcc_d256", "pmc_tcc_d4096"):
    acc = collections.defaultdict(list)
    for p in glob.glob(f"gpurun_out/r06c6/{tag}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(p)):
            if "brotlig_decode_kernel(" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(tag, {k: "%.4g" % (sum(v) / len(v)) for k, v in sorted(acc.items())})
PY
find $out -name '*.csv' -size +4M -delete; find $out -name '*agent_info*' -delete
tail -3 $out/err.log

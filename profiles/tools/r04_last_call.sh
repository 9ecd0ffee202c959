mkdir -p gpurun_out/fin gpurun_out/final
export TMPDIR=/tmp
timeout 100 python profiles/tools/soak.py 150000 8000 1200 1600 0 > gpurun_out/fin/soak_auto.json 2> gpurun_out/fin/soak_auto.err; echo soak rc=$?
tail -c 300 gpurun_out/fin/soak_auto.json
timeout 60 python -m pytest tests/test_gpu_differential.py -m gpu -x -q -k "below_zero" 2>&1 | tail -1
root=$(pwd); out=$root/gpurun_out/final
timeout 90 python bench.py > "$out/bench.json" 2> "$out/bench.err"
cd /tmp
timeout 40 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -o f -- python "$root/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-alt-parse > "$out/pmc_fetch.log" 2>&1
timeout 40 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write" -o f -- python "$root/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-alt-parse > "$out/pmc_write.log" 2>&1
find "$out" -name '*_kernel_trace.csv' -size +8M -delete; find "$out" -name '*agent_info*' -delete
tail -1 "$out/bench.json" | cut -c1-160

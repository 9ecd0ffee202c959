timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
bash profiles/tools/collect_traffic.sh final > gpurun_out/collect_traffic.log 2>&1
python bench.py --workload bc3 --streams 256 --no-cpu-baseline > gpurun_out/final/bench_bc3.json 2>> gpurun_out/final/bench.err
tail -1 gpurun_out/final/bench_bc3.json | cut -c1-200

set -x
mkdir -p gpurun_out/solo
timeout 600 python profiles/tools/ab_run.py --workloads mixed text records samples16 --reps 3 --out gpurun_out/solo/ab.json > gpurun_out/solo/ab.txt 2>&1
for v in prev base; do
  BROTLIG_HIP_SO=$(pwd)/build/abv/lib_$v.so timeout 200 python profiles/tools/latency.py > gpurun_out/solo/latency_$v.json 2> gpurun_out/solo/latency_$v.err
  BROTLIG_HIP_SO=$(pwd)/build/abv/lib_$v.so timeout 200 python bench.py --workload runs --streams 1 --no-cpu-baseline --no-alt-parse --steps 10 --warmup 3 > gpurun_out/solo/config2_$v.json 2> gpurun_out/solo/config2_$v.err
done
timeout 900 python -m pytest tests/test_gpu_decode.py tests/test_gpu_differential.py -m gpu -x -q > gpurun_out/solo/pytest.txt 2>&1
tail -3 gpurun_out/solo/pytest.txt
cat gpurun_out/solo/ab.txt

set -x
mkdir -p gpurun_out/duo
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/duo/pytest_all.txt 2>&1
tail -4 gpurun_out/duo/pytest_all.txt
for v in base duo8 duo2; do
  BROTLIG_HIP_SO=$(pwd)/build/abv/lib_$v.so timeout 300 python profiles/tools/latency.py auto two_wavefronts > gpurun_out/duo/latency_$v.json 2> gpurun_out/duo/latency_$v.err
done
true

set -x
mkdir -p gpurun_out/duo
timeout 600 python profiles/tools/latency.py one_wavefront two_wavefronts > gpurun_out/duo/latency.json 2> gpurun_out/duo/latency.err
for m in one_wavefront two_wavefronts; do timeout 200 python profiles/tools/page_latency.py $m 2>gpurun_out/duo/pl_$m.err | grep "^{" > gpurun_out/duo/page_latency_$m.jsonl; done
true

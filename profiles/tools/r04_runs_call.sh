mkdir -p gpurun_out/runsab
timeout 600 python profiles/tools/ab_run.py --workloads mixed text runs records --reps 2 --out gpurun_out/runsab/ab.json 2>&1 | grep -E "^(mixed|text|records|samples16|runs)"
for v in prev base; do BROTLIG_HIP_SO=$(pwd)/build/abv/lib_$v.so timeout 200 python bench.py --workload runs --streams 1 --no-cpu-baseline --no-alt-parse --steps 10 --warmup 3 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('config2 $v', d['value'], d['roofline']['kernel_ms'], d['bit_exact'])
"; done

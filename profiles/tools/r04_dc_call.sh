for v in it1 base it3 base; do BROTLIG_HIP_SO=$(pwd)/build/abv/lib_$v.so timeout 200 python bench.py --workload bc3 --streams 256 --no-cpu-baseline --no-alt-parse --steps 5 --warmup 2 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('config4 $v', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['bit_exact'])
"; done

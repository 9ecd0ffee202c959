for so in build/ablate/*.so; do for w in $WL; do
BROTLIG_HIP_SO=$(pwd)/$so python bench.py --workload $w --no-cpu-baseline --steps 4 --warmup 1 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$so', '$w', d['value'], 'kernel_ms', d['roofline']['kernel_ms'], 'exact', d['bit_exact'])
"; done; done

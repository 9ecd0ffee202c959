#!/bin/bash
# Round 4, GPU call 1: parity of the changed kernel, A/B of the early near pieces, config 5 projection, PC sampling attempt.
out=gpurun_out/r04c1; mkdir -p $out
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $out/pytest_gpu.log; tail -3 $out/pytest_gpu.log
bash profiles/tools/ab_variants.sh run mixed text records samples16 2>&1 | tail -20
timeout 900 python profiles/tools/config5_projection.py --out $out/config5_projection.json > $out/config5.log 2>&1; tail -5 $out/config5.log
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for m in host_trap stochastic; do
  unit=time; iv=1000; [ $m = stochastic ] && { unit=cycles; iv=1048576; }
  rm -rf /tmp/pcs_$m
  timeout 240 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit $unit --pc-sampling-method $m --pc-sampling-interval $iv --kernel-trace \
      -d /tmp/pcs_$m --output-format csv -- python $R/bench.py --steps 10 --warmup 1 --no-cpu-baseline --no-alt-parse > $R/$out/pcs_$m.log 2>&1
  echo "pcs $m rc=$?"; tail -2 $R/$out/pcs_$m.log
  ls -la /tmp/pcs_$m/* 2>/dev/null | head
  python $R/profiles/tools/pcs_aggregate.py /tmp/pcs_$m $R/$out/pcs_$m.json
done

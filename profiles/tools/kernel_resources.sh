#!/bin/bash
# Register / LDS / scratch usage of the decode kernel for a set of -D flags (hipcc remarks; no GPU needed).
#   bash profiles/tools/kernel_resources.sh [-DFLAG=..]...
cd "$(dirname "$0")/../../brotli_g_sdk_amd/csrc" || exit 1
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-unroll-loops --cuda-device-only -c -Rpass-analysis=kernel-resource-usage "$@" -I ../../include -I . brotlig_hip.hip -o /dev/null 2>&1 |
  awk '/Function Name: .*brotlig_decode_kernelE/ {on=1} /Function Name: .*brotlig_decode_kernel_timed/ {on=0} on && /remark/ {sub(/.*remark: [^:]*:[0-9]+:[0-9]+: /,""); sub(/.*remark: /,""); printf "%s; ", $0} END {print ""}'

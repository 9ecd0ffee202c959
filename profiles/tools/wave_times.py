#!/usr/bin/env python3
"""How a launch of the decode kernel ramps up and ends: the first and last tick (100 MHz) of every wavefront of ONE launch of the
phase-timer twin (BrotligDecodePhaseProfile, entries beyond the phase sums).  Prints one JSON line: launch length, when the first /
median / last wavefront left, and the wavefront-time lost to the tail (sum over wavefronts of `launch end - my end`) as a fraction of
grid x launch length -- the part of the machine that idles because the work ran out at different times.

  python profiles/tools/wave_times.py [--workload mixed] [--streams 16]
"""
import argparse, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="mixed")
    ap.add_argument("--streams", type=int, default=16)
    a = ap.parse_args()
    import bench
    from brotli_g_sdk_amd import api
    streams, expected = bench.build_streams(a.workload, list(range(a.streams)), 4096, 256)
    dec = api.BatchDecoder(streams)
    dec.decode()
    t = dec.wave_times().astype(np.int64)
    t = t[t[:, 1] > 0]
    t0, t1 = t[:, 0].min(), t[:, 1].max()
    span = float(t1 - t0)
    ends = np.sort(t[:, 1] - t0) / span
    starts = np.sort(t[:, 0] - t0) / span
    print(json.dumps({"workload": a.workload, "streams": a.streams, "wavefronts": int(len(t)), "launch_us": round(span / 100.0, 1),
                      "last_start_frac": round(float(starts[-1]), 4),
                      "end_frac_percentiles": {str(p): round(float(np.percentile(ends, p)), 4) for p in (1, 5, 25, 50, 75, 95)},
                      "idle_tail_frac": round(float(np.mean(1.0 - ends)), 4),
                      "idle_head_frac": round(float(np.mean(starts)), 4)}))


if __name__ == "__main__":
    main()

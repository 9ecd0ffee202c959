#!/usr/bin/env python3
"""Damaged streams in BATCHES under AddressSanitizer + UndefinedBehaviorSanitizer on the CPU simulator, every output region sized by what the
damaged header claims -- the way profiles/tools/soak.py drives the device (round 4: that is how the two faults of the final soak reproduce).
  (cd tests/sim && g++ -O1 -g -std=c++17 -fPIC -shared -fsanitize=address,undefined -fno-sanitize-recover=undefined -DBROTLIG_WITH_SPLIT \
      -I . -I ../../brotli_g_sdk_amd/csrc -o /tmp/libbrotlig_sim_asan.so sim_decode.cpp sim_runtime.cpp)
  LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 python profiles/tools/sim_sanitize_batches.py <first seed> <end seed> <0|1: two wavefronts per page> <plain|precon>
Round 4, final source: seeds 200000-200800 and 230000-230800 (plain, both kernels), 210000-210800 (plain, one wavefront), 220000-220400 and
240000-240400 (pre-conditioned, both kernels): clean -- after load_u32 stopped assuming 4-byte alignment (a damaged page table places pages
at any byte address; same gfx950 code)."""
import ctypes, os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import test_sim_decode as T
from brotli_g_sdk_amd import encoder as E
from brotli_g_sdk_amd.api import DecompressedSize
from fuzzcases import random_plain, random_precon, corrupt
L=ctypes.CDLL('/tmp/libbrotlig_sim_asan.so')
L.sim_decode_batch.restype = ctypes.c_int
L.sim_decode_batch.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
lo,hi,duo,kind=int(sys.argv[1]),int(sys.argv[2]),int(sys.argv[3]),sys.argv[4]
L.sim_set_duo(duo)
for c in range(lo,hi,20):
    streams,sizes=[],[]; pre_any=False
    for s in range(c,c+20):
        if kind=="precon":
            t,pre,kw=random_precon(s); st=E.encode(t,precondition=pre,**kw); n0=len(t); pre_any=True
        else:
            d,kw=random_plain(s); st=E.encode(d,**kw); n0=len(d)
        b,k=corrupt(st,s)
        try: n=int(DecompressedSize(b))
        except Exception: n=n0
        streams.append(b); sizes.append(n if 0<n<=(16<<20) else n0)
    pre_any = pre_any or any((int(x[6])>>4)&1 for x in streams)
    print("batch",c,kind,duo,flush=True)
    T.run_batch(L,streams,sizes,grid=32,precon=pre_any)
print("clean",lo,hi,duo,kind)

#!/usr/bin/env python3
"""The kernel source on the CPU simulator under AddressSanitizer + UndefinedBehaviorSanitizer: every case class, 60 random streams and
their damaged twins, 20 random pre-conditioned streams and theirs.  Build and run (container, no GPU):
  (cd tests/sim && g++ -O1 -g -std=c++17 -fPIC -shared -fsanitize=address,undefined -fno-sanitize-recover=undefined -DBROTLIG_WITH_SPLIT \
      -I . -I ../../brotli_g_sdk_amd/csrc -o /tmp/libbrotlig_sim_asan.so sim_decode.cpp sim_runtime.cpp)
  LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 python profiles/tools/sim_sanitize.py"""
import ctypes, sys, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import test_sim_decode as T
from brotli_g_sdk_amd import encoder as E, datagen as D
from cases import plain_cases, raw_stress_cases, precon_cases, symbol_overflow_cases
from fuzzcases import random_plain, random_precon, corrupt
L=ctypes.CDLL('/tmp/libbrotlig_sim_asan.so')
L.sim_decode_batch.restype = ctypes.c_int
L.sim_decode_batch.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
n=0
for name,thunk,kw in plain_cases()+raw_stress_cases()+symbol_overflow_cases():
    d=np.ascontiguousarray(thunk(),dtype=np.uint8); outs,st=T.run_batch(L,[E.encode(d,**kw)],[len(d)]); assert st==0 and np.array_equal(outs[0],d),name; n+=1
for name,thunk,pre in precon_cases():
    t=thunk(); outs,st=T.run_batch(L,[E.encode(t,precondition=pre)],[len(t)],precon=True); assert st==0; n+=1
for seed in range(60):
    d,kw=random_plain(seed); s=E.encode(d,**kw); outs,st=T.run_batch(L,[s],[len(d)]); assert st==0 and np.array_equal(outs[0],d),seed
    bad,kind=corrupt(s,seed); T.run_batch(L,[bad],[len(d)]); n+=2
for seed in range(20):
    t,pre,kw=random_precon(seed); s=E.encode(t,precondition=pre,**kw); T.run_batch(L,[s],[len(t)],precon=True); bad,kind=corrupt(s,seed); T.run_batch(L,[bad],[len(t)],precon=True); n+=2
print("asan/ubsan clean over", n, "runs")

"""Whole-matrix K^-1 product / triangular inverse at 4096 and 8192 rows with the 128x128-tile kernels (default) and with
the 64x64-tile forms forced for every launch (TUNE_LAUUM_WM2_MAX / TUNE_TRTRI_WM2_MAX raised): python tools/wm2_probe.py"""
import ctypes as C, sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cugp_amd import capi
L=capi.lib()
L.cugp_bench_la.argtypes=[C.c_int,C.c_int,C.c_int,C.c_int,C.POINTER(C.c_double)]
def run(op,n):
    ms=C.c_double(); rc=L.cugp_bench_la(op,n,0,5,C.byref(ms)); return ms.value
for n in (4096,8192):
    for key,name,op in ((0,"lauum",2),(1,"trtri",1)):
        base=run(op,n)
        capi.check(L.cugp_set_tuning(key,1<<30))
        alt=run(op,n)
        capi.check(L.cugp_set_tuning(key,{0:768,1:1200}[key]))
        print("n=%d %s: 128x128 tiles %.3f ms, 64x64 tiles %.3f ms"%(n,name,base,alt),flush=True)

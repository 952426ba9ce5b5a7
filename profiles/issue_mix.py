import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
pkg=bench.importlib_pkg(); eng=pkg.Engine(0)
v=eng.probe_valu_rate(); s=eng.probe_salu_rate()
print("alone: valu %.3f salu %.3f G/s per SIMD"%(v,s))
for w in (1,2,4,5,6,8):
    mv,ms=eng.probe_mixed_rate(w)
    print("mixed 3:1, %d waves/SIMD: valu %.3f salu %.3f | time vs sum-of-alone %.2f, vs max-of-alone %.2f"%(w,mv,ms,(mv/v+ms/s),max(mv/v,ms/s)))

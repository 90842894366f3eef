import sys; sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))), "cuda-pathtracer_amd", "python"))
import ptmi, numpy as np
r = ptmi.Renderer(0)
# positive floats by exponent: bits = e<<23 .. ; check each binade
tot=0
for e in range(0, 256):
    bad, first = r.debug_rcp_check(e << 23, 1 << 23)
    if bad: print("exp", e, "2^%d"%(e-127), "bad", bad, "first", hex(first), np.uint32(first).view(np.float32))
    tot+=bad
print("positive total bad", tot)
bad, first = r.debug_rcp_check(0x80000000 + (27 << 23), (227-27) << 23)
print("negative normal-range [2^-100,2^100) bad", bad, hex(first))

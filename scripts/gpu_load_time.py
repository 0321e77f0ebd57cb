import time, sys, os
sys.path.insert(0, '/root/repo')
from totalsegmentator2d_amd.arch import UNetArch
from totalsegmentator2d_amd import weights
from totalsegmentator2d_amd.engine import Engine
a = UNetArch.canonical()
blob = weights.pack_blob(a, weights.synthetic_state_dict(a, 1))
t = time.time()
e = Engine(a, blob)
print('engine create', time.time() - t)
e.close()

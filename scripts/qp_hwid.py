import os, sys
sys.path.insert(0, "/root/repo")
os.environ["NEP_QP_PROFILE"] = "1"
os.environ["NEP_BACKEND_LIB"] = "/root/repo/neptune_amd/libneptune_backend_prof.so"
import numpy as np
from neptune_amd import scene, dist as ndist
from neptune_amd.backend import BatchBackend
N, M, S = 64, 20, 32
scs = [scene.make_scene(N, M, seed=s) for s in range(S)]
com, gue = ndist.stack_scenes(scs)
be = BatchBackend(scs[0]["par"], scs[0]["statics"], n_scenes=S)
for s in range(1, S): be.set_scene_statics(s, scs[s]["statics"])
d_com, d_gue = be.to_device(com), be.to_device(gue)
for _ in range(3): be.replan(d_com, d_gue)
c = np.array([be.debug_phase_cycles(i) for i in range(S * N)], dtype=np.int64)
h = c[:, 11]
simd = (h >> 4) & 3; wave = h & 15; cu = (h >> 8) & 15; sh = (h >> 12) & 1; se = (h >> 13) & 7; tg = (h >> 16) & 15
print("simd_id histogram of thread 0's wave:", np.bincount(simd, minlength=4))
print("wave_id histogram:", np.bincount(wave, minlength=16))
print("tg_id histogram:", np.bincount(tg, minlength=16))
print("cu_id histogram:", np.bincount(cu, minlength=16)); print("se", np.bincount(se, minlength=8), "sh", np.bincount(sh, minlength=2))
for i in list(range(0, 24)) + list(range(1024, 1040)):
    print(i, "se", se[i], "sh", sh[i], "cu", cu[i], "simd", simd[i], "wave", wave[i], "tg", tg[i], "life", c[i, 10])

"""Per-phase shader cycles of the front-end kernel for two slots (build with make PROFILE=1, NEP_QP_PROFILE=1)."""
import os, sys
os.environ["NEP_QP_PROFILE"] = "1"
os.environ.setdefault("NEP_BACKEND_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "neptune_amd", "libneptune_backend_prof.so"))
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from neptune_amd import scene, backend, abi
W = int(sys.argv[1]) if len(sys.argv) > 1 else 32
scs = [scene.make_scene(64, 20, seed=s) for s in range(8)]
for s in scs[1:]: s["statics"] = scs[0]["statics"]
p = scs[0]["par"]
bb = backend.BatchBackend(p, scs[0]["statics"], n_scenes=8)
com = np.stack([s["committed"] for s in scs]); st = np.stack([scene.frontend_starts(s) for s in scs])
fe = scene.frontend_cfg(p, beam_width=W)
d_g = torch.zeros(8 * 64 * abi.GUESS_DTYPE.itemsize, dtype=torch.uint8, device=bb.device)
for _ in range(3): bb.frontend(fe, bb.to_device(com), bb.to_device(st), d_g)
torch.cuda.synchronize()
names = ["loop top/barrier", "parent boxes + clear", "shortlist", "children pass 2 (GJK)", "voxel dedup", "compact", "rank + install", "children pass 1"]
for slot in (0, 100):
    c = bb.debug_phase_cycles(2 * slot); tot = sum(c[:8]); d = max(c[8], 1)
    print("slot %d: depths %d total %d" % (slot, d, tot))
    print('   per depth: GJK work list %d, shortlist %d, children %d' % (c[9] // d, c[10] // d, c[11] // d))
    for k, n in enumerate(names[:8]): print("   %-22s %9d  per-depth %7d  %5.1f%%" % (n, c[k], c[k] // d, 100.0 * c[k] / max(tot, 1)))

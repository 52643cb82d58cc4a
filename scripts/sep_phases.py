"""Development aid: where separator_packed_kernel's waves spend their cycles (make EXTRA=-DNEP_SEP_PROF B=build_sp OUT=../libsp.so;
NEP_BACKEND_LIB=neptune_amd/libsp.so python scripts/sep_phases.py [scenes])."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from neptune_amd import scene, dist as ndist, _lib
from neptune_amd.backend import BatchBackend


def main():
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    M = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    scs = scene.make_scenes(N, M, range(S), workers=min(S, 64))
    com, gue = ndist.stack_scenes(scs)
    be = BatchBackend(scs[0]["par"], scs[0]["statics"], n_scenes=S)
    for s in range(1, S):
        be.set_scene_statics(s, scs[s]["statics"])
    d_com, d_gue = be.to_device(com), be.to_device(gue)
    L = C.CDLL(_lib.LIB_PATH)
    out = (C.c_ulonglong * 16)()
    for _ in range(3):
        be.replan(d_com, d_gue)
    be.torch.cuda.synchronize()
    L.nep_debug_sep_prof(out, 1)
    R = 10
    for _ in range(R):
        be.replan(d_com, d_gue)
    be.torch.cuda.synchronize()
    L.nep_debug_sep_prof(out, 0)
    v = np.array(list(out), dtype=np.float64)
    waves = v[15]
    names = ["prologue", "A1 hull boxes", "A2 bases", "A3 statics", "list walk (B)", "stage point sets", "separator_impl", "line placement", "epilogue"]
    tot = v[:9].sum()
    print("waves %d; cycles per wave %.0f" % (waves, tot / waves))
    for n_, c_ in zip(names, v[:9]):
        print("  %-18s %8.0f cycles per wave  %5.1f %%" % (n_, c_ / waves, 100 * c_ / tot))


if __name__ == "__main__":
    main()

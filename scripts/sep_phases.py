"""Development aid: where separator_packed_kernel's waves spend their cycles (make EXTRA=-DNEP_SEP_PROF B=build_sp OUT=../libsp.so;
NEP_BACKEND_LIB=neptune_amd/libsp.so python scripts/sep_phases.py [scenes])."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from neptune_amd import scene, dist as ndist, _lib
from neptune_amd.backend import BatchBackend


def main_c5(S):
    """config 5 (256 agents + 100 obstacles, entangle rows): python scripts/sep_phases.py c5 [scenes]"""
    import dataclasses, torch
    from bench_legs import config5
    made = config5.ScenePool(256, 100, range(S), 64).wait()
    sc5 = [m[0] for m in made]
    p5 = dataclasses.replace(sc5[0]["par"], enable_entangle=True)
    be = BatchBackend(p5, sc5[0]["statics"], n_scenes=S)
    for s in range(S):
        be.set_scene_statics(s, sc5[s]["statics"])
    com5, gue5 = ndist.stack_scenes(sc5)
    d_com, d_gue = be.to_device(com5), be.to_device(gue5)
    d_ent = torch.from_numpy(np.ascontiguousarray(np.stack([m[1] for m in made])).reshape(-1)).to(be.device)
    report(be, lambda: be.replan(d_com, d_gue, d_ent=d_ent))


def report(be, step):
    L = C.CDLL(_lib.LIB_PATH)
    out = (C.c_ulonglong * 16)()
    for _ in range(3):
        step()
    be.torch.cuda.synchronize()
    L.nep_debug_sep_prof(out, 1)
    for _ in range(10):
        step()
    be.torch.cuda.synchronize()
    L.nep_debug_sep_prof(out, 0)
    v = np.array(list(out), dtype=np.float64)
    waves = v[15]
    names = ["prologue", "A1 hull boxes", "A2 bases", "A3 statics (+ case ids)", "list walk (B)", "stage point sets", "separator_impl", "line placement", "epilogue", "entangle agents (in B)", "entangle pairs (in B)"]
    tot = v[:11].sum()
    print("waves %d; cycles per wave %.0f" % (waves, tot / waves))
    for n_, c_ in zip(names, v[:11]):
        print("  %-24s %8.0f cycles per wave  %5.1f %%" % (n_, c_ / waves, 100 * c_ / tot))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "c5":
        return main_c5(int(sys.argv[2]) if len(sys.argv) > 2 else 8)
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    M = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    scs = scene.make_scenes(N, M, range(S), workers=min(S, 64))
    com, gue = ndist.stack_scenes(scs)
    be = BatchBackend(scs[0]["par"], scs[0]["statics"], n_scenes=S)
    for s in range(1, S):
        be.set_scene_statics(s, scs[s]["statics"])
    d_com, d_gue = be.to_device(com), be.to_device(gue)
    report(be, lambda: be.replan(d_com, d_gue))


if __name__ == "__main__":
    main()

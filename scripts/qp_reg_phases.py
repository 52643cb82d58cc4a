"""Per-phase shader cycles of qp_reg_kernel over a whole bench launch (make -C neptune_amd/csrc PROFILE=1 builds
libneptune_backend_prof.so; thread 0's clock, so barrier waits are charged to the phase that ends at the barrier)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ["NEP_QP_PROFILE"] = "1"
os.environ.setdefault("NEP_BACKEND_LIB", os.path.join(ROOT, "neptune_amd", "libneptune_backend_prof.so"))
import numpy as np
from neptune_amd import scene, dist as ndist
from neptune_amd.backend import BatchBackend
N, M, S = 64, 20, int(sys.argv[1]) if len(sys.argv) > 1 else 32
scs = [scene.make_scene(N, M, seed=s) for s in range(S)]
com, gue = ndist.stack_scenes(scs)
be = BatchBackend(scs[0]["par"], scs[0]["statics"], n_scenes=S)
for s in range(1, S): be.set_scene_statics(s, scs[s]["statics"])
d_com, d_gue = be.to_device(com), be.to_device(gue)
for _ in range(3): be.replan(d_com, d_gue)
c = np.array([be.debug_phase_cycles(i) for i in range(S * N)], dtype=np.int64)
names = ["A1+A2 row sweeps -> b1", "combine + ball row -> b2", "rd/rhs + M (MFMA) -> b3", "test + chol + pred solve (wave 0)", "wait for b4", "P2 sweep -> b5",
         "corrector rhs -> b6", "corrector solve -> b7", "P5 sweep -> b8", "step + z"]
it = c[:, 12].astype(float); life = c[:, 10].astype(float); loop = c[:, :10].sum(1).astype(float)
print("kernel: %s; slots %d, iterations mean %.2f (max %d)" % (be.qp_kernel_name(), len(c), it.mean(), it.max()))
print("workgroup lifetime: mean %.0f cycles, p50 %.0f, p99 %.0f, max %.0f  (clock64 ticks)" % (life.mean(), np.percentile(life, 50), np.percentile(life, 99), life.max()))
print("  iteration loop %.1f%% | line gather %.1f%% | mode staging %.1f%% | start point %.1f%% | rest (outputs) %.1f%%" % (
    100 * loop.sum() / life.sum(), 100 * c[:, 13].sum() / life.sum(), 100 * c[:, 14].sum() / life.sum(), 100 * c[:, 15].sum() / life.sum(),
    100 * (life.sum() - loop.sum() - c[:, 13:16].sum()) / life.sum()))
for k, nme in enumerate(names):
    print("   %-36s per iteration %8.0f  %5.1f%% of the loop" % (nme, c[:, k].sum() / max(it.sum(), 1), 100.0 * c[:, k].sum() / loop.sum()))
print("   per iteration, all phases: %.0f" % (loop.sum() / it.sum()))
w0 = c[:, 11] >> 20; wd = c[:, 11] & ((1 << 20) - 1)
t0 = w0.min(); st = (w0 - t0) / 100.0; en = st + wd / 100.0          # microseconds (100 MHz wall clock)
print("wall clock: kernel span %.1f us; workgroup duration mean %.1f us (p50 %.1f, p99 %.1f, max %.1f); shader clock = %.2f GHz" % (
    en.max(), (wd / 100.0).mean(), np.percentile(wd / 100.0, 50), np.percentile(wd / 100.0, 99), (wd / 100.0).max(), life.sum() / (wd.sum() * 10.0)))
late = st > 1.0
print("  workgroups starting at t = 0: %d; later: %d (start p50 %.1f us, last start %.1f us); last finish %.1f us; finish p50 %.1f p90 %.1f" % (
    int((~late).sum()), int(late.sum()), np.percentile(st[late], 50) if late.any() else 0, st.max(), en.max(), np.percentile(en, 50), np.percentile(en, 90)))
hist, edges = np.histogram(en, bins=12, range=(0, en.max()))
print("  finish-time histogram (us):", " ".join("%d:%d" % (int(edges[i + 1]), hist[i]) for i in range(12)))
busy = np.zeros(int(en.max()) + 2)
for a, b in zip(st, en): busy[int(a):int(b) + 1] += 1
print("  resident workgroups over time (every 40 us):", " ".join(str(int(busy[i])) for i in range(0, len(busy), 40)))
if os.environ.get("NEP_PH_LAST"):
    idx = np.argsort(-en)[:24]
    print("  last finishers (slot, iterations, start us, duration us, gather+staging+start cycles):")
    for i in idx: print("   ", int(i), int(c[i, 12]), "%.1f" % st[i], "%.1f" % (wd[i] / 100.0), int(c[i, 13] + c[i, 14] + c[i, 15]))
    print("  iterations vs duration (mean us by iteration count):", {int(k): round(float((wd[c[:, 12] == k] / 100.0).mean()), 1) for k in np.unique(c[:, 12])})
    print("  start time by iteration count (mean us):", {int(k): round(float(st[c[:, 12] == k].mean()), 1) for k in np.unique(c[:, 12])})

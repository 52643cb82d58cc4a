"""Development aid: the closed-loop fleet of the GPU test (64 agents + 20 obstacles flown to their goals, neptune_amd/loop.py) — how many
of its replans the QP gives up on, under the give-up rule in force (NEP_CORR_FROM / NEP_CORR_MAX override the defaults for an A/B).
python scripts/closed_loop_failures.py [seeds=0,1,2]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from neptune_amd import scene
from neptune_amd.loop import FleetLoop


def main():
    seeds = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "0,1,2").split(",")]
    tot = {}
    for seed in seeds:
        sc = scene.make_scene(64, 20, seed=seed)
        loop = FleetLoop(sc["par"], sc["statics"], sc["starts"], scene.reachable_goals(sc), beam_width=32)
        st = loop.run(max_rounds=400); loop.close()
        for k in ("replans", "accepted", "qp_failed", "fe_no_solution", "reached"):
            tot[k] = tot.get(k, 0) + int(st.get(k, 0))
    print("rule from %s max %s: %r  qp_failed %.3f %% of replans" % (os.environ.get("NEP_CORR_FROM", "default"), os.environ.get("NEP_CORR_MAX", "default"), tot, 100.0 * tot["qp_failed"] / max(tot["replans"], 1)))


if __name__ == "__main__":
    main()

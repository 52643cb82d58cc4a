#!/usr/bin/env python3
"""Closed-loop flight of a fleet on the device path (neptune_amd/loop.py): every agent flies from its
start next to its base to a random goal, replanning in bulk-synchronous rounds, the way the reference's
benchmark driver logs a run (scripts/benchmark_mtlp.py:215-223: elapsed time, distance, success).
  python scripts/closed_loop.py [--agents 16 --obstacles 8 --seed 0 --beam 32]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neptune_amd import scene
from neptune_amd.loop import FleetLoop

ap = argparse.ArgumentParser()
ap.add_argument("--agents", type=int, default=16); ap.add_argument("--obstacles", type=int, default=8)
ap.add_argument("--seed", type=int, default=0); ap.add_argument("--beam", type=int, default=32); ap.add_argument("--max-rounds", type=int, default=400)
a = ap.parse_args()
sc = scene.make_scene(a.agents, a.obstacles, seed=a.seed)
t0 = time.perf_counter()
loop = FleetLoop(sc["par"], sc["statics"], sc["starts"], scene.reachable_goals(sc), beam_width=a.beam)
st = loop.run(a.max_rounds)
st["wall_s"] = time.perf_counter() - t0
st["success"] = bool(st["reached"] == a.agents)
print(json.dumps({k: (float(v) if hasattr(v, "dtype") else v) for k, v in st.items()}))

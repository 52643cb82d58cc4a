"""The legs of bench.py: bench.py parses the command line, launches the ranks and prints ONE short JSON line;
the modules here do the timing.

  context.py   process group, barriers, HIP-graph capture, the timed region (`Ctx.run_leg`)
  account.py   algorithmic bytes / flops of one replan, the committed PMC summaries, executed fp64 work
  headline.py  BASELINE "64 agents + 20 obstacles" step (any N: agents block-sharded by id) + long_run / launch_order_off /
               reference_tolerances / presolve
  chain.py     front end -> lines -> QP -> safety check (chain), the closed loop (moving, crossing, two_groups)
  small.py     single_scene, BASELINE configs[1] / [2] batched, the per-agent drop-in call
  config5.py   BASELINE configs[4]: the single-GPU legs and the N-GPU workload (`--workload config5`)
  cpu.py       the CPU oracle timed on the host cores (cpu_baseline), the reference's own solvers where a box has them
  compact.py   the short line (<= 4 KB) the driver parses; everything else goes to bench_detail.json
"""

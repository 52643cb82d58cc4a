import json,sys
d=json.load(open(sys.argv[1]))   # bench_detail.json
fr = d.get("full_rows") or d.get("presolve")
print(sys.argv[1], "value %.0f ms/step %.3f qp %.4f sep %.4f hull %.4f iters %.2f" % (d["value"], d["ms_per_step"], d["kernel_ms"]["qp"], d["kernel_ms"]["separator"], d["kernel_ms"]["hull"], d["solver"]["ipm_iters_mean"]),
      "cull", d["solver"].get("line_cull_radius_m"), "full_rows/presolve leg", fr and round(fr["value"]))

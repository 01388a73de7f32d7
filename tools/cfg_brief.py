import json, sys
for line in sys.stdin:
    if line.startswith("{"):
        d = json.loads(line)
        print(sys.argv[1] if len(sys.argv) > 1 else "", "value %.4g M  L3 %.4g M" % (d["value"] / 1e6, d["roofline"]["l3_resident"]["frames_per_s"] / 1e6))
        for k, v in d.get("configs", {}).items():
            print("  ", k, "1-stream %.1f us" % v["us_per_batch"], {a: round(b, 1) for a, b in v["kernels_us"].items()}, "3-stream %.1f us" % v["us_per_batch_3_streams"],
                  "frac %.3f" % v["frac_of_hbm_peak_pipeline"])
        if "end_to_end" in d:
            print("  e2e", {k: (round(v) if isinstance(v, float) else v) for k, v in d["end_to_end"].items() if k != "what"})

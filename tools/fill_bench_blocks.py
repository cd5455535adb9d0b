"""Fill the counter blocks of a committed bench line (profiles/rNN/bench_default.json) from the PMC summaries next to it.

    python tools/fill_bench_blocks.py profiles/r02/bench_default.json

The bench run that produced the line PRECEDES the counter passes of the same tools/profile_bench.sh call, so its
`traffic` / `hbm_measured` / `valu` are null (bench.py only reports counter figures stamped with the tvl1.hip it runs).
This applies bench.pmc_blocks -- the function bench.py itself uses -- to the line's own timings."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main(path):
    d = json.load(open(path))
    r, cfg, K = d["roofline"], d["config"], d["steps"]
    px = sum(l["px_iters_per_step"] for l in r["per_level"])
    traffic, hbm, valu = bench.pmc_blocks(dict(block_iters=cfg["block_iters"], flow_streams=cfg["flow_streams"]),
                                          r["kernel_ms_per_step"], px, r["launches"] / K)
    if traffic is None or valu is None:
        raise SystemExit("the committed summaries are not stamped with this tvl1.hip: re-run tools/summarize_profiles.py")
    r["traffic"], r["hbm_measured"], r["valu"] = traffic, hbm, valu
    json.dump(d, open(path, "w"), indent=1)
    print("hbm: %.1f GB/step, %.2f TB/s, %.2f of peak | valu: %.3f instr/px-it, useful %.2f, min %.2f, busy %.2f"
          % (hbm["GB_per_step"], hbm["GBps"] / 1e3, hbm["frac_of_peak"], valu["wave_instr_per_px_iter"], valu["frac_useful"],
             valu["frac_min_work"], valu["frac_busy"]))


if __name__ == "__main__":
    main(sys.argv[1])

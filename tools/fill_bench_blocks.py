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
    r, K = d["roofline"], d["steps"]
    pmc = bench.pmc_blocks(d["config"], r["kernel_ms_per_step"], r["px_iters_per_step"], r["launches"] / K)
    hbm, valu = pmc["hbm"], pmc["valu"]
    if hbm is None or valu is None:
        raise SystemExit("the committed summaries are not stamped with this tvl1.hip and bench configuration: re-run tools/summarize_profiles.py")
    r.update(traffic=pmc["traffic"], hbm_measured=hbm, valu=valu, frac_valu_issued=valu["frac_issued"], frac_valu_busy=valu["frac_busy"],
             valu_instr_per_px_iter=valu["wave_instr_per_px_iter"], frac_hbm_measured=hbm["frac_of_peak"], hbm_GBps_measured=hbm["GBps"])
    json.dump(d, open(path, "w"), indent=1)
    print("hbm: %.1f GB/step, %.2f TB/s, %.2f of peak | valu: %.3f instr/px-it, issued %.2f, needed %.2f, busy %.2f"
          % (hbm["GB_per_step"], hbm["GBps"] / 1e3, hbm["frac_of_peak"], valu["wave_instr_per_px_iter"], valu["frac_issued"],
             r["frac_valu_needed"], valu["frac_busy"]))


if __name__ == "__main__":
    main(sys.argv[1])

"""VALU wave-instructions per launch from the SQ_INSTS_VALU pass of tools/profile_round.sh (csv by tools/rocpd_pmc.py):
pmc_valu.py pmc_SQ_INSTS_VALU.csv out.json  ->  {"kernels": {bench kernel name: {"valu_wave_insts_per_launch": n}}}
(what bench.py reports for the VALU-bound kernels: counters cannot be read from inside the timed run)."""
import csv
import json
import sys

from pmc_traffic import short_name


def main(path, out_json):
    rows = list(csv.DictReader(open(path)))
    # launches of a kernel = SQ_WAVES summed / waves of one launch is not known here; the csv carries the number of distinct dispatches
    # when the rocpd schema exposes it, else SQ counters come as 32 instance rows (8 XCD x 4 SE) per dispatch on MI355X
    out = {}
    for r in rows:
        if r["counter"] != "SQ_INSTS_VALU":
            continue
        k = short_name(r["kernel"])
        if not k:
            continue
        per_launch = float(r["per_launch"]) if r.get("per_launch") else float(r["per_dispatch"]) * 32.0
        out[k] = max(out.get(k, 0.0), per_launch)  # several template instantiations share a name; the ones that exit at once count next to nothing
    note = ("rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU ... on `python bench.py --steps 2 --warmup 1 --cpu-sample 0 --overlap 0`; per dispatch; "
            "wave-level instruction count of the working template instantiation")
    json.dump({"note": note, "kernels": {k: {"valu_wave_insts_per_launch": v} for k, v in sorted(out.items())}}, open(out_json, "w"), indent=1)


if __name__ == "__main__":
    sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
    main(sys.argv[1], sys.argv[2])

"""HBM traffic per launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), summarised by tools/rocpd_pmc.py.

Usage: pmc_traffic.py FETCH_SIZE.csv WRITE_SIZE.csv out.json
Counters are in KB (x1024).  FETCH_SIZE is doubled: gfx950 reports half of wide coalesced reads (MI355X_MICROARCH.md,
HBM / rocprofv3 section); WRITE_SIZE is used as reported.  Several template instantiations share a short name (variants that
exit at once are launched next to the one that works): the largest per-dispatch value is the working variant."""
import csv
import json
import re
import sys

# kernel function -> the timing name bench.py reports it under.  A timing name may cover several kernels of one stage (their traffic
# adds up); several template instantiations of ONE function are launched side by side and all but one exit at once (the largest is the work)
SHORT = {"d3_energy_kernel": "d3_energy", "d3_energy_kernel_w5": "d3_energy", "d3_cn_kernel": "d3_cn", "d3_chain_kernel": "d3_chain", "ewald_real_kernel": "ewald_real",
         "spline_spread_kernel": "spline_spread", "spread_tiled_kernel": "spline_spread", "spread_key_kernel": "spline_spread", "spread_box_kernel": "spline_spread",
         "spread_box_reduce_kernel": "spline_spread", "pme_convolve_kernel": "pme_convolve", "pme_gather_finish_kernel": "pme_gather_finish",
         "pme_gather_box_kernel": "pme_gather_finish", "pme_solve_fwd_kernel": "pme_solve_fwd", "pme_solve_inv_kernel": "pme_solve_inv",
         "pme_solve_fwd_cols_kernel": "pme_solve_cols", "pme_solve_inv_cols_kernel": "pme_solve_cols",
         "nl_query_tiled_kernel": None, "nl_query_kernel": None}
MODE = {"0": "nl_query_matrix", "1": "nl_query_count", "2": "nl_query_csr"}


def parse(sym: str):
    """(function name, template arguments) of a demangled kernel symbol, or (None, None)."""
    m = re.search(r"(?:::)?(\w+)<([^>]*)>", sym.replace("(anonymous namespace)::", ""))
    if not m:
        m2 = re.search(r"(\w+)\(", sym.replace("(anonymous namespace)::", ""))
        return (m2.group(1), []) if m2 else (None, None)
    return m.group(1), [a.strip() for a in m.group(2).split(",")]


def short_name(sym: str):
    fn, args = parse(sym)
    if fn not in SHORT:
        return None
    if SHORT[fn] is not None:
        return SHORT[fn]
    name = MODE.get(args[1])
    if name == "nl_query_matrix":  # the two matrix lists of the step differ in dtype (PME: double, D3: float)
        name += "_f64" if args[0] == "double" else "_f32"
    return name


def load(path, counter):
    """({timing name: bytes per launch}, {function<dtype>: bytes per launch}).  Within one function the largest instantiation is the one that
    worked; the functions of one timing name add up."""
    per_fn, per_sym = {}, {}
    for r in csv.DictReader(open(path)):
        if r["counter"] != counter:
            continue
        fn, args = parse(r["kernel"])
        if fn is None:
            continue
        val = float(r["per_dispatch"]) * 1024.0
        dt = next((a for a in (args or []) if a in ("float", "double")), "")
        key = f"{fn}<{dt}>" if dt else fn
        per_sym[key] = max(per_sym.get(key, 0.0), val)
        k = short_name(r["kernel"])
        if k:
            per_fn.setdefault(k, {})
            per_fn[k][fn] = max(per_fn[k].get(fn, 0.0), val)
    return {k: sum(v.values()) for k, v in per_fn.items()}, per_sym


def main(fetch_csv, write_csv, out_json):
    (f, fs), (w, ws) = load(fetch_csv, "FETCH_SIZE"), load(write_csv, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(f) | set(w)):
        fr, wr = f.get(k, 0.0), w.get(k, 0.0)
        kernels[k] = {"fetch_bytes_raw": fr, "fetch_bytes_corrected_x2": 2 * fr, "write_bytes": wr, "hbm_bytes_per_launch": 2 * fr + wr}
    note = ("rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `python bench.py --steps 2 --warmup 1 "
            "--cpu-sample 0 --overlap 0`, per dispatch; KB counters x1024; FETCH_SIZE doubled (gfx950 reports 1/2 of wide coalesced "
            "reads, MI355X_MICROARCH.md HBM section); WRITE_SIZE as reported.")
    symbols = {k: {"fetch_bytes_corrected_x2": 2 * fs.get(k, 0.0), "write_bytes": ws.get(k, 0.0), "hbm_bytes_per_launch": 2 * fs.get(k, 0.0) + ws.get(k, 0.0)}
               for k in sorted(set(fs) | set(ws)) if 2 * fs.get(k, 0.0) + ws.get(k, 0.0) >= 1e6}
    json.dump({"note": note + "  `kernels`: by bench.py timing name (the kernels of one stage summed); `symbols`: every kernel function above 1 MB per launch.",
               "kernels": kernels, "symbols": symbols}, open(out_json, "w"), indent=1)


if __name__ == "__main__":
    main(*sys.argv[1:4])

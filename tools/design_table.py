"""Rows of DESIGN.md section 3's numbers table from ONE bench line (profiles/<tag>_bench_default.json): in-step and isolated milliseconds, the
8(d) bytes, the bytes moved by design, the fraction of 8 TB/s and the PMC traffic of every kernel group the line prices.
    python tools/design_table.py profiles/r06_bench_default.json"""
import json
import sys


def main(path):
    r = json.loads([ln for ln in open(path) if ln.startswith("{")][-1])
    k = r["kernels"]
    gb = lambda v: "—" if not v else f"{v / 1e9:.2f} GB"  # noqa: E731
    print(f"headline {r['ms_per_step']:.3f} ms  value {r['value'] / 1e6:.2f} M atom-steps/s  processes {[round(p['ms_per_step'], 3) for p in r['processes']['each']]}")
    cal = r.get("calibration") or {}
    print("calibration", {a: round(b, 1) for a, b in cal.items() if isinstance(b, float)})
    print("serial", r["stats"].get("step_ms_median_serial_untimed"))
    print("| kernel | in-step ms | isolated ms | algorithmic bytes (§8d) | bytes moved by design | of 8 TB/s (isolated) | PMC traffic |")
    for name, v in sorted(k.items(), key=lambda kv: -(kv[1].get("isolated_median_ms") or 0)):
        frac = v.get("frac_of_hbm_peak")
        print(f"| `{name}` | {v['avg_ms_timed_region']:.3f} | {(v.get('isolated_median_ms') or 0):.3f} | {gb(v.get('algorithmic_bytes'))} | {gb(v.get('design_bytes'))} | "
              f"{'—' if frac is None else f'{frac:.2f}'} | {gb(v.get('traffic_bytes'))} | {v.get('bound')} {v.get('valu_wave_insts') or ''} {v.get('frac_of_hbm_peak_list_only') or ''}")
    for name, c in (r.get("configs") or {}).items():
        if isinstance(c, dict):
            print(name, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in c.items() if a in ("ms", "ms_per_step", "value", "error")},
                  (c.get("roofline") or {}).get("kernel"), (c.get("roofline") or {}).get("frac"))
            for row in c.get("rows", []) or []:
                print("    ", {a: (round(b, 4) if isinstance(b, float) else b) for a, b in row.items() if a in ("atoms", "n_atoms", "median_ms", "reference_h100_median_ms", "speedup_vs_reference_h100", "cutoff", "method")})
    print("cpu", {a: b for a, b in (r.get("cpu_baseline") or {}).items() if a in ("value", "cores", "seconds")})
    print("parity", {a: b for a, b in (r.get("parity") or {}).items() if a in ("d3_abs_dE_Ha", "d3_rel_dE", "pme_abs_dE", "d3_energy_gpu_Ha", "pme_energy_gpu")})


if __name__ == "__main__":
    main(sys.argv[1])

"""DESIGN.md section 2's full-size parity table, generated FROM THE RECORD the GPU tests write (tests/conftest.py ->
gpurun_out/parity_r06.json, copied to profiles/parity_r06.json as the last action of the round):

    python profiles/make_parity_table.py [profiles/parity_r06.json]            # prints the markdown table
    python profiles/make_parity_table.py profiles/parity_r06.json --write     # ... and replaces the block between the markers in DESIGN.md

Per configuration: the HIP path's error against the fp64 oracle (max, p99), the two envelopes apart -- the CPU oracle evaluated in fp32
and the ATen-only GPU evaluation of the reference's op sequence (tests/test_gpu_parity.py::_aten_only: no kernel of this package runs,
asserted by a launch counter) --, the ratio of the HIP maximum to the larger envelope maximum, and the energy error over its bound."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BEGIN, END = "<!-- parity-table:begin (profiles/make_parity_table.py) -->", "<!-- parity-table:end -->"
WANT = ["qm9_1024", "md17_4096", "qm9_8192", "qm9_8192_chunked", "water_512", "qm9_1024, reference initialisation",
        "qm9_1024, well-conditioned model (0e block of update_V bounded away from 0), plain tolerance"]


def table(path):
    recs = json.load(open(path))["records"]
    by = {}
    for r in recs:
        by.setdefault(r["config"], r)
    rows = ["| configuration | atoms / edges (compared graphs) | max abs dF (HIP) | p99 | CPU oracle fp32: max / p99 | ATen-only GPU fp32: max / p99 | HIP max / envelope max | max dE / bound |",
            "|---|---|---|---|---|---|---|---|"]
    f = lambda v: "-" if v is None else f"{v:.1e}"
    for name in WANT:
        r = by.get(name)
        if r is None:
            continue
        cpu, gpu = r.get("cpu_oracle32_max_abs_dF"), r.get("aten_gpu32_max_abs_dF")
        env = max(v for v in (cpu, gpu, 0.0) if v is not None)
        ratio = f"{r['max_abs_dF'] / env:.2f}" if env > 0 else "no envelope: bound 1e-4"
        size = f"{r.get('atoms', r.get('compared_atoms', '-'))} / {r.get('edges', '-')} ({r.get('compared_graphs', '-')})"
        de = r.get("max_dE_over_bound")
        rows.append(f"| {name} | {size} | {f(r['max_abs_dF'])} | {f(r.get('p99_abs_dF'))} | {f(cpu)} / {f(r.get('cpu_oracle32_p99_abs_dF'))} | "
                    f"{f(gpu)} / {f(r.get('aten_gpu32_p99_abs_dF'))} | {ratio} | {'-' if de is None else f'{de:.3f}'} |")
    small = [r for r in recs if r["config"].startswith("model check") or "golden edge list" in r["config"]]
    if small:
        worst = max(small, key=lambda r: r["max_abs_dF"] / max(r.get("oracle32_max_abs_dF", 0.0), 1e-30))
        rows.append("")
        rows.append(f"Model-level checks below full size ({len(small)} records): worst HIP max / envelope max = "
                    f"{worst['max_abs_dF'] / max(worst.get('oracle32_max_abs_dF', 0.0), 1e-30):.2f} ({worst['config']}: HIP {f(worst['max_abs_dF'])}, "
                    f"CPU fp32 {f(worst.get('cpu_oracle32_max_abs_dF'))}, ATen-only GPU {f(worst.get('aten_gpu32_max_abs_dF'))}); asserted factor 1.5.")
    return "\n".join(rows)


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    path = args[0] if args else os.path.join(ROOT, "profiles", "parity_r06.json")
    t = table(path)
    print(t)
    if "--write" in sys.argv:
        p = os.path.join(ROOT, "DESIGN.md")
        s = open(p).read()
        i, j = s.index(BEGIN) + len(BEGIN), s.index(END)
        open(p, "w").write(s[:i] + f"\n(generated from `{os.path.relpath(path, ROOT)}`)\n\n" + t + "\n" + s[j:])

#!/usr/bin/env python3
"""Turns rocprofv3 CSV output (kernel trace [+ counter collection]) into the markdown/JSON summaries committed under profiles/.
usage: summarize_profile.py <trace_dir> <out.md> [--pmc-fetch DIR --pmc-write DIR --pmc-json OUT.json]"""
import argparse
import collections
import csv
import json
import os


def load(path):
    with open(path) as f:
        return list(csv.DictReader(f))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace_dir")
    ap.add_argument("out_md")
    ap.add_argument("--title", default="rocprofv3 --kernel-trace --stats")
    ap.add_argument("--command", default="")
    ap.add_argument("--pmc-fetch")
    ap.add_argument("--pmc-write")
    ap.add_argument("--pmc-json")
    a = ap.parse_args()
    rows = load(os.path.join(a.trace_dir, "run_kernel_trace.csv"))
    per = collections.defaultdict(list)
    grid = collections.defaultdict(list)
    for r in rows:
        name = r["Kernel_Name"].split("(")[0]
        us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0
        per[name].append(us)
        grid[(name, int(r["Grid_Size_X"]))].append(us)
    total = sum(sum(v) for v in per.values())
    with open(a.out_md, "w") as f:
        f.write(f"# {a.title}\n\nCommand: `{a.command}`\n\nDurations in microseconds.\n\n")
        f.write("| kernel | calls | total us | avg us | min us | % |\n|---|---|---|---|---|---|\n")
        for name, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
            f.write(f"| `{name[:80]}` | {len(v)} | {sum(v):.1f} | {sum(v)/len(v):.2f} | {min(v):.2f} | {100*sum(v)/total:.2f} |\n")
        f.write("\n## Sum-check kernels by launch size (grid = threads)\n\n| kernel | grid | calls | avg us | min us |\n|---|---|---|---|---|\n")
        for (name, g), v in sorted(grid.items(), key=lambda kv: (kv[0][0], -kv[0][1])):
            if any(k in name for k in ("bind", "eval", "sum_partials")):
                f.write(f"| `{name[:60]}` | {g} | {len(v)} | {sum(v)/len(v):.2f} | {min(v):.2f} |\n")
        if a.pmc_fetch and a.pmc_write:
            def counters(d):
                out = collections.defaultdict(list)
                for r in load(os.path.join(d, "run_counter_collection.csv")):
                    out[(r["Kernel_Name"].split("(")[0], int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
                return out
            fe, wr = counters(a.pmc_fetch), counters(a.pmc_write)
            f.write("\n## HBM traffic from PMC (separate passes: `--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`)\n\n"
                    "Correction per MI355X_MICROARCH.md (HBM section): counters are in KiB; on gfx950 FETCH_SIZE reports half of a wide coalesced "
                    "streaming read, so read bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE * 1024 taken as is (it matches the algorithmic write bytes here).\n\n"
                    "| kernel | grid | FETCH_SIZE KiB | WRITE_SIZE KiB | corrected traffic MB |\n|---|---|---|---|---|\n")
            summary = {}
            for key in sorted(fe, key=lambda k: (k[0], -k[1])):
                if key in wr and ("stream" in key[0] or "k_bind_top" in key[0] or "k_eval" in key[0]):
                    fv, wv = sum(fe[key]) / len(fe[key]), sum(wr[key]) / len(wr[key])
                    traffic = (2 * fv + wv) * 1024
                    f.write(f"| `{key[0][:60]}` | {key[1]} | {fv:.1f} | {wv:.1f} | {traffic/1e6:.2f} |\n")
                    summary[f"{key[0]}@{key[1]}"] = {"fetch_kib": fv, "write_kib": wv, "traffic_bytes": traffic}
            if a.pmc_json:
                with open(a.pmc_json, "w") as jf:
                    json.dump(summary, jf, indent=1)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Joins tools/fetch_calib's known byte counts with the rocprofv3 FETCH_SIZE / WRITE_SIZE values of its dispatches (counter unit: KiB as rocprofv3 reports
it, i.e. TCC_EA0_RDREQ-derived) and derives, per access pattern, counter bytes per known byte; then prices the poly_ABC walk's own counters with those
factors instead of the guide's blanket x2."""
import argparse
import collections
import csv
import json
import re

ap = argparse.ArgumentParser()
for k in ("cases", "fetch", "write", "polyabc-fetch", "polyabc-write", "out"):
    ap.add_argument("--" + k, required=True)
a = ap.parse_args()


def counters(path):
    per = collections.defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            per[r["Kernel_Name"].split("(")[0].replace("void ", "")].append((int(r.get("Dispatch_Id", 0)), float(r["Counter_Value"]), int(r["Grid_Size"])))
    for v in per.values():
        v.sort()
    return per


cases = []  # in launch order, three repetitions
for line in open(a.cases):
    m = re.match(r"case (\S+) (.*)", line.strip())
    if m:
        kv = m.group(2).split()
        cases.append((m.group(1), {kv[i]: int(kv[i + 1]) for i in range(0, len(kv), 2)}))
kernel_of = lambda c: "k_stream16" if c == "stream16" else ("k_stream4" if c == "stream4" else "k_stream32" if c == "stream32" else "k_gather32_sorted" if c.startswith("gather32_sorted") else ("k_gather32" if c.startswith("gather32") else ("k_lane_streams" if c == "lane_streams" else "k_store32")))
fe, wr = counters(a.fetch), counters(a.write)
# the flush launches are k_stream16 dispatches with grid 8192*256 over 1 GiB: the measured stream case is the one over 512 MiB; tell them apart by order
res = collections.defaultdict(lambda: {"known_bytes": 0, "fetch_kib": [], "write_kib": []})


def take(per, kname, which):
    lst = per.get(kname, [])
    return lst[which][1] if which < len(lst) else None


# order of k_stream16 dispatches per repetition: flush, CASE, flush, flush, flush, flush, flush, flush (one flush before every case: 7 per rep + 1 case)
seen = collections.Counter()
stream_idx = 0
for name, kv in cases:
    k = kernel_of(name)
    if k == "k_stream16":
        # per repetition 10 k_stream16 dispatches (a flush in front of each of the 9 cases + the case itself): index 1 is the case
        rep = seen[name]
        which = rep * 10 + 1
    else:
        which = seen[k]
        seen[k] += 1
    if k == "k_stream16":
        seen[name] += 1
    f, w = take(fe, k, which), take(wr, k, which)
    known = kv.get("bytes", kv.get("bytes32", 0))
    r = res[name]
    r["known_bytes"] = known
    if "gathers" in kv:
        r["gathers"] = kv["gathers"]
    if f is not None:
        r["fetch_kib"].append(f)
    if w is not None:
        r["write_kib"].append(w)
out = {"unit_note": "FETCH_SIZE / WRITE_SIZE as rocprofv3 prints them, taken as KiB; factor = counter bytes / known bytes (median of 3 repetitions)", "patterns": {}}
for name, r in res.items():
    med = lambda v: sorted(v)[len(v) // 2] if v else None
    f, w = med(r["fetch_kib"]), med(r["write_kib"])
    e = {"known_bytes": r["known_bytes"], "FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w}
    if f is not None and r["known_bytes"]:
        e["fetch_bytes_per_known_byte"] = f * 1024 / r["known_bytes"]
    if "gathers" in r and f is not None:
        e["fetch_bytes_per_gather"] = f * 1024 / r["gathers"]
    if name == "store32" and w is not None:
        e["write_bytes_per_known_byte"] = w * 1024 / r["known_bytes"]
    out["patterns"][name] = e
pf, pw = counters(a.polyabc_fetch), counters(a.polyabc_write)
for kname in ("spk::k_polyabc_short_and_long", "spk::k_spmv3", "spk::k_rowmat_vec_tall"):
    if kname in pf:
        vals = [v for _, v, _ in pf[kname]]
        wv = [v for _, v, _ in pw.get(kname, [])]
        out.setdefault("kernels_alone", {})[kname] = {"FETCH_SIZE_KiB_median": sorted(vals)[len(vals) // 2], "WRITE_SIZE_KiB_median": sorted(wv)[len(wv) // 2] if wv else None,
                                                      "launches": len(vals)}
json.dump(out, open(a.out, "w"), indent=1)
print(json.dumps(out, indent=1))

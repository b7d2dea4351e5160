#!/usr/bin/env python3
"""profiles/<round>_kernel_stats.md from the rocprofv3 CSVs of tools/profile_job.sh.
For every kernel VERDICT r3 item 7 names (and the roofline kernel), selected by (kernel, grid size) so that a row is ONE call site of the config-2 prove:
the kernel-only duration inside a prove (headline driver and reference-order driver), the same launch run alone, how much of the in-prove duration other
kernels were resident on the GPU (from the trace's own timestamps), the SURVEY 8(d) algorithmic bytes, GB/s, and PMC traffic."""
import argparse
import collections
import csv
import json
import os

M = 1 << 20  # config 2: num_cons = num_vars = 2^20


def load(d, name):
    with open(os.path.join(d, name)) as f:
        return list(csv.DictReader(f))


def launches(d):
    out = []
    for r in load(d, "run_kernel_trace.csv"):
        out.append((r["Kernel_Name"].split("(")[0].replace("void ", ""), int(r["Grid_Size_X"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    out.sort(key=lambda x: x[2])
    return out


def by_site(ls):
    per = collections.defaultdict(list)
    for i, (name, grid, s, e) in enumerate(ls):
        per[(name, grid)].append(i)
    return per


def overlap_stats(ls, idxs):
    """mean fraction of the launch's duration during which at least one OTHER kernel was running, and the names of those kernels"""
    fr, names = [], collections.Counter()
    starts = [x[2] for x in ls]
    import bisect
    for i in idxs:
        _, _, s, e = ls[i]
        iv = []
        lo = bisect.bisect_left(starts, s - 3_000_000)  # kernels that started up to 3 ms earlier (resident tails run ~0.15 ms)
        for j in range(lo, len(ls)):
            if j == i:
                continue
            n2, _, s2, e2 = ls[j]
            if s2 >= e:
                break
            if e2 > s:
                iv.append((max(s, s2), min(e, e2)))
                names[n2] += 1
        iv.sort()
        cov, cur = 0, s
        for a, b in iv:
            if b > cur:
                cov += b - max(a, cur)
                cur = b
        fr.append(cov / max(1, e - s))
    return sum(fr) / max(1, len(fr)), names


# kernel -> (what, algorithmic bytes as a function of the launch's grid (threads), the rule). Geometry from capi_core.hip: the fused stream kernels launch
# len / 4 threads, the evaluation kernels one thread per pair (len / 2) or per four pairs (lowhi<4>).
def sites(M_):
    return [
        ("k_bind_eval_cubic_stream<1, false, true>", None, "outer: bind round r + evaluate round r+1, queued behind its mailbox gate (ROOFLINE kernel at grid 262144)",
         lambda g: 48 * (4 * g) * 3, "3 tables of len = 4 grid: read 32 len, write 16 len each"),
        ("k_bind_eval_cubic_stream<1, false, false>", None, "the same launched behind its challenge (reference order without a device mailbox; the streaming probe)", lambda g: 48 * (4 * g) * 3,
         "3 tables of len = 4 grid: read 32 len, write 16 len each"),
        ("k_eval_cubic_stream<1>", None, "outer round 0 evaluation (no round-0 products: reference order)", lambda g: 160 * g, "grid = pairs; A, B, C pairs + eq: 160 B per pair"),
        ("k_eval_products_stream<1, 4>", None, "outer round 0 from the round-0 products (headline driver; four 256-pair chunks per block)", lambda g: 64 * (4 * g), "grid = pairs / 4; p0, p1: 64 B per pair"),
        ("k_bind_eval_quad_stream_sparse", None, "inner: first bind (effective ranges) + evaluate", lambda g: 64 * (4 * g), "len = 4 grid; live low halves only: 64 B per entry of len / 2, x 2 tables"),
        ("k_bind_eval_quad_stream", None, "inner: bind round r + evaluate round r+1", lambda g: 48 * (4 * g) * 2, "2 tables of len = 4 grid: read 32 len, write 16 len each"),
        ("k_eval_quad_stream_lowhi<4>", None, "inner round 0 evaluation (effective ranges)", lambda g: 64 * (4 * g), "grid = pairs / 4; 64 B per live pair (+ 32 B per live high entry, < 1 KB here)"),
        ("k_rowmat_vec_tall", None, "PCS::prove L^T W (512 x 2048)", lambda g: 32 * (M_ + 512 + 2048), "reads W once: 32 (rows cols + rows + cols)"),
        ("k_polyabc_short_and_long", None, "bind_and_prepare_poly_ABC", None, "8(d) as the library accounts it: 12 B per nonzero + 32 B per live row of eq(r_x) + 32 B per output column"),
        ("k_spmv3<false>", None, "multiply_vec (incremental: rest columns only)", None, "8(d) as the library accounts it: 12 B per nonzero + 3 x 32 B per row written + witness gathers"),
        ("k_eq_outer_lastk", None, "evals_rx outer product (pyramids started under the last four rounds)", lambda g: 32 * (8 * g), "grid = entries / 8 (512 threads per 4096 entries); writes 32 B per entry"),
        ("k_round0_products", None, "round-0 products of the outer sum-check (headline driver)", lambda g: 224 * (M_ // 2), "reads Az, Bz, Cz (96 B per row), writes p0, p1 (32 B per row)"),
    ]


def main():
    ap = argparse.ArgumentParser()
    for k in ("prove", "reford", "solo", "fetch", "write", "out", "pmc-json", "sites-json"):
        ap.add_argument("--" + k)
    ap.add_argument("--round", default="r05")
    ap.add_argument("--bench-json", help="a bench.py line: poly_abc / spmv bytes (the library's 8(d) accounting) are read from its other_kernels")
    a = ap.parse_args()
    P, R, S = launches(a.prove), launches(a.reford), launches(a.solo)
    sp, sr, ss = by_site(P), by_site(R), by_site(S)

    def counters(d):
        out = collections.defaultdict(list)
        for r in load(d, "run_counter_collection.csv"):
            out[(r["Kernel_Name"].split("(")[0].replace("void ", ""), int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
        return {k: sum(v) / len(v) for k, v in out.items()}

    fe, wr = counters(a.fetch), counters(a.write)
    lib_bytes = {}
    if a.bench_json and os.path.exists(a.bench_json):
        j = json.loads(open(a.bench_json).read().strip().splitlines()[-1])
        ok = j["roofline"]["other_kernels"]
        for cls, kn in (("poly_abc", "k_polyabc_short_and_long"), ("spmv_incremental", "k_spmv3<false>")):
            if cls in ok:
                lib_bytes[kn] = ok[cls]["alg_GBps"] * 1e9 * ok[cls]["avg_us"] * 1e-6
    # median, not mean: a call site's first launch in a process (cold caches, first touch of a workspace) can be several times the steady state
    st = lambda ls, idx: (sorted(ls[i][3] - ls[i][2] for i in idx)[len(idx) // 2] / 1e3, min(ls[i][3] - ls[i][2] for i in idx) / 1e3) if idx else (None, None)
    fmt = lambda v: "-" if v is None else f"{v:.1f}"
    pmc, site_rows = {}, {}
    with open(a.out, "w") as f:
        f.write(f"# {a.round}: per-kernel evidence for the config-2 prove (sha256 2048 B, num_cons = num_vars = 2^20)\n\n"
                "Commands (tools/profile_job.sh): `rocprofv3 --kernel-trace --stats -- python tools/kernel_evidence.py prove` (headline driver, 12 proves, nothing else "
                "in the process), `... prove --reference-order`, `... solo` (the same kernels, one ABI call at a time with a device sync in between), and two `--pmc` passes "
                "(FETCH_SIZE, WRITE_SIZE) of the first command.\n\n"
                "A row is ONE call site: kernel name + grid size (threads). `in prove` = rocprofv3 kernel-only duration (End - Start of the dispatch) inside the headline prove; "
                "`ref order` = the same inside the one-thread reference-order prove; `solo` = the same launch with the GPU otherwise empty. `others resident` = mean fraction of "
                "the in-prove duration during which at least one other kernel was running (from the trace's timestamps) and which. GB/s = algorithmic bytes / median duration. "
                "`PMC MB` = (2 x FETCH_SIZE + WRITE_SIZE) KiB per launch, the gfx950 correction of MI355X_MICROARCH.md's HBM section.\n\n"
                "| kernel @ grid | what | launches/prove | alg MB | in prove median / min us | GB/s | ref order median / min us | solo median / min us | solo GB/s | others resident | PMC MB |\n"
                "|---|---|---|---|---|---|---|---|---|---|---|\n")
        nproves = 12
        rows_done = set()
        for sub, _, what, bytes_fn, why in sites(M):
            keys = sorted({k for k in list(sp) + list(sr) + list(ss) if k[0].endswith(sub) or k[0] == "spk::" + sub}, key=lambda k: -k[1])
            for key in keys:
                if key in rows_done:
                    continue
                rows_done.add(key)
                name, grid = key
                # entries the launch covers: stream kernels use 1 thread per 4 pairs or similar; bytes are given per table entry count inferred from grid
                pa, pm = st(P, sp.get(key, []))
                ra, rm = st(R, sr.get(key, []))
                sa, sm = st(S, ss.get(key, []))
                n_l = len(sp.get(key, [])) / nproves
                if pa is None and ra is None:
                    continue
                by = None
                if name.split("::")[-1] in lib_bytes:
                    by = lib_bytes[name.split("::")[-1]]
                elif bytes_fn is not None:
                    by = bytes_fn(grid)
                gb = lambda us: "-" if (by is None or us is None) else f"{by / us / 1e3:.0f}"
                ov, names = overlap_stats(P, sp.get(key, [])) if key in sp else (None, {})
                top = ", ".join(n.split("::")[-1][:28] for n, _ in collections.Counter(names).most_common(3))
                tr = None
                if key in fe and key in wr:
                    tr = (2 * fe[key] + wr[key]) * 1024
                    pmc[f"{name}@{grid}"] = {"fetch_kib": fe[key], "write_kib": wr[key], "traffic_bytes": tr, "alg_bytes": by}
                    if "polyabc" in name or "spmv3" in name:
                        # gather-shaped reads: profiles/r06_pmc_calibration.json (tools/fetch_calib.hip) - FETCH_SIZE counts a random 32-byte gather as the 64-byte
                        # request it is (no halving; only coalesced streams are tallied at half their bytes), so the guide's x2 over-counts these kernels.
                        # Bounds: every read a 64-byte request counted 1:1 (lower) .. the index / pointer / order streams perfectly coalesced and halved (upper)
                        lo = fe[key] * 1024 + wr[key] * 1024
                        pmc[f"{name}@{grid}"].update({"traffic_bytes_note": "traffic_bytes applies the guide's x2 to FETCH_SIZE; calibrated (r06_pmc_calibration.json): gathers are "
                                                      "counted 1:1, so the true traffic lies in traffic_bytes_calibrated_range",
                                                      "traffic_bytes_calibrated_range": [lo, lo + min(fe[key] * 1024, 59e6 / 2)]})
                site_rows[f"{name.split('::')[-1]}@{grid}"] = {"alg_bytes": by, "in_prove_median_us": pa, "in_prove_min_us": pm, "reference_order_median_us": ra, "solo_median_us": sa,
                                                               "solo_min_us": sm, "others_resident_frac": ov, "pmc_traffic_bytes": tr}
                f.write(f"| `{name.split('::')[-1]}` @ {grid} | {what} | {n_l:.1f} | {'-' if by is None else f'{by / 1e6:.1f}'} | {fmt(pa)} / {fmt(pm)} | {gb(pa)} | {fmt(ra)} / {fmt(rm)} | "
                        f"{fmt(sa)} / {fmt(sm)} | {gb(sa)} | {'-' if ov is None else f'{100 * ov:.0f} %'} {top} | {'-' if tr is None else f'{tr / 1e6:.1f}'} |\n")
        f.write("\nAlgorithmic bytes: " + "; ".join(f"`{s[0]}`: {s[4]}" for s in sites(M)) + ".\n")
        # whole-prove accounting
        for title, L in (("headline driver", P), ("reference-order driver", R)):
            per = collections.defaultdict(list)
            for name, grid, s, e in L:
                per[name].append((e - s) / 1e3)
            tot = sum(sum(v) for v in per.values())
            f.write(f"\n## All kernels, {title} ({nproves} proves + setup in the trace)\n\n| kernel | calls | total us | avg us | min us | % |\n|---|---|---|---|---|---|\n")
            for name, v in sorted(per.items(), key=lambda kv: -sum(kv[1]))[:45]:
                f.write(f"| `{name[:80]}` | {len(v)} | {sum(v):.1f} | {sum(v) / len(v):.2f} | {min(v):.2f} | {100 * sum(v) / tot:.2f} |\n")
    if a.sites_json:
        with open(a.sites_json, "w") as jf:
            json.dump(site_rows, jf, indent=1)
    if a.pmc_json:
        with open(a.pmc_json, "w") as jf:
            json.dump(pmc, jf, indent=1)


if __name__ == "__main__":
    main()

"""Register-spill gate: the gfx950 code objects inside spartan2_amd/lib/*.o are read with tools/spill_report.py (llvm-objcopy ->
clang-offload-bundler -> llvm-readelf --notes) and the kernels on the hot list below must not spill a single VGPR; the two kernel families that are
KEPT at three waves a SIMD with spills - measured faster than their spill-free two-wave forms, profiles/r06_spills.md - are pinned to their measured
counts so that a regression (or a silent change of the trade) shows up here. Runs without a GPU."""
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import spill_report  # noqa: E402

# kernel name prefixes (demangled, without parameter lists) on the path of a prove / prep_prove / commit that must stay spill-free
HOT_NO_SPILL = [
    "spk::k_bind_eval_cubic_stream", "spk::k_bind_eval_quad_stream", "spk::k_bind_eval_quad_stream_sparse", "spk::k_bind_eval_cubic", "spk::k_bind_eval_quad",
    "spk::k_eval_quad_stream_lowhi", "spk::k_eval_products_stream", "spk::k_round0_products", "spk::k_spmv3", "spk::k_polyabc_short_and_long",
    "spk::k_eq_outer_lastk", "spk::k_eq_levels_pair", "spk::k_sum_partials", "spk::k_sum_partials_lazy", "spk::k_mail_gate",
    "spk::k_sumcheck_tail",  # <true> spilled 22 VGPRs in round 5: eight hoisted check-word weights per slot store (slot_chk_add)
    "spk::k_rowmat_vec_tall",  # 26 in round 5 under capi_group.hip's max-ilp scheduling; now in capi_bulk.hip
    "spk::k_fixed_base_table", "spk::k_fixed_base_tables", "spk::k_fb_ladder", "spk::k_fb_fill", "spk::k_jac_to_affine_batch",
    "spk::k_msm_binary_rows", "k_expand_u64", "k_expand_bits", "spk::k_nifs_fold_prove",
]
# kept WITH spills on purpose: (prefix, waves-per-SIMD template argument, max spilled VGPRs). The <.., 2> forms of both must be spill-free.
MEASURED_TRADE = [("spk::k_comb_rows", 3, 68), ("spk::k_pip_bucket_tasks", 3, 66)]


def _base(name):
    name = re.sub(r"^void ", "", name)
    return re.sub(r"[<(].*$", "", name)


@pytest.fixture(scope="module")
def rows():
    lib = os.path.join(ROOT, "spartan2_amd", "lib")
    if not os.path.isdir(lib) or not [f for f in os.listdir(lib) if f.endswith(".o")]:
        pytest.skip("spartan2_amd/lib/*.o not built (run __graft_entry__.build())")
    r = spill_report.kernels(lib)
    assert len(r) > 100, "the code objects were not read"
    return r


def test_hot_kernels_do_not_spill(rows):
    seen = set()
    bad = []
    for r in rows:
        b = _base(r["name"])
        if b in HOT_NO_SPILL:
            seen.add(b)
            if r.get("vgpr_spill_count", 0):
                bad.append((r["name"], r["vgpr_spill_count"], r.get("private_segment_fixed_size", 0)))
    assert not bad, f"hot kernels spill VGPRs: {bad}"
    missing = [h for h in HOT_NO_SPILL if h not in seen]  # names on the list that no longer exist would make the gate vacuous
    assert not missing, f"hot-list names not found in the code objects: {missing}"


def test_measured_spill_trades_are_what_was_measured(rows):
    for prefix, minw, max_spills in MEASURED_TRADE:
        kept = [r for r in rows if _base(r["name"]) == prefix and re.search(rf", {minw}>", r["name"])]
        free = [r for r in rows if _base(r["name"]) == prefix and re.search(r", 2>", r["name"])]
        assert kept and free, f"{prefix}: both wave-count forms must be compiled"
        assert all(r.get("vgpr_spill_count", 0) <= max_spills for r in kept), [(r["name"], r.get("vgpr_spill_count")) for r in kept]
        assert all(r.get("vgpr_spill_count", 0) == 0 for r in free), [(r["name"], r.get("vgpr_spill_count")) for r in free]


def test_no_other_kernel_spills_more_than_a_couple(rows):
    """Everything outside the two measured trades: at most 2 spilled VGPRs (the cooperative-addition kernels of capi_group.hip sit at the 256-register
    limit of their 512-thread blocks with 1-2 cold spills each)."""
    allowed = tuple(p for p, _, _ in MEASURED_TRADE)
    worst = [(r["name"], r["vgpr_spill_count"]) for r in rows if r.get("vgpr_spill_count", 0) > 2 and _base(r["name"]) not in allowed]
    assert not worst, worst

"""CPU-side checks of the drop-in boundary: libspartan_hip.so loads, exports every symbol include/spartan_hip.h
declares, and refuses to run without a gfx950 device (no CPU fallback). No compute calls here."""
import ctypes
import os

import pytest

from spartan2_amd import hip


def test_library_is_built_in_tree():
    assert os.path.exists(hip.LIB_PATH), "run __graft_entry__.build() first"


def test_every_declared_symbol_is_exported():
    L = hip.lib()
    names = hip.declared_symbols()
    assert len(names) >= 25
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing


def test_no_cpu_fallback_without_a_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present; the refusal path is only observable on a CPU-only box")
    h = ctypes.c_void_p()
    rc = hip.lib().sp_ctx_create(0, ctypes.byref(h))
    assert rc == -100  # SP_ERR_NO_DEVICE
    assert b"no HIP device" in hip.lib().sp_last_error() or b"hip" in hip.lib().sp_last_error().lower()


def test_transcript_entry_points_match_reference_kat_on_cpu():
    """The transcript is host code inside the library, so its KAT can run without a GPU... but it squeezes into the
    bench field (T256 scalar), for which the reference holds no KAT; cross-check against the oracle instead."""
    import numpy as np

    import oracle_lib as ol

    t = hip.Transcript(None, b"test")
    o = ol.Transcript(b"test")
    for lbl, val in ((b"s1", 2), (b"s2", 5)):
        be = val.to_bytes(32, "big")
        t.absorb(lbl, be)
        o.absorb(lbl, be)
    assert (t.squeeze(b"c1") == o.squeeze(b"c1", fid=0)).all()
    t.absorb(b"s3", (128).to_bytes(32, "big"))
    o.absorb(b"s3", (128).to_bytes(32, "big"))
    t.dom_sep(b"inner product argument (linear)")
    ol.lib().orc_transcript_dom_sep(o.h, b"inner product argument (linear)")
    for _ in range(3):
        assert (t.squeeze(b"c2") == o.squeeze(b"c2", fid=0)).all()
    big = bytes(range(256)) * 5
    t.absorb(b"blob", big)
    o.absorb(b"blob", big)
    assert (t.squeeze(b"r") == o.squeeze(b"r", fid=0)).all()


def test_rust_ffi_module_is_generated_from_the_header():
    """integration/hip_ffi.rs (the `extern "C"` module of the reference-side binding) is what tools/gen_rust_ffi.py emits for the current header, and it
    declares every function of include/spartan_hip.h."""
    import importlib.util
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("gen_rust_ffi", os.path.join(root, "tools", "gen_rust_ffi.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    text, names = gen.emit()
    with open(os.path.join(root, "integration", "hip_ffi.rs")) as f:
        assert f.read() == text, "run python tools/gen_rust_ffi.py"
    from spartan2_amd import hip

    assert sorted(names) == hip.declared_symbols()
    for n in names:
        assert f"pub fn {n}(" in text
    # the hand-written halves of the binding (uncompiled here: no rustc) at least call functions that exist, with the declared number of arguments
    import re

    arity = {m.group(1): (0 if not m.group(2).strip() else m.group(2).count(":")) for m in re.finditer(r"pub fn (sp_\w+)\((.*?)\)(?: ->|;)", text)}
    for fname in ("hip_provider.rs", "hip_r1cs_pcs.rs"):
        src = open(os.path.join(root, "integration", fname)).read()
        src = re.sub(r"//[^\n]*", "", src)
        for m in re.finditer(r"\b(sp_[a-z0-9_]+)\s*\(", src):
            name = m.group(1)
            assert name in arity, (fname, name)
            # argument count of the call: split the balanced parenthesis contents at top-level commas
            i, depth, args, cur = m.end(), 1, 0, ""
            while depth:
                ch = src[i]
                depth += ch in "([{"
                depth -= ch in ")]}"
                if depth == 1 and ch == ",":
                    args += 1
                    cur = ""
                elif depth >= 1:
                    cur += ch
                i += 1
            args += 1 if cur.strip() else 0
            assert args == arity[name], (fname, name, args, arity[name])

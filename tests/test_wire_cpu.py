"""CPU: the PRODUCT's wire formats (include/spartan_hip.h "wire formats", spartan2_amd/csrc/capi_wire.hip — host code of libspartan_hip.so, no device call)
against the independent Python writer tests/pywire.py, hashlib, and the oracle's bytes: SHA-256 (both block functions), SpartanSNARK bincode bytes from the
flat word layout and back, the SpartanVerifierKey digest stream, rejection of malformed input. The proofs fed in are the oracle's (the GPU tests feed the
device's own: tests/test_gpu_wire.py)."""
import hashlib
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle_lib as ol
import pywire
from spartan2_amd import frontend, hip, host
from test_oracle_wire import CASES, _layout, _prove


def test_sha256_both_block_functions():
    rng = np.random.default_rng(11)
    for n in list(range(0, 200)) + [4095, 4096, 65535, 65536, 65537, (1 << 20) + 3]:
        data = rng.integers(0, 256, size=n, dtype=np.uint8).tobytes()
        assert hip.sha256(data) == hashlib.sha256(data).digest(), n
    # the portable block function, in a process of its own (the choice is made once per process)
    code = ("import hashlib, numpy as np\nfrom spartan2_amd import hip\nassert hip.lib().sp_sha256_accelerated() == 0\n"
            "rng = np.random.default_rng(12)\n"
            "for n in list(range(0, 130)) + [4097, 70001]:\n"
            "    d = rng.integers(0, 256, size=n, dtype=np.uint8).tobytes()\n"
            "    assert hip.sha256(d) == hashlib.sha256(d).digest(), n\n")
    env = dict(os.environ, SPARTAN_SHA_PORTABLE="1", PYTHONPATH=hip.ROOT)
    subprocess.check_call([sys.executable, "-c", code], env=env)


def _product_layout(sp):
    lay = _layout(sp)
    return dict(rows_shared=lay["rows_shared"], rows_precommitted=lay["rows_pre"], rows_rest=lay["rows_rest"], num_public=lay["num_public"],
                num_challenges=lay["num_challenges"], rounds_x=lay["lx"], rounds_y=lay["ly"], z_len=lay["nz"])


@pytest.mark.parametrize("name", list(CASES))
def test_proof_serialize_matches_python_writer_and_oracle(name):
    sp, words = _prove(CASES[name](), 4)
    L = _product_layout(sp)
    data = hip.proof_serialize(L, words)
    assert data == pywire.spartan_proof_bytes(words, **_layout(sp))
    assert data == sp.proof_to_bytes(words)
    L2, back = hip.proof_deserialize(data)
    assert L2 == L and (back == words).all()
    with pytest.raises(hip.SpartanHipError):
        hip.proof_serialize(L, words[:-4])
    for bad in (data + b"\0", data[:-1], data[:-33], data[: len(data) // 2], b""):
        with pytest.raises(hip.SpartanHipError):
            hip.proof_deserialize(bad)
    b = bytearray(data)
    b[-32:] = pywire.P_SCALAR.to_bytes(32, "little")
    with pytest.raises(hip.SpartanHipError, match="non-canonical"):
        hip.proof_deserialize(bytes(b))
    b = bytearray(data)
    off = 1 if L["rows_shared"] == 0 else 0
    b[off + 1:off + 9] = (1 << 61).to_bytes(8, "little")
    with pytest.raises(hip.SpartanHipError, match="length prefix"):
        hip.proof_deserialize(bytes(b))


def test_proof_deserialize_accepts_any_jacobian_representative_and_checks_the_curve():
    sp, words = _prove(frontend.synthetic_circuit(6, 21, num_public=1), 8)
    data = bytearray(hip.proof_serialize(_product_layout(sp), words))
    off, P = 1 + 1 + 8, pywire.P_BASE
    x, y, z = (int.from_bytes(data[off + 32 * i:off + 32 * i + 32], "little") for i in range(3))
    assert z == 1
    lam = 0xABCDEF0123456789ABCDEF
    for i, v in enumerate((x * lam * lam % P, y * pow(lam, 3, P) % P, lam)):
        data[off + 32 * i:off + 32 * i + 32] = v.to_bytes(32, "little")
    _, back = hip.proof_deserialize(bytes(data))
    assert (back == words).all()
    data[off + 32:off + 64] = ((y * pow(lam, 3, P) + 1) % P).to_bytes(32, "little")
    with pytest.raises(hip.SpartanHipError, match="curve"):
        hip.proof_deserialize(bytes(data))
    # the identity: z = 0 whatever x and y say
    data = bytearray(hip.proof_serialize(_product_layout(sp), words))
    data[off + 64:off + 96] = bytes(32)
    _, back = hip.proof_deserialize(bytes(data))
    assert not back[:8].any() and (back[8:] == words[8:]).all()


@pytest.mark.parametrize("name", list(CASES))
def test_vk_digest_matches_python_stream_and_oracle(name):
    inst = CASES[name]()
    mats, dims = host.pad_shape(inst)
    gens, gens_s = host.from_label(b"ck", 2049), host.from_label(b"ck_s", 2)
    got = hip.vk_digest([dims[k] for k in host.DIM_NAMES], mats, gens[:2048], gens[2048], gens_s[:1], gens_s[1])
    assert got == pywire.spartan_vk_digest(inst, gens[:2048], gens[2048], gens_s[:1], gens_s[1])
    assert got == ol.OracleSpartan(inst).export_keys()[4].tobytes()

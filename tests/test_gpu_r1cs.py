"""GPU parity for the sparse R1CS kernels: multiply_vec, incremental multiply_vec, poly_ABC — C ABI vs the CPU oracle on a
seeded synthetic circuit and a one-block SHA-256 circuit (the bench circuit's shape at small size)."""
import ctypes

import numpy as np
import pytest

import oracle_lib as ol
from oracle_lib import lib as olib, p64
from spartan2_amd import frontend, hip
from spartan2_amd.host import pad_shape

pytestmark = pytest.mark.gpu
SEED = 0xDEADBEEF


@pytest.fixture(scope="module")
def ctx():
    c = hip.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("which", ["synthetic", "sha256_1block"])
def test_spmv_and_poly_abc(ctx, which):
    inst = frontend.synthetic_circuit(40, SEED, num_public=3) if which == "synthetic" else frontend.sha256_circuit(b"abc")
    oshape = ol.OracleShape(inst)
    mats, dims = pad_shape(inst)
    assert dims["num_cons"] == oshape.num_cons and dims["num_precommitted"] == oshape.num_precommitted and dims["num_rest"] == oshape.num_rest
    shape = hip.Shape(ctx, mats, dims)
    rng = np.random.default_rng(SEED)
    N, M, ncols = oshape.num_cons, oshape.num_vars, oshape.num_vars + oshape.num_extra
    # the real assignment z = [W | 1 | X] and a random z
    W = np.zeros((M, 4), dtype=np.uint64)
    W[oshape.num_shared : oshape.num_shared + len(inst.witness)] = ol.mont_array([int(x) for x in inst.witness])
    z_real = np.concatenate([W, ol.mont_array([1] + [int(x) for x in inst.publics])])
    for z in (z_real, ol.random_field_array(rng, ncols)):
        want = [np.zeros((N, 4), dtype=np.uint64) for _ in range(3)]
        assert olib().orc_shape_multiply_vec(oshape.h, p64(z), *(p64(w) for w in want)) == 0
        outs = [hip.Table.zeros(ctx, N) for _ in range(3)]
        shape.multiply_vec(hip.Table.from_host(ctx, z), *outs)
        for o, w in zip(outs, want):
            assert (o.read() == w).all()
    # satisfied instance: Az o Bz == Cz
    az, bz, cz = (o.read() for o in outs) if False else want
    # incremental: cached products of the precommitted part + the rest/public columns
    zc = z_real.copy()
    zc[oshape.num_shared + oshape.num_precommitted :] = 0
    cached = [np.zeros((N, 4), dtype=np.uint64) for _ in range(3)]
    assert olib().orc_shape_multiply_vec(oshape.h, p64(zc), *(p64(w) for w in cached)) == 0
    want = [np.zeros((N, 4), dtype=np.uint64) for _ in range(3)]
    assert olib().orc_shape_multiply_vec_incremental(oshape.h, p64(z_real), *(p64(c) for c in cached), *(p64(w) for w in want)) == 0
    full = [np.zeros((N, 4), dtype=np.uint64) for _ in range(3)]
    olib().orc_shape_multiply_vec(oshape.h, p64(z_real), *(p64(w) for w in full))
    for a, b in zip(want, full):
        assert (a == b).all()
    outs = [hip.Table.zeros(ctx, N) for _ in range(3)]
    shape.multiply_vec_incremental(hip.Table.from_host(ctx, z_real), *(hip.Table.from_host(ctx, c) for c in cached), *outs)
    for o, w in zip(outs, want):
        assert (o.read() == w).all()
    # poly_ABC, compact and full-size variants
    rx = ol.random_field_array(rng, N)
    r = ol.random_field_array(rng, 1)[0]
    for out_len in (ncols, 2 * M):
        want = np.zeros((out_len, 4), dtype=np.uint64)
        assert olib().orc_shape_poly_abc(oshape.h, p64(rx), p64(r), ctypes.c_size_t(out_len), p64(want)) == 0
        out = hip.Table.from_host(ctx, ol.random_field_array(rng, 16).repeat((2 * M + 15) // 16, axis=0)[: 2 * M])  # dirty buffer
        shape.poly_abc(hip.Table.from_host(ctx, rx), r, out_len, out)
        assert (out.read(0, out_len) == want).all()
    # twice more in a row on a dirty buffer (the per-column arrival counters of the long columns must be back at zero after every call)
    for _ in range(2):
        out = hip.Table.from_host(ctx, ol.random_field_array(rng, 16).repeat((2 * M + 15) // 16, axis=0)[: 2 * M])
        shape.poly_abc(hip.Table.from_host(ctx, rx), r, 2 * M, out)
        assert (out.read(0, 2 * M) == want).all()

"""TEST INFRASTRUCTURE ONLY: Python-integer stand-ins for sp_sumcheck_cubic3_sharded / sp_sumcheck_quad_sharded (include/spartan_hip.h), so the
slice-sharding logic of spartan2_amd.dist runs in CPU-only multi-process tests. Follows src/sumcheck.rs:502-571 + :1276-1324 (claim-derived
evaluations) and :190-247; tiny sizes only. The product path never imports this."""
import numpy as np

import oracle_lib as ol

P = ol.MODULI[0]
INV2, INV6 = pow(2, -1, P), pow(6, -1, P)


def _ints(a):
    return ol.ints_of(np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4))


def _limbs(v):
    return ol.mont_array([x % P for x in v])


def _le32(v):
    return (v % P).to_bytes(32, "little")


def _eq_table(taus):
    t = [1]
    for tau in taus:  # first variable = MSB
        t = [y for x in t for y in (x * (1 - tau) % P, x * tau % P)]
    return t


def cubic_sharded(tr, claim, p, taus, A, B, C, scale, reduce):
    claim, p = ol.from_mont(np.asarray(claim)), ol.from_mont(np.asarray(p))
    taus_i, a, b, c = _ints(taus), _ints(A), _ints(B), _ints(C)
    sc = ol.from_mont(np.asarray(scale)) if scale is not None else 1
    polys, rs = [], []
    for rnd, tau in enumerate(taus_i):
        half = len(a) // 2
        E = _eq_table(taus_i[rnd + 1 :])
        t0 = sum(E[i] * (a[i] * b[i] - c[i]) for i in range(half)) % P
        tinf = sum(E[i] * (a[half + i] - a[i]) * (b[half + i] - b[i]) for i in range(half)) % P
        sums = [t0 * sc % P, tinf * sc % P]
        if reduce is not None:
            sums = _ints(reduce(_limbs(sums)))
        t0, tinf = sums
        eq0, slope = (1 - tau) % P, (2 * tau - 1) % P
        eqm1 = (eq0 - slope) % P
        s0 = eq0 * p * t0 % P
        s1 = (claim - s0) % P
        t1 = s1 * pow(tau * p % P, -1, P) % P
        slead = slope * p * tinf % P
        sm1 = eqm1 * p * ((2 * tinf + 2 * t0 - t1) % P) % P
        c1 = ((s1 - sm1) * INV2 - slead) % P
        c2 = ((s1 + sm1) * INV2 - s0) % P
        e2 = (s0 + 2 * (c1 + 2 * (c2 + 2 * slead))) % P
        e3 = (s0 + 3 * (c1 + 3 * (c2 + 3 * slead))) % P
        ev = [s0, (claim - s0) % P, e2, e3]
        d = ev[0]
        ca = (ev[3] - 3 * ev[2] + 3 * ev[1] - ev[0]) * INV6 % P
        cb = ((ev[2] - 2 * ev[1] + ev[0]) * INV2 - 3 * ca) % P
        cc = (ev[1] - d - cb - ca) % P
        tr.absorb(b"p", _le32(d) + _le32(cb) + _le32(ca))  # compressed: constant, then degree >= 2 (src/polys/univariate.rs:147-190)
        r = ol.from_mont(tr.squeeze(b"c"))
        polys.append([d, cb, ca])
        rs.append(r)
        claim = (d + cc * r + cb * r * r + ca * r ** 3) % P
        a, b, c = ([(x[i] + r * (x[half + i] - x[i])) % P for i in range(half)] for x in (a, b, c))
        p = p * ((1 - tau - r + 2 * r * tau) % P) % P
    return (np.stack([_limbs(x) for x in polys]), _limbs(rs), _limbs([a[0], b[0], c[0]]), ol.to_mont(claim), ol.to_mont(p))


def quad_sharded(tr, claim, rounds, A, B, reduce):
    claim = ol.from_mont(np.asarray(claim))
    a, b = _ints(A), _ints(B)
    polys, rs = [], []
    for _ in range(rounds):
        half = len(a) // 2
        sums = [sum(a[i] * b[i] for i in range(half)) % P, sum((a[half + i] - a[i]) * (b[half + i] - b[i]) for i in range(half)) % P]
        if reduce is not None:
            sums = _ints(reduce(_limbs(sums)))
        e0, tinf = sums
        e2 = (2 * claim - 3 * e0 + 2 * tinf) % P
        e1 = (claim - e0) % P
        ca = (e0 - 2 * e1 + e2) * INV2 % P
        cb = (e1 - e0 - ca) % P
        tr.absorb(b"p", _le32(e0) + _le32(ca))
        r = ol.from_mont(tr.squeeze(b"c"))
        polys.append([e0, ca])
        rs.append(r)
        claim = (e0 + cb * r + ca * r * r) % P
        a, b = ([(x[i] + r * (x[half + i] - x[i])) % P for i in range(half)] for x in (a, b))
    return np.stack([_limbs(x) for x in polys]), _limbs(rs), _limbs([a[0], b[0]]), ol.to_mont(claim)

"""TEST INFRASTRUCTURE ONLY: a pure-Python-integer stand-in for spartan2_amd.hip.Nifs (the sp_nifs_* C ABI), so the sharding logic of
spartan2_amd.dist.nifs_rounds_sharded can run in CPU-only multi-process tests. Same interface, same semantics as the device object
(include/spartan_hip.h "NeutronNova NIFS rounds"), tiny sizes only. The product path never imports this."""
import numpy as np

import oracle_lib as ol

P = ol.MODULI[0]


def _ints(a):
    return ol.ints_of(np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4))


def _limbs(vals):
    return ol.mont_array([v % P for v in vals])


class _View:
    def __init__(self, store, key):
        self.store, self.key = store, key

    def free(self):
        pass


class PyNifs:
    def __init__(self, n_padded, left, right):
        self.n_padded, self.left, self.right, self.total = n_padded, left, right, left * right
        self.L = {(w, b): [0] * self.total for w in range(3) for b in range(n_padded)}  # original layers
        self.cur = None

    # layer access (the tests' read_layer / write_layer callbacks use these views)
    def layer(self, which, idx):
        return _View(self.L, (which, idx))

    def current_layer(self, which, idx):
        return _View(self.cur, (which, idx))

    @staticmethod
    def read_layer(view):
        return _limbs(view.store[view.key])

    @staticmethod
    def write_layer(view, arr):
        view.store[view.key] = _ints(arr)

    def _E(self, k):
        return self.e[k % self.left] * self.e[self.left + k // self.left] % P

    def _weight(self, t, pair):
        w, k = 1, pair
        for s in range(t + 1, self.ell_b):
            w = w * (self.rhos[s] if k & 1 else (1 - self.rhos[s])) % P
            k >>= 1
        return w

    def begin_shard(self, E_eq, rhos, first_instance, small_values=False):
        self.e, self.rhos = _ints(E_eq), _ints(rhos)
        self.ell_b, self.first = len(self.rhos), first_instance
        self.cur = {(w, b): list(self.L[(w, b)]) for w in range(2) for b in range(self.n_padded)}
        self.m, self.t_done, self.r_bs, self.prefix = self.n_padded, 0, [], []
        self.T_cur, self.acc_eq, self.fold_pending_, self.poly = 0, 1, False, None
        self.cv = [0] * (1 << self.ell_b)
        for b in range(self.n_padded):
            self.cv[first_instance + b] = sum(self._E(k) * self.L[(2, b)][k] for k in range(self.total)) % P

    def begin(self, E_eq, rhos, small_values=False):
        self.begin_shard(E_eq, rhos, 0, small_values)

    def cvals(self):
        return _limbs(self.cv[self.first : self.first + self.n_padded])

    def set_cvals(self, allv):
        self.cv = _ints(allv)

    def _fold(self):
        r = self.r_bs[-1]
        new = {}
        for w in range(2):
            for i in range(self.m // 2):
                lo, hi = self.cur[(w, 2 * i)], self.cur[(w, 2 * i + 1)]
                new[(w, i)] = [(a + r * (b - a)) % P for a, b in zip(lo, hi)]
        self.cur, self.m, self.fold_pending_ = new, self.m // 2, False

    def fold_pending(self):
        self._fold()

    def round_sums(self, t):
        assert t == self.t_done and self.poly is None
        if t > 0 and self.fold_pending_:
            self._fold()
        pairs = self.m // 2
        base = self.first >> (t + 1)
        e0 = quad = 0
        for p in range(pairs):
            a0, a1, b0, b1 = (self.cur[(0, 2 * p)], self.cur[(0, 2 * p + 1)], self.cur[(1, 2 * p)], self.cur[(1, 2 * p + 1)])
            w = self._weight(t, base + p)
            q = sum(self._E(k) * (a1[k] - a0[k]) * (b1[k] - b0[k]) for k in range(self.total)) % P
            quad = (quad + w * q) % P
            if t > 0:
                ab = sum(self._E(k) * a0[k] * b0[k] for k in range(self.total)) % P
                npre = len(self.prefix)
                cval = sum(self.prefix[v] * self.cv[2 * (base + p) * npre + v] for v in range(npre)) % P
                e0 = (e0 + w * (ab - cval)) % P
        return _limbs([e0, quad])

    def round_finish(self, t, sums):
        e0, quad = _ints(sums)
        rho = self.rhos[t]
        c, a = e0 * self.acc_eq % P, quad * self.acc_eq % P
        abc = (self.T_cur - c * (1 - rho)) * pow(rho, -1, P) % P
        b = (abc - a - c) % P
        two = (2 * rho - 1) % P
        self.poly = [c * (1 - rho) % P, (c * two + b * (1 - rho)) % P, (b * two + a * (1 - rho)) % P, a * two % P]
        return _limbs(self.poly)

    def round(self, t):
        return self.round_finish(t, self.round_sums(t))

    def challenge(self, r_b):
        r = ol.from_mont(np.ascontiguousarray(r_b, dtype=np.uint64))
        rho = self.rhos[self.t_done]
        self.r_bs.append(r)
        self.acc_eq = self.acc_eq * ((1 - r) * (1 - rho) + r * rho) % P
        self.T_cur = sum(c * pow(r, i, P) for i, c in enumerate(self.poly)) % P
        self.prefix = [(1 - r) % P, r] if not self.prefix else [x * (1 - r) % P for x in self.prefix] + [x * r % P for x in self.prefix]
        self.t_done += 1
        self.poly, self.fold_pending_ = None, True

    def state(self):
        return ol.to_mont(self.T_cur), ol.to_mont(self.acc_eq)

    def resume(self, E_eq, rhos, t_start, r_bs, T_cur, acc_eq, c_vals_all):
        self.e, self.rhos = _ints(E_eq), _ints(rhos)
        self.ell_b, self.first = len(self.rhos), 0
        self.cur = {(w, b): list(self.L[(w, b)]) for w in range(2) for b in range(self.n_padded)}
        self.m, self.t_done, self.r_bs = self.n_padded, t_start, _ints(r_bs)
        self.prefix = []
        for r in self.r_bs:
            self.prefix = [(1 - r) % P, r] if not self.prefix else [x * (1 - r) % P for x in self.prefix] + [x * r % P for x in self.prefix]
        self.T_cur, self.acc_eq = ol.from_mont(np.ascontiguousarray(T_cur)), ol.from_mont(np.ascontiguousarray(acc_eq))
        self.cv, self.fold_pending_, self.poly = _ints(c_vals_all), False, None

    def finish_ab(self):
        """final fold -> (A, B) limb arrays, T_out, eq_rho_at_rb (the C fold is per shard)."""
        self._fold()
        T_out = self.T_cur * pow(self.acc_eq, -1, P) % P
        return _limbs(self.cur[(0, 0)]), _limbs(self.cur[(1, 0)]), ol.to_mont(T_out), ol.to_mont(self.acc_eq)

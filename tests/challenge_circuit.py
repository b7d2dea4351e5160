"""A circuit with a VERIFIER CHALLENGE (SpartanCircuit::num_challenges > 0, src/traits/circuit.rs; witness staging bellpepper/r1cs.rs:429-461), built by
hand as an integer R1CS: K independent gadgets, each
    precommitted x, x2 with x * x = x2;       rest w, u with x * c = w and (w + x2) * 1 = u;       one public y per gadget with y * 1 = x2,
where c is the single challenge the prover squeezes AFTER committing to the precommitted segment, so the rest segment (w, u) can only be
synthesized inside prove(). Columns: [x.., x2.. | w.., u.. | 1 | y.. | c]."""
import numpy as np

P = 0xFFFFFFFF00000001000000000000000000000000FFFFFFFFFFFFFFFFFFFFFFFF


class ChallengeCircuit:
    def __init__(self, K=300, seed=5):
        rng = np.random.default_rng(seed)
        self.K = K
        self.x = [int(v) for v in rng.integers(2, 1 << 20, size=K)]
        self.num_cons = 4 * K
        self.num_shared, self.num_precommitted, self.num_rest = 0, 2 * K, 2 * K
        self.num_public, self.num_challenges = K, 1
        X, X2, W, U = (lambda i: i), (lambda i: K + i), (lambda i: 2 * K + i), (lambda i: 3 * K + i)
        ONE, Y, C = 4 * K, (lambda i: 4 * K + 1 + i), 4 * K + 1 + K
        rows = []
        for i in range(K):
            rows.append(([(X(i), 1)], [(X(i), 1)], [(X2(i), 1)]))
            rows.append(([(X(i), 1)], [(C, 1)], [(W(i), 1)]))
            rows.append(([(X2(i), 1), (W(i), 1)], [(ONE, 1)], [(U(i), 1)]))
            rows.append(([(Y(i), 1)], [(ONE, 1)], [(X2(i), 1)]))
        self.csr = []
        for m in range(3):
            data, idx, ptr = [], [], [0]
            for r in rows:
                for col, val in sorted(r[m]):
                    idx.append(col)
                    data.append(val)
                ptr.append(len(idx))
            self.csr.append((np.array(data, dtype=np.int64), np.array(idx, dtype=np.uint32), np.array(ptr, dtype=np.uint64)))
        self.witness = np.array(self.x + [v * v for v in self.x] + [0] * (2 * K), dtype=np.uint64)  # the rest segment is synthesized in prove()
        self.publics = np.array([v * v for v in self.x], dtype=np.uint64)

    def synthesize(self, to_mont, from_mont):
        """-> callback(challenges (1, 4) limbs) -> rest witness (2K, 4) limbs"""

        def cb(ch):
            c = from_mont(ch[0])
            w = [c * v % P for v in self.x]
            u = [(wi + v * v) % P for wi, v in zip(w, self.x)]
            return np.stack([to_mont(t) for t in w + u])

        return cb

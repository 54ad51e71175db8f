"""Deterministic synthetic RNA inputs for tests and bench.py.

A self-contained splitmix64 generator is used instead of a library RNG so that the GPU box
regenerates byte-identical FASTA whatever numpy/Python version it has; tests/golden holds the
checksums of the standard sets.
"""
import hashlib

_MASK = (1 << 64) - 1


class SplitMix64:
    def __init__(self, seed):
        self.s = seed & _MASK

    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & _MASK
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _MASK
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _MASK
        return z ^ (z >> 31)

    def below(self, n):
        return self.next() % n

    def uniform(self):
        return (self.next() >> 11) / float(1 << 53)


def random_set(n, length, seed=12345, jitter=0.07):
    """n unrelated sequences, i.i.d. uniform ACGU, lengths uniform in [0.93L, 1.07L] (jitter=0: exact)."""
    rng = SplitMix64(seed)
    lo = int(round(length * (1.0 - jitter)))
    hi = int(round(length * (1.0 + jitter)))
    out = []
    for i in range(n):
        L = lo + rng.below(hi - lo + 1)
        out.append(("s%d" % i, "".join("ACGU"[rng.below(4)] for _ in range(L))))
    return out


def family_set(n, length, seed=12346, sub=0.15, indel=0.05):
    """one random root, each member mutated independently (substitutions + indels)."""
    rng = SplitMix64(seed)
    root = ["ACGU"[rng.below(4)] for _ in range(length)]
    out = []
    for i in range(n):
        s = []
        for ch in root:
            u = rng.uniform()
            if u < indel / 2:
                continue  # deletion
            if u < indel:
                s.append(ch)
                s.append("ACGU"[rng.below(4)])  # insertion
                continue
            if u < indel + sub:
                s.append("ACGU"[rng.below(4)])
            else:
                s.append(ch)
        if not s:
            s = ["A"]
        out.append(("f%d" % i, "".join(s)))
    return out


def to_fasta(records):
    return "".join(">%s\n%s\n" % (n, s) for n, s in records)


def checksum(records):
    return hashlib.sha256(to_fasta(records).encode()).hexdigest()

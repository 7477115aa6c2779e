"""Seeded synthetic inputs shared by tests, golden generation and bench.py.

``planted_topics_csr`` is the host (numpy) form of the planted-topic count
model of SURVEY.md §8(d): every cell belongs to one of T topics, every topic
up-weights a random 5 % subset of peaks over a Gamma(2,1) background, and the
count of (cell, peak) is Poisson with mean depth_i * p_topic(i)[peak].  It gives
a spectrum with T well separated singular values followed by a bulk, which is
what makes the LSI subspace-angle target well posed (SURVEY.md §7 hard part 1).
The device-side generator used at full scale (csrc/synth.hip) draws from the
same model with counter-based hashing instead of a numpy Generator.
"""
import numpy as np
import scipy.sparse as sp


def planted_topics_csr(n, d, n_topics=50, density=0.03, seed=0, dtype=np.float32, chunk=2048):
    rng = np.random.default_rng(seed)
    bg = rng.gamma(2.0, 1.0, size=d)
    bg /= bg.sum()
    p = np.empty((n_topics, d))
    for t in range(n_topics):
        w = np.zeros(d)
        sel = rng.choice(d, size=max(1, int(0.05 * d)), replace=False)
        w[sel] = rng.gamma(2.0, 1.0, size=sel.size)
        w /= w.sum()
        p[t] = 0.5 * bg + 0.5 * w
    topic = rng.integers(0, n_topics, size=n)
    # Poisson thinning loses ~15% of draws to multiplicity; aim slightly above density
    depth = np.maximum(50.0 if d >= 2000 else 5.0, rng.lognormal(np.log(1.15 * density * d), 0.3, size=n))
    blocks = []
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        lam = depth[s:e, None] * p[topic[s:e]]
        c = rng.poisson(lam)
        # no empty rows (the reference divides by the row sum)
        empty = np.nonzero(c.sum(axis=1) == 0)[0]
        for r in empty:
            c[r, rng.integers(0, d)] = 1
        blocks.append(sp.csr_matrix(c.astype(dtype)))
    X = sp.vstack(blocks, format="csr")
    X.sort_indices()
    X.indices = X.indices.astype(np.int32)
    X.indptr = X.indptr.astype(np.int64)
    return X


def unstructured_csr(n, d, density=0.03, seed=0, dtype=np.float32):
    """Uniformly random pattern, values 1+Poisson(0.5)."""
    rng = np.random.default_rng(seed)
    X = sp.random(n, d, density=density, format="csr", random_state=rng, dtype=np.float64)
    X.data = (1 + rng.poisson(0.5, size=X.nnz)).astype(dtype)
    X = X.astype(dtype)
    X.sort_indices()
    X.indices = X.indices.astype(np.int32)
    X.indptr = X.indptr.astype(np.int64)
    return X

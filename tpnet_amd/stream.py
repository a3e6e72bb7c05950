"""Synthetic temporal edge streams shaped like the reference's datasets (the datasets themselves are
not in the reference tree: DG_data/ holds only a README).  Host-side numpy only."""
import numpy as np

# name -> (U users, I items, E edges, time span [s], lambda, d, batch)   -- BASELINE.json configs C1..C5
CONFIGS = {
    "C1": dict(U=8227, I=1000, E=157474, span=2.678e6, lam=1e-6, d=64, B=200, desc="Wikipedia-shape d=64 B=200"),
    "C2": dict(U=8227, I=1000, E=157474, span=2.678e6, lam=1e-6, d=128, B=1000, desc="Wikipedia-shape d=128 B=1000"),
    "C3": dict(U=10000, I=984, E=672447, span=2.678e6, lam=1e-6, d=256, B=10000, desc="Reddit-shape d=256 B=10000"),
    # C4 on ONE GPU: 10 M nodes, d=256 -> 72 GB of state, far beyond the 256 MB Infinity Cache: the HBM-bound case
    "C4": dict(U=5000000, I=5000000, E=200000000, span=2.678e6 * 64, lam=1e-7, d=256, B=10000,
               desc="synthetic 10 M nodes power-law stream d=256 B=10000"),
    "C5": dict(U=980, I=1000, E=1293103, span=1.37e8, lam=1e-7, d=512, B=10000, desc="LastFM-shape d=512 B=10000"),
}


def synthetic_stream(U: int, I: int, E: int, span: float, seed: int = 0, pu: float = 2.0, pi: float = 3.0):
    """S(U, I, E, span, seed): bipartite power-law stream.  ids: users 1..U, items U+1..U+I, row 0 = padding
    (preprocess_data/preprocess_data.py:56-81,101-108).  Returns src, dst (int64), t (sorted float64), N.
    pu / pi: exponents of the degree law of users / items (SURVEY.md section 8d fixes 2.0 / 3.0; 1.0 = uniform; other
    values are for the sensitivity runs of tools/degree_sensitivity.py)."""
    rng = np.random.RandomState(seed)
    perm_u = np.random.RandomState(seed + 100).permutation(U)
    perm_i = np.random.RandomState(seed + 200).permutation(I)
    su = np.minimum(np.floor(U * rng.random_sample(E) ** pu).astype(np.int64), U - 1)
    si = np.minimum(np.floor(I * rng.random_sample(E) ** pi).astype(np.int64), I - 1)
    src = (1 + perm_u[su]).astype(np.int64)
    dst = (U + 1 + perm_i[si]).astype(np.int64)
    t = np.sort(rng.uniform(0, span, E)).astype(np.float64)
    return src, dst, t, U + I + 1


def synthetic_negatives(U: int, N: int, E: int, B: int, seed: int = 1):
    """Uniform random item per edge, drawn batch by batch from one advancing RandomState (the reference's
    NegativeEdgeSampler.random_sample draws per batch: utils/utils.py:388-400)."""
    rng = np.random.RandomState(seed)
    out = np.empty(E, dtype=np.int64)
    for b in range(0, E, B):
        n = min(B, E - b)
        out[b:b + n] = rng.randint(U + 1, N, n)
    return out


def bytes_per_edge(d: int, L: int = 3) -> int:
    """Algorithmic bytes per edge unit: (9L+3) rows of 4d bytes + two (2L+2)^2 f32 outputs +
    32 bytes of inputs (src, dst, neg ids + timestamp)."""
    return 4 * d * (9 * L + 3) + 8 * (2 * L + 2) ** 2 + 32

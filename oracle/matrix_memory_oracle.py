"""TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy) of the reference's `MatrixMemory`, PINT's walk-matrix state
(models/MemoryModel.py:364-420).  Pinned by tests/golden/g9_*.npz, which were produced by importing
the reference's own class (tests/golden/make_golden_f4.py).  Only tests/ may import this module.

State: matrix [N, N, H+1] f32; matrix[:, :, 0] = I after reset (MemoryModel.py:381-385 -- reset rewrites hop 0 only).
The shift matrix P has P[j][j-1] = 1 (j = 1..H) (:376-377), so a message row is the partner's row moved DOWN one hop:
    message[..., k] = matrix[partner][..., k+1]  (k < H),   message[..., H] = 0
which makes every message zero as long as only hop 0 is populated: from the reset state the reference's matrix never
changes.  That behaviour is reproduced, not repaired.
"""
import numpy as np


def shift_matrix(H: int) -> np.ndarray:
    """MemoryModel.py:376-377."""
    P = np.zeros((H + 1, H + 1), dtype=np.float32)
    P[1:, :-1] = np.eye(H, dtype=np.float32)
    return P


def reset(N: int, H: int, matrix: np.ndarray = None) -> np.ndarray:
    """MemoryModel.py:374,381-385: zeros at construction; reset_memory only sets hop 0 to the identity."""
    if matrix is None:
        matrix = np.zeros((N, N, H + 1), dtype=np.float32)
    matrix[:, :, 0] = np.eye(N, dtype=np.float32)
    return matrix


def update(matrix: np.ndarray, src: np.ndarray, dst: np.ndarray) -> None:
    """MemoryModel.py:387-394, in place.  Messages are computed from the PRE-batch matrix for [dst; src], then added at
    [src; dst] in index order (scatter_add_ on the CPU walks the index sequentially)."""
    H = matrix.shape[2] - 1
    ids = np.concatenate([src, dst])
    partners = np.concatenate([dst, src])
    msg = np.matmul(matrix[partners], shift_matrix(H)[None, :, :])          # [2B, N, H+1], f32
    np.add.at(matrix, ids, msg)


def get_memory(matrix: np.ndarray, src: np.ndarray, dst: np.ndarray) -> np.ndarray:
    """MemoryModel.py:396-405: matrix[src, dst] / (sum over hops + 1e-4)."""
    m = matrix[src, dst]
    return m / (m.sum(axis=1, keepdims=True, dtype=np.float32) + np.float32(1e-4))

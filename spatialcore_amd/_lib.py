"""ctypes binding of libspatialcore_hip.so (include/spatialcore_hip.h).

There is no CPU fallback anywhere in this package: if the shared library is missing, or no gfx950
device is visible, the first call that needs the GPU raises.
"""

from __future__ import annotations

import ctypes
import os
import threading
from ctypes import POINTER, byref, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_void_p
from typing import Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# (SPATIALCORE_HIP_LIB: another build of the same library -- the sanitizer build of `make asan`, a variant of scripts/build_variant.sh)
LIB_PATH = os.environ.get("SPATIALCORE_HIP_LIB") or os.path.join(_HERE, "libspatialcore_hip.so")

SC_OK, SC_ERR_INVALID, SC_ERR_STATE, SC_ERR_HIP, SC_ERR_NOMEM, SC_ERR_EMPTY = 0, 1, 2, 3, 4, 5
SC_F32, SC_F64 = 0, 1
K_MORAN_PERM, K_LAG, K_KNN, K_PERMGEN, K_LEE_PERM, K_PERM_SCAN, K_PERM_SWAP = 0, 1, 2, 3, 4, 5, 6

# every symbol include/spatialcore_hip.h declares: (name, argtypes); restype is always int
_P = c_void_p
SYMBOLS = {
    "sc_version": [],
    "sc_device_count": [POINTER(c_int)],
    "sc_ctx_create": [c_int, POINTER(c_void_p)],
    "sc_ctx_destroy": [_P],
    "sc_ctx_sync": [_P],
    "sc_ctx_kernel_time": [_P, c_int, POINTER(c_double), POINTER(c_int64)],
    "sc_ctx_reset_timers": [_P],
    "sc_ctx_set_timing": [_P, c_int],
    "sc_ctx_set_permgen_mode": [_P, c_int],
    "sc_ctx_permgen_note": [_P, POINTER(c_char_p)],
    "sc_ctx_probe_streams": [_P, _P, _P],
    "sc_ctx_permgen_form": [_P, c_int64, POINTER(c_char_p)],
    "sc_ctx_set_moran_source_bits": [_P, c_int],
    "sc_ctx_moran_source_bits": [_P, _P],
    "sc_ctx_moran_lag_bits": [_P, _P],
    "sc_ctx_moran_row_groups": [_P, _P],
    "sc_ctx_permgen_stats": [_P, _P, _P, _P, _P, _P],
    "sc_ctx_device_mem": [_P, POINTER(c_int64)],
    "sc_debug_copy": [_P, c_int, c_int64, _P, c_int64],
    "sc_knn_2d": [_P, _P, c_int64, c_int, c_int, _P, _P],
    "sc_knn_fetch": [_P, _P, _P],
    "sc_radius_count_2d": [_P, _P, c_int64, c_double, _P],
    "sc_radius_fill_2d": [_P, c_int64, _P],
    "sc_graph_set_csr": [_P, _P, _P, _P, c_int64, c_int64],
    "sc_graph_from_knn": [_P, c_double],
    "sc_graph_get": [_P, _P, _P, _P],
    "sc_graph_shape": [_P, POINTER(c_int64), POINTER(c_int64)],
    "sc_graph_moments": [_P, POINTER(c_double), POINTER(c_double), POINTER(c_double)],
    "sc_expr_set_csr": [_P, _P, _P, _P, c_int, c_int64, c_int64, _P, c_int64],
    "sc_expr_set_dense": [_P, _P, c_int, c_int64, c_int64, _P, c_int64],
    "sc_expr_stats": [_P, _P, _P],
    "sc_perm_numpy_host": [_P, c_int64, c_int64, _P],
    "sc_perm_generate": [_P, _P, c_int64, c_int64, _P],
    "sc_perm_set": [_P, _P, c_int64, c_int64],
    "sc_perm_generate_counter": [_P, ctypes.c_uint64, c_int64, c_int64, c_int64, _P],
    "sc_perm_counter_host": [ctypes.c_uint64, c_int64, c_int64, c_int64, _P],
    "sc_moran": [_P, c_int64, _P, _P, _P, _P, _P],
    "sc_moran_seeded": [_P, _P, c_int64, _P, _P, _P, _P, _P],
    "sc_moran_seeded_begin": [_P, _P, c_int64, c_int64, c_int64],
    "sc_moran_seeded_finish": [_P, _P, _P, _P, _P, _P, _P],
    "sc_moran_seeded_abort": [_P],
    "sc_lee": [_P, _P, _P, _P, c_int64, c_int64, _P, _P, _P],
    "sc_lee_seeded": [_P, _P, _P, _P, c_int64, c_int64, _P, _P, _P],
    "sc_lee_shared": [_P, _P, _P, c_int32, _P, c_int32, c_int64, _P, _P, _P],
    "sc_lee_observed_f32": [_P, _P, _P, c_int64, _P, _P, _P],
    "sc_local_moran": [_P, c_int64, c_int64, _P, _P, _P, _P, _P],
    "sc_local_moran_seeded": [_P, _P, c_int64, _P, _P, _P, _P, _P],
    "sc_local_moran_hist": [_P, _P],
    "sc_local_moran_classify": [_P, _P, _P, _P, c_float, _P, _P, _P],
    "sc_lee_local": [_P, c_int32, c_int32, c_int64, c_int64, _P, _P, _P, _P],
    "sc_lee_local_seeded": [_P, _P, c_int32, c_int32, c_int64, c_int64, _P, _P, _P, _P, _P, _P],
    "sc_nearest_2d": [_P, _P, c_int64, _P, c_int64, _P, _P],
    "sc_pairwise_2d": [_P, _P, c_int64, _P, c_int64, POINTER(c_double), POINTER(c_double)],
    "sc_nearest_excluding_2d": [_P, _P, _P, c_int64, _P, _P, c_int64, _P, _P],
    "sc_pair_table_2d": [_P, _P, _P, c_int32, _P, _P, c_int32, _P, _P],
    "sc_profile_counts": [_P, _P, c_int64, c_int32, _P, POINTER(c_int64)],
    "sc_enrichment_counts": [_P, _P, c_int64, c_int32, c_int64, c_int64, _P],
    "sc_enrichment_counter": [_P, _P, c_int64, c_int32, ctypes.c_uint64, c_int64, c_int64, c_int64, _P, _P],
    "sc_comm_unique_id": [_P],
    "sc_comm_create": [_P, _P, c_int, c_int, POINTER(c_void_p)],
    "sc_comm_destroy": [_P],
    "sc_allgather": [_P, _P, c_int64, _P],
    "sc_allreduce_max": [_P, _P, c_int64],
    "sc_allreduce_sum_i64": [_P, _P, c_int64],
    "sc_comm_info": [_P, POINTER(c_int), POINTER(c_int), POINTER(c_int)],
}

_lib = None
_lock = threading.Lock()


class SpatialCoreHipError(RuntimeError):
    """HIP runtime / library-state failure reported by libspatialcore_hip.so."""


def load_library() -> ctypes.CDLL:
    """Load the shared library (no GPU needed for loading).  Raises if it has not been built."""
    global _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise ImportError(
                    f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                    "or `make -C spatialcore_amd/csrc`.  There is no CPU fallback.")
            lib = ctypes.CDLL(LIB_PATH)
            for name, argtypes in SYMBOLS.items():
                fn = getattr(lib, name)  # AttributeError if the header and the library diverge
                fn.argtypes = argtypes
                fn.restype = c_int
            lib.sc_last_error.argtypes = []
            lib.sc_last_error.restype = c_char_p
            _lib = lib
    return _lib


def source_hash(names=("sc_moran.hip", "sc_ctx.h")) -> str:
    """sha256 over the native sources that hold the scoring kernels: ties a measured artefact
    (profiles/*_pmc_traffic.json) to the kernel code it was measured on."""
    import hashlib

    h = hashlib.sha256()
    for name in names:
        h.update(name.encode())
        with open(os.path.join(_HERE, "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def _check(rc: int) -> None:
    if rc == SC_OK:
        return
    msg = load_library().sc_last_error().decode("utf-8", "replace")
    if rc in (SC_ERR_INVALID, SC_ERR_EMPTY):
        raise ValueError(msg)
    if rc == SC_ERR_NOMEM:
        raise MemoryError(msg)
    raise SpatialCoreHipError(msg)


def device_count() -> int:
    n = c_int(0)
    rc = load_library().sc_device_count(byref(n))
    return n.value if rc == SC_OK else 0


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(c_void_p)


def _c(a, dtype) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=dtype)


def rng_state_words(rng: np.random.Generator) -> np.ndarray:
    """PCG64 Generator state as the 6 uint64 words sc_perm_* take.  The seed -> state expansion
    (SeedSequence) stays in numpy; only the stream itself is re-implemented natively."""
    st = rng.bit_generator.state
    if st.get("bit_generator") != "PCG64":
        raise ValueError("only numpy PCG64 generators (np.random.default_rng) are supported")
    s, inc = st["state"]["state"], st["state"]["inc"]
    m = 0xFFFFFFFFFFFFFFFF
    return np.array([s >> 64, s & m, inc >> 64, inc & m, st["has_uint32"], st["uinteger"]], dtype=np.uint64)


def set_rng_state(rng: np.random.Generator, words: np.ndarray) -> None:
    w = [int(x) for x in words]
    rng.bit_generator.state = {"bit_generator": "PCG64",
                               "state": {"state": (w[0] << 64) | w[1], "inc": (w[2] << 64) | w[3]},
                               "has_uint32": w[4], "uinteger": w[5]}


def perm_numpy_host(words: np.ndarray, n: int, n_perm: int) -> np.ndarray:
    """Host-only numpy-exact permutation table (no GPU needed).  `words` is updated in place."""
    out = np.empty((n_perm, n), dtype=np.int32)
    _check(load_library().sc_perm_numpy_host(_ptr(words), n, n_perm, _ptr(out)))
    return out


def perm_counter_host(seed: int, n: int, n_perm: int, p_first: int = 0) -> np.ndarray:
    """Host-only counter-based permutations p_first .. p_first + n_perm - 1 (no GPU needed); see
    Context.generate_permutations_counter."""
    out = np.empty((n_perm, n), dtype=np.int32)
    _check(load_library().sc_perm_counter_host(int(seed) & 0xFFFFFFFFFFFFFFFF, n, int(p_first), n_perm, _ptr(out)))
    return out


class Context:
    """One GPU + one HIP stream + the device-resident operands of the hot path."""

    def __init__(self, device: int = 0):
        self._lib = load_library()
        h = c_void_p()
        _check(self._lib.sc_ctx_create(int(device), byref(h)))
        self._h = h
        self.device = int(device)

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.sc_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ---- plumbing ---------------------------------------------------------------------------
    def sync(self) -> None:
        _check(self._lib.sc_ctx_sync(self._h))

    def kernel_time(self, kernel_id: int) -> Tuple[float, int]:
        ms, cnt = c_double(0), c_int64(0)
        _check(self._lib.sc_ctx_kernel_time(self._h, kernel_id, byref(ms), byref(cnt)))
        return ms.value, cnt.value

    def reset_timers(self) -> None:
        _check(self._lib.sc_ctx_reset_timers(self._h))

    def set_timing(self, enabled: bool) -> None:
        _check(self._lib.sc_ctx_set_timing(self._h, int(enabled)))

    def set_permgen_mode(self, mode: int) -> None:
        """0 automatic, 1 sequential rejection scan only, 2 fault injection (tests). Results never differ."""
        _check(self._lib.sc_ctx_set_permgen_mode(self._h, int(mode)))

    def set_moran_source_bits(self, min_bits: int) -> None:
        """Narrowest exact source copy the permutation kernels may gather: 4 (opt-in: nibble slots for count data when they
        take fewer rows than uint8), 8 (default: uint8 when every value is an integer count < 256), 16 (uint16, counts <
        65536), 32 (float32) or 64 (the fp64 tiles)."""
        _check(self._lib.sc_ctx_set_moran_source_bits(self._h, int(min_bits)))

    def moran_source_bits(self) -> int:
        """Source width the last scoring call gathered: 4 (nibble slots) / 8 / 16 (uint8 / uint16 counts), 32 (float32 raw
        values) or 64 (fp64 kernel)."""
        v = c_int(0)
        _check(self._lib.sc_ctx_moran_source_bits(self._h, byref(v)))
        return v.value

    def moran_row_groups(self) -> int:
        """128-byte rows the last scoring call gathered per (permutation, cell)."""
        v = c_int(0)
        _check(self._lib.sc_ctx_moran_row_groups(self._h, byref(v)))
        return v.value

    def moran_lag_bits(self) -> int:
        """Element width of the lag rows the last scoring call streamed: 16 (neighbour sums of a count batch) or 64."""
        v = c_int(0)
        _check(self._lib.sc_ctx_moran_lag_bits(self._h, byref(v)))
        return v.value

    def permgen_stats(self) -> Tuple[int, int, int, int, int]:
        """(jobs by the block-parallel scan, jobs by the sequential scan, verification fallbacks,
        blocks resolved by prepared table lookup, blocks computed by the chain workgroup)."""
        v = [c_int64(0) for _ in range(5)]
        _check(self._lib.sc_ctx_permgen_stats(self._h, *[byref(x) for x in v]))
        return tuple(x.value for x in v)

    def permgen_note(self) -> str:
        """Why the permutation generator left its block-parallel form ("" while it is in use); logged once as a warning."""
        msg = c_char_p()
        _check(self._lib.sc_ctx_permgen_note(self._h, byref(msg)))
        note = (msg.value or b"").decode("utf-8", "replace")
        if note and note != getattr(self, "_note_logged", ""):
            self._note_logged = note
            from spatialcore_amd._logging import get_logger

            get_logger("device").warning(note)
        return note

    def probe_streams(self) -> Tuple[bool, int]:
        """(the generator's streams run concurrently, GPU_MAX_HW_QUEUES as this process's environment has it; 0 = unset)."""
        ok, q = c_int(0), c_int(0)
        _check(self._lib.sc_ctx_probe_streams(self._h, byref(ok), byref(q)))
        self.permgen_note()
        return bool(ok.value), q.value

    def permgen_form(self, n: int) -> str:
        """Which scan a permutation job of length n takes right now: "block-parallel", "sequential (...)" or
        "sequential: <reason>" -- recorded in the provenance entry of every drop-in call that draws permutations."""
        msg = c_char_p()
        _check(self._lib.sc_ctx_permgen_form(self._h, int(n), byref(msg)))
        return (msg.value or b"").decode("utf-8", "replace")

    def debug_copy(self, which: int, offset_bytes: int, count: int, dtype) -> np.ndarray:
        out = np.empty(count, dtype=dtype)
        _check(self._lib.sc_debug_copy(self._h, int(which), int(offset_bytes), _ptr(out), out.nbytes))
        return out

    def device_mem(self) -> int:
        v = c_int64(0)
        _check(self._lib.sc_ctx_device_mem(self._h, byref(v)))
        return v.value

    # ---- A1 / A2 ----------------------------------------------------------------------------
    def knn(self, coords, k: int, include_self: bool = False, return_distance: bool = False,
            fetch: bool = True):
        xy = _c(coords, np.float64)
        if xy.ndim != 2 or xy.shape[1] != 2:
            raise ValueError(f"coordinates must have shape (n, 2), got {xy.shape}")
        n = xy.shape[0]
        idx = np.empty((n, k), dtype=np.int32) if fetch else None
        rd = np.empty((n, k), dtype=np.float64) if (return_distance and fetch) else None
        self._xy_in_flight = xy      # fetch=False returns without waiting: the upload may still be reading this array
        _check(self._lib.sc_knn_2d(self._h, _ptr(xy), n, int(k), int(include_self), _ptr(idx), _ptr(rd)))
        self._knn_shape = (n, int(k))
        return (idx, rd) if return_distance else idx

    def knn_fetch(self, return_distance: bool = True):
        """Neighbour lists of the last knn(..., fetch=False), copied out on the library's copy stream (callable from a
        second thread while the context's stream works on something else)."""
        n, k = self._knn_shape
        idx = np.empty((n, k), dtype=np.int32)
        rd = np.empty((n, k), dtype=np.float64) if return_distance else None
        _check(self._lib.sc_knn_fetch(self._h, _ptr(idx), _ptr(rd)))
        return (idx, rd) if return_distance else idx

    def radius_graph(self, coords, radius: float):
        xy = _c(coords, np.float64)
        if xy.ndim != 2 or xy.shape[1] != 2:
            raise ValueError(f"coordinates must have shape (n, 2), got {xy.shape}")
        n = xy.shape[0]
        indptr = np.empty(n + 1, dtype=np.int64)
        _check(self._lib.sc_radius_count_2d(self._h, _ptr(xy), n, float(radius), _ptr(indptr)))
        nnz = int(indptr[-1])
        indices = np.empty(nnz, dtype=np.int32)
        _check(self._lib.sc_radius_fill_2d(self._h, nnz, _ptr(indices)))
        return indptr, indices

    # ---- A3 ---------------------------------------------------------------------------------
    def set_graph_csr(self, indptr, indices, data, n: int) -> None:
        indptr = _c(indptr, np.int64)
        indices = _c(indices, np.int32)
        data = _c(data, np.float64)
        _check(self._lib.sc_graph_set_csr(self._h, _ptr(indptr), _ptr(indices), _ptr(data), int(n), int(indices.size)))

    def graph_from_knn(self, weight: float) -> None:
        _check(self._lib.sc_graph_from_knn(self._h, float(weight)))

    def graph_shape(self) -> Tuple[int, int]:
        n, nnz = c_int64(0), c_int64(0)
        _check(self._lib.sc_graph_shape(self._h, byref(n), byref(nnz)))
        return n.value, nnz.value

    def get_graph(self):
        n, nnz = self.graph_shape()
        indptr = np.empty(n + 1, dtype=np.int64)
        indices = np.empty(nnz, dtype=np.int32)
        data = np.empty(nnz, dtype=np.float64)
        _check(self._lib.sc_graph_get(self._h, _ptr(indptr), _ptr(indices), _ptr(data)))
        return indptr, indices, data

    def graph_moments(self) -> Tuple[float, float, float]:
        a, b, d = c_double(0), c_double(0), c_double(0)
        _check(self._lib.sc_graph_moments(self._h, byref(a), byref(b), byref(d)))
        return a.value, b.value, d.value

    # ---- expression -------------------------------------------------------------------------
    def set_expression(self, X, gene_cols) -> None:
        """X: scipy sparse (any format) or dense (cells x n_vars); gene_cols: distinct column ids."""
        from scipy import sparse

        cols = _c(gene_cols, np.int32)
        if sparse.issparse(X):
            Xc = X.tocsr()
            if not Xc.has_canonical_format:
                Xc = Xc.copy()
                Xc.sum_duplicates()
            dt = SC_F32 if Xc.dtype == np.float32 else SC_F64
            data = _c(Xc.data, np.float32 if dt == SC_F32 else np.float64)
            indptr = _c(Xc.indptr, np.int64)
            indices = _c(Xc.indices, np.int32)
            _check(self._lib.sc_expr_set_csr(self._h, _ptr(indptr), _ptr(indices), _ptr(data), dt, Xc.shape[0],
                                             Xc.shape[1], _ptr(cols), cols.size))
        else:
            A = np.asarray(X)
            if A.ndim != 2:
                raise ValueError("expression matrix must be 2-D")
            # ship only the requested columns when they are a small part of a wide matrix
            if A.shape[1] > 2 * cols.size:
                A = A[:, cols]
                cols = np.arange(cols.size, dtype=np.int32)
            dt = SC_F32 if A.dtype == np.float32 else SC_F64
            A = _c(A, np.float32 if dt == SC_F32 else np.float64)
            _check(self._lib.sc_expr_set_dense(self._h, _ptr(A), dt, A.shape[0], A.shape[1], _ptr(cols), cols.size))
        self._n_genes = int(cols.size)

    def expr_stats(self):
        mean = np.empty(self._n_genes, dtype=np.float64)
        var = np.empty(self._n_genes, dtype=np.float64)
        _check(self._lib.sc_expr_stats(self._h, _ptr(mean), _ptr(var)))
        return mean, var

    # ---- A4 ---------------------------------------------------------------------------------
    def generate_permutations(self, words: np.ndarray, n: int, n_perm: int, fetch: bool = False):
        out = np.empty((n_perm, n), dtype=np.int32) if fetch else None
        _check(self._lib.sc_perm_generate(self._h, _ptr(words), int(n), int(n_perm), _ptr(out)))
        self.permgen_note()
        return out

    def generate_permutations_counter(self, seed: int, n: int, n_perm: int, p_first: int = 0, fetch: bool = False):
        """EXTENSION: counter-based permutations p_first .. p_first + n_perm - 1 as the resident table (rows 0 ..):
        permutation p is a pure function of (seed, p) -- Fisher-Yates with Philox4x32-10 + Lemire draws -- so ranks and
        batches can take disjoint ranges.  For the paths that have no reference seed semantics only."""
        out = np.empty((n_perm, n), dtype=np.int32) if fetch else None
        _check(self._lib.sc_perm_generate_counter(self._h, int(seed) & 0xFFFFFFFFFFFFFFFF, int(n), int(p_first), int(n_perm),
                                                  _ptr(out)))
        return out

    def set_permutations(self, perm) -> None:
        perm = _c(perm, np.int32)
        if perm.ndim != 2:
            raise ValueError("permutation table must have shape (n_perm, n)")
        _check(self._lib.sc_perm_set(self._h, _ptr(perm), perm.shape[1], perm.shape[0]))

    # ---- A5-A8 ------------------------------------------------------------------------------
    def moran(self, n_perm: int, return_sims: bool = True):
        G = self._n_genes
        I = np.empty(G, dtype=np.float64)
        sims = np.empty((n_perm, G), dtype=np.float64) if (n_perm > 0 and return_sims) else None
        cnt = np.zeros(G, dtype=np.int64) if n_perm > 0 else None
        ssum = np.zeros(G, dtype=np.float64) if n_perm > 0 else None
        ssq = np.zeros(G, dtype=np.float64) if n_perm > 0 else None
        _check(self._lib.sc_moran(self._h, int(n_perm), _ptr(I), _ptr(sims), _ptr(cnt), _ptr(ssum), _ptr(ssq)))
        return {"I": I, "sims": sims, "count_ge": cnt, "sim_sum": ssum, "sim_sumsq": ssq}

    def moran_seeded(self, words: np.ndarray, n_perm: int, return_sims: bool = True):
        """sc_perm_generate + sc_moran, pipelined on the device; `words` is advanced in place."""
        G = self._n_genes
        I = np.empty(G, dtype=np.float64)
        sims = np.empty((n_perm, G), dtype=np.float64) if return_sims else None
        cnt = np.zeros(G, dtype=np.int64)
        ssum = np.zeros(G, dtype=np.float64)
        ssq = np.zeros(G, dtype=np.float64)
        _check(self._lib.sc_moran_seeded(self._h, _ptr(words), int(n_perm), _ptr(I), _ptr(sims), _ptr(cnt),
                                         _ptr(ssum), _ptr(ssq)))
        self.permgen_note()
        return {"I": I, "sims": sims, "count_ge": cnt, "sim_sum": ssum, "sim_sumsq": ssq}

    def moran_seeded_begin(self, words: np.ndarray, n_cells: int, n_perm: int, ahead_chunks: int = 0) -> None:
        """First half of moran_seeded: the generator job (it needs only n_cells and the state) starts and runs while the
        caller builds the graph and uploads the expression; moran_seeded_finish scores.  ahead_chunks: generator chunks
        enqueued before this returns (0 = all: for callers with an upload in front of the finish; else >= 2)."""
        _check(self._lib.sc_moran_seeded_begin(self._h, _ptr(words), int(n_cells), int(n_perm), int(ahead_chunks)))
        self._begun_perms = int(n_perm)

    def moran_seeded_abort(self) -> None:
        _check(self._lib.sc_moran_seeded_abort(self._h))
        self._begun_perms = 0

    def moran_seeded_finish(self, words: np.ndarray, return_sims: bool = True):
        n_perm, G = self._begun_perms, self._n_genes
        I = np.empty(G, dtype=np.float64)
        sims = np.empty((n_perm, G), dtype=np.float64) if return_sims else None
        cnt = np.zeros(G, dtype=np.int64)
        ssum = np.zeros(G, dtype=np.float64)
        ssq = np.zeros(G, dtype=np.float64)
        self._begun_perms = 0
        _check(self._lib.sc_moran_seeded_finish(self._h, _ptr(words), _ptr(I), _ptr(sims), _ptr(cnt), _ptr(ssum), _ptr(ssq)))
        self.permgen_note()
        return {"I": I, "sims": sims, "count_ge": cnt, "sim_sum": ssum, "sim_sumsq": ssq}

    def lee(self, pair_x, pair_y, perm_offset, n_perm: int, return_perms: bool = False):
        px, py = _c(pair_x, np.int32), _c(pair_y, np.int32)
        off = _c(perm_offset, np.int64) if perm_offset is not None else None
        q = px.size
        L = np.empty(q, dtype=np.float64)
        cnt = np.zeros(q, dtype=np.int64)
        Lp = np.empty((q, n_perm), dtype=np.float64) if return_perms else None
        _check(self._lib.sc_lee(self._h, _ptr(px), _ptr(py), _ptr(off), q, int(n_perm), _ptr(L), _ptr(cnt), _ptr(Lp)))
        return {"L": L, "count_abs_ge": cnt, "L_perm": Lp}

    def lee_seeded(self, words: np.ndarray, pair_x, pair_y, n_perm: int, return_perms: bool = False):
        """All pairs of a lees_l call: observed L (fp64 matrix cores) + per-pair permutation counts from the one
        numpy-exact stream `words` (advanced in place), generator and scoring pipelined on the device."""
        px, py = _c(pair_x, np.int32), _c(pair_y, np.int32)
        q = px.size
        L = np.empty(q, dtype=np.float64)
        cnt = np.zeros(q, dtype=np.int64)
        Lp = np.empty((q, n_perm), dtype=np.float64) if return_perms else None
        _check(self._lib.sc_lee_seeded(self._h, _ptr(words), _ptr(px), _ptr(py), q, int(n_perm), _ptr(L), _ptr(cnt), _ptr(Lp)))
        return {"L": L, "count_abs_ge": cnt, "L_perm": Lp}

    def lee_shared(self, words, genes_x, genes_y, n_perm: int, return_perms: bool = False):
        """EXTENSION: the full grid genes_x x genes_y under one shared block of permutations (fp64 MFMA contractions)."""
        gx, gy = _c(genes_x, np.int32), _c(genes_y, np.int32)
        L = np.empty((gx.size, gy.size), dtype=np.float64)
        cnt = np.zeros((gx.size, gy.size), dtype=np.int64)
        Lp = np.empty((n_perm, gx.size, gy.size), dtype=np.float64) if return_perms else None
        # words = None: rows [0, n_perm) of the resident table (e.g. generate_permutations_counter's)
        _check(self._lib.sc_lee_shared(self._h, _ptr(words), _ptr(gx), gx.size, _ptr(gy), gy.size, int(n_perm), _ptr(L), _ptr(cnt), _ptr(Lp)))
        return {"L": L, "count_abs_ge": cnt, "L_perm": Lp}

    def lee_observed_f32(self, pair_x, pair_y) -> np.ndarray:
        """Observed L of each pair exactly as the reference's float32 arithmetic yields it (float32 matrices only)."""
        px, py = _c(pair_x, np.int32), _c(pair_y, np.int32)
        out = np.empty(px.size, dtype=np.float32)
        _check(self._lib.sc_lee_observed_f32(self._h, _ptr(px), _ptr(py), px.size, _ptr(out), None, None))
        return out

    # ---- N1 / N2 ----------------------------------------------------------------------------
    def local_moran(self, n_cells: int, n_perm: int, perm_row0: int = 0, fetch_counts: bool = True):
        G = self._n_genes
        z = np.empty((n_cells, G), dtype=np.float32)
        lag = np.empty((n_cells, G), dtype=np.float32)
        I = np.empty((n_cells, G), dtype=np.float32)
        cnt = np.zeros((n_cells, G), dtype=np.int32) if (n_perm > 0 and fetch_counts) else None
        zero = np.zeros(G, dtype=np.uint8)
        _check(self._lib.sc_local_moran(self._h, int(n_perm), int(perm_row0), _ptr(z), _ptr(lag), _ptr(I), _ptr(cnt),
                                        _ptr(zero)))
        return {"z": z, "lag": lag, "I": I, "count": cnt, "zero_var": zero.astype(bool)}

    def local_moran_seeded(self, words: np.ndarray, n_cells: int, n_perm: int, fetch_counts: bool = True):
        """local_moran with its n_perm permutations drawn from `words` (advanced in place) inside the call: generator and
        per-cell counts run as one pipeline."""
        G = self._n_genes
        z = np.empty((n_cells, G), dtype=np.float32)
        lag = np.empty((n_cells, G), dtype=np.float32)
        I = np.empty((n_cells, G), dtype=np.float32)
        cnt = np.zeros((n_cells, G), dtype=np.int32) if fetch_counts else None
        zero = np.zeros(G, dtype=np.uint8)
        _check(self._lib.sc_local_moran_seeded(self._h, _ptr(words), int(n_perm), _ptr(z), _ptr(lag), _ptr(I), _ptr(cnt),
                                               _ptr(zero)))
        self.permgen_note()
        return {"z": z, "lag": lag, "I": I, "count": cnt, "zero_var": zero.astype(bool)}

    def local_moran_hist(self, n_perm: int) -> np.ndarray:
        """hist[g][c] = cells of gene g whose permutation count is c (of the last local_moran call)."""
        hist = np.zeros((self._n_genes, n_perm + 1), dtype=np.int64)
        _check(self._lib.sc_local_moran_hist(self._h, _ptr(hist)))
        return hist

    def local_moran_classify(self, n_cells: int, p_tab, padj_tab, force_ns, alpha):
        """Per-cell p, adjusted p (table lookups by permutation count) and LISA quadrants of the last local_moran."""
        G = self._n_genes
        with_p = p_tab is not None
        p = np.empty((n_cells, G), dtype=np.float32) if with_p else None
        padj = np.empty((n_cells, G), dtype=np.float32) if with_p else None
        q = np.empty((n_cells, G), dtype=np.int8)
        pt = _c(p_tab, np.float32) if with_p else None
        at = _c(padj_tab, np.float32) if with_p else None
        f = _c(np.asarray(force_ns).astype(np.uint8), np.uint8)
        _check(self._lib.sc_local_moran_classify(self._h, _ptr(pt), _ptr(at), _ptr(f), c_float(float(np.float32(alpha))),
                                                 _ptr(p), _ptr(padj), _ptr(q)))
        return p, padj, q

    def lee_local(self, n_cells: int, gene_x: int, gene_y: int, n_perm: int = 0, perm_row0: int = 0):
        zx = np.empty(n_cells, dtype=np.float64)
        lag = np.empty(n_cells, dtype=np.float64)
        L = np.empty(n_cells, dtype=np.float64)
        cnt = np.zeros(n_cells, dtype=np.int32) if n_perm > 0 else None
        _check(self._lib.sc_lee_local(self._h, int(gene_x), int(gene_y), int(n_perm), int(perm_row0), _ptr(zx),
                                      _ptr(lag), _ptr(L), _ptr(cnt)))
        return {"zx": zx, "lag": lag, "L_local": L, "count": cnt}

    def lee_local_seeded(self, words: np.ndarray, n_cells: int, gene_x: int, gene_y: int, n_perm_global: int, n_perm_local: int):
        """generate_permutations + lee (rows [0, n_perm_global)) + lee_local (the rows behind) for one pair, as one
        pipeline behind the generator; `words` (numpy generator state) is advanced in place."""
        zx = np.empty(n_cells, dtype=np.float64)
        lag = np.empty(n_cells, dtype=np.float64)
        L_local = np.empty(n_cells, dtype=np.float64)
        cnt = np.zeros(n_cells, dtype=np.int32) if n_perm_local > 0 else None
        L = np.zeros(1, dtype=np.float64)
        ge = np.zeros(1, dtype=np.int64)
        w = _c(words, np.uint64)
        _check(self._lib.sc_lee_local_seeded(self._h, _ptr(w), int(gene_x), int(gene_y), int(n_perm_global), int(n_perm_local),
                                             _ptr(L), _ptr(ge), _ptr(zx), _ptr(lag), _ptr(L_local), _ptr(cnt)))
        if w is not words:
            words[...] = w
        return {"L": float(L[0]), "count_abs_ge": int(ge[0]), "zx": zx, "lag": lag, "L_local": L_local, "count": cnt}

    # ---- N3 ---------------------------------------------------------------------------------
    def nearest(self, targets, queries):
        t, q = _c(targets, np.float64), _c(queries, np.float64)
        idx = np.empty(q.shape[0], dtype=np.int32)
        dist = np.empty(q.shape[0], dtype=np.float64)
        _check(self._lib.sc_nearest_2d(self._h, _ptr(t), t.shape[0], _ptr(q), q.shape[0], _ptr(idx), _ptr(dist)))
        return dist, idx

    def nearest_excluding(self, targets, target_code, queries, query_excluded_code):
        """Nearest target whose group code differs from the query's excluded code; (inf, -1) when none is left."""
        t, q = _c(targets, np.float64), _c(queries, np.float64)
        tc, qc = _c(target_code, np.int32), _c(query_excluded_code, np.int32)
        if tc.size != t.shape[0] or qc.size != q.shape[0]:
            raise ValueError("one group code per target and per query is required")
        idx = np.empty(q.shape[0], dtype=np.int32)
        dist = np.empty(q.shape[0], dtype=np.float64)
        _check(self._lib.sc_nearest_excluding_2d(self._h, _ptr(t), _ptr(tc), t.shape[0], _ptr(q), _ptr(qc), q.shape[0],
                                                 _ptr(idx), _ptr(dist)))
        return dist, idx

    def pair_table(self, a_sorted, a_off, b_sorted, b_off):
        """(sum, min) of the pairwise distances of every (group of a, group of b) block; points sorted by group."""
        a, b = _c(a_sorted, np.float64), _c(b_sorted, np.float64)
        ao, bo = _c(a_off, np.int64), _c(b_off, np.int64)
        ga, gb = ao.size - 1, bo.size - 1
        if ao[-1] != a.shape[0] or bo[-1] != b.shape[0]:
            raise ValueError("group offsets do not cover the point arrays")
        tot = np.empty((ga, gb), dtype=np.float64)
        mn = np.empty((ga, gb), dtype=np.float64)
        _check(self._lib.sc_pair_table_2d(self._h, _ptr(a), _ptr(ao), ga, _ptr(b), _ptr(bo), gb, _ptr(tot), _ptr(mn)))
        return tot, mn

    def pairwise(self, a, b):
        a, b = _c(a, np.float64), _c(b, np.float64)
        mean, mn = c_double(0), c_double(0)
        _check(self._lib.sc_pairwise_2d(self._h, _ptr(a), a.shape[0], _ptr(b), b.shape[0], byref(mean), byref(mn)))
        return mean.value, mn.value

    # ---- A9 ---------------------------------------------------------------------------------
    def profile_counts(self, labels, n_types: int) -> np.ndarray:
        lab = _c(labels, np.int32)
        out = np.empty((lab.size, n_types), dtype=np.float32)
        empty = c_int64(0)
        _check(self._lib.sc_profile_counts(self._h, _ptr(lab), lab.size, int(n_types), _ptr(out), byref(empty)))
        return out


    # ---- N4 (extension) ---------------------------------------------------------------------
    def enrichment_counts(self, labels, n_types: int, n_perm: int, perm_row0: int = 0) -> np.ndarray:
        lab = _c(labels, np.int32)
        out = np.empty((n_perm + 1, n_types, n_types), dtype=np.int64)
        _check(self._lib.sc_enrichment_counts(self._h, _ptr(lab), lab.size, int(n_types), int(n_perm), int(perm_row0),
                                              _ptr(out)))
        return out


    def enrichment_counter(self, labels, n_types: int, seed: int, p_first: int, n_perm: int, batch: int = 512):
        """Observed T x T edge counts and the integer sums (deviation, squared deviation, exceedances) over the
        counter-based label permutations p_first .. p_first + n_perm - 1, generation overlapped with counting."""
        lab = _c(labels, np.int32)
        obs = np.empty((n_types, n_types), dtype=np.int64)
        sums = np.empty((3, n_types, n_types), dtype=np.int64)
        _check(self._lib.sc_enrichment_counter(self._h, _ptr(lab), lab.size, int(n_types), int(seed) & 0xFFFFFFFFFFFFFFFF,
                                               int(p_first), int(n_perm), int(batch), _ptr(obs), _ptr(sums)))
        return obs, sums


class RcclComm:
    """One rank's end of the RCCL communicator (include/spatialcore_hip.h, "multi-GPU")."""

    ID_BYTES = 128

    @staticmethod
    def unique_id() -> bytes:
        buf = np.zeros(RcclComm.ID_BYTES, dtype=np.uint8)
        _check(load_library().sc_comm_unique_id(_ptr(buf)))
        return buf.tobytes()

    def __init__(self, ctx: Context, unique_id: bytes, world: int, rank: int):
        if len(unique_id) != self.ID_BYTES:
            raise ValueError(f"an RCCL unique id is {self.ID_BYTES} bytes, got {len(unique_id)}")
        self._lib = load_library()
        self._ctx = ctx                     # keeps the context (stream, device) alive
        self.world, self.rank = int(world), int(rank)
        h = c_void_p()
        idb = np.frombuffer(unique_id, dtype=np.uint8).copy()
        _check(self._lib.sc_comm_create(ctx._h, _ptr(idb), self.world, self.rank, byref(h)))
        self._h = h

    def all_gather(self, block: np.ndarray) -> np.ndarray:
        """(world, *block.shape) float64: every rank's equally shaped block, in rank order."""
        mine = _c(block, np.float64)
        out = np.empty((self.world,) + mine.shape, dtype=np.float64)
        if mine.size:
            _check(self._lib.sc_allgather(self._h, _ptr(mine), mine.size, _ptr(out)))
        return out

    def max_over_ranks(self, values) -> np.ndarray:
        v = np.array(values, dtype=np.float64, ndmin=1)
        _check(self._lib.sc_allreduce_max(self._h, _ptr(v), v.size))
        return v

    def sum_over_ranks_i64(self, values) -> np.ndarray:
        """Element-wise integer sum over ranks (exact; used to merge exceedance counts of permutation shards)."""
        v = np.array(values, dtype=np.int64, ndmin=1)
        shape = v.shape
        v = np.ascontiguousarray(v.reshape(-1))
        if v.size:
            _check(self._lib.sc_allreduce_sum_i64(self._h, _ptr(v), v.size))
        return v.reshape(shape)

    def info(self) -> Tuple[int, int, int]:
        """(ranks, this rank, device) as RCCL itself reports them for this communicator."""
        w, r, d = c_int(0), c_int(0), c_int(0)
        _check(self._lib.sc_comm_info(self._h, byref(w), byref(r), byref(d)))
        return w.value, r.value, d.value

    def barrier(self) -> None:
        self.max_over_ranks([0.0])

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.sc_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_ctx = {}


def release_default_contexts() -> None:
    """Destroy the cached per-device contexts (their device memory and streams); the next default_context() creates anew.
    A process's streams share hardware queues once they outnumber GPU_MAX_HW_QUEUES, so a program that is done with the
    default context and goes on with contexts of its own (the test suite) gives it back."""
    for ctx in list(_default_ctx.values()):
        if ctx is not None and ctx._h is not None:
            ctx.close()
    _default_ctx.clear()


def default_context(device: int = 0) -> Context:
    """Process-wide context per device (created on first use; raises without a gfx950 GPU)."""
    ctx = _default_ctx.get(device)
    if ctx is None or ctx._h is None:
        ctx = Context(device)
        _default_ctx[device] = ctx
    return ctx

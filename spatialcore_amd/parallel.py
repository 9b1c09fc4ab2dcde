"""Gene sharding across the GPUs of one node; no PyTorch anywhere.

The path shards embarrassingly: rank r owns a contiguous slice of the gene list, builds the same
graph and the same permutation table (same seed) on its own GPU, and there is no data-path
collective.  ONE all-gather of the small per-gene result rows (I, expected_I, z, p) at the end makes
the full table available on every rank: ``sc_allgather`` = ``ncclAllGather`` over RCCL (xGMI inside a
node), reached through the C ABI like everything else (``_lib.RcclComm``).

Launch contract: one process per GPU with ``RANK`` / ``WORLD_SIZE`` / ``LOCAL_RANK`` (and, for the
rendezvous name, ``MASTER_PORT``) in the environment -- what ``python -m torch.distributed.run``,
``mpirun`` wrappers or a plain shell loop provide.  The 128-byte RCCL id travels from rank 0 to the
others through a small file (``rendezvous_file()``).

``FileComm`` moves the same blocks through files instead of RCCL.  It exists for rehearsing the
N > 1 code path where RCCL cannot run -- several ranks sharing ONE GPU (RCCL refuses duplicate
devices) or no GPU at all (CPU tests of the shard/merge logic) -- and is only used when asked for
(``SC_COMM_TRANSPORT=file`` or an explicit ``comm=``).  It is a transport, not a compute fallback.
"""

from __future__ import annotations

import os
import time
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np
import pandas as pd

_RESULT_COLUMNS = ["I", "expected_I", "z_score", "p_value"]
_connects = 0   # communicators opened by this process (all ranks open them in the same order)


def shard_bounds(n_items: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous balanced split: the first ``n_items % world`` ranks get one extra item."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world: {rank}/{world}")
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def world_info() -> Tuple[int, int, int]:
    """(rank, world, local_rank) from the launcher's environment (1 process = 1 GPU)."""
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    return rank, world, int(os.environ.get("LOCAL_RANK", rank))


def _launcher_identity() -> str:
    """Same string in every rank of one launch, different across launches: the parent process (the
    launcher all ranks were forked from) with its start time, plus the rendezvous port."""
    ppid = os.getppid()
    started = "0"
    try:
        with open(f"/proc/{ppid}/stat") as f:
            started = f.read().rsplit(")", 1)[1].split()[19]   # field 22: start time in clock ticks
    except (OSError, IndexError):
        pass
    return f"{os.getuid()}_{ppid}_{started}_{os.environ.get('MASTER_PORT', '0')}"


def _private_rendezvous_dir() -> str:
    """A directory only this user can write to (0700, owned by us, not a symlink): the default home of the rendezvous
    files, so that another local user can neither plant an id nor pre-create the temp name (r03 advisor finding: the
    name and the nonce are computable from public facts)."""
    import stat
    import tempfile

    path = os.path.join(tempfile.gettempdir(), f"sc_rdv_{os.getuid()}")
    try:
        os.mkdir(path, 0o700)
    except FileExistsError:
        pass
    st = os.lstat(path)
    if not stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid() or (st.st_mode & 0o077):
        raise PermissionError(f"rendezvous directory {path} is not a private directory of this user; "
                              "set SC_RENDEZVOUS_DIR to one")
    return path


def rendezvous_file(seq: int = 0) -> str:
    """Where rank 0 leaves the RCCL id for the other ranks (``SC_RENDEZVOUS_FILE`` / ``SC_RENDEZVOUS_DIR`` override;
    the default is a 0700 per-user directory under the system's temp directory)."""
    base = os.environ.get("SC_RENDEZVOUS_FILE")
    if not base:
        d = os.environ.get("SC_RENDEZVOUS_DIR") or _private_rendezvous_dir()
        base = os.path.join(d, f"sc_rccl_{_launcher_identity()}")
    return f"{base}.{seq}"


def _launch_nonce() -> bytes:
    """16 bytes that every rank of ONE launch computes alike and no other launch does (launcher pid + start time +
    port + uid): rank 0 prefixes the RCCL id with it, the others ignore files that carry another one.  It guards
    against STALE files (a crashed launch that left its id behind) only -- every input is public, so it is no secret;
    protection against other local users comes from where the file lives (``_private_rendezvous_dir``)."""
    import hashlib

    return hashlib.sha256(("sc-rdv-v1:" + _launcher_identity()).encode()).digest()[:16]


def _await_id(path: str, nonce: bytes, id_bytes: int, timeout_s: float) -> bytes:
    """Poll `path` until it holds `nonce` + an id of `id_bytes` bytes; stale or foreign files are skipped."""
    deadline = time.monotonic() + timeout_s
    while True:
        try:
            with open(path, "rb") as f:
                raw = f.read()
            if len(raw) == len(nonce) + id_bytes and raw[:len(nonce)] == nonce:
                return raw[len(nonce):]
        except FileNotFoundError:
            pass
        if time.monotonic() > deadline:
            raise TimeoutError(f"rendezvous file {path} with this launch's nonce did not appear within {timeout_s:.0f}s")
        time.sleep(0.005)


def _publish(path: str, payload: bytes) -> None:
    tmp = f"{path}.tmp{os.getpid()}"
    try:
        os.unlink(tmp)           # (a leftover of a crashed launch with our pid)
    except FileNotFoundError:
        pass
    fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL | getattr(os, "O_NOFOLLOW", 0), 0o600)   # never through a planted link
    with os.fdopen(fd, "wb") as f:
        f.write(payload)
    os.replace(tmp, path)        # atomic: a reader sees nothing or everything


def _await_file(path: str, timeout_s: float) -> bytes:
    deadline = time.monotonic() + timeout_s
    while True:
        try:
            with open(path, "rb") as f:
                return f.read()
        except FileNotFoundError:
            if time.monotonic() > deadline:
                raise TimeoutError(f"rendezvous file {path} did not appear within {timeout_s:.0f}s") from None
            time.sleep(0.005)


class FileComm:
    """All-gather through a directory (rehearsal / CPU tests only; see the module docstring)."""

    def __init__(self, directory: str, world: int, rank: int, timeout_s: float = 300.0):
        self.world, self.rank, self._dir, self._round, self._timeout = int(world), int(rank), directory, 0, timeout_s
        os.makedirs(directory, exist_ok=True)

    def _name(self, rnd: int, rank: int) -> str:
        return os.path.join(self._dir, f"{rnd}_{rank}.bin")

    def all_gather(self, block: np.ndarray) -> np.ndarray:
        mine = np.ascontiguousarray(block, dtype=np.float64)
        rnd = self._round
        _publish(self._name(rnd, self.rank), mine.tobytes())
        out = np.empty((self.world,) + mine.shape, dtype=np.float64)
        for r in range(self.world):
            raw = mine.tobytes() if r == self.rank else _await_file(self._name(rnd, r), self._timeout)
            out[r] = np.frombuffer(raw, dtype=np.float64).reshape(mine.shape)
        # everybody has written round `rnd`, hence finished reading round `rnd - 1`
        if rnd > 0:
            try:
                os.unlink(self._name(rnd - 1, self.rank))
            except OSError:
                pass
        self._round += 1
        return out

    def max_over_ranks(self, values) -> np.ndarray:
        return self.all_gather(np.array(values, dtype=np.float64, ndmin=1)).max(axis=0)

    def sum_over_ranks_i64(self, values) -> np.ndarray:
        v = np.array(values, dtype=np.int64, ndmin=1)
        # the blocks travel as float64 bytes: split into 32-bit halves so that every integer arrives exactly
        halves = np.stack([v >> 32, v & 0xFFFFFFFF]).astype(np.float64)
        got = self.all_gather(halves).astype(np.int64)
        return ((got[:, 0] << 32) + got[:, 1]).sum(axis=0)

    def info(self) -> Tuple[int, int, int]:
        return self.world, self.rank, -1

    def barrier(self) -> None:
        self.all_gather(np.zeros(1))

    def close(self) -> None:
        """A rank is done once its last all_gather has returned, i.e. it has read everything it will ever read;
        rank 0 waits for every rank's marker and then removes the directory."""
        import shutil

        _publish(os.path.join(self._dir, f"done_{self.rank}"), b"")
        if self.rank == 0:
            for r in range(self.world):
                _await_file(os.path.join(self._dir, f"done_{r}"), self._timeout)
            shutil.rmtree(self._dir, ignore_errors=True)
            try:
                os.unlink(rendezvous_file(0) + ".turn")     # the rehearsal's GPU-turn lock file, if one was used
            except OSError:
                pass


class device_turn:
    """``with device_turn(path):`` -- ranks that SHARE one GPU in a rehearsal take turns on it (an exclusive file
    lock).  Two processes running kernels on one MI355X at the same time were measured to corrupt in-flight wave
    state now and then (profiles/r02_gpu_sharing_raw_stream_corruption.txt); with one process per GPU -- the
    deployment model -- nothing is shared and this is never used."""

    def __init__(self, path: Optional[str] = None):
        self._path = path or (rendezvous_file(0) + ".turn")
        self._fd = None

    def __enter__(self):
        import fcntl

        self._fd = os.open(self._path, os.O_CREAT | os.O_RDWR, 0o600)
        fcntl.flock(self._fd, fcntl.LOCK_EX)
        return self

    def __exit__(self, *exc):
        import fcntl

        fcntl.flock(self._fd, fcntl.LOCK_UN)
        os.close(self._fd)
        self._fd = None
        return False


class _SoloComm:
    world, rank = 1, 0

    def all_gather(self, block):
        return np.ascontiguousarray(block, dtype=np.float64)[None]

    def max_over_ranks(self, values):
        return np.array(values, dtype=np.float64, ndmin=1)

    def sum_over_ranks_i64(self, values):
        return np.array(values, dtype=np.int64, ndmin=1)

    def info(self):
        return 1, 0, -1

    def barrier(self):
        pass

    def close(self):
        pass


def connect(ctx=None, transport: Optional[str] = None, timeout_s: Optional[float] = None):
    """Communicator of this rank among ``WORLD_SIZE`` ranks.

    ``transport``: ``"rccl"`` (default; needs ``ctx``, the rank's ``_lib.Context`` on its own GPU) or
    ``"file"`` (``FileComm``); ``SC_COMM_TRANSPORT`` sets the default.  Every rank must call this the
    same number of times in the same order."""
    global _connects
    rank, world, _ = world_info()
    if world == 1:
        return _SoloComm()
    if timeout_s is None:
        timeout_s = float(os.environ.get("SC_COMM_TIMEOUT_S", 300.0))
    transport = transport or os.environ.get("SC_COMM_TRANSPORT", "rccl")
    seq, _connects = _connects, _connects + 1
    path = rendezvous_file(seq)
    if transport == "file":
        return FileComm(path + ".d", world, rank, timeout_s)
    if transport != "rccl":
        raise ValueError(f"unknown transport '{transport}' (expected 'rccl' or 'file')")
    if ctx is None:
        raise ValueError("the RCCL transport needs the rank's GPU context")
    from spatialcore_amd import _lib

    nonce = _launch_nonce()
    if rank == 0:
        uid = _lib.RcclComm.unique_id()
        _publish(path, nonce + uid)
    else:
        uid = _await_id(path, nonce, _lib.RcclComm.ID_BYTES, timeout_s)
    comm = _lib.RcclComm(ctx, uid, world, rank)     # returns once every rank has joined, i.e. has read the id
    if rank == 0:
        try:
            os.unlink(path)
        except OSError:
            pass
    return comm


def all_gather_rows(local: np.ndarray, n_total: int, comm) -> np.ndarray:
    """All-gather (rows_local, F) float64 blocks whose row ranges follow ``shard_bounds``: blocks are padded
    to the longest shard (the collective wants equal counts) and trimmed again on arrival."""
    local = np.ascontiguousarray(local, dtype=np.float64)
    world = comm.world
    if world == 1:
        return local
    spans = [shard_bounds(n_total, world, r) for r in range(world)]
    longest = max(b - a for a, b in spans)
    padded = np.zeros((longest, local.shape[1]), dtype=np.float64)
    padded[: local.shape[0]] = local
    parts = comm.all_gather(padded)                  # the single collective of the path
    return np.concatenate([parts[r, : b - a] for r, (a, b) in enumerate(spans)], axis=0)


def morans_i_sharded(adata, genes: Optional[Sequence[str]] = None, key_added: str = "morans_i", copy: bool = False,
                     compute: Optional[Callable] = None, comm=None, device: Optional[int] = None, **kwargs):
    """``morans_i`` with the gene list sharded over the ranks of the launch.

    Every rank ends with the complete ``adata.uns[key_added]`` table (input gene order), identical
    to an unsharded call: the permutation table depends only on ``seed`` and ``n_cells``, so each
    gene sees the same permutations whichever rank computes it.  ``kwargs`` go to ``morans_i``
    (``n_neighbors``, ``n_permutations``, ``seed``, ``layer``, ``spatial_key``, ``use_existing_graph``).
    ``device`` defaults to ``LOCAL_RANK``; ``comm`` defaults to ``connect()`` on that device's context
    and is then closed before returning.  ``compute(adata, gene_list, **kwargs) -> DataFrame`` replaces
    the per-shard HIP call in the CPU tests of the shard/merge logic.
    """
    rank, world, local_rank = world_info()
    device = local_rank if device is None else int(device)
    if copy:
        adata = adata.copy()
    names: List[str] = list(adata.var_names) if genes is None else ([genes] if isinstance(genes, str) else list(genes))
    own_comm = comm is None
    if compute is None:
        from spatialcore_amd import _lib
        from spatialcore_amd.spatial.autocorrelation import morans_i

        def compute(ad, gene_list, **kw):
            return morans_i(ad, genes=gene_list, key_added="_shard", device=device, **kw).uns.pop("_shard")

        if own_comm:
            comm = connect(_lib.default_context(device))
    elif own_comm:
        comm = connect(None, transport=os.environ.get("SC_COMM_TRANSPORT", "file"))
    lo, hi = shard_bounds(len(names), comm.world, comm.rank)
    mine = names[lo:hi]
    if mine:
        local = compute(adata, mine, **kwargs)[_RESULT_COLUMNS].to_numpy(dtype=np.float64)
    else:
        local = np.zeros((0, len(_RESULT_COLUMNS)))
    full = all_gather_rows(local, len(names), comm)
    if own_comm:
        comm.close()
    table = pd.DataFrame(full, columns=_RESULT_COLUMNS)
    table.insert(0, "gene", names)
    adata.uns[key_added] = table
    return adata

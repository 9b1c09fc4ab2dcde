"""spatialcore_amd -- MI355X-native drop-in for the spatialcore.spatial hot path."""
from spatialcore_amd._adata import SimpleAnnData

__all__ = ["SimpleAnnData", "init"]
__version__ = "0.1.0"


def init(device: int = 0) -> dict:
    """Create the process-wide device context NOW and say which form the numpy-exact permutation generator will take.

    The fast (block-parallel) generator needs its HIP streams on separate hardware queues.  The library asks the runtime
    for 24 (``GPU_MAX_HW_QUEUES``) when it is loaded -- which only works if nothing initialised HIP in this process
    before, and if the variable was not set to something smaller.  Rather than find out inside the first call (same
    results, about half the speed), call this once after import::

        report = spatialcore_amd.init()
        # {'device': 0, 'hw_queues_requested': 24, 'streams_concurrent': True, 'generator': 'block-parallel'}

    ``streams_concurrent`` is MEASURED (a five-round stream probe, a few milliseconds); a ``False`` comes with a warning
    on the ``spatialcore_amd`` logger and ``generator`` = ``"sequential: <reason>"``.  Every drop-in function also records the
    form it ran with as ``permgen_form`` in its ``adata.uns["spatialcore_metadata"]`` entry.  Raises without a gfx950 GPU
    (there is no CPU fallback)."""
    from spatialcore_amd import _lib

    ctx = _lib.default_context(device)
    ok, queues = ctx.probe_streams()
    return {"device": device, "hw_queues_requested": queues, "streams_concurrent": ok,
            "generator": ctx.permgen_form(1 << 20)}

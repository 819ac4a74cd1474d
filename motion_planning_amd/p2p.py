"""Set-up of the engine's one-shot peer-to-peer all-gather (include/mppi_hip.h mppi_p2p_*) across the ranks
of a torch.distributed group on ONE node: IPC handles of the mailboxes are exchanged through the group, a probe
round-trip is checked on every rank, and only if all ranks pass does the ticker switch from RCCL to p2p."""


def setup(engine, group, rank, world, local_rank, required=False):
    """Returns True when the p2p exchange is live on every rank (False: keep RCCL)."""
    lib = engine._lib
    if not hasattr(lib, "mppi_p2p_create"):
        if required:
            raise RuntimeError("this libmppi_hip.so has no p2p exchange")
        return False
    from . import _p2p_impl
    return _p2p_impl.setup(engine, group, rank, world, local_rank, required)

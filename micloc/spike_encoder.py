"""Drop-in alias: `micloc.spike_encoder` -> haghighatshoarmuir2024_amd.spike_encoder (MI355X implementation)."""
from haghighatshoarmuir2024_amd.spike_encoder import *  # noqa: F401,F403
from haghighatshoarmuir2024_amd import spike_encoder as _impl

__all__ = [n for n in dir(_impl) if not n.startswith("_")]

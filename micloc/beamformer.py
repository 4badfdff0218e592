"""Drop-in alias: `micloc.beamformer` -> haghighatshoarmuir2024_amd.beamformer (MI355X implementation)."""
from haghighatshoarmuir2024_amd.beamformer import *  # noqa: F401,F403
from haghighatshoarmuir2024_amd import beamformer as _impl

__all__ = [n for n in dir(_impl) if not n.startswith("_")]

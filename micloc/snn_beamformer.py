"""Drop-in alias: `micloc.snn_beamformer` -> haghighatshoarmuir2024_amd.snn_beamformer (MI355X implementation)."""
from haghighatshoarmuir2024_amd.snn_beamformer import *  # noqa: F401,F403
from haghighatshoarmuir2024_amd import snn_beamformer as _impl

__all__ = [n for n in dir(_impl) if not n.startswith("_")]

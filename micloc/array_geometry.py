"""Drop-in alias: `micloc.array_geometry` -> haghighatshoarmuir2024_amd.array_geometry (MI355X implementation)."""
from haghighatshoarmuir2024_amd.array_geometry import *  # noqa: F401,F403
from haghighatshoarmuir2024_amd import array_geometry as _impl

__all__ = [n for n in dir(_impl) if not n.startswith("_")]

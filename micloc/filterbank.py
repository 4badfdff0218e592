"""Drop-in alias: `micloc.filterbank` -> haghighatshoarmuir2024_amd.filterbank (MI355X implementation)."""
from haghighatshoarmuir2024_amd.filterbank import *  # noqa: F401,F403
from haghighatshoarmuir2024_amd import filterbank as _impl

__all__ = [n for n in dir(_impl) if not n.startswith("_")]

"""Drop-in alias: `micloc.utils` -> haghighatshoarmuir2024_amd.utils (MI355X implementation)."""
from haghighatshoarmuir2024_amd.utils import *  # noqa: F401,F403
from haghighatshoarmuir2024_amd import utils as _impl

__all__ = [n for n in dir(_impl) if not n.startswith("_")]

"""Drop-in alias: `micloc.xylo_snn_localization` -> haghighatshoarmuir2024_amd.xylo_snn_localization (MI355X implementation,
no rockpool / samna; the integer LIF is parity-unpinned, see that module)."""
from haghighatshoarmuir2024_amd.xylo_snn_localization import *  # noqa: F401,F403
from haghighatshoarmuir2024_amd.xylo_snn_localization import Demo, signal_from_template, xylo_lif, xylo_specification  # noqa: F401

"""Drop-in alias: `micloc.localization_demo_snn` -> haghighatshoarmuir2024_amd.localization_demo_snn (hardware-free Demo)."""
from haghighatshoarmuir2024_amd.localization_demo_snn import Demo  # noqa: F401

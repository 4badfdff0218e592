"""MI355X-native implementation of the micloc hot path (STHT -> RZCC -> SNN beamforming).

The public class surface mirrors the reference's `micloc` package; `import micloc` at the repository root
resolves to thin re-exports of these modules so the paper_plots scripts run unchanged.
"""
from . import _lib  # noqa: F401

__all__ = ["array_geometry", "beamformer", "filterbank", "snn_beamformer", "spike_encoder", "utils", "runtime", "sweep", "flac",
           "xylo_snn_localization", "localization_demo_snn"]
__version__ = "0.1.0"

"""Drop-in `micloc` package: the reference's import paths (micloc.snn_beamformer, micloc.beamformer,
micloc.spike_encoder, micloc.array_geometry, micloc.utils, micloc.filterbank) backed by the MI355X
implementation in haghighatshoarmuir2024_amd."""

"""MI355X photon-beam radiance path: Python binding of the C ABI in
include/photon_beams.h (libphoton_beams.so, hand-written HIP for gfx950).

There is no CPU path in this package: importing works anywhere, but creating a
:class:`~raytracer_amd.api.Context` needs the built library and a GPU.
"""
from .api import (Context, Control, PhotonBeamsError, BEAM_DTYPE, NODE_DTYPE, SEGMENT_DTYPE,  # noqa: F401
                  MODE_721, MODE_73, MODE_73HET, MODE_DEFAULT, MODE_GENERIC, control_defaults, device_count, library, library_path,
                  load_test_model, quantise_argb, read_segments_file, write_segments_file, source_stamp, transformation_matrix, create_surface, make_grid, reload_tuning)

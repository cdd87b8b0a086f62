"""
directdemod_amd -- MI355X-native (gfx950, HIP) implementation of DirectDemod's
per-sample hot path behind the reference's own class surface.

Drop-in modules (same names, signatures and error behaviour as the reference's
``directdemod`` package):

    comm.commSignal   filters.*   demod_fm.demod_fm   demod_am.demod_am
    chunker.chunker   constants   source (IQwav/IQdat readers + device ingest)

All arithmetic runs in hand-written HIP kernels reached through the C-ABI
library ``libdirectdemod_hip.so`` (include/directdemod_hip.h) via ctypes.  There
is no CPU fallback: without the library or without a GPU the compute calls raise.
"""
from . import constants  # noqa: F401

# `from directdemod_amd import *` brings in the drop-in modules (imported lazily: nothing here loads the HIP library)
__all__ = ["comm", "filters", "demod_fm", "demod_am", "chunker", "constants", "source", "resample", "afsk", "noaa_sync", "shard", "stream"]
__version__ = "0.1.0"

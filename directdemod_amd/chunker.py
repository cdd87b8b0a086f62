"""
chunker -- host-side bookkeeping for chunked processing (drop-in for the
reference's directdemod/chunker.py:15-84: same constructor, ``getChunks``, ``get``
and ``set``).

It only produces index ranges and stores the named integers the operators use to
continue across chunk borders (NCO sample index, decimation phase).  On the GPU
path those integers are passed by value into the fused kernel and every other
carried quantity (FIR history, last FM sample) stays in device memory inside the
operator objects, so a chunk loop never synchronises with the device.
"""
from . import constants


class chunker:
    """Index ranges ``[start, stop]`` covering ``sigsrc.length`` samples."""

    def __init__(self, sigsrc, chunkSize=constants.PROC_CHUNKSIZE):
        total = sigsrc.length                     # the only thing needed from the source (chunker.py:30)
        self.__vars = {}
        # Rule of chunker.py:36-45: full chunks are emitted only while at least one
        # more sample would remain; the remainder (which is a *full-size* chunk when
        # total is an exact multiple) closes the list.  An empty or short source
        # yields the single range [0, total].
        size = int(chunkSize)
        nfull = 0 if total <= size else (total - 1) // size
        edges = [k * size for k in range(nfull + 1)] + [total]
        self.__chunks = [[edges[k], edges[k + 1]] for k in range(len(edges) - 1)]

    @property
    def getChunks(self):
        """list of ``[start, stop]`` pairs"""
        return self.__chunks

    def set(self, name, value):
        """store a named value for the following chunks"""
        self.__vars[name] = value

    def get(self, name, init=None):
        """fetch a named value; with ``init`` given, define it first if missing.
        Without ``init`` a missing name raises KeyError (chunker.py:77-78)."""
        if init is not None and name not in self.__vars:
            self.__vars[name] = init
        return self.__vars[name]

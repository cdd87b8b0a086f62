"""
Module-level tunables read by the hot path.  Names and values follow the
reference's constants module (directdemod/constants.py:4-40) because user code
addresses them by name; the grouping and commentary are this package's.
"""

# --- enumerations used as plain ints by filters.butter / sources ----------------
FLT_LP, FLT_HP, FLT_BP, FLT_BS = range(4)          # low/high/band-pass, band-stop
SOURCE_IQWAV, SOURCE_IQDAT = range(2)

# --- chunker variable names (state carried chunk to chunk) ----------------------
CHUNK_FREQOFFSET = "freqoffset"                    # running NCO sample index (comm.py:75-76)
CHUNK_BWLIM = "bwlim"                              # + uniq: decimation phase (comm.py:123-125)

# --- processing ------------------------------------------------------------------
PROC_CHUNKSIZE = 20 * 1000 * 1000                  # samples per chunk (160 MB of complex64)
IQ_SDRSAMPRATE = 2.048e6
IQ_FREQOFFSET = 30000

# --- NOAA APT ----------------------------------------------------------------------
NOAA_FREQ = 137620000
NOAA_SATS = {137620000: "NOAA 15", 137100000: "NOAA 19", 137912500: "NOAA 18"}
NOAA_FMBW = 60000                                  # FM channel bandwidth -> decimate to this
NOAA_AUDSAMPRATE = 20800
NOAA_CRUDESYNCSAMPRATE = 40960
NOAA_T = 1.0 / 4160                                # one APT word, seconds


def _bits(s):
    return [int(ch) for ch in s]


# 40-word sync patterns: A = 7 cycles of 1040 Hz, B = 7 cycles of 832 pps
NOAA_SYNCA = _bits("0000" + "1100" * 7 + "00000000")
NOAA_SYNCB = _bits("0000" + "11100" * 7 + "0")
NOAA_PEAKHEIGHTWIGGLE = 0.25                       # threshold slack below the mean peak height
NOAA_MINPEAKDIST = 0.45                            # seconds between two syncs of one kind
NOAA_DETECTMAXCHANGE = 5                           # samples of jitter tolerated ...
NOAA_DETECTCONSSYNCSNUM = 10                       # ... over this many consecutive syncs
NOAA_COLORCORRECT_FIFOLEN = 10000

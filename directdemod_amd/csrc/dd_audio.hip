// placeholder: audio-rate rows (R2, A1, X1, X2)
#include "dd_common.h"
extern "C" int dd_resample_fft_f64(const double*, double*, int64_t, int64_t, void*) { dd_set_error("not implemented"); return DD_ERR_UNSUPPORTED; }
extern "C" int dd_am_envelope_f64(const double*, double*, int64_t, int64_t, void*) { dd_set_error("not implemented"); return DD_ERR_UNSUPPORTED; }
extern "C" int dd_xcorr_norm_f64(const double*, int64_t, const double*, int, double*, void*) { dd_set_error("not implemented"); return DD_ERR_UNSUPPORTED; }
extern "C" int dd_find_peaks_f64(const double*, int64_t, double, int, int64_t*, int, int*, void*) { dd_set_error("not implemented"); return DD_ERR_UNSUPPORTED; }

// Process-wide modes and study knobs of the convolution family, shared by its translation units (defined in csrc/awr_conv.hip).
#pragma once
#include <stdlib.h>

namespace awr {
inline int env_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
extern int g_force_tm, g_force_tn, g_products, g_wgrad_split, g_staging, g_accum, g_accum_auto_k, g_accum_auto_dgrad;
extern int g_knob_deep, g_knob_deep_1x1, g_knob_fast_stats;
inline int wg_products() { return (g_products == 6 && g_wgrad_split) ? 6 : 1; }
}  // namespace awr

// The one definition of the kernels that live in shared headers (elementwise preparation / reduction kernels, the any-shape fallbacks, the
// feature contraction's product and reduce kernels, the low-rank draw's kernels): every other translation unit sees their declarations only.
// (Rounds 1-4 had them `static` in the headers: 273 copies in 122 units -- profiles/r04_register_report.txt -- that only ever launched one.)
#define GPSIG_KERNEL_DEFS
#include "ctx.hpp"
#include "aux_kernels.hpp"
#include "sig_feat_kernel.hpp"
#include "tvs_tile_kernel.hpp"
#include "tvs_grad_tile_kernel.hpp"

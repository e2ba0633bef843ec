// placeholder until the halo form lands (replaced below in this round)
#include "ief_common.h"
#include "ief_params.h"
int ief_conv_halo_x3p_dispatch(const IefGemmX3pParams& p, hipStream_t st) { (void)p; (void)st; return IEF_ESHAPE; }

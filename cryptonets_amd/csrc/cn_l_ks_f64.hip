// key-switch kernels, arithmetic policy ArF64
#define KS_POLICY ArF64
#define KS_NAME cn_ks_f64
#include "cn_l_ks.inc.h"

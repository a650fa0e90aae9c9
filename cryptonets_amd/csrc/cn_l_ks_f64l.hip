// key-switch kernels, arithmetic policy ArF64L
#define KS_POLICY ArF64L
#define KS_NAME cn_ks_f64l
#include "cn_l_ks.inc.h"

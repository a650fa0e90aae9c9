// key-switch kernels, arithmetic policy ArU64
#define KS_POLICY ArU64
#define KS_NAME cn_ks_u64
#include "cn_l_ks.inc.h"

// register-radix kernels, arithmetic policy ArU64
#define RR_POLICY ArU64
#define RR_NAME cn_rr_u64
#define RR_ENC_TAIL 1
#include "cn_l_rr.inc.h"

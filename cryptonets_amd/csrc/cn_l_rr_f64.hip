// register-radix kernels, arithmetic policy ArF64
#define RR_POLICY ArF64
#define RR_NAME cn_rr_f64
#define RR_ENC_TAIL 1
#include "cn_l_rr.inc.h"

// register-radix kernels, arithmetic policy ArF64L
#define RR_POLICY ArF64L
#define RR_NAME cn_rr_f64l
#include "cn_l_rr.inc.h"

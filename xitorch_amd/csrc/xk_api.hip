// xitorch_amd :: ABI bookkeeping
#include "xk_common.h"

extern "C" int xk_abi_version(void) { return 1; }

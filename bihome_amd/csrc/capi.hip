// Library probe entry points.
#include "common.h"
#include <string.h>

extern "C" {

int bh_version(void) { return 1; }

int bh_device_arch(char* buf, int buflen) {
    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, 0);
    if (e != hipSuccess) return (int)e;
    if (buf && buflen > 0) {
        strncpy(buf, prop.gcnArchName, (size_t)buflen - 1);
        buf[buflen - 1] = 0;
    }
    return BH_OK;
}

}  // extern "C"

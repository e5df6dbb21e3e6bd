// libvidc.so: error state, version, device info.
#include "common.h"
#include <cstring>

namespace vidc {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace vidc

extern "C" int vidc_version(void) { return 1; }
extern "C" const char* vidc_last_error(void) { return vidc::g_err; }

extern "C" int vidc_device_info(int* n_cu, int* lds_bytes_per_cu, char* arch_name, int arch_name_len) {
    int dev = 0;
    VIDC_HIP(hipGetDevice(&dev));
    hipDeviceProp_t p;
    VIDC_HIP(hipGetDeviceProperties(&p, dev));
    if (n_cu) *n_cu = p.multiProcessorCount;
    if (lds_bytes_per_cu) *lds_bytes_per_cu = (int)p.maxSharedMemoryPerMultiProcessor;
    if (arch_name && arch_name_len > 0) {
        strncpy(arch_name, p.gcnArchName, arch_name_len - 1);
        arch_name[arch_name_len - 1] = 0;
    }
    return VIDC_OK;
}

// libvidc.so: error state, version, device info, a shader-clock stamp.
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

// ---- shader-clock stamp ----------------------------------------------------------------------------------------------------------------
// The roofline of the conv kernels is quoted against the guide's nominal fp32 MFMA peak (256 CUs x 256 FLOP/clk x 2.4 GHz = 157.3 TFLOP/s).
// Under the real mix of MFMA, LDS and HBM traffic the chip clocks lower (round 5, in-kernel stamps of the layer-3 launches: 2.03 GHz;
// a register-only MFMA loop alone holds 2.42 GHz), which caps what ANY kernel can reach.  A stamp = the shader-clock counter and the
// 100 MHz wall clock; bench.py enqueues one behind every item of its steady-state stream, and the slope between two stamps far apart
// is the average shader clock WHILE the frame programs run.  A measurement aid, not part of the data path.
namespace {
// The cycle counter is per XCD (eight counters with unrelated offsets): one single-thread workgroup per XCD, each tagging its stamp with the
// XCC id it ran on, so that two stamps are compared XCD by XCD.
__global__ void clock_stamp_kernel(long long* __restrict__ out) {
    unsigned xcc;
#if defined(__gfx942__) || defined(__gfx950__)
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
#else
    xcc = blockIdx.x & 7u;      // (an ARCH override of the Makefile without that hardware register: the dispatch order stands in for the id)
#endif
    long long* o = out + (size_t)blockIdx.x * 4;
    o[0] = (long long)(xcc & 0xF);
    o[1] = (long long)__builtin_readcyclecounter();
    o[2] = (long long)__builtin_amdgcn_s_memrealtime();
    o[3] = 1;
}
}  // namespace

extern "C" int vidc_clock_stamp(long long* out, vidc_stream_t stream) {
    VIDC_REQUIRE(out, VIDC_ERR_NULL, "vidc_clock_stamp: null pointer");
    hipLaunchKernelGGL(clock_stamp_kernel, dim3(VIDC_CLOCK_STAMP_WGS), dim3(1), 0, vidc::as_stream(stream), out);
    VIDC_CHECK_LAUNCH("clock_stamp_kernel");
    return VIDC_OK;
}

// Direct-W form of the 16-row teams of the fp16x2 block stack (h2_phase.hpp h2_stackd_kernel), a translation unit of its own
// so that it compiles beside h2_gemm.hip / h2n_gemm.hip.
#include <stdlib.h>

#include <mutex>

#include "h2_phase.hpp"

namespace mpl {

int launch_h2d_stack(const H2StackArgs& a, int grid, hipStream_t s) {
    static std::atomic<bool> ready[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return MPL_E_LAUNCH;
    if (!ready[dev].load(std::memory_order_acquire)) {
        int per_cu = 0;
        if (hipFuncSetAttribute((const void*)h2_stackd_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, H2_LDS_BYTES) != hipSuccess)
            return MPL_E_LAUNCH;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)h2_stackd_kernel<2>, 512, H2_LDS_BYTES) != hipSuccess || per_cu < 1)
            return MPL_E_UNSUPPORTED;
        ready[dev].store(true, std::memory_order_release);
    }
    if (a.rgs != 1) return MPL_E_INVALID;
    hipLaunchKernelGGL(h2_stackd_kernel<2>, dim3(grid), dim3(512), H2_LDS_BYTES, s, a);
    return MPL_OK;
}

}  // namespace mpl

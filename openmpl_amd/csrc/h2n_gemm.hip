// Row-narrow form of the fp16x2 block stack (h2_phase.hpp h2_stackn_kernel), a translation unit of its own so that it compiles
// beside h2_gemm.hip: the persistent stack kernel for launches that would leave most of the chip idle -- the reference's
// shipped call shape (configs/h36m/mpl_amass/h36m.yaml:37-39,107: TEST.BATCH_SIZE 256, two views; loop
// lib/core/function_mpl.py:334-351) is 512 token rows = 8 row tiles x 4 column groups = 32 of 256 compute units for
// h2_stack_kernel, which walks them in the 0.83 ms it needs for ANY number of tiles.  Here a 64-row tile is shared by 4 (2)
// teams of 16 (32) rows; same arithmetic per output element, bitwise the same poses (tests/test_h2_gpu.py).
#include <stdlib.h>

#include <mutex>

#include "h2_phase.hpp"

namespace mpl {

int launch_h2n_stack(const H2StackArgs& a, int grid, hipStream_t s) {
    static std::atomic<bool> ready[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return MPL_E_LAUNCH;
    if (!ready[dev].load(std::memory_order_acquire)) {
        int per_cu = 0;
        if (hipFuncSetAttribute((const void*)h2_stackn_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, H2_LDS_BYTES) != hipSuccess)
            return MPL_E_LAUNCH;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)h2_stackn_kernel<2>, 512, H2_LDS_BYTES) != hipSuccess || per_cu < 1)
            return MPL_E_UNSUPPORTED;
        ready[dev].store(true, std::memory_order_release);
    }
    if (a.rgs != 1 && a.rgs != 2) return MPL_E_INVALID;
    hipLaunchKernelGGL(h2_stackn_kernel<2>, dim3(grid), dim3(512), H2_LDS_BYTES, s, a);
    return MPL_OK;
}

}  // namespace mpl

// LDS-DMA implicit GEMM (forward / data gradients), workgroup tile 64 x 128 waves-tiles: every epilogue / prologue / accumulation variant of
// this tile shape (csrc/awr_gemm_launch.inc).  Split by tile so that the four shapes compile in parallel.
#include "awr_gemm_launch.inc"

namespace awr {
template void launch_dma_tile<1, 2>(const awr_conv_args*, dim3, hipStream_t, int, int, bool, int);
}  // namespace awr

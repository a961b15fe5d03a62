// lab (tdtk_lab_icp_device_solve): the solve of the host-free ICP loop (loop_dev.h) run ONCE on the rows of pair sums it is handed, so
// that the device solver can be compared with tdtk_align on the same sums.
#include <hip/hip_runtime.h>

#include "kernels.h"
#ifdef TDTK_LAB
#include "loop_dev.h"

namespace tdtk {

__global__ void __launch_bounds__(256) k_solve_once(const double* __restrict__ partials, int rows, double s0, double s1, double s2, IcpLoopDev* st)
{
  __shared__ double red[4][ICP_LOOP_COLS];
  __shared__ double sums[ICP_LOOP_COLS];
  loop_reduce_rows(partials, rows, red, sums);
  const double shift[3] = {s0, s1, s2};
  const LoopSolve o = loop_solve(sums, shift, 0.0, 0.0, -1.0, 0, 2);
  if (threadIdx.x == 0) loop_write_row(st, 0, o, sums);
}

hipError_t launch_solve_once(const double* partials, int rows, const double shift[3], IcpLoopDev* st, hipStream_t s)
{
  hipLaunchKernelGGL(k_solve_once, dim3(1), dim3(256), 0, s, partials, rows, shift[0], shift[1], shift[2], st);
  return hipGetLastError();
}

}  // namespace tdtk
#endif

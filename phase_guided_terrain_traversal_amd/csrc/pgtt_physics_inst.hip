// One physics_kernel instantiation per translation unit (-DPG_MODE, -DPG_DR, -DPG_TERRAIN, -DPG_SUBS) so that the
// variants compile in parallel; each exports a plain host launcher used by pgtt_api.hip.
#include <hip/hip_runtime.h>
#include "pgtt_kernels.hip.h"

#define PG_CAT_(a, s, b, c, d) a##s##_##b##_##c##_##d
#define PG_CAT(a, s, b, c, d) PG_CAT_(a, s, b, c, d)

void PG_CAT(pgtt_launch_physics_s, PG_SUBS, PG_MODE, PG_DR, PG_TERRAIN)(int nblocks, hipStream_t st, const pgtt::KArgs& a, const float* action) {
  hipLaunchKernelGGL((pgtt::physics_kernel<PG_MODE, (PG_DR != 0), (PG_TERRAIN != 0), PG_SUBS>), dim3(nblocks), dim3(64), 0, st, a, action);
}

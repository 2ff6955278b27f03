/* Development tool: the C ABI from a plain C host (no Python, no torch: the HIP runtime is the
 * system's /opt/rocm one, as under a MATLAB mex host).
 *   gcc -Iinclude tools/abi_smoke.c -Lstereo_amd -lstereo_hip -Wl,-rpath,$PWD/stereo_amd -o /tmp/abi_smoke && /tmp/abi_smoke */
#include <stdio.h>
#include <stdint.h>
#include "stereo_hip.h"

int main(void) {
  char err[512] = "";
  printf("devices %d, abi %d\n", stereo_hip_device_count(), stereo_hip_abi_version());
  printf("warm-up rc %d (%s)\n", stereo_hip_warm_up(), stereo_hip_last_error());
  /* a 2 x 2 grid, one frustrated cycle */
  const double U0[4] = {0, 1, 0, 2}, U1[4] = {1, 0, 2, 0};
  const double E00[4] = {0, 0, 0, 0}, E01[4] = {1, 1, 1, 1}, E10[4] = {1, 1, 1, 1}, E11[4] = {0, 0, 0, 0};
  const uint32_t conn[8] = {0, 1, 1, 3, 0, 2, 2, 3};
  double lab[4], en = 0, lb = 0, unl = 0;
  int rc = stereo_rd(U0, U1, E00, E01, E10, E11, conn, 4, 4, 1, lab, &en, &lb, &unl, err, sizeof(err));
  printf("stereo_rd rc %d energy %g bound %g unlabelled %g labels %g %g %g %g %s\n", rc, en, lb, unl, lab[0], lab[1], lab[2], lab[3], err);
  /* TRW-S on the same grid, 3 labels */
  const double unary[12] = {0, 1, 2, 2, 0, 1, 1, 2, 0, 0, 2, 1};
  double q[12], alphas[4] = {1, 1, 1, 1}, tl[4], ten = 0, tlb = 0, it = 0;
  for (int e = 0; e < 4; ++e) for (int k = 0; k < 3; ++k) q[3 * e + k] = k;
  rc = stereo_trws(1, unary, conn, q, q, alphas, 1.0, 5, -1.0, 3, 4, 4, tl, &ten, &tlb, &it, err, sizeof(err));
  printf("stereo_trws rc %d energy %g bound %g iterations %g labels %g %g %g %g %s\n", rc, ten, tlb, it, tl[0], tl[1], tl[2], tl[3], err);
  return rc;
}

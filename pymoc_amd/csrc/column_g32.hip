// column_g32.hip -- k_column_steps<32, P, *> instantiations (see column.hip.h).
#include "column.hip.h"

namespace pm {

int column_steps_g32(int P, const pm_columns &c, const double *wA, const double *vdx,
                     const double *bin, double dt, int nsteps, int ops, hipStream_t st) {
  switch (P) {
    case 1: return launch_column_steps<32, 1>(c, wA, vdx, bin, dt, nsteps, ops, st);
    case 2: return launch_column_steps<32, 2>(c, wA, vdx, bin, dt, nsteps, ops, st);
    case 3: return launch_column_steps<32, 3>(c, wA, vdx, bin, dt, nsteps, ops, st);
    case 4: return launch_column_steps<32, 4>(c, wA, vdx, bin, dt, nsteps, ops, st);
    case 5: return launch_column_steps<32, 5>(c, wA, vdx, bin, dt, nsteps, ops, st);
    case 6: return launch_column_steps<32, 6>(c, wA, vdx, bin, dt, nsteps, ops, st);
    case 7: return launch_column_steps<32, 7>(c, wA, vdx, bin, dt, nsteps, ops, st);
    case 8: return launch_column_steps<32, 8>(c, wA, vdx, bin, dt, nsteps, ops, st);
  }
  return fail(PM_EINVAL, "unsupported levels-per-lane %d for %d-lane groups", P, 32);
}

}  // namespace pm

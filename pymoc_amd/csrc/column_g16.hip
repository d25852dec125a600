// column_g16.hip -- k_column_steps<16, P, *> instantiations (see column.hip.h).
#include "column.hip.h"

namespace pm {

int column_steps_g16(int P, const pm_columns &c, const double *wA, const double *vdx,
                     const double *bin, double dt, int nsteps, int ops, hipStream_t st) {
  switch (P) {
    case 1: return launch_column_steps<16, 1>(c, wA, vdx, bin, dt, nsteps, ops, st);
    case 2: return launch_column_steps<16, 2>(c, wA, vdx, bin, dt, nsteps, ops, st);
    case 3: return launch_column_steps<16, 3>(c, wA, vdx, bin, dt, nsteps, ops, st);
    case 4: return launch_column_steps<16, 4>(c, wA, vdx, bin, dt, nsteps, ops, st);
    case 5: return launch_column_steps<16, 5>(c, wA, vdx, bin, dt, nsteps, ops, st);
    case 6: return launch_column_steps<16, 6>(c, wA, vdx, bin, dt, nsteps, ops, st);
    case 7: return launch_column_steps<16, 7>(c, wA, vdx, bin, dt, nsteps, ops, st);
    case 8: return launch_column_steps<16, 8>(c, wA, vdx, bin, dt, nsteps, ops, st);
  }
  return fail(PM_EINVAL, "unsupported levels-per-lane %d for %d-lane groups", P, 16);
}

}  // namespace pm

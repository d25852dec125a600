// column_g64.hip -- k_column_steps<64, P, *> instantiations (see column.hip.h).
#include "column.hip.h"

namespace pm {

int column_steps_g64(int P, const pm_columns &c, const double *wA, const double *vdx,
                     const double *bin, double dt, int nsteps, int ops, hipStream_t st) {
  switch (P) {
    case 1: return launch_column_steps<64, 1>(c, wA, vdx, bin, dt, nsteps, ops, st);
    case 2: return launch_column_steps<64, 2>(c, wA, vdx, bin, dt, nsteps, ops, st);
    case 3: return launch_column_steps<64, 3>(c, wA, vdx, bin, dt, nsteps, ops, st);
    case 4: return launch_column_steps<64, 4>(c, wA, vdx, bin, dt, nsteps, ops, st);
    case 5: return launch_column_steps<64, 5>(c, wA, vdx, bin, dt, nsteps, ops, st);
    case 6: return launch_column_steps<64, 6>(c, wA, vdx, bin, dt, nsteps, ops, st);
    case 7: return launch_column_steps<64, 7>(c, wA, vdx, bin, dt, nsteps, ops, st);
    case 8: return launch_column_steps<64, 8>(c, wA, vdx, bin, dt, nsteps, ops, st);
    case 10: return launch_column_steps<64, 10>(c, wA, vdx, bin, dt, nsteps, ops, st);
    case 13: return launch_column_steps<64, 13>(c, wA, vdx, bin, dt, nsteps, ops, st);
    case 16: return launch_column_steps<64, 16>(c, wA, vdx, bin, dt, nsteps, ops, st);
  }
  return fail(PM_EINVAL, "unsupported levels-per-lane %d for %d-lane groups", P, 64);
}

}  // namespace pm

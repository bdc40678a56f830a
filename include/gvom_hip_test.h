/*
 * gvom_hip_test.h -- what lib/libgvom_hip_test.so (make -C g-vom_amd test-lib; -DGVOM_TEST_HOOKS) adds to the C ABI of
 * include/gvom_hip.h.  The test library is the production library -- same sources, same kernels, same entry points --
 * plus three TEST HOOKS that the production library does not contain (there gvom_set_tuning returns GVOM_ERR_INVALID for the
 * two names and the environment variable is never read).  Tests load it through GVOM_HIP_LIBRARY or gvom.Gvom(_library=...).
 *
 *   gvom_set_tuning(h, "epoch_bias", v)   advances the handle's 32-bit tile-epoch counter by v, e.g. to just below its wrap:
 *                                         the renumbering of live maps then runs within a few steps
 *                                         instead of after ~3 days (tests/test_hip_parity.py::test_tile_epoch_renumbering_*).
 *   gvom_set_tuning(h, "churn", 1)        sharded handles: the endpoint send region is re-allocated -- and therefore exported
 *                                         and mapped by every peer again -- on EVERY scan (the abuse that found the three rules of
 *                                         inter-process memory, csrc/gvom_comm.hip; tests/test_hip_sharded.py).
 *   GVOM_TEST_IPC_REFUSE="export:N" | "import:N" [",rank:R"]   (environment, read once per process) the N-th
 *                                         hipIpcGetMemHandle / hipIpcOpenMemHandle this process (of rank R) attempts is answered
 *                                         as the HSA runtime answers when it refuses an allocation, every repetition too: the
 *                                         peer transport's recovery (fresh allocation, collective retry) runs.
 */
#ifndef GVOM_HIP_TEST_H
#define GVOM_HIP_TEST_H
#include "gvom_hip.h"
#define GVOM_TEST_KNOB_EPOCH_BIAS "epoch_bias"
#define GVOM_TEST_KNOB_CHURN      "churn"
#define GVOM_TEST_ENV_IPC_REFUSE  "GVOM_TEST_IPC_REFUSE"
#endif /* GVOM_HIP_TEST_H */

/* mci_debug.h -- development and test hooks of libmci_hip.so.  NOT part of the drop-in boundary (include/mci.h): nothing here stands
 * in for reference code, tests and tools/ may use it, a binding must not. */
#ifndef MCI_DEBUG_H
#define MCI_DEBUG_H
#include "../../include/mci.h"
#ifdef __cplusplus
extern "C" {
#endif
/* tools/persist_trace.py: the persistent kernel's counter words and, in builds with -DMCI_PERSIST_TRACE, the wall-clock stamps of
 * three of its workgroups over the first eight turns of the last launch */
int mci_debug_persist_words(mci_problem *prob, unsigned long long *out, int32_t n);
/* out[0] = serial walks of train! (mci_set_train_walk mode 1) this problem has run as slots with given decisions, out[1] = walks in
 * the general form (mode 2, a decision that did not hold, grids too long for the slots' LDS).  Synchronises the stream. */
int mci_debug_walk_counts(mci_problem *prob, int64_t *out);
/* test hook: the serial walk of train! with one decision deliberately wrong, so that its check and the fall-back to the general form
 * run (same results as mci_set_train_walk(prob, 1)); on = 0 takes it back */
int mci_debug_plant_wrong_decision(mci_problem *prob, int32_t on);
/* test hook: ticks of the 100 MHz clock a grid-wide wait of the persistent :vegas launch may take before it gives up (default 2 s =
 * 200000000); a tiny value forces the stall so that the fall-back to the launch chain can be tested */
int mci_debug_persist_spin_ticks(mci_problem *prob, unsigned long long ticks);
#ifdef __cplusplus
}
#endif
#endif

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
/* Layout decisions of mci_problem_create that tests and A/B tools force; process-wide, consulted by the NEXT mci_problem_create; on = 0
 * takes an override back.  Keys:
 *   table_mode      0 .. 3   placement of grids / histograms
 *   hist_tile_bins  n        bins of an LDS histogram tile (forces several tiles)
 *   no_split_all    1        tiled :vegas: tile 0 stays in the sample pass, only the other tiles are replayed
 *   l1_phase        0 | 1    dimension-major gather phase of the many-grid sample pass
 *   hist_copies     n        interleaved histogram copies of the :vegas sample kernel (1 = none)
 *   train_walk      0 | 1 | 2  = mci_set_train_walk on every new problem
 *   fresh_floors    n        (consulted per launch) length of automatic :vegasmc chains that start afresh, in burn-in floors (8)
 *   fresh_burnin_pct n       (per launch) ... and the least part of such a chain that is not measured, in per cent (profiles/r05_bias.txt A5)
 *   split_chunk      n       (per launch) samples per chunk, over all blocks, of a many-grid :vegas launch (mci_debug_split_chunks)
 *   spec_self_check  0 | 1   (per launch) the self-check of a several-lanes-per-chain code object (mci_chain_speculation_status): never |
 *                            also when the kernel cache holds the marker of an earlier pass (the guard test of tests/test_hip_spec.py)
 * (The library reads two environment variables and no others: MCI_KERNEL_CACHE -- the directory code objects are cached in -- and
 * MCI_JIT_FLAGS -- extra hiprtc options; INTEGRATION.md.) */
int mci_debug_override(const char *key, int64_t value, int32_t on);
/* the last many-grid (several histogram tiles) :vegas launch of the problem: how many chunks of the blocks' samples it ran as (sample pass
 * -> replay per chunk; override key split_chunk = samples per chunk over all blocks, default min(2^27, 7.5 GB of stream)) and the bytes of
 * parked (weights, bins) stream it held at a time */
int mci_debug_split_chunks(const mci_problem *prob, int64_t *chunks, int64_t *bytes);
/* what mci_jit.h puts into the kernel-cache key for "which compiler made this code object" (hiprtc version, the files of libhiprtc and
 * libamd_comgr, the target): set != NULL overrides it for this process ("" takes the override back); out: the identity in force */
int mci_debug_compiler_id(const char *set, char *out, int32_t n);
/* the constants of the automatic :mcmc chain length (DESIGN.md "Chains"), process-wide, for A/B campaigns (tools/mcmc_policy.py): measured
 * steps of a first launch (4096) | how much longer than the chains that measured the holds a launch's chains may be (2) | length of a
 * carried chain in longest holds (4) | its minimum in half burn-in floors (2); <= 0 keeps a value */
int mci_debug_mcmc_policy(int64_t pilot_steps, int64_t grow, int64_t carry_holds, int64_t carry_half_floors);
#ifdef __cplusplus
}
#endif
#endif

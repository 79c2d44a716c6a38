/* Development hooks of libsynthsr_hip.so -- NOT part of the drop-in boundary (include/synthsr_hip.h).
 *
 * The boundary contract is stateless and re-entrant: every entry point of synthsr_hip.h depends only on its
 * arguments.  The one exception is kept out of that header on purpose: a PROCESS-WIDE A/B switch that the profiling
 * scripts under tools/ (ab.py, conv_ablate.py, persist_check.py, ...) use to time kernel variants against each other.
 * Nothing in synthsr_amd/, scripts/ or bench.py calls it; it is not thread-safe; options that change the launch geometry
 * must be set before weights are packed (a packed weight set is only valid for the plan it was packed under). */
#ifndef SYNTHSR_HIP_TUNING_H
#define SYNTHSR_HIP_TUNING_H
#ifdef __cplusplus
extern "C" {
#endif

/* option 0 = persistent forward kernel on the large levels (default 1),
 * 1 = diagnostic ablation mask, 2 = force MT, 3 = EXPERIMENTAL MFMA+VALU co-execution for Cout % 16 == 8 (default 0),
 * 4 = 4x4x1-MFMA kernels, 5 = split-K workgroup target, 6 = brick tiles, 7 = parity split of small up-conv data gradients.  Options
 * that change the launch geometry must be set before weights are packed. */
int synthsr_conv3d_set_option(int option, int value);

/* Deterministic mode (process-wide, single stream; synchronises the device).  on = 1: every cross-workgroup float
 * accumulation (weight / bias / BatchNorm gradients, BatchNorm statistics, losses, the critic's dense layers) is flushed in
 * workgroup-id order instead of arrival order, in-workgroup LDS float atomics are replaced by ordered sums, and the split-K /
 * parity-split forward variants (partial sums meeting in atomics) are not selected: the same inputs give bit-identical
 * results run after run.  Slower (flushes are serialised); the default (0) keeps plain atomics.  Not covered: channel counts
 * C with 384 % (C / 4) != 0 and the Dice sums of the segmentation-regularised loss (data-indexed LDS atomics).
 * Reference: SURVEY.md section 5 (determinism); the reference itself relies on TF's non-deterministic GPU reductions. */
int synthsr_set_deterministic(int on);
/* 0 = off, 1 = on and every ordered wait completed, 2 = on but a wait timed out (results may be unordered), -1 = error */
int synthsr_deterministic_status(void);

#ifdef __cplusplus
}
#endif
#endif

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

#ifdef __cplusplus
}
#endif
#endif

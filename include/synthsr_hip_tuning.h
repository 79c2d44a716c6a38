/* Process-wide switches of libsynthsr_hip.so -- NOT part of the drop-in boundary (include/synthsr_hip.h).
 *
 * The boundary contract is stateless and re-entrant: every entry point of synthsr_hip.h depends only on its
 * arguments.  The exceptions are kept out of that header on purpose:
 *  - synthsr_set_conv_arithmetic: which matrix instructions evaluate the fp32 convolutions (below).
 *  - synthsr_conv3d_set_option: an A/B switch that the profiling scripts under tools/ (ab.py, conv_ablate.py, ...) use to
 *    time kernel variants against each other.  Nothing in synthsr_amd/, scripts/ or bench.py calls it.
 *  - synthsr_set_deterministic: called by synthsr_amd.ops.set_deterministic, training(deterministic=True) and the parity
 *    tests.  Its state (a device block holding scratch for ordered reductions, installed in every translation unit's
 *    g_syn_det symbol) is per PROCESS and per CURRENT DEVICE and assumes ONE stream: every network, critic and predictor
 *    of the process is switched together, and kernels of two streams must not overlap while it is on.
 * Neither is thread-safe; options that change the launch geometry must be set before weights are packed (a packed weight
 * set is only valid for the plan it was packed under). */
#ifndef SYNTHSR_HIP_TUNING_H
#define SYNTHSR_HIP_TUNING_H
#ifdef __cplusplus
extern "C" {
#endif

/* option 0 = persistent forward kernel on the large levels (default 1),
 * 1 = diagnostic ablation mask, 2 = force MT, 3 = EXPERIMENTAL MFMA+VALU co-execution for Cout % 16 == 8 (default 0),
 * 4 = 4x4x1-MFMA kernels, 5 = split-K workgroup target, 6 = brick tiles, 7 = parity split of small up-conv data gradients,
 * 8 = generation of the split forward kernel (0 round 3; 1 default: conversion inside the K loop, LDS-weights kernel where it
 * quantises better; 2 LDS-weights kernel everywhere), 9 = smallest layer (4x4x16 tiles x co-chunks) planned on the split
 * kernels (default 200), 10 = stacked weight layout of the plain Cout = 24 split convs (default 1: the three bf16 pieces share
 * row tiles, 10 instead of 12 MFMAs per K step; 0: two padded 16-row tiles per piece), 11 = smallest layer (4x4x16 tiles) whose
 * weight gradient takes the split kernel (default 1; rounds 1-3: 256), 12 = bit mask over the split weight gradient (default
 * 1): bit 0 stacked column tiles for Cout = 24 (the dz pieces are read as five column tiles, 10 instead of 12 MFMAs per row
 * tile and K step; off: two padded tiles per piece), bit 1 (A/B only) 24-column workgroups also where Cout % 48 == 0, bit 2
 * (A/B only) 8 instead of 16 input channels per 48-column workgroup where Cin % 16 == 0, bit 3 (A/B only) 8 instead of all 24
 * input channels per stacked 24-column workgroup (Cin = 24).  The environment variable SYNTHSR_CONV_OPTIONS="12=9,8=3" applies
 * options when the Python host loads the library (profiling tools only).  Options that change the launch geometry
 * or a packed layout must be set before weights are packed. */
int synthsr_conv3d_set_option(int option, int value);

/* Arithmetic of the fp32 3x3x3 convolutions (process-wide; set it BEFORE weights are packed: a packed weight set is only valid
 * for the arithmetic it was packed under -- synthsr_amd.unet re-packs when the mode changed).
 *   1 (default) "split": every fp32 operand is the exact sum of three bf16 numbers (round to nearest even on what the previous
 *      pieces left); a product a*b is accumulated as a0 b0 + a0 b1 + a1 b0 + a1 b1 + a0 b2 + a2 b0 on v_mfma_f32_16x16x32_bf16,
 *      each partial product exact in the fp32 accumulator; what is left out (a1 b2 + a2 b1 + a2 b2) is < 2^-23 |a b| in the
 *      worst case, 2^-24 at most / 2^-27 rms over random operands: within the rounding of an fp32 multiply-add.  Inputs, outputs, accumulation, BatchNorm statistics, gradients and weights stay fp32; against a
 *      float64 convolution the result is as accurate as the fp32-MFMA kernels' (tests/test_split_gpu.py).  Used for the
 *      layers with >= 256 tiles of 4x4x16 voxels and channel counts that are multiples of 8 (csrc/conv_split.hip: forward,
 *      data gradient and weight gradient of plain convs, forward and data gradient of the folded decoder / stride-2 parity
 *      convs); the rest (first layer, deep levels, the folded convs' weight gradient) runs on the fp32 matrix instructions
 *      in either mode.
 *   2 "split9": the same kernels and packed weights with ALL nine partial products a_i b_j: an fp32 product is reproduced
 *      exactly (no term dropped) at 1.5x the matrix instructions of "split".
 *   0 "fp32_mfma": v_mfma_f32_4x4x1 / 16x16x4 kernels everywhere (csrc/conv3d.hip), the round-1/2 path.
 * The reference computes in fp32 on TensorFlow (SynthSR/training.py:330-341); both modes are fp32 computations of it. */
int synthsr_set_conv_arithmetic(int mode);
int synthsr_conv_arithmetic(void);
/* A counter that moves whenever the arithmetic or a plan-changing option above takes a new value: weights packed under an
 * older epoch must be packed again (synthsr_amd.unet / synthsr_amd.critic compare it at every repack()). */
int synthsr_conv3d_layout_epoch(void);
/* 1 / 0: whether the weight gradient of a plain 3x3x3 conv of this shape runs on the split kernels under the current
 * arithmetic (the dispatcher's own condition; forward / data-gradient plans: synthsr_conv3d_plan) -- what benchmarks price a
 * layer against.  Negative: SYNTHSR_EINVAL. */
int synthsr_conv3d_wgrad_runs_split(const int shape[3], int Cin, int Cout);

/* Diagnostic (host only, no device work): the launch width and tile range of one workgroup of the split kernels -- the same
 * function the kernels evaluate (csrc/conv_split.hip: tile_walk_of).  kernel 0 = forward / data gradient (ny = output-channel
 * chunks), 1 = weight gradient (ny = input-channel chunks x column groups), 2 = folded forward (ny = parity groups).
 * out = {grid width, first tile, end, stride}: workgroup (block_x, linear y/z index block_yz) walks first, first + stride, ...
 * < end.  tests/test_host_cpu.py checks that every tile of every launch geometry is visited exactly once, evenly. */
int synthsr_split_tile_schedule(int kernel, int ntiles, int ny, int block_x, int block_yz, int out[4]);

/* Deterministic mode (process-wide, per current device, single stream; synchronises the device).  on = 1: every
 * cross-workgroup float accumulation is performed in a fixed order -- small partials (channel sums, BatchNorm statistics,
 * losses, the critic's dense outputs) are parked per workgroup and added up in workgroup-id order by the workgroup that
 * arrives last; weight gradients go to one private copy of dW per workgroup column which a second kernel sums in column
 * order (no serialisation: ~1.1x the default step time at 160^3); in-workgroup LDS float atomics are replaced by ordered
 * sums, and the split-K / parity-split forward variants (partial sums meeting in atomics) are not selected: the same
 * inputs give bit-identical results run after run.  The default (0) keeps plain atomics.  Not covered: channel counts
 * C with 384 % (C / 4) != 0.  An
 * allocation failure leaves the mode off and nothing half-installed.
 * Reference: SURVEY.md section 5 (determinism); the reference itself relies on TF's non-deterministic GPU reductions. */
int synthsr_set_deterministic(int on);
/* 0 = off, 1 = on and every ordered wait completed, 2 = on but a wait timed out (results may be unordered), -1 = error */
int synthsr_deterministic_status(void);

#ifdef __cplusplus
}
#endif
#endif

/* The one mode of libsynthsr_hip.so that is NOT an argument -- kept out of the drop-in boundary (include/synthsr_hip.h) on purpose.
 *
 * The boundary is stateless and re-entrant: every entry point of synthsr_hip.h depends only on its arguments (the
 * arithmetic of the convolutions is a field of the caller's synthsr_conv_ctx; the A/B option switch and the arithmetic setter
 * of rounds 1-4 no longer exist).  The exception:
 *  - synthsr_set_deterministic: called by synthsr_amd.ops.set_deterministic, training(deterministic=True) and the parity
 *    tests.  Its state (a device block holding tickets and scratch for ordered reductions, installed in every translation
 *    unit's g_syn_det symbol, plus the caller-provided private dW planes of the weight-gradient flush) is per DEVICE -- the device that is
 *    current when it is called; another device of the same process keeps its own setting -- and assumes ONE stream per device:
 *    kernels of two streams must not overlap while it is on.  Not thread-safe.
 * Also here: a host-only query of the tile schedule (tests). */
#ifndef SYNTHSR_HIP_TUNING_H
#define SYNTHSR_HIP_TUNING_H
#ifdef __cplusplus
extern "C" {
#endif

/* Diagnostic (host only, no device work): the launch width and tile range of one workgroup of the split kernels -- the same
 * function the kernels evaluate (csrc/conv_split.hip: tile_walk_of).  kernel 0 = forward / data gradient (ny = output-channel
 * chunks), 1 = weight gradient (ny = input-channel chunks x column groups), 2 = folded forward (ny = parity groups).
 * out = {grid width, first tile, end, stride}: workgroup (block_x, linear y/z index block_yz) walks first, first + stride, ...
 * < end.  tests/test_host_cpu.py checks that every tile of every launch geometry is visited exactly once, evenly. */
int synthsr_split_tile_schedule(int kernel, int ntiles, int ny, int block_x, int block_yz, int out[4]);

/* Deterministic mode (per device -- the CURRENT one --, single stream; synchronises the device).  on = 1: every
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
/* The private dW planes of the deterministic weight-gradient flush are CALLER memory (torch allocates): `planes` = device
 * buffer of `bytes` on the current device, kept until replaced or withdrawn (NULL, 0); synchronises the device.  A weight
 * gradient that needs more than is registered returns SYNTHSR_EWORKSPACE and does nothing;
 * synthsr_deterministic_workspace_demand() = the largest demand (bytes) any call on this device has had so far -- register
 * at least that much and call again (synthsr_amd/ops.py does exactly this).  The library itself allocates device memory in
 * ONE place: the ticket block + 64 MB of ordered-reduction scratch inside synthsr_set_deterministic(1). */
int synthsr_set_deterministic_workspace(void* planes, unsigned long long bytes);
unsigned long long synthsr_deterministic_workspace_demand(void);
/* 0 = off, 1 = on and every ordered wait completed, 2 = on but a wait timed out (results may be unordered), -1 = error */
int synthsr_deterministic_status(void);

#ifdef __cplusplus
}
#endif
#endif
